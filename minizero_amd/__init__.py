"""minizero_amd — MI355X-native self-play worker for MiniZero (host-side Python mirror of the C ABI).

The product is libmzgpu.so (hand-written HIP for gfx950, built in-tree by `__graft_entry__.build()`).
This package only binds include/mzgpu.h with ctypes; there is no Python or CPU fallback: every entry
point raises MzError when the HIP library is missing or no GPU is visible.
"""
from .lib import (MzError, NetDesc, SearchCfg, WorkerStats, Net, Pool, Worker, Env, DataLoader, load, lib_path, make_desc,  # noqa: F401
                  param_count, generate_weights, device_count, usable_cpus, godev_playout, envdev_playout, sort_candidates, invert_values_device, read_pt, read_weight_file, read_weights_once, weight_file_reads, DESCS, CONFIGS)
