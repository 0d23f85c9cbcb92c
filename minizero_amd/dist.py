"""Multi-GPU plumbing of the worker: one process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI on
ROCm; "gloo" in the CPU tests).  Self-play games are independent, so the data path has NO collective
(SURVEY.md §8e); what is here is the optional synchronous weight broadcast on load_model, the barriers/
reductions bench.py needs, and the shard arithmetic (which games / which seed a rank gets)."""
import os

import numpy as np


def env_rank():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def shard_seed(program_seed, rank, streams=1):
    """ref actor_group.cpp:66-70: slave thread `id` seeds its generator with program_seed + id.  A rank with `streams` generators (mz_rng_streams) owns the ids
    rank * streams .. rank * streams + streams - 1, so the generators of a node are program_seed + 0 .. program_seed + ranks * streams - 1, each used once."""
    return program_seed + rank * max(1, streams)


def games_for_rank(total_games, rank, world):
    """ref actor_group.cpp:185: actor i uses network i % nGPU  ->  rank r owns games r, r+world, ..."""
    return len(range(rank, total_games, world))


class Group:
    def __init__(self, backend="nccl", device=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.rank, self.local_rank, self.world = env_rank()
        self.device = device
        if self.world > 1 and not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            kw = {}
            if backend == "nccl":
                kw["device_id"] = device if device is not None else torch.device("cuda", self.local_rank)
            dist.init_process_group(backend, **kw)
        self.backend = backend

    def _dev(self):
        if self.backend != "nccl":
            return "cpu"  # gloo reduces host tensors (the GPU work of the ranks is still fenced by barrier()'s synchronize)
        return self.device if self.device is not None else "cuda"

    def broadcast_weights(self, weights, src=0):
        """load_model fan-out: rank `src` holds the parsed blob; one flat f32 broadcast (<= 6.1 MB)."""
        if self.world == 1:
            return weights
        t = self.torch.from_numpy(np.ascontiguousarray(weights, np.float32)).to(self._dev())
        self.dist.broadcast(t, src=src)
        return t.cpu().numpy()

    def barrier(self):
        gpu = self.torch.cuda.is_available()
        if gpu:
            self.torch.cuda.synchronize()
        if self.world > 1:
            self.dist.barrier()
        if gpu:
            self.torch.cuda.synchronize()

    def reduce(self, values, op="sum"):
        v = np.asarray(values, np.float64)
        if self.world == 1:
            return v
        t = self.torch.from_numpy(v.copy()).to(self._dev())
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX if op == "max" else self.dist.ReduceOp.SUM)
        return t.cpu().numpy()

    def close(self):
        if self.world > 1 and self.dist.is_initialized():
            self.dist.barrier()
            self.dist.destroy_process_group()
