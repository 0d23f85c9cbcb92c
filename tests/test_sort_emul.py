"""sort_emul.h (the device's restatement of libstdc++ std::sort) against the real std::sort on the host — CPU only."""
import os
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_sort_emulation_matches_libstdcxx():
    with tempfile.TemporaryDirectory() as tmp:
        exe = os.path.join(tmp, "sort_emul_check")
        subprocess.run(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "csrc", "sort_emul_check.cpp")], check=True, timeout=120)
        out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
        assert out.returncode == 0, out.stdout + out.stderr
        assert out.stdout.startswith("OK "), out.stdout
