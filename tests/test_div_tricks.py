"""The division shortcuts of the PUCT selection kernel (reciprocal table + Markstein correction, pool_body.h) against real IEEE
division on the host — CPU only (IEEE mul / fma / conversions are the same on both sides)."""
import os
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_division_shortcuts_are_exact():
    with tempfile.TemporaryDirectory() as tmp:
        exe = os.path.join(tmp, "div_tricks_check")
        subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-o", exe, os.path.join(ROOT, "tests", "csrc", "div_tricks_check.cpp")],
                       check=True, timeout=120)
        out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
        assert out.returncode == 0 and out.stdout.startswith("OK "), out.stdout + out.stderr


def test_powf_with_exponent_one_is_the_identity():
    """worker.cpp selectChildBySoftmaxCount skips std::pow for the default temperature 1."""
    with tempfile.TemporaryDirectory() as tmp:
        exe = os.path.join(tmp, "pow_one_check")
        subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-o", exe, os.path.join(ROOT, "tests", "csrc", "pow_one_check.cpp")], check=True, timeout=120)
        out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
        assert out.returncode == 0 and out.stdout.startswith("OK "), out.stdout + out.stderr
