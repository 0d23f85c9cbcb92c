"""transformValue / invertValue (ref utils/utils.h:93-108) — the scale of the 601-bin value / reward heads — pinned to their CLOSED FORM,
evaluated here in numpy with the reference's own promotions, not to another transcription:

    h(x)    = float( sign(x) * (sqrt(|double(x)| + 1) - 1) + double(float(0.001f * x)) )
    h^-1(y) = sign(y) * ( powf( float( (sqrt(1 + double(0.004f) * (|double(y)| + 1 + double(0.001f))) - 1) / double(0.002f) ), 2 ) - 1 )

(the unqualified sqrt / fabs are the C double functions there: tests/csrc/overload_check.cpp is compiled with utils.h's includes and says so;
powf(x, 2.0f) is the correctly rounded square: x * x is exact in double).  IEEE sqrt, +, *, / are correctly rounded in numpy and in libm alike,
so the closed form has one value per input.  Checked against it: the oracle, the product's host functions (mz_invert_value /
mz_transform_value: the lock-step path and the learner-side sampler) and — in tests/test_gpu_net.py — the device function of the simulation kernel."""
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def closed_form_transform(x):
    x = np.asarray(x, np.float32)
    sign = np.sign(x).astype(np.float64)
    eps = np.float32(0.001)
    prod = (eps * x).astype(np.float32)  # float * float
    return (sign * (np.sqrt(np.abs(x.astype(np.float64)) + 1.0) - 1.0) + prod.astype(np.float64)).astype(np.float32)


def closed_form_invert(y):
    y = np.asarray(y, np.float32)
    eps = np.float32(0.001)
    four_eps = np.float64(np.float32(4) * eps)   # 4 * epsilon: int * float -> float
    two_eps = np.float64(np.float32(2) * eps)
    inner = 1.0 + four_eps * (np.abs(y.astype(np.float64)) + 1.0 + np.float64(eps))
    x = ((np.sqrt(inner) - 1.0) / two_eps).astype(np.float32)        # the argument powf receives
    sq = (x.astype(np.float64) * x.astype(np.float64)).astype(np.float32)  # powf(x, 2.0f): exact product, one rounding
    return (np.sign(y).astype(np.float32) * (sq - np.float32(1))).astype(np.float32)


def vectors():
    rng = np.random.default_rng(20260928)
    v = [0.0, -0.0, 1.0, -1.0, 300.0, -300.0, 1e-30, -1e-30, 1e-8, 0.5, 2.5, 17.25, 299.999, 0.001, 1e4, -1e4, 1e6]
    v += list(rng.uniform(-300, 300, 96)) + list(rng.uniform(-1, 1, 48)) + list(10.0 ** rng.uniform(-6, 5, 48) * rng.choice([-1, 1], 48))
    return np.array(v, np.float32)


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def test_unqualified_sqrt_and_fabs_are_the_double_functions(tmp_path):
    exe = str(tmp_path / "overload_check")
    subprocess.check_call(["g++", "-std=c++17", os.path.join(ROOT, "tests", "csrc", "overload_check.cpp"), "-o", exe])
    assert subprocess.check_output([exe], text=True).split() == ["8", "8"]


def test_oracle_matches_the_closed_form(oracle):
    L = oracle.lib()
    v = vectors()
    assert len(v) >= 64
    inv = np.array([L.mzo_invert_value(float(x)) for x in v], np.float32)
    tr = np.array([L.mzo_transform_value(float(x)) for x in v], np.float32)
    assert np.array_equal(_bits(inv), _bits(closed_form_invert(v)))
    assert np.array_equal(_bits(tr), _bits(closed_form_transform(v)))
    # and the two are inverse to each other to within the rounding of the f32 round trip
    mid = v[(np.abs(v) > 1e-3) & (np.abs(v) < 1e4)]
    back = closed_form_invert(closed_form_transform(mid))
    assert np.all(np.abs(back - mid) <= 2e-3 * np.maximum(1.0, np.abs(mid)))


def test_product_host_functions_match_the_closed_form(mz):
    L = mz.load()
    v = vectors()
    inv = np.array([L.mz_invert_value(float(x)) for x in v], np.float32)
    tr = np.array([L.mz_transform_value(float(x)) for x in v], np.float32)
    assert np.array_equal(_bits(inv), _bits(closed_form_invert(v)))
    assert np.array_equal(_bits(tr), _bits(closed_form_transform(v)))
