"""qa_device_math.h compiled for the host (same source as the device code) against glibc.
sinf/cosf restate glibc's algorithm and must return its bits on [0, 2*pi] - the only range the
integrator uses (phi = 2*pi*r, r in [0,1]).  expf/powf restate glibc's table-driven algorithms
(including where its x86-64 FMA build fuses) and must return its bits too: tests/cpp/math_exhaustive.c
sweeps them against the host libm (the full sweep - every float for expf - passes; the suite runs a
strided one)."""
import ctypes as C
import os
import subprocess

import numpy as np

from qaray_amd import hip


def _host(fn, x, y=None):
    L = hip.lib()
    L.qa_test_math_host.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    out = np.zeros_like(x)
    rc = L.qa_test_math_host(fn, x.ctypes.data, y.ctypes.data if y is not None else None, x.size, out.ctypes.data)
    assert rc == 0
    return out


def _libm(name, x, y=None):
    libm = C.CDLL("libm.so.6")
    f = getattr(libm, name)
    f.restype = C.c_float
    f.argtypes = [C.c_float] * (2 if y is not None else 1)
    if y is None:
        return np.array([f(float(v)) for v in x], np.float32)
    return np.array([f(float(a), float(b)) for a, b in zip(x, y)], np.float32)


def test_sincos_bit_exact_on_integrator_range():
    rng = np.random.default_rng(5)
    # every float in a few exponent ranges is too slow from python; sample densely instead:
    x = np.concatenate([
        rng.random(120000, dtype=np.float32) * np.float32(6.2831855),
        np.linspace(0, 6.2831855, 60000, dtype=np.float32),
        np.float32(2 * np.pi) * (np.float32(1.0) - rng.random(20000, dtype=np.float32) ** 8),   # near 2*pi
        rng.random(20000, dtype=np.float32) ** 8 * np.float32(1e-2),                              # near 0
        np.array([0.0, 6.2831855, np.pi, np.pi / 2, np.pi / 4, 0.75, 0.7853982, 2.0 ** -12, 2.0 ** -13], np.float32)])
    assert np.array_equal(_host(0, x).view(np.uint32), _libm("sinf", x).view(np.uint32))
    assert np.array_equal(_host(1, x).view(np.uint32), _libm("cosf", x).view(np.uint32))


def test_powf_expf_bit_exact():
    rng = np.random.default_rng(6)
    x = rng.random(40000, dtype=np.float32)
    y = rng.choice(np.array([2, 5, 10, 20, 50, 100, 1, 0, 80, 0.5], np.float32), 40000)
    assert np.array_equal(_host(2, x, y).view(np.uint32), _libm("powf", x, y).view(np.uint32))
    xe = -rng.random(40000, dtype=np.float32) * np.float32(50)
    assert np.array_equal(_host(3, xe).view(np.uint32), _libm("expf", xe).view(np.uint32))
    # edge cases the shading code relies on
    e = _host(2, np.array([0, 0, 1, 0.5], np.float32), np.array([5, 0, 7, 0], np.float32))
    assert e.tolist() == [0.0, 1.0, 1.0, 1.0]


def test_powf_expf_strided_sweep(tmp_path):
    """expf over every 16th block of 65536 float bit patterns (incl. NaN / inf / overflow ranges), powf over
    positive bases below 2 x 17 exponents plus random pairs - hundreds of millions of comparisons."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "math_exhaustive")
    subprocess.run(["gcc", "-O2", "-fopenmp", os.path.join(root, "tests", "cpp", "math_exhaustive.c"), "-o", exe, "-ldl", "-lm"],
                   check=True)
    for mode in ("0", "1"):
        r = subprocess.run([exe, hip.HIP_LIB_PATH, mode, "16"], capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr
        assert " 0 mismatches" in r.stdout

