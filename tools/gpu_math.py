"""Device build of qa_device_math.h against the host build of the same source (which the CPU suite pins to libm)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from qaray_amd import hip
L = hip.lib()
for f in (L.qa_test_math_device, L.qa_test_math_host):
    f.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
rng = np.random.default_rng(1)
n = 1 << 22
def run(fn, x, y):
    a = np.zeros_like(x); b = np.zeros_like(x)
    assert L.qa_test_math_device(fn, x.ctypes.data, y.ctypes.data, x.size, a.ctypes.data) == 0, L.qa_last_error()
    assert L.qa_test_math_host(fn, x.ctypes.data, y.ctypes.data, x.size, b.ctypes.data) == 0
    return a, b
for name, fn, x, y in [
    ("powf (0,1)^gloss", 2, rng.random(n, dtype=np.float32), rng.choice(np.array([2, 5, 10, 20, 50, 80, 100, 0.5, 1], np.float32), n)),
    ("powf (0,8)^(-50,250)", 2, (rng.random(n, dtype=np.float32) * 8), (rng.random(n, dtype=np.float32) * 300 - 50)),
    ("powf near 1", 2, np.float32(1) + (rng.random(n, dtype=np.float32) - np.float32(0.5)) * np.float32(1e-5), np.full(n, 80, np.float32)),
    ("expf (-100,100)", 3, (rng.random(n, dtype=np.float32) * 200 - 100), np.zeros(n, np.float32)),
    ("expf (-1,0)", 3, -rng.random(n, dtype=np.float32), np.zeros(n, np.float32)),
    ("sinf [0,2pi]", 0, rng.random(n, dtype=np.float32) * np.float32(6.2831855), np.zeros(n, np.float32)),
    ("cosf [0,2pi]", 1, rng.random(n, dtype=np.float32) * np.float32(6.2831855), np.zeros(n, np.float32)),
]:
    a, b = run(fn, x.astype(np.float32), y.astype(np.float32))
    bad = a.view(np.uint32) != b.view(np.uint32)
    bad &= ~(np.isnan(a) & np.isnan(b))
    print(f"{name}: {int(bad.sum())} of {n} differ", flush=True)
    for i in np.nonzero(bad)[0][:4]:
        print("   x", float(x[i]).hex(), "y", float(y[i]), "device", float(a[i]).hex(), "host", float(b[i]).hex())
