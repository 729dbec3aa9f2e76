"""Isolated axis-aligned right triangles far from the object-space origin: how many pixels differ between the default
kernel (own tree) and the counting kernel (reference walk)?  python tools/gpu_edge_probe.py n offset w h spp [pipeline]"""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from test_gpu_parity import _write_edge_scene
from qaray_amd.host import load_scene_blob
from qaray_amd import hip
n, off, w, h, spp = int(sys.argv[1]), float(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
tmp = tempfile.mkdtemp()
xml = _write_edge_scene(tmp, n, off)
ctx = hip.Context(0)
ctx.set_pipeline(sys.argv[6] if len(sys.argv) > 6 else "mega")
ctx.upload_scene(load_scene_blob(xml, size=(w, h), asset_root=tmp))
a = ctx.render_region((0, 0, w, h), spp)
ctx.reset_counters()
b = ctx.render_region((0, 0, w, h), spp, stats=True)
c = ctx.counters()
d = int((a[0].view(np.uint32) != b[0].view(np.uint32)).any(axis=2).sum()), int((a[1].view(np.uint32) != b[1].view(np.uint32)).sum())
print(f"{os.environ.get('QA_HIP_LIB', 'product')[-30:]} n={n} offset={off} {w}x{h}@{spp} [{ctx.kernel_name()[:40]}]: differing pixels rgb {d[0]} depth {d[1]}; casts {c['casts_normal']}+{c['casts_shadow']}")
