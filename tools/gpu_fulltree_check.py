"""Full BASELINE frame (1920x1080 @ 512 spp, 1.6e9 casts): default kernel (own search tree) vs counting kernel
(reference tree walk) must agree bit for bit."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from qaray_amd.host import load_scene_blob
from qaray_amd import hip
ctx = hip.Context(0)
ctx.upload_scene(load_scene_blob("example_project12_box.xml", size=(1920, 1080)))
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 512
ctx.reset_counters(); ctx.reset_kernel_time()
a = ctx.render_region((0, 0, 1920, 1080), spp); ca = ctx.counters(); ta = ctx.kernel_time()[0]
ctx.reset_counters(); ctx.reset_kernel_time()
b = ctx.render_region((0, 0, 1920, 1080), spp, stats=True); cb = ctx.counters(); tb = ctx.kernel_time()[0]
print(f"spp {spp}: casts {ca['casts_normal']}; default {ta:.1f} ms, counting {tb:.1f} ms; rgb bit-equal {np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32))}, "
      f"depth equal {np.array_equal(a[1], b[1])}; reference-tree work: {cb['bvh_nodes'] / cb['casts_normal']:.2f} nodes, {cb['tri_tests'] / cb['casts_normal']:.2f} triangle tests per cast")
d = a[0] != b[0]
ys, xs = np.nonzero(d.any(axis=2))
print("differing pixels:", len(ys), "max abs diff", float(np.abs(a[0] - b[0]).max()))
for y, x in list(zip(ys, xs))[:10]:
    print("  pixel", int(x), int(y), a[0][y, x], b[0][y, x])
