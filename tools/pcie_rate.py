"""PCIe-inclusive rate of the host-buffer entry point (qa_render_region): scene upload from host
memory + render + D2H of rgb/depth/ns, wall clock."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qaray_amd.host import load_scene_blob
from qaray_amd import hip
W, H, spp = 1920, 1080, 512
blob = load_scene_blob("example_project12_box.xml", size=(W, H))
ctx = hip.Context(0)
ctx.upload_scene(blob); ctx.render_region((0, 0, W, H), 8)   # warm-up
for rep in range(3):
    t0 = time.perf_counter()
    ctx.upload_scene(blob)
    rgb, depth, ns = ctx.render_region((0, 0, W, H), spp)
    dt = time.perf_counter() - t0
    print(f"host-buffer path: {dt*1e3:.2f} ms per frame incl. upload + D2H of {rgb.nbytes + depth.nbytes + ns.nbytes} B -> {W*H*spp/dt*1e-6:.1f} Msamples/s")
