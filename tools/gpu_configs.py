"""Throughput of every BASELINE.json config on one MI355X at its own frame size (spp reduced where the full
count would take minutes; throughput is flat in spp).  C3-C5 use the synthetic stand-in assets."""
import sys, os, subprocess, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scenes", "gen_assets.py")], check=True, stdout=subprocess.DEVNULL)
from qaray_amd.host import load_scene_blob
from qaray_amd import hip
ctx = hip.Context(0)
for tag, scene, size, spp_full, spp in [("C1", "example_project3_sphere.xml", (256, 256), 1, 1), ("C1 x1024spp", "example_project3_sphere.xml", (256, 256), 1, 1024),
                                        ("C2", "example_project12_box.xml", (1920, 1080), 512, 512), ("C3", "example_project7_object.xml", (1920, 1080), 256, 64),
                                        ("C4", "example_project12_caustics_glossy.xml", (3840, 2160), 1024, 64), ("C5", "trc_scene_tower.xml", (3840, 2160), 2048, 16)]:
    blob = load_scene_blob(scene, size=size)
    ctx.upload_scene(blob)
    ctx.render_region((0, 0) + size, 1)
    ctx.reset_kernel_time(); ctx.reset_counters()
    t0 = time.time(); ctx.render_region((0, 0) + size, spp); wall = time.time() - t0
    ms, _ = ctx.kernel_time(); c = ctx.counters()
    print(f"{tag}: {scene} {size[0]}x{size[1]} @ {spp} spp (config: {spp_full}): kernel {ms:.1f} ms, {c['samples'] / ms * 1e-3:.1f} Msamples/s, "
          f"{(c['casts_normal'] + c['casts_shadow']) / c['samples']:.2f} casts/sample", flush=True)
