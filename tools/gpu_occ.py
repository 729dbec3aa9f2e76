import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import subprocess
subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scenes", "gen_assets.py")], check=True)
from qaray_amd.host import load_scene_blob
from qaray_amd import hip
ctx = hip.Context(0)
for name in ["example_project7_object.xml", "example_project12_caustics_glossy.xml", "trc_scene_tower.xml", "trc_scene_xmas.xml", "example_project12_box.xml"]:
    blob = load_scene_blob(name, size=(960, 540))
    ctx.upload_scene(blob); ctx.reset_counters()
    ctx.render_region((0, 0, 960, 540), 4, stats=True)
    c = ctx.counters()
    lanes = c["bvh_nodes"] // 1000000; waves64 = c["tri_tests"] // 1000000
    casts = c["casts_normal"] + c["casts_shadow"]
    print(f"{name}: mesh traversals per cast {lanes / casts:.2f}; lanes per wave-level traversal {64 * lanes / max(waves64, 1):.1f} of 64 ({lanes / max(waves64, 1) * 100:.0f} %)")
