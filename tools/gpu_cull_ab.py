"""Instance culling of the cooperative kernel (option "cs_cull") on / off: bitwise comparison and frame times on the BASELINE
scenes and on a synthetic scene with many nodes (a field of spheres and mesh instances under two point lights).
   python tools/gpu_cull_ab.py [spp]"""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
subprocess.run([sys.executable, os.path.join(ROOT, "scenes", "gen_assets.py")], check=True, stdout=subprocess.DEVNULL)
from qaray_amd.host import load_scene_blob
from qaray_amd import hip
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 8


def many_nodes_xml(path, n_side=6):
    objs = ['<object type="plane" name="floor" material="floor"><scale value="60"/></object>']
    for i in range(n_side):
        for j in range(n_side):
            x, y = (i - (n_side - 1) / 2) * 7.0, (j - (n_side - 1) / 2) * 7.0
            if (i + j) % 5 == 0:
                objs.append(f'<object type="obj" name="teapot-low.obj" material="m{(i * 7 + j) % 3}"><scale value="0.25"/><rotate angle="{(i * 37 + j * 11) % 360}" z="1"/>'
                            f'<translate x="{x}" y="{y}" z="0"/></object>')
            else:
                objs.append(f'<object type="sphere" name="s{i}_{j}" material="m{(i * 7 + j) % 3}"><scale value="{1.5 + ((i * 3 + j) % 4) * 0.4}"/><translate x="{x}" y="{y}" z="2.2"/></object>')
    mats = ('<material type="blinn" name="floor"><diffuse r="0.8" g="0.8" b="0.8"/><specular value="0"/></material>'
            '<material type="blinn" name="m0"><diffuse r="0.8" g="0.2" b="0.2"/><specular value="0.6"/><glossiness value="30"/></material>'
            '<material type="blinn" name="m1"><diffuse r="0.2" g="0.7" b="0.3"/><specular value="0.5"/><glossiness value="20"/><reflection value="0.4"/></material>'
            '<material type="blinn" name="m2"><diffuse r="0.1" g="0.1" b="0.1"/><specular value="0.8"/><glossiness value="50"/><refraction value="0.9" index="1.5"/></material>')
    lights = ('<light type="ambient" name="amb"><intensity value="0.1"/></light>'
              '<light type="point" name="p1"><intensity value="0.7"/><position x="20" y="-30" z="40"/></light>'
              '<light type="point" name="p2"><intensity value="0.5"/><position x="-25" y="10" z="30"/></light>')
    cam = '<camera><position x="0" y="-70" z="35"/><target x="0" y="0" z="0"/><up x="0" y="0" z="1"/><fov value="35"/><width value="800"/><height value="600"/></camera>'
    open(path, "w").write("<xml><scene>" + "".join(objs) + mats + lights + "</scene>" + cam + "</xml>")


tmp = tempfile.mkdtemp()
many = os.path.join(tmp, "many_nodes.xml")
many_nodes_xml(many)
CASES = [("many (38 nodes)", many, (1920, 1080), os.path.join(ROOT, "scenes")), ("c3", "example_project7_object.xml", (1920, 1080), None),
         ("c4", "example_project12_caustics_glossy.xml", (3840, 2160), None), ("c5", "trc_scene_tower.xml", (3840, 2160), None)]
ctx = hip.Context(0)
for tag, scene, size, assets in CASES:
    blob = load_scene_blob(scene, size=size, asset_root=assets)
    out = {}
    for cull in (0, 1):
        ctx.set_option("cs_cull", cull)
        ctx.upload_scene(blob)
        ctx.render_region((0, 0, 64, 64), 1)
        ctx.reset_kernel_time(); ctx.reset_counters()
        out[cull] = ctx.render_region((0, 0) + size, spp)
        ms, _ = ctx.kernel_time(); c = ctx.counters()
        print(f"{tag:16s} cs_cull={cull}: {ms:8.1f} ms  {c['samples'] / ms * 1e-3:8.1f} Msamples/s  [{ctx.kernel_name()}]", flush=True)
    same = all(np.array_equal(a.view(np.uint32), b.view(np.uint32)) for a, b in zip(out[0], out[1]))
    print(f"{tag:16s} culled frame == unculled frame, bit for bit: {same}", flush=True)
ctx.close()
