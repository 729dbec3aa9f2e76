"""Two (or more) library builds on the same scenes: frame time, kernel, hash of the frame (colour | depth | sample counts), one process
   per build.  python tools/gpu_lib_ab.py lib_base,lib "scene.xml:W:H:SPP,..." [reps]"""
import hashlib, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--one":
    sys.path.insert(0, ROOT)
    from qaray_amd.host import load_scene_blob
    from qaray_amd import hip
    ctx = hip.Context(0)
    ctx.set_option("verbose", 1) if os.environ.get("QA_AB_VERBOSE") else None
    for spec in sys.argv[2].split(","):
        scene, w, h, spp = spec.split(":")
        w, h, spp = int(w), int(h), int(spp)
        ctx.upload_scene(load_scene_blob(scene, size=(w, h)))
        ctx.render_region((0, 0, 64, 64), 1)
        best, sha = 1e30, ""
        for rep in range(int(sys.argv[3])):
            ctx.reset_kernel_time(); ctx.reset_counters()
            out = ctx.render_region((0, 0, w, h), spp)
            ms, _ = ctx.kernel_time(); c = ctx.counters()
            best = min(best, ms)
            hsh = hashlib.sha256()
            for a in out: hsh.update(a.tobytes())
            sha = hsh.hexdigest()[:12]
        print(f"  {scene:44s} {w}x{h}@{spp}: {best:9.2f} ms {c['samples'] / best * 1e-3:9.1f} Msamples/s  sha {sha}  [{ctx.kernel_name()}]", flush=True)
    ctx.close()
    sys.exit(0)
subprocess.run([sys.executable, os.path.join(ROOT, "scenes", "gen_assets.py")], check=True, stdout=subprocess.DEVNULL)
libs = sys.argv[1].split(",")
reps = sys.argv[3] if len(sys.argv) > 3 else "2"
for rnd in range(2):
    for d in libs:
        print(f"{d} (pass {rnd}):", flush=True)
        env = dict(os.environ, QA_HIP_LIB=os.path.join(ROOT, "qaray_amd", d, "libqaray_hip.so"))
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--one", sys.argv[2], reps], env=env)
        if r.returncode != 0:
            sys.exit(r.returncode)    # (a GPU step failed: no further GPU step)
