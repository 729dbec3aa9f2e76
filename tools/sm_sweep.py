"""Sweep the state-machine kernel's scheduling knobs (each config in a fresh process: env vars)."""
import os, subprocess, sys, json
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
subprocess.run([sys.executable, os.path.join(R, "scenes", "gen_assets.py")], check=True, stdout=subprocess.DEVNULL)
def run(scene, spp, env):
    e = dict(os.environ, **{k: str(v) for k, v in env.items()})
    r = subprocess.run([sys.executable, os.path.join(R, "bench.py"), "--scene", scene, "--spp", str(spp), "--steps", "2", "--warmup", "1", "--cpu-spp", "0"],
                       env=e, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
    try:
        return json.loads(r.stdout.strip().splitlines()[-1])["value"]
    except Exception:
        return float("nan")
scenes = [("example_project12_box.xml", 128), ("trc_scene_tower.xml", 8), ("example_project7_object.xml", 8)]
print("lockstep", [round(run(s, spp, {"QA_KERNEL": "lockstep"}), 1) for s, spp in scenes], flush=True)
for g, i, t in [(24, 24, 4), (8, 8, 4), (16, 16, 8), (32, 32, 8), (48, 48, 8), (32, 16, 16), (48, 32, 16), (56, 48, 16), (32, 32, 2), (60, 60, 32), (40, 8, 8)]:
    print((g, i, t), [round(run(s, spp, {"QA_KERNEL": "sm", "QA_SM_GEN": g, "QA_SM_INST": i, "QA_SM_TRAV": t}), 1) for s, spp in scenes], flush=True)
