import json
import os
import subprocess
import sys

import numpy as np
import pytest

# the staged integrator's tile groups want a hardware queue per stream; the HIP runtime reads this at its first call
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """The product libraries and the oracle are built in-tree by __graft_entry__.build(); build them
    here when a fresh checkout runs the tests directly."""
    from qaray_amd import hip, host
    from oracle import binding as oracle
    if not (os.path.exists(host.HOST_LIB_PATH) and os.path.exists(hip.HIP_LIB_PATH) and os.path.exists(oracle.LIB_PATH)):
        import __graft_entry__
        __graft_entry__.build()


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    meta = json.loads(bytes(z["meta"]).decode())
    return z["rgb"], z["depth"], z["ns"], meta


def photon_golden_names():
    d = os.path.join(GOLDEN, "photon")
    return sorted(f[:-4] for f in os.listdir(d) if f.endswith(".npz"))


def load_photon_golden(name):
    """-> dict with rgb, depth, ns, photon, caustics (balanced qa_photon arrays without the [0] slot), meta."""
    z = np.load(os.path.join(GOLDEN, "photon", name + ".npz"))
    out = {k: z[k] for k in ("rgb", "depth", "ns", "photon", "caustics")}
    out["meta"] = json.loads(bytes(z["meta"]).decode())
    return out


def golden_names():
    return sorted(f[:-4] for f in os.listdir(GOLDEN) if f.endswith(".npz"))


def ensure_assets():
    """Synthetic stand-in meshes / textures (scenes/gen_assets.py) are generated on demand."""
    subprocess.run([sys.executable, os.path.join(ROOT, "scenes", "gen_assets.py")], check=True, stdout=subprocess.DEVNULL)


def golden_blob(meta):
    from qaray_amd.host import load_scene_blob
    if meta["scene"] in ("example_project7_object.xml", "example_project12_caustics_glossy.xml", "trc_scene_tower.xml"):
        ensure_assets()
    return load_scene_blob(meta["scene"], size=(meta["width"], meta["height"]))


def reference_input_names():
    """The reference's 28 inputs/*.xml, kept unchanged under scenes/ (SURVEY.md 8 f1)."""
    d = os.path.join(ROOT, "scenes")
    return sorted(f for f in os.listdir(d) if f.endswith(".xml") and (f.startswith("example_") or f.startswith("trc_")))


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)
