"""qaray_amd — MI355X-native hot path of the qaray path tracer.

Python is only the harness layer (ctypes over the two C-ABI libraries, torch for device memory
and torch.distributed); the product is the C++ host layer (libqaray_host.so) and the HIP
integrator (libqaray_hip.so) under qaray_amd/csrc.
"""
from .host import HostScene, FrameBuffer, load_scene_blob, HOST_LIB_PATH  # noqa: F401
