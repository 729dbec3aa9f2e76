"""ctypes binding of libqaray_host.so (include/qaray_host.h): XML scene loading + flattening,
the FrameBuffer sink and the tasking stop flag.  No GPU needed."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
HOST_LIB_PATH = os.path.join(_HERE, "lib", "libqaray_host.so")
REPO_ROOT = os.path.dirname(_HERE)
SCENES_DIR = os.path.join(REPO_ROOT, "scenes")

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(HOST_LIB_PATH):
            raise RuntimeError(
                f"{HOST_LIB_PATH} is missing - run `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C qaray_amd/csrc host`")
        L = C.CDLL(HOST_LIB_PATH)
        L.qa_host_last_error.restype = C.c_char_p
        L.qa_host_scene_load.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_void_p)]
        L.qa_host_scene_destroy.argtypes = [C.c_void_p]
        L.qa_host_scene_destroy.restype = None
        L.qa_host_scene_set_size.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.qa_host_scene_get_size.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.qa_host_scene_flatten.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
        L.qa_host_free.argtypes = [C.c_void_p]
        L.qa_host_free.restype = None
        L.qa_fb_create.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        L.qa_fb_destroy.argtypes = [C.c_void_p]
        L.qa_fb_destroy.restype = None
        L.qa_fb_deposit.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_int, C.c_int]
        for name, rt in (("qa_fb_pixels", C.POINTER(C.c_uint8)), ("qa_fb_zbuffer", C.POINTER(C.c_float)),
                         ("qa_fb_sample_count", C.POINTER(C.c_uint8)), ("qa_fb_mask", C.POINTER(C.c_uint8)),
                         ("qa_fb_z_image", C.POINTER(C.c_uint8)), ("qa_fb_sample_count_image", C.POINTER(C.c_uint8))):
            getattr(L, name).argtypes = [C.c_void_p]
            getattr(L, name).restype = rt
        L.qa_fb_num_rendered_pixels.argtypes = [C.c_void_p]
        L.qa_fb_place_strips.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        L.qa_strip_row_range.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        for name in ("qa_fb_save_image", "qa_fb_save_z_image", "qa_fb_save_sample_count_image"):
            getattr(L, name).argtypes = [C.c_void_p, C.c_char_p]
        L.qa_tasking_init.restype = None
        L.qa_tasking_get_num_of_threads.restype = C.c_uint64
        L.qa_tasking_set_num_of_threads.argtypes = [C.c_uint64]
        L.qa_tasking_set_num_of_threads.restype = None
        L.qa_tasking_parallel_for.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p]
        L.qa_tasking_signal_start.restype = None
        L.qa_tasking_signal_stop.restype = None
        _lib = L
    return _lib


class HostError(RuntimeError):
    pass


def _check(rc):
    if rc != 0:
        raise HostError(f"libqaray_host error {rc}: {lib().qa_host_last_error().decode()}")


class HostScene:
    """A loaded XML scene (qaray's LoadScene) that can be flattened for the HIP layer."""

    def __init__(self, xml_path, asset_root=None, size=None):
        self._h = C.c_void_p()
        if asset_root is None:
            asset_root = os.path.dirname(os.path.abspath(xml_path))
        _check(lib().qa_host_scene_load(os.fsencode(xml_path), os.fsencode(asset_root), C.byref(self._h)))
        if size is not None:
            self.set_size(*size)

    def set_size(self, width, height):
        _check(lib().qa_host_scene_set_size(self._h, int(width), int(height)))

    @property
    def size(self):
        w, h = C.c_int(), C.c_int()
        _check(lib().qa_host_scene_get_size(self._h, C.byref(w), C.byref(h)))
        return w.value, h.value

    def flatten(self):
        """-> numpy uint8 array holding the relocatable scene blob (include/qa_flat_scene.h)."""
        p, n = C.c_void_p(), C.c_uint64()
        _check(lib().qa_host_scene_flatten(self._h, C.byref(p), C.byref(n)))
        try:
            return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(n.value,)).copy()
        finally:
            lib().qa_host_free(p)

    def close(self):
        if self._h:
            lib().qa_host_scene_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def load_scene_blob(xml_name_or_path, size=None, asset_root=None):
    """Convenience: scene file (absolute path, or a name under <repo>/scenes) -> flat blob."""
    path = xml_name_or_path if os.path.isabs(xml_name_or_path) else os.path.join(SCENES_DIR, xml_name_or_path)
    s = HostScene(path, asset_root=asset_root, size=size)
    try:
        return s.flatten()
    finally:
        s.close()


class FrameBuffer:
    """qaray's renderImage: 8-bit RGB, float z, sample-count byte, mask, PNG dumps."""

    def __init__(self, width, height):
        self.width, self.height = int(width), int(height)
        self._h = C.c_void_p()
        _check(lib().qa_fb_create(self.width, self.height, C.byref(self._h)))

    def deposit(self, x0, y0, x1, y1, rgb, depth, nsamples, spp_max, use_srgb=True):
        rgb = np.ascontiguousarray(rgb, dtype=np.float32)
        depth = np.ascontiguousarray(depth, dtype=np.float32)
        nsamples = np.ascontiguousarray(nsamples, dtype=np.uint32)
        n = (x1 - x0) * (y1 - y0)
        assert rgb.size == 3 * n and depth.size == n and nsamples.size == n
        _check(lib().qa_fb_deposit(self._h, x0, y0, x1, y1, rgb.ctypes.data, depth.ctypes.data,
                                   nsamples.ctypes.data, int(spp_max), int(bool(use_srgb))))

    def place_strips(self, world, rank, rgb, depth, nsamples, spp_max, use_srgb=True):
        """Deposit rank `rank`'s PACKED strips (8-row strips rank, rank + world, ...) into the rows they belong to -> strips placed."""
        rgb = np.ascontiguousarray(rgb, dtype=np.float32)
        depth = np.ascontiguousarray(depth, dtype=np.float32)
        nsamples = np.ascontiguousarray(nsamples, dtype=np.uint32)
        n = lib().qa_fb_place_strips(self._h, int(world), int(rank), rgb.ctypes.data, depth.ctypes.data, nsamples.ctypes.data,
                                     int(spp_max), int(bool(use_srgb)))
        if n < 0:
            raise HostError(n, "qa_fb_place_strips: bad arguments")
        return n

    def _arr(self, fn, shape, dtype):
        p = fn(self._h)
        return np.ctypeslib.as_array(p, shape=shape).astype(dtype, copy=True)

    @property
    def pixels(self):
        return self._arr(lib().qa_fb_pixels, (self.height, self.width, 3), np.uint8)

    @property
    def zbuffer(self):
        return self._arr(lib().qa_fb_zbuffer, (self.height, self.width), np.float32)

    @property
    def sample_count(self):
        return self._arr(lib().qa_fb_sample_count, (self.height, self.width), np.uint8)

    @property
    def z_image(self):
        """FrameBuffer::ComputeZBufferImage's 8-bit visualisation (src/fb/framebuffer.cpp:62-84)."""
        return self._arr(lib().qa_fb_z_image, (self.height, self.width), np.uint8)

    @property
    def sample_count_image(self):
        """FrameBuffer::ComputeSampleCountImage (src/fb/framebuffer.cpp:86-107)."""
        return self._arr(lib().qa_fb_sample_count_image, (self.height, self.width), np.uint8)

    @property
    def mask(self):
        return self._arr(lib().qa_fb_mask, (self.height, self.width), np.uint8)

    @property
    def num_rendered_pixels(self):
        return lib().qa_fb_num_rendered_pixels(self._h)

    def save_image(self, path):
        _check(lib().qa_fb_save_image(self._h, os.fsencode(path)))

    def save_z_image(self, path):
        _check(lib().qa_fb_save_z_image(self._h, os.fsencode(path)))

    def save_sample_count_image(self, path):
        _check(lib().qa_fb_save_sample_count_image(self._h, os.fsencode(path)))

    def close(self):
        if self._h:
            lib().qa_fb_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
