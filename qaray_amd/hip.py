"""ctypes binding of libqaray_hip.so (include/qaray_hip.h) - the MI355X integrator.

There is no CPU fallback: creating a context without the library or without a GPU raises."""
import ctypes as C
import os

import numpy as np

# (No environment variable is changed here.  The staged integrator's tile groups - Context.set_option("staged_groups", 4) -
# only pay when the PROCESS exported GPU_MAX_HW_QUEUES=8 before the HIP runtime initialised: bench.py and the tools do.)

_HERE = os.path.dirname(os.path.abspath(__file__))
HIP_LIB_PATH = os.environ.get("QA_HIP_LIB") or os.path.join(_HERE, "lib", "libqaray_hip.so")  # QA_HIP_LIB: A/B builds

QA_RENDER_STATS = 1
DEFAULT_SEED = 0x51A7A7


class Counters(C.Structure):
    _fields_ = [("samples", C.c_uint64), ("casts_normal", C.c_uint64), ("casts_shadow", C.c_uint64),
                ("bvh_nodes", C.c_uint64), ("tri_tests", C.c_uint64), ("pixels", C.c_uint64)]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


class PhotonMapParams(C.Structure):   # include/qa_photon.h
    _fields_ = [("size", C.c_uint32), ("bounce", C.c_uint32), ("radius", C.c_float)]


class PhotonParams(C.Structure):
    _fields_ = [("photon", PhotonMapParams), ("caustics", PhotonMapParams)]


# qa_photon: byte-compatible with the reference's cy::PhotonMap::Photon (24 bytes)
PHOTON_DTYPE = np.dtype([("pos", np.float32, 3), ("power", np.float32), ("rgb", np.uint8, 3), ("plane_dirz", np.uint8),
                         ("dirx", np.int16), ("diry", np.int16)])


class HipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libqaray_hip error {code}: {msg}")
        self.code = code


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(HIP_LIB_PATH):
            raise RuntimeError(
                f"{HIP_LIB_PATH} is missing: the HIP extension is the product path and has no fallback - "
                "build it with `python -c 'import __graft_entry__ as g; g.build()'` or `make -C qaray_amd/csrc hip`")
        # PyTorch wheels bundle their own HIP runtime (torch/lib/libamdhip64.so).  Two HIP runtimes in one
        # process do not share the device: whichever initialises second reports "No HIP GPUs are
        # available".  Loading torch first makes libqaray_hip.so bind to the runtime torch already
        # loaded, so device pointers and streams can be exchanged; without torch the system runtime is used.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(HIP_LIB_PATH)
        L.qa_last_error.restype = C.c_char_p
        L.qa_ctx_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
        L.qa_ctx_destroy.argtypes = [C.c_void_p]
        L.qa_scene_upload.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
        L.qa_scene_upload_device.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
        L.qa_render_region.argtypes = [C.c_void_p] + [C.c_int] * 7 + [C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        L.qa_render_region_device.argtypes = [C.c_void_p] + [C.c_int] * 7 + [C.c_uint32, C.c_uint32, C.c_void_p,
                                                                            C.c_void_p, C.c_void_p, C.c_void_p]
        L.qa_render_strips_device.argtypes = [C.c_void_p] + [C.c_int] * 9 + [C.c_uint32, C.c_uint32, C.c_void_p,
                                                                            C.c_void_p, C.c_void_p, C.c_void_p]
        L.qa_strip_count.argtypes = [C.c_int] * 4
        L.qa_synchronize.argtypes = [C.c_void_p]
        L.qa_request_stop.argtypes = [C.c_void_p]
        L.qa_clear_stop.argtypes = [C.c_void_p]
        L.qa_get_counters.argtypes = [C.c_void_p, C.POINTER(Counters)]
        L.qa_reset_counters.argtypes = [C.c_void_p]
        L.qa_get_kernel_time.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
        L.qa_reset_kernel_time.argtypes = [C.c_void_p]
        L.qa_set_launch_config.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.qa_debug_scrub_scratch.argtypes = [C.c_void_p, C.c_uint32]
        L.qa_set_pipeline.argtypes = [C.c_void_p, C.c_int]
        L.qa_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_longlong]
        L.qa_get_kernel_name.argtypes = [C.c_void_p]
        L.qa_get_kernel_name.restype = C.c_char_p
        L.qa_get_staged_stats.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        L.qa_photon_maps_build.argtypes = [C.c_void_p, C.POINTER(PhotonParams), C.c_uint32]
        L.qa_photon_maps_clear.argtypes = [C.c_void_p]
        L.qa_photon_maps_info.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.qa_photon_maps_download.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint64]
        _lib = L
    return _lib


def _check(rc):
    if rc != 0:
        raise HipError(rc, lib().qa_last_error().decode())


def strip_count(y0, y1, first_strip, strip_step):
    return int(lib().qa_strip_count(y0, y1, first_strip, strip_step))


class Context:
    """One integrator context = one GPU (the reference's Renderer instance on one MPI rank)."""

    def __init__(self, device_id=0):
        self._h = C.c_void_p()
        _check(lib().qa_ctx_create(int(device_id), C.byref(self._h)))
        self.device_id = int(device_id)
        self.size = None
        self._photon_sizes = None

    # -- scene ---------------------------------------------------------------------------------
    def upload_scene(self, blob):
        """blob: numpy uint8 array (host) holding a flat scene."""
        blob = np.ascontiguousarray(blob, dtype=np.uint8)
        _check(lib().qa_scene_upload(self._h, blob.ctypes.data, blob.size))
        self._remember_size(blob)
        self._photon_sizes = None   # a new scene drops the photon maps

    def upload_scene_device(self, dev_tensor):
        """dev_tensor: torch uint8 CUDA tensor holding a flat scene (e.g. after a broadcast)."""
        assert dev_tensor.is_cuda and dev_tensor.is_contiguous()
        _check(lib().qa_scene_upload_device(self._h, dev_tensor.data_ptr(), dev_tensor.numel()))
        self._remember_size(dev_tensor[:256].cpu().numpy())
        self._photon_sizes = None   # a new scene drops the photon maps

    def _remember_size(self, blob_head):
        # qa_flat_header: width/height follow magic,version(8) total_bytes(8) 6 vec3 (72) dof (4)
        w, h = np.frombuffer(bytes(bytearray(blob_head[92:100])), dtype=np.uint32)
        self.size = (int(w), int(h))

    # -- photon / caustics maps (the reference's -use-photon-map) ---------------------------------
    def build_photon_maps(self, photon=(10000, 20, 0.2), caustics=(1000, 20, 1.0), seed=DEFAULT_SEED):
        """Trace and balance both maps on the GPU; later renders shade with Scene::usePhotonMap = true.
        photon / caustics: (size, bounce, radius), defaults = RendererParam (src/renderers/renderer.h:51-57)."""
        pp = PhotonParams(PhotonMapParams(*photon), PhotonMapParams(*caustics))
        _check(lib().qa_photon_maps_build(self._h, C.byref(pp), seed))
        self._photon_sizes = (int(photon[0]), int(caustics[0]))

    def clear_photon_maps(self):
        _check(lib().qa_photon_maps_clear(self._h))
        self._photon_sizes = None

    def photon_maps_info(self):
        """-> (emitted[2], emissions[2]): numOfEmittedRays and emission-loop iterations per map."""
        emitted, emissions = (C.c_uint64 * 2)(), (C.c_uint64 * 2)()
        _check(lib().qa_photon_maps_info(self._h, emitted, emissions))
        return list(emitted), list(emissions)

    def download_photon_map(self, which):
        """-> the balanced qa_photon records as they sit in HBM, size + 1 entries ([0] unused)."""
        if self._photon_sizes is None:
            raise HipError(-5, "no photon maps built")
        out = np.zeros(self._photon_sizes[which] + 1, PHOTON_DTYPE)
        _check(lib().qa_photon_maps_download(self._h, which, out.ctypes.data, out.size))
        return out

    # -- rendering -----------------------------------------------------------------------------
    def render_region(self, region, spp, max_bounce=5, seed=DEFAULT_SEED, spp_max=None, stats=False):
        """Synchronous render into host arrays: -> (rgb[h,w,3] f32, depth[h,w] f32, ns[h,w] u32)."""
        x0, y0, x1, y1 = region
        h, w = y1 - y0, x1 - x0
        rgb = np.zeros((h, w, 3), np.float32)
        depth = np.zeros((h, w), np.float32)
        ns = np.zeros((h, w), np.uint32)
        spp_max = spp if spp_max is None else spp_max
        _check(lib().qa_render_region(self._h, x0, y0, x1, y1, spp, spp_max, max_bounce, seed,
                                      QA_RENDER_STATS if stats else 0, rgb.ctypes.data, depth.ctypes.data,
                                      ns.ctypes.data))
        return rgb, depth, ns

    def render_region_device(self, region, spp, rgb, depth, ns, max_bounce=5, seed=DEFAULT_SEED, spp_max=None,
                             stats=False, stream=None):
        """Asynchronous render into torch CUDA tensors (float32 [h,w,3], float32 [h,w], int32/uint32 [h,w]).
        stream: a HIP stream handle (e.g. torch.cuda.Stream().cuda_stream).  None or 0 - which is also the
        handle of torch's DEFAULT stream - means the context's own non-blocking stream: call synchronize()
        before anything else consumes the outputs, or pass a real stream and keep the consumers on it."""
        x0, y0, x1, y1 = region
        n = (x1 - x0) * (y1 - y0)
        assert rgb.is_cuda and rgb.is_contiguous() and rgb.numel() == 3 * n and rgb.element_size() == 4
        assert depth.is_cuda and depth.is_contiguous() and depth.numel() == n and depth.element_size() == 4
        assert ns.is_cuda and ns.is_contiguous() and ns.numel() == n and ns.element_size() == 4
        spp_max = spp if spp_max is None else spp_max
        sptr = self._stream_arg(stream, rgb)
        _check(lib().qa_render_region_device(self._h, x0, y0, x1, y1, spp, spp_max, max_bounce, seed,
                                             QA_RENDER_STATS if stats else 0, rgb.data_ptr(), depth.data_ptr(),
                                             ns.data_ptr(), sptr))

    def render_strips_device(self, region, first_strip, strip_step, spp, rgb, depth, ns, max_bounce=5,
                             seed=DEFAULT_SEED, spp_max=None, stats=False, stream=None):
        """Render strips first_strip, first_strip+strip_step, ... (8 rows each) of `region` into PACKED
        torch CUDA tensors of strip_count(...)*8 rows."""
        x0, y0, x1, y1 = region
        n = strip_count(y0, y1, first_strip, strip_step) * 8 * (x1 - x0)
        assert rgb.is_cuda and rgb.is_contiguous() and rgb.numel() == 3 * n and rgb.element_size() == 4
        assert depth.is_cuda and depth.is_contiguous() and depth.numel() == n and depth.element_size() == 4
        assert ns.is_cuda and ns.is_contiguous() and ns.numel() == n and ns.element_size() == 4
        spp_max = spp if spp_max is None else spp_max
        sptr = self._stream_arg(stream, rgb)
        _check(lib().qa_render_strips_device(self._h, x0, y0, x1, y1, first_strip, strip_step, spp, spp_max,
                                             max_bounce, seed, QA_RENDER_STATS if stats else 0, rgb.data_ptr(),
                                             depth.data_ptr(), ns.data_ptr(), sptr))

    @staticmethod
    def _stream_arg(stream, tensor):
        """No stream given: the render runs on the context's own non-blocking stream, which is not ordered against
        torch's streams - so first wait for whatever torch has queued on the tensor's device (e.g. the zero-fill of
        freshly created output tensors).  The caller still has to synchronize() before reading the outputs."""
        if stream:
            return C.c_void_p(stream)
        import torch
        torch.cuda.current_stream(tensor.device).synchronize()
        return None

    def synchronize(self):
        _check(lib().qa_synchronize(self._h))

    def request_stop(self):
        _check(lib().qa_request_stop(self._h))

    def clear_stop(self):
        _check(lib().qa_clear_stop(self._h))

    # -- measurement ---------------------------------------------------------------------------
    def counters(self):
        c = Counters()
        _check(lib().qa_get_counters(self._h, C.byref(c)))
        return c.as_dict()

    def reset_counters(self):
        _check(lib().qa_reset_counters(self._h))

    def kernel_time(self):
        """-> (total_ms, launches) of the integrator kernel since the last reset (HIP events)."""
        ms, n = C.c_double(), C.c_uint64()
        _check(lib().qa_get_kernel_time(self._h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    PIPELINES = {"mega": 0, "staged": 1, "auto": 2}

    def set_pipeline(self, mode):
        """'mega' | 'staged' | 'auto' (= mega): which integrator renders scenes that do not fit LDS (same bits either way)."""
        _check(lib().qa_set_pipeline(self._h, self.PIPELINES[mode]))

    def set_option(self, name, value):
        """qa_set_option: 'coop', 'cs_cull', 'cs_force_exact', 'walk_zero_terms', 'cs_pool_limit', 'chunk_spp', 'chunk_tail', 'sync_samples', 'tile_order', 'staged_groups', 'verbose'."""
        _check(lib().qa_set_option(self._h, name.encode(), int(value)))

    def kernel_name(self):
        """The integrator the uploaded scene runs on (megakernel variant or the staged pipeline)."""
        return lib().qa_get_kernel_name(self._h).decode()

    STAGED_FIELDS = ("passes", "rays_closest", "rays_shadow", "jobs_queued", "rays_redone", "jobs_done", "node_steps", "leaf_steps",
                     "tri_tests", "order_check_failed", "jobs_suspended", "lane_slots", "wave_rounds")

    def staged_stats(self):
        """Diagnostics of the staged integrator since the last reset_counters()."""
        v = (C.c_uint64 * len(self.STAGED_FIELDS))()
        _check(lib().qa_get_staged_stats(self._h, v))
        d = {k: int(x) for k, x in zip(self.STAGED_FIELDS, v)}
        d["lane_utilisation"] = d["lane_slots"] / (64.0 * d["wave_rounds"]) if d["wave_rounds"] else 0.0
        d["geometry_bytes"] = d["node_steps"] * 64 + d["tri_tests"] * 48
        return d

    def reset_kernel_time(self):
        _check(lib().qa_reset_kernel_time(self._h))

    def scrub_scratch(self, pattern):
        """qa_debug_scrub_scratch: every wave slot's private segment filled with `pattern` (tests: frames must not depend on it)."""
        _check(lib().qa_debug_scrub_scratch(self._h, pattern & 0xFFFFFFFF))

    def set_launch_config(self, blocks_per_cu=0, threads_per_block=0):
        _check(lib().qa_set_launch_config(self._h, blocks_per_cu, threads_per_block))

    def close(self):
        if self._h:
            lib().qa_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
