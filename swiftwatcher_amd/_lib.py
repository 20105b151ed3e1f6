"""ctypes binding of libswk.so (include/swk.h; the A/B switches and counters of include/swk_debug.h).  No torch, no numpy arithmetic: this module
only marshals pointers.  There is no CPU fallback -- if the HIP library is missing or no
gfx950 device is usable, importing is fine but creating a Context raises SwkError."""
import ctypes
import os
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SWK_LIB", os.path.join(_HERE, "libswk.so"))     # SWK_LIB: A/B another build of the same ABI

ABI_VERSION = 2
MEM_HOST, MEM_DEVICE = 0, 1
ERR_STALE = -6
STAGES = ("gray", "rpca", "bilateral", "thresh", "opened", "labels")
ORDER_RASTER, ORDER_BLOCK2X2 = 0, 1
GRAY_Q14, GRAY_Q15 = 0, 1
K_GRAY, K_IALM_STATS, K_IALM_PASS, K_IALM_SMALL, K_FILTER, K_CCL, K_PROPS, K_COPY = range(8)
KERNEL_FAMILIES = ["gray", "ialm_stats", "ialm_pass", "ialm_small", "filter", "ccl", "props", "copy"]

c_u8p = ctypes.POINTER(ctypes.c_uint8)
c_f64p = ctypes.POINTER(ctypes.c_double)
c_i32p = ctypes.POINTER(ctypes.c_int32)


class SwkError(RuntimeError):
    pass


class StaleBatch(SwkError):
    """What a batch left on the device has been overwritten by a later call on the same context."""


class Params(ctypes.Structure):
    _fields_ = [("lmbda", ctypes.c_double), ("tol", ctypes.c_double), ("maxiter", ctypes.c_int32),
                ("bil_d", ctypes.c_int32), ("bil_sigma_color", ctypes.c_double),
                ("bil_sigma_space", ctypes.c_double), ("bil_fma", ctypes.c_int32),
                ("thresh", ctypes.c_int32), ("open_kh", ctypes.c_int32), ("open_kw", ctypes.c_int32),
                ("connectivity", ctypes.c_int32), ("label_order", ctypes.c_int32),
                ("gray_mode", ctypes.c_int32), ("reserved_", ctypes.c_int32)]


class Segment(ctypes.Structure):
    _fields_ = [("label", ctypes.c_int32), ("r0", ctypes.c_int32), ("c0", ctypes.c_int32),
                ("r1", ctypes.c_int32), ("c1", ctypes.c_int32), ("reserved_", ctypes.c_int32),
                ("area", ctypes.c_int64), ("sum_r", ctypes.c_int64), ("sum_c", ctypes.c_int64)]


SEGMENT_DTYPE = np.dtype([("label", "<i4"), ("r0", "<i4"), ("c0", "<i4"), ("r1", "<i4"), ("c1", "<i4"),
                          ("reserved_", "<i4"), ("area", "<i8"), ("sum_r", "<i8"), ("sum_c", "<i8")])
assert SEGMENT_DTYPE.itemsize == ctypes.sizeof(Segment) == 48


class Input(ctypes.Structure):
    _fields_ = [("frames", ctypes.c_void_p), ("mem", ctypes.c_int32), ("channels", ctypes.c_int32),
                ("nwin", ctypes.c_int32), ("n", ctypes.c_int32), ("Hc", ctypes.c_int32), ("Wc", ctypes.c_int32),
                ("x0", ctypes.c_int32), ("y0", ctypes.c_int32),
                ("frame_stride", ctypes.c_int64), ("row_stride", ctypes.c_int64)]


class Output(ctypes.Structure):
    _fields_ = [("mem", ctypes.c_int32), ("seg_cap", ctypes.c_int32),
                ("gray", ctypes.c_void_p), ("rpca", ctypes.c_void_p), ("bilateral", ctypes.c_void_p),
                ("thresh", ctypes.c_void_p), ("opened", ctypes.c_void_p), ("labels", ctypes.c_void_p),
                ("A", ctypes.c_void_p), ("E", ctypes.c_void_p),
                ("iters", ctypes.c_void_p), ("nseg", ctypes.c_void_p), ("segs", ctypes.c_void_p),
                ("planes_on_device", ctypes.c_int32), ("reserved_", ctypes.c_int32)]


_lib = None
_lib_lock = threading.Lock()

_SIGS = {
    "swk_abi_version": (ctypes.c_int32, []),
    "swk_params_default": (None, [ctypes.POINTER(Params)]),
    "swk_ctx_create": (ctypes.c_int32, [ctypes.c_int32] * 5 + [ctypes.POINTER(ctypes.c_void_p)]),
    "swk_ctx_destroy": (None, [ctypes.c_void_p]),
    "swk_last_error": (ctypes.c_char_p, [ctypes.c_void_p]),
    "swk_ctx_device_bytes": (ctypes.c_int64, [ctypes.c_void_p]),
    "swk_batch_run": (ctypes.c_int32, [ctypes.c_void_p, ctypes.POINTER(Input), ctypes.POINTER(Params), ctypes.POINTER(Output)]),
    "swk_bgr2gray": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p]),
    "swk_ialm": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_double, ctypes.c_double, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "swk_rpca_epilogue": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]),
    "swk_bilateral_u8": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_double, ctypes.c_double, ctypes.c_int32, ctypes.c_void_p]),
    "swk_thresh_tozero_u8": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_void_p]),
    "swk_grey_open3x3_u8": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p]),
    "swk_grey_open_u8": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int32] * 5 + [ctypes.c_void_p]),
    "swk_resize_linear_u8": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int32] * 6 + [ctypes.c_void_p]),
    "swk_ccl_u8": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p]),
    "swk_regionprops_u8": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p]),
    "swk_classifier_input": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32]),
    "swk_set_sparse_speculation": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_double]),
    "swk_set_integer_start": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_int32]),
    "swk_last_integer_start_windows": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_void_p]),
    "swk_last_eig_sweeps": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_void_p]),
    "swk_set_norm_speculation": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_double]),
    "swk_set_norm_guard": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_double]),
    "swk_prof_guard_windows": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_void_p]),
    "swk_last_stopping_norms": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32]),
    "swk_set_start_refine": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_double]),
    "swk_prof_refined_windows": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "swk_prof_pass_bytes_per_element": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_void_p]),
    "swk_prof_redo_batches": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_void_p]),
    "swk_prof_redo_windows": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_void_p]),
    "swk_nhwc_bias_relu_place": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int32] * 8 + [ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int32] * 6),
    "swk_set_cnn_tuning": (ctypes.c_int32, [ctypes.c_int32, ctypes.c_int32]),
    "swk_nhwc_conv7x7s2_bias_relu": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p]),
    "swk_nhwc_conv1x1_bias_relu_place": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int32] * 8 + [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p] + [ctypes.c_int32] * 6),
    "swk_nhwc_conv3x3_bias_relu_place": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p] + [ctypes.c_int32] * 6),
    "swk_nhwc_conv3x3_winograd_bias_relu_place": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p] + [ctypes.c_int32] * 6),
    "swk_winograd_f2x2_3x3_weights": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p]),
    "swk_nhwc_maxpool3s2_conv1x1_bias_relu_place": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p] + [ctypes.c_int32] * 5 + [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32]),
    "swk_nhwc_maxpool3s2": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p]),
    "swk_nhwc_head2_relu_mean": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p,
                                                  ctypes.c_void_p, ctypes.c_void_p, ctypes.c_float, ctypes.c_void_p]),
    "swk_segment_inputs": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "swk_segment_inputs_last": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "swk_device_alloc": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_int64, ctypes.POINTER(ctypes.c_void_p)]),
    "swk_device_free": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_void_p]),
    "swk_device_read": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64]),
    "swk_classifier_input_window": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32]),
    "swk_track_costs": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p]),
    "swk_lsap": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p]),
    "swk_median_blur_u8": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p]),
    "swk_otsu_threshold_u8": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]),
    "swk_canny_u8": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p]),
    "swk_dilate_up_u8": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p]),
    "swk_roi_mask": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64]),
    "swk_pinned_alloc": (ctypes.c_int32, [ctypes.c_int32, ctypes.c_int64, ctypes.POINTER(ctypes.c_void_p)]),
    "swk_pinned_free": (ctypes.c_int32, [ctypes.c_void_p]),
    "swk_cut_boxes": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "swk_stage_frames": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int32]),
    "swk_prof_enable": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_int32]),
    "swk_prof_reset": (ctypes.c_int32, [ctypes.c_void_p]),
    "swk_prof_get": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_int32, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int64)]),
    "swk_prof_window_iters": (ctypes.c_int32, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int64)]),
    "swk_set_ialm_variant": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_int32]),
    "swk_set_pass_tuning": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_int32]),
    "swk_set_eig_method": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_int32]),
}
EXPORTS = sorted(_SIGS)


def _preload_torch_hip_runtime():
    """PyTorch-ROCm wheels bundle their own libamdhip64.so (SONAME libamdhip64.so.7, like /opt/rocm's).  If libswk.so
    is loaded first it pulls in the system runtime, torch then loads its bundled copy as a SECOND HIP runtime in the
    process and sees no device (measured: torch.cuda.is_available() turns False).  Loading torch's copy first makes it
    the process's one runtime: libswk's request for libamdhip64.so.7 binds to it by SONAME and a later `import torch`
    finds its own file already mapped -- which is also what happens when torch is imported before this module.
    torch itself is NOT imported here."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)
        except OSError:
            pass


def load():
    """dlopen libswk.so and type its entry points.  Raises SwkError when the library has not
    been built (python swiftwatcher_amd/csrc/build.py) -- never falls back to anything."""
    global _lib
    with _lib_lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise SwkError("libswk.so not built: run `python swiftwatcher_amd/csrc/build.py` "
                               "(there is no CPU fallback)")
            _preload_torch_hip_runtime()
            lib = ctypes.CDLL(LIB_PATH)
            for name, (res, args) in _SIGS.items():
                fn = getattr(lib, name)          # AttributeError = ABI mismatch, let it surface
                fn.restype = res
                fn.argtypes = args
            if lib.swk_abi_version() != ABI_VERSION:
                raise SwkError("libswk.so ABI version mismatch")
            _lib = lib
    return _lib


def default_params(**overrides):
    p = Params()
    load().swk_params_default(ctypes.byref(p))
    for k, v in overrides.items():
        if not hasattr(p, k):
            raise TypeError("unknown swk parameter %r" % k)
        setattr(p, k, v)
    return p


def _ptr(a):
    return ctypes.c_void_p(a.ctypes.data) if a is not None else None


class _SerialisedLib:
    """The library's entry points with a lock held for the duration of every call.  A swk_ctx is not thread-safe
    (include/swk.h), ctypes drops the GIL while a call runs, and the counting loop's producer thread (pipeline.py,
    windows_per_call > 1) segments the next windows while the consumer thread cuts classifier inputs on the same
    context: every call on one Context is therefore serialised here."""

    def __init__(self, lib, lock):
        self._lib, self._lock = lib, lock

    def __getattr__(self, name):
        fn = getattr(self._lib, name)
        lock = self._lock

        def call(*args):
            with lock:
                return fn(*args)
        call.__name__ = name
        setattr(self, name, call)
        return call


class Context:
    """One per process per GPU (swk_ctx).  The C object is not thread-safe; this wrapper serialises calls on it."""

    def __init__(self, device=0, max_windows=0, max_n=0, max_Hc=0, max_Wc=0):
        self._lock = threading.RLock()
        self._lib = _SerialisedLib(load(), self._lock)
        self._h = ctypes.c_void_p()
        rc = self._lib.swk_ctx_create(device, max_windows, max_n, max_Hc, max_Wc, ctypes.byref(self._h))
        if rc != 0:
            msg = self._lib.swk_last_error(None)
            self._h = None
            raise SwkError("swk_ctx_create failed (%d): %s" % (rc, msg.decode() if msg else "?"))
        self.device = device
        self.generation = 0          # batches run on this context: what a batch left on the device is gone once it moves on
        self._plane_pool = {}        # size -> free device buffers of that size (stage images kept on the GPU, see DevicePlanes)

    def close(self):
        if getattr(self, "_h", None):
            for ptrs in self._plane_pool.values():
                for p in ptrs:
                    self._lib.swk_device_free(self._h, ctypes.c_void_p(p))
            self._plane_pool = {}
            self._lib.swk_ctx_destroy(self._h)
            self._h = None

    # ---- device memory the caller keeps (stage images that are only copied to the host when somebody reads them) ----
    def device_alloc(self, nbytes):
        ptr = ctypes.c_void_p()
        self._check(self._lib.swk_device_alloc(self._h, int(nbytes), ctypes.byref(ptr)))
        return ptr.value

    def device_free(self, ptr):
        if getattr(self, "_h", None) and ptr:
            self._check(self._lib.swk_device_free(self._h, ctypes.c_void_p(ptr)))

    def device_read(self, ptr, shape, dtype=np.uint8):
        out = np.empty(shape, dtype)
        self._check(self._lib.swk_device_read(self._h, ctypes.c_void_p(ptr), _ptr(out), out.nbytes))
        return out

    def staging(self, shape):
        """A page-locked uint8 staging array of this shape, kept for the calling thread's next call with the same shape (calls on a
        context are synchronous: the upload out of it has finished when batch_run returns; a thread that segments ahead and the
        loop's own thread never share one)."""
        key = (threading.get_ident(), tuple(shape))
        with self._lock:
            cache = self.__dict__.setdefault("_staging", {})
            arr = cache.get(key)
            if arr is None:
                if len(cache) >= 3:
                    cache.pop(next(iter(cache)))          # (an evicted array lives on while its user holds it)
                arr = cache[key] = pinned_empty(shape, np.uint8, device=self.device)
        return arr

    def take_planes(self, nbytes):
        """A device buffer of nbytes from the context's free list (or a new one): see DevicePlanes."""
        with self._lock:
            free = self._plane_pool.get(nbytes)
            if free:
                return free.pop()
        return self.device_alloc(nbytes)

    def give_planes(self, ptr, nbytes, keep=4):
        if not getattr(self, "_h", None):
            return
        with self._lock:
            free = self._plane_pool.setdefault(nbytes, [])
            if len(free) < keep:
                free.append(ptr)
                return
        self.device_free(ptr)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            msg = self._lib.swk_last_error(self._h)
            raise SwkError("libswk error %d: %s" % (rc, msg.decode() if msg else "?"))

    @property
    def device_bytes(self):
        return int(self._lib.swk_ctx_device_bytes(self._h))

    # ---- profiling hooks ----
    def prof_enable(self, on=True):
        self._check(self._lib.swk_prof_enable(self._h, int(bool(on))))

    def prof_reset(self):
        self._check(self._lib.swk_prof_reset(self._h))

    def prof(self):
        out = {}
        for i, name in enumerate(KERNEL_FAMILIES):
            ms, n = ctypes.c_double(), ctypes.c_int64()
            self._check(self._lib.swk_prof_get(self._h, i, ctypes.byref(ms), ctypes.byref(n)))
            out[name] = (ms.value, n.value)
        wi = ctypes.c_int64()
        self._check(self._lib.swk_prof_window_iters(self._h, ctypes.byref(wi)))
        out["window_iters"] = wi.value
        return out

    def set_ialm_variant(self, variant):
        self._check(self._lib.swk_set_ialm_variant(self._h, int(variant)))

    def set_pass_tuning(self, flags):
        self._check(self._lib.swk_set_pass_tuning(self._h, int(flags)))

    def set_sparse_speculation(self, factor):
        self._check(self._lib.swk_set_sparse_speculation(self._h, float(factor)))

    def set_integer_start(self, on):
        self._check(self._lib.swk_set_integer_start(self._h, int(bool(on))))

    def set_norm_guard(self, rel):
        self._check(self._lib.swk_set_norm_guard(self._h, float(rel)))

    @property
    def guard_windows(self):
        v = ctypes.c_int64(0)
        self._check(self._lib.swk_prof_guard_windows(self._h, ctypes.byref(v)))
        return v.value

    def set_start_refine(self, tau):
        """Estimated first-iteration error above which a window gets the accurate start (csrc/ialm_refine.hip); <= 0: never."""
        self._check(self._lib.swk_set_start_refine(self._h, float(tau)))

    @property
    def refined_windows(self):
        """(windows whose first iteration was refined, windows that wanted it and did not get it) since the context was made."""
        a, b = ctypes.c_int64(0), ctypes.c_int64(0)
        self._check(self._lib.swk_prof_refined_windows(self._h, ctypes.byref(a), ctypes.byref(b)))
        return a.value, b.value

    def last_stopping_norms(self, cap=4096):
        """(ratio ||Z||_F / ||X||_F of the last stopping test, bound on its relative error) per window of the last batch."""
        r, e = np.zeros(cap), np.zeros(cap)
        n = self._lib.swk_last_stopping_norms(self._h, _ptr(r), _ptr(e), cap)
        if n < 0:
            self._check(n)
        return r[:min(n, cap)], e[:min(n, cap)]

    def set_norm_speculation(self, factor):
        self._check(self._lib.swk_set_norm_speculation(self._h, float(factor)))

    @property
    def pass_bytes_per_element(self):
        v = ctypes.c_double(0)
        self._check(self._lib.swk_prof_pass_bytes_per_element(self._h, ctypes.byref(v)))
        return v.value

    @property
    def last_eig_sweeps(self):
        v = ctypes.c_int32(0)
        self._check(self._lib.swk_last_eig_sweeps(self._h, ctypes.byref(v)))
        return v.value

    @property
    def last_integer_start_windows(self):
        v = ctypes.c_int32(0)
        self._check(self._lib.swk_last_integer_start_windows(self._h, ctypes.byref(v)))
        return v.value

    @property
    def redo_batches(self):
        v = ctypes.c_int64(0)
        self._check(self._lib.swk_prof_redo_batches(self._h, ctypes.byref(v)))
        return v.value

    @property
    def redo_windows(self):
        v = ctypes.c_int64(0)
        self._check(self._lib.swk_prof_redo_windows(self._h, ctypes.byref(v)))
        return v.value

    def set_eig_method(self, method):
        self._check(self._lib.swk_set_eig_method(self._h, int(method)))

    # ---- hot path ----
    def batch_run_raw(self, inp, params, out):
        """swk_batch_run.  Returns the batch's generation: the value Context.generation has while what this batch left on the device
        is still there (another thread's batch may already have moved it on by the time the caller looks at the attribute)."""
        with self._lock:
            self.generation += 1
            generation = self.generation
            self._check(self._lib.swk_batch_run(self._h, ctypes.byref(inp), ctypes.byref(params), ctypes.byref(out)))
        return generation

    def batch_run(self, frames, nwin, n, crop=None, params=None, stages=STAGES, want_A=False, want_E=False, seg_cap=255,
                  device_stages=False, reverse_frames=False):
        """Host-buffer convenience wrapper.

        frames: u8 array (nwin*n, H, W, 3) or (nwin*n, H, W), C-contiguous in the last two/three axes
        crop:   (x0, y0, Wc, Hc) inside each frame, or None for the whole frame
        Returns dict with the requested stage stacks (nwin*n, Hc, Wc) u8, 'iters' (nwin,),
        'nseg' (nwin*n,), 'segs' structured array (nwin*n, seg_cap), 'generation' (see batch_run_raw), optionally 'A'/'E' (nwin, P, n).
        device_stages=True: the stage stacks stay on the GPU -- res['planes'] is a DevicePlanes whose read(stage, frame)
        copies one image to the host when somebody asks for it (swk_output.planes_on_device).
        reverse_frames=True: frame j of the batch is frames[F - 1 - j] (negative frame stride): a window that lies in memory in
        the order it was read becomes a queue (position 0 = newest) without being reversed.
        """
        params = params or default_params()
        F = nwin * n
        if frames.dtype != np.uint8 or frames.shape[0] != F or frames.ndim not in (3, 4):
            raise ValueError("frames must be uint8 (nwin*n, H, W[, 3])")
        ch = 1 if frames.ndim == 3 else frames.shape[3]
        if frames.ndim == 4 and (frames.strides[3] != 1 or frames.strides[2] != ch):
            raise ValueError("frames must be contiguous along columns/channels")
        if frames.ndim == 3 and frames.strides[2] != 1:
            raise ValueError("frames must be contiguous along columns")
        H, W = frames.shape[1], frames.shape[2]
        x0, y0, Wc, Hc = crop if crop is not None else (0, 0, W, H)
        if x0 < 0 or y0 < 0 or x0 + Wc > W or y0 + Hc > H:
            raise ValueError("crop rectangle outside the frame")
        inp = Input(frames=frames.ctypes.data + ((F - 1) * frames.strides[0] if reverse_frames else 0), mem=MEM_HOST, channels=ch,
                    nwin=nwin, n=n, Hc=Hc, Wc=Wc, x0=x0, y0=y0,
                    frame_stride=-frames.strides[0] if reverse_frames else frames.strides[0], row_stride=frames.strides[1])
        res = {}
        out = Output(mem=MEM_HOST, seg_cap=seg_cap)
        if device_stages and stages:
            planes = res["planes"] = DevicePlanes(self, tuple(stages), F, Hc, Wc)
            out.planes_on_device = 1
            for name in stages:
                setattr(out, name, planes.pointer(name))
        else:
            for name in stages:
                res[name] = np.empty((F, Hc, Wc), np.uint8)
                setattr(out, name, res[name].ctypes.data)
        P = Hc * Wc
        if want_A:
            res["A"] = np.empty((nwin, P, n), np.float64)
            out.A = res["A"].ctypes.data
        if want_E:
            res["E"] = np.empty((nwin, P, n), np.float64)
            out.E = res["E"].ctypes.data
        res["iters"] = np.zeros(nwin, np.int32)
        res["nseg"] = np.zeros(F, np.int32)
        res["segs"] = np.zeros((F, seg_cap), SEGMENT_DTYPE)
        out.iters = res["iters"].ctypes.data
        out.nseg = res["nseg"].ctypes.data
        out.segs = res["segs"].ctypes.data
        res["generation"] = self.batch_run_raw(inp, params, out)          # for swk_segment_inputs_last (segment_inputs_last)
        return res

    # ---- stage-level entry points ----
    def bgr2gray(self, bgr, gray_mode=GRAY_Q14):
        bgr = np.ascontiguousarray(bgr, np.uint8)
        single = bgr.ndim == 3
        b4 = bgr[None] if single else bgr
        cnt, H, W, _ = b4.shape
        out = np.empty((cnt, H, W), np.uint8)
        self._check(self._lib.swk_bgr2gray(self._h, _ptr(b4), cnt, H, W, gray_mode, _ptr(out)))
        return out[0] if single else out

    def ialm(self, planes, lmbda=0.01, tol=0.001, maxiter=100, want_E=True):
        """planes: u8 (n, P).  Returns A (P, n), E (P, n) or None, iterations."""
        planes = np.ascontiguousarray(planes, np.uint8)
        n, P = planes.shape
        A = np.empty((P, n), np.float64)
        E = np.empty((P, n), np.float64) if want_E else None
        it = np.zeros(1, np.int32)
        self._check(self._lib.swk_ialm(self._h, _ptr(planes), n, P, lmbda, tol, maxiter, _ptr(A), _ptr(E), _ptr(it)))
        return A, E, int(it[0])

    def rpca_epilogue(self, E):
        E = np.ascontiguousarray(E, np.float64)
        out = np.empty(E.shape, np.uint8)
        self._check(self._lib.swk_rpca_epilogue(self._h, _ptr(E), E.size, _ptr(out)))
        return out

    def _planes(self, a):
        a = np.ascontiguousarray(a, np.uint8)
        return (a[None], True) if a.ndim == 2 else (a, False)

    def bilateral_u8(self, src, d=7, sigma_color=15.0, sigma_space=1.0, use_fma=False):
        s, single = self._planes(src)
        out = np.empty_like(s)
        self._check(self._lib.swk_bilateral_u8(self._h, _ptr(s), s.shape[0], s.shape[1], s.shape[2], d,
                                               sigma_color, sigma_space, int(bool(use_fma)), _ptr(out)))
        return out[0] if single else out

    def thresh_tozero_u8(self, src, thresh=15):
        s = np.ascontiguousarray(src, np.uint8)
        out = np.empty_like(s)
        self._check(self._lib.swk_thresh_tozero_u8(self._h, _ptr(s), s.size, thresh, _ptr(out)))
        return out

    def grey_open3x3_u8(self, src):
        s, single = self._planes(src)
        out = np.empty_like(s)
        self._check(self._lib.swk_grey_open3x3_u8(self._h, _ptr(s), s.shape[0], s.shape[1], s.shape[2], _ptr(out)))
        return out[0] if single else out

    def grey_open_u8(self, src, size):
        """scipy.ndimage.grey_opening(size=(kh, kw)) on u8 images, any window."""
        s, single = self._planes(src)
        out = np.empty_like(s)
        self._check(self._lib.swk_grey_open_u8(self._h, _ptr(s), s.shape[0], s.shape[1], s.shape[2], int(size[0]), int(size[1]), _ptr(out)))
        return out[0] if single else out

    def resize_linear_u8(self, frame, dsize):
        """cv2.resize(frame, dsize=(width, height)) (INTER_LINEAR) for one (H, W[, C]) u8 frame."""
        f = np.ascontiguousarray(frame, np.uint8)
        H, W = f.shape[:2]
        ch = 1 if f.ndim == 2 else f.shape[2]
        dW, dH = int(dsize[0]), int(dsize[1])
        out = np.empty((dH, dW) + ((ch,) if f.ndim == 3 else ()), np.uint8)
        self._check(self._lib.swk_resize_linear_u8(self._h, _ptr(f), 1, H, W, ch, dH, dW, _ptr(out)))
        return out

    def ccl_u8(self, src, connectivity=8, label_order=ORDER_BLOCK2X2):
        s, single = self._planes(src)
        lab = np.empty(s.shape, np.int32)
        nc = np.zeros(s.shape[0], np.int32)
        self._check(self._lib.swk_ccl_u8(self._h, _ptr(s), s.shape[0], s.shape[1], s.shape[2], connectivity,
                                         label_order, _ptr(lab), _ptr(nc)))
        return (int(nc[0]), lab[0]) if single else (nc, lab)

    def classifier_input(self, crops, mean, std, want_patches=False, net_ptr=None, pad=100, channels_last=False):
        """swk_classifier_input_window for a list of HxWx3 uint8 crops.  Returns (patches or None, net or None):
        net is a float32 host array (n, 3, S, S), S = 24 + 2 pad, unless net_ptr (a device pointer with room for
        it) is given.  pad = 100 is the reference's full 224x224 input.  channels_last: the network input is written
        [S][S][3] per segment (a torch.channels_last tensor) instead of planes."""
        n = len(crops)
        flat = [np.ascontiguousarray(c, np.uint8).reshape(-1) for c in crops]
        sizes = np.array([f.size for f in flat], np.int64)
        offsets = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int64)
        packed = np.concatenate(flat) if n else np.zeros(0, np.uint8)
        hw = np.array([[c.shape[0], c.shape[1]] for c in crops], np.int32)
        m = np.asarray(mean, np.float32)
        s = np.asarray(std, np.float32)
        patches = np.empty((n, 24, 24, 3), np.uint8) if want_patches else None
        net = None
        if net_ptr is None:
            net = np.empty((n, 3, 24 + 2 * pad, 24 + 2 * pad), np.float32)
            nptr, nmem = _ptr(net), MEM_HOST
        else:
            nptr, nmem = ctypes.c_void_p(net_ptr), MEM_DEVICE
        self._check(self._lib.swk_classifier_input_window(self._h, _ptr(packed), packed.size, _ptr(offsets), _ptr(hw), n,
                                                          _ptr(m), _ptr(s), int(pad), 1 if channels_last else 0, _ptr(patches), nptr,
                                                          nmem))
        return patches, net

    def segment_inputs(self, inp, frame_hw, segs_ptr, nseg_ptr, seg_cap, mean, std, net_ptr, net_cap, first=0, pad=8,
                       min_seg_size=(24, 24), seg_frame_ptr=None, channels_last=False):
        """swk_segment_inputs: classifier inputs cut on the device from the frames (inp, device-resident BGR) and the
        region records of a batch_run with device outputs.  Returns (total segments in the batch, skipped boxes)."""
        m = np.asarray(mean, np.float32)
        s = np.asarray(std, np.float32)
        total = ctypes.c_int32(0)
        skipped = ctypes.c_int32(0)
        self._check(self._lib.swk_segment_inputs(self._h, ctypes.byref(inp), int(frame_hw[0]), int(frame_hw[1]),
                                                 ctypes.c_void_p(segs_ptr), ctypes.c_void_p(nseg_ptr), int(seg_cap),
                                                 int(min_seg_size[0]), int(min_seg_size[1]), _ptr(m), _ptr(s), int(pad),
                                                 1 if channels_last else 0, int(first), int(net_cap), ctypes.c_void_p(net_ptr),
                                                 ctypes.c_void_p(seg_frame_ptr) if seg_frame_ptr else None,
                                                 ctypes.byref(total), ctypes.byref(skipped)))
        return total.value, skipped.value

    def segment_inputs_last(self, generation, mean, std, net_ptr, net_cap, first=0, pad=8, min_seg_size=(24, 24),
                            seg_frame_ptr=None, channels_last=False, known_total=-1):
        """swk_segment_inputs_last: the same for the batch this context ran LAST (its own device copy of the frames, its own
        region records).  generation = Context.generation right after that batch_run: raises StaleBatch when the context
        has run another batch since (the buffers hold something else now).  Returns (total, skipped)."""
        m = np.asarray(mean, np.float32)
        s = np.asarray(std, np.float32)
        total = ctypes.c_int32(int(known_total))
        skipped = ctypes.c_int32(0)
        with self._lock:                 # nothing may slip in between the generation test and the call
            if generation != self.generation:
                raise StaleBatch("the context has run another batch since")
            rc = self._lib.swk_segment_inputs_last(self._h, int(min_seg_size[0]), int(min_seg_size[1]), _ptr(m), _ptr(s), int(pad),
                                                   1 if channels_last else 0, int(first), int(net_cap), ctypes.c_void_p(net_ptr),
                                                   ctypes.c_void_p(seg_frame_ptr) if seg_frame_ptr else None,
                                                   ctypes.byref(total), ctypes.byref(skipped))
            if rc == ERR_STALE:
                raise StaleBatch("the context no longer holds that batch")
            self._check(rc)
        return total.value, skipped.value

    def regionprops_u8(self, labels, seg_cap=255):
        s, single = self._planes(labels)
        segs = np.zeros((s.shape[0], seg_cap), SEGMENT_DTYPE)
        nseg = np.zeros(s.shape[0], np.int32)
        self._check(self._lib.swk_regionprops_u8(self._h, _ptr(s), s.shape[0], s.shape[1], s.shape[2], seg_cap,
                                                 _ptr(segs), _ptr(nseg)))
        return (segs[0, :nseg[0]], int(nseg[0])) if single else (segs, nseg)


def stage_frames(frames, y0, y1, x0, x1, dst, threads=4):
    """dst[i] = frames[i][y0:y1, x0:x1] for a list of equally shaped uint8 frames (H, W[, C]) whose rows are contiguous, by
    swk_stage_frames (threads of the library, GIL released); frames that do not qualify are copied by numpy."""
    f0 = frames[0]
    px = f0.strides[1] if f0.ndim >= 2 else 1
    ok = all(f.dtype == np.uint8 and f.shape == f0.shape and f.strides == f0.strides for f in frames) and \
        f0.ndim in (2, 3) and (f0.ndim == 2 or (f0.strides[2] == 1 and f0.strides[1] == f0.shape[2])) and \
        (f0.ndim == 3 or f0.strides[1] == 1) and f0.strides[0] >= f0.shape[1] * px and dst.flags.c_contiguous
    if not ok:
        for i, f in enumerate(frames):
            dst[i] = f[y0:y1, x0:x1]
        return
    ptrs = (ctypes.c_void_p * len(frames))(*[f.ctypes.data for f in frames])
    rc = load().swk_stage_frames(ptrs, len(frames), f0.strides[0], y0, y1 - y0, x0 * px, (x1 - x0) * px, dst.ctypes.data, threads)
    if rc:
        raise SwkError("swk_stage_frames failed (%d)" % rc)


def cut_boxes(arrays, frame_of, boxes):
    """swk_cut_boxes: boxes (count, 4) int32 rows [r0, r1) x columns [c0, c1) of arrays[frame_of[i]] (equally shaped, row-contiguous
    uint8 arrays), copied densely into ONE new buffer.  Returns (buffer, offsets int64 (count,))."""
    a0 = arrays[0]
    px = a0.strides[1]
    h = np.maximum(boxes[:, 1] - boxes[:, 0], 0).astype(np.int64)
    w = np.maximum(boxes[:, 3] - boxes[:, 2], 0).astype(np.int64)
    sizes = h * w * px
    offsets = np.zeros(len(boxes), np.int64)
    if len(boxes) > 1:
        np.cumsum(sizes[:-1], out=offsets[1:])
    out = np.empty(int(sizes.sum()) if len(boxes) else 0, np.uint8)
    if len(boxes):
        ptrs = (ctypes.c_void_p * len(arrays))(*[a.ctypes.data for a in arrays])
        fo = np.ascontiguousarray(frame_of, np.int32)
        bx = np.ascontiguousarray(boxes, np.int32)
        rc = load().swk_cut_boxes(ptrs, len(arrays), a0.strides[0], px, len(boxes), fo.ctypes.data, bx.ctypes.data, offsets.ctypes.data,
                                  out.ctypes.data)
        if rc:
            raise SwkError("swk_cut_boxes failed (%d)" % rc)
    return out, offsets


class DevicePlanes:
    """The u8 stage images of one batch_run kept on the GPU: one device buffer [stage][frame][Hc][Wc] taken from the
    context's free list and handed back when the last reader is gone.  read(stage, frame) copies ONE image to the
    host (swk_device_read).  The reference stores six images per frame (data_structures.py:183-208) and its counting
    loop reads none of them."""

    def __init__(self, ctx, stages, F, Hc, Wc):
        self.ctx, self.stages, self.F, self.Hc, self.Wc = ctx, stages, F, Hc, Wc
        self.plane = F * Hc * Wc
        self.nbytes = (self.plane + 3) // 4 * 4 * len(stages)          # every stack starts on a dword
        self.ptr = ctx.take_planes(self.nbytes)

    def pointer(self, stage):
        return self.ptr + self.stages.index(stage) * ((self.plane + 3) // 4 * 4)

    def read(self, stage, frame):
        return self.ctx.device_read(self.pointer(stage) + frame * self.Hc * self.Wc, (self.Hc, self.Wc))

    def read_stack(self, stage):
        return self.ctx.device_read(self.pointer(stage), (self.F, self.Hc, self.Wc))

    def __del__(self):
        try:
            if self.ptr:
                self.ctx.give_planes(self.ptr, self.nbytes)
                self.ptr = 0
        except Exception:
            pass


def track_costs(prev_c, prev_hist0, prev_has_hist, curr_c):
    """swk_track_costs: (n_prev + n_curr)^2 float64 cost matrix (host-side, no GPU needed)."""
    n_prev, n_curr = len(prev_c), len(curr_c)
    pc = np.ascontiguousarray(prev_c, np.float64).reshape(n_prev, 2)
    ph = np.ascontiguousarray(prev_hist0, np.float64).reshape(n_prev, 2)
    hh = np.ascontiguousarray(prev_has_hist, np.uint8).reshape(n_prev)
    cc = np.ascontiguousarray(curr_c, np.float64).reshape(n_curr, 2)
    packed = np.concatenate([pc.ravel(), ph.ravel(), cc.ravel()])
    return track_costs_packed(packed, hh.tobytes(), n_prev, n_curr)


def track_costs_packed(packed, has_hist, n_prev, n_curr):
    """The same from ONE float64 array [prev centroids (n_prev, 2) | first-of-history centroids (n_prev, 2) | current centroids
    (n_curr, 2)] and a bytes object of n_prev flags: the per-frame call of the tracker (two array objects per call instead of
    nine -- at a few dozen segments the marshalling, not the arithmetic, is the cost)."""
    n = n_prev + n_curr
    cost = np.empty((n, n), np.float64)
    base = packed.ctypes.data
    rc = _track_costs_fn()(base, base + 16 * n_prev, has_hist, base + 32 * n_prev, n_prev, n_curr, cost.ctypes.data)
    if rc:
        raise SwkError("swk_track_costs failed (%d)" % rc)
    return cost


_fast = {}


def _track_costs_fn():
    fn = _fast.get("costs")
    if fn is None:
        fn = _fast["costs"] = load().swk_track_costs
    return fn


def lsap(cost):
    """swk_lsap: column assigned to every row (same result as scipy.optimize.linear_sum_assignment)."""
    if cost.dtype != np.float64 or not cost.flags.c_contiguous:
        cost = np.ascontiguousarray(cost, np.float64)
    nr, nc = cost.shape
    out = np.empty(nr, np.int32)
    fn = _fast.get("lsap")
    if fn is None:
        fn = _fast["lsap"] = load().swk_lsap
    rc = fn(cost.ctypes.data, nr, nc, out.ctypes.data)
    if rc:
        raise SwkError("swk_lsap failed (%d)" % rc)
    return out


def pinned_empty(shape, dtype=np.uint8, device=0):
    """numpy array in page-locked host memory (swk_pinned_alloc on `device`): staging buffer for host -> device input,
    freed when the array (and every view of it) is gone.  Falls back to ordinary memory when pinning fails: pinning is
    an optimisation, never a requirement."""
    import weakref
    count = int(np.prod(shape))
    nbytes = max(count * np.dtype(dtype).itemsize, 1)
    lib = load()
    ptr = ctypes.c_void_p()
    if lib.swk_pinned_alloc(int(device), nbytes, ctypes.byref(ptr)) != 0 or not ptr.value:
        return np.empty(shape, dtype)
    buf = (ctypes.c_uint8 * nbytes).from_address(ptr.value)      # numpy keeps this object alive as .base
    weakref.finalize(buf, lib.swk_pinned_free, ctypes.c_void_p(ptr.value))
    return np.frombuffer(buf, dtype=dtype, count=count).reshape(shape)


# ---- ROI mask of a video (host side, no GPU): image_filtering.py:99-180 ----
def _host_call(name, *args):
    rc = getattr(load(), name)(*args)
    if rc:
        raise SwkError("%s failed (%d)" % (name, rc))


def median_blur_u8(image, ksize):
    a = np.ascontiguousarray(image, np.uint8)
    ch = 1 if a.ndim == 2 else a.shape[2]
    out = np.empty_like(a)
    _host_call("swk_median_blur_u8", _ptr(a), a.shape[0], a.shape[1], ch, int(ksize), _ptr(out))
    return out


def otsu_threshold_u8(image):
    """(threshold, binary image): cv2.threshold(image, 0, 255, THRESH_BINARY + THRESH_OTSU)."""
    a = np.ascontiguousarray(image, np.uint8)
    out = np.empty_like(a)
    t = ctypes.c_int32(0)
    _host_call("swk_otsu_threshold_u8", _ptr(a), a.size, _ptr(out), ctypes.byref(t))
    return t.value, out


def canny_u8(image, low, high):
    a = np.ascontiguousarray(image, np.uint8)
    out = np.empty_like(a)
    _host_call("swk_canny_u8", _ptr(a), a.shape[0], a.shape[1], int(low), int(high), _ptr(out))
    return out


def dilate_up_u8(image, N):
    a = np.ascontiguousarray(image, np.uint8)
    out = np.empty_like(a)
    _host_call("swk_dilate_up_u8", _ptr(a), a.shape[0], a.shape[1], int(N), _ptr(out))
    return out


def roi_mask(frame, corners):
    """swk_roi_mask: (crop_region [(x0, y0), (x1, y1)], mask uint8 (Hc, Wc)) from the first BGR frame and the two
    chimney corners ((x1, y1), (x2, y2))."""
    if hasattr(frame, "as_full_frame"):          # a ROI-stream frame (io_roi_stream.RoiFrame): the stored rectangle pasted into a blank frame
        frame = frame.as_full_frame()
    if frame.dtype != np.uint8 or frame.ndim != 3 or frame.shape[2] != 3 or frame.strides[2] != 1 or frame.strides[1] != 3:
        frame = np.ascontiguousarray(frame, np.uint8)
    c = np.array([corners[0][0], corners[0][1], corners[1][0], corners[1][1]], np.int32)
    crop = np.zeros(4, np.int32)
    left, right = min(c[0], c[2]), max(c[0], c[2])
    width = int(right - left)
    hc, wc = int(0.5 * width) + int(0.125 * width), width + 2 * int(0.125 * width)
    mask = np.empty((max(hc, 1), max(wc, 1)), np.uint8)
    _host_call("swk_roi_mask", _ptr(frame), frame.shape[0], frame.shape[1], frame.strides[0], _ptr(c), _ptr(crop), _ptr(mask), mask.size)
    return [(int(crop[0]), int(crop[1])), (int(crop[2]), int(crop[3]))], mask


_default_ctx = {}
_ctx_lock = threading.Lock()


def default_context(device=0):
    """Process-wide context used by the image_filtering.* drop-in functions."""
    with _ctx_lock:
        ctx = _default_ctx.get(device)
        if ctx is None:
            ctx = _default_ctx[device] = Context(device)
    return ctx
