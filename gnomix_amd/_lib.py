"""ctypes binding of libgnomix_hip.so (include/gnomix_hip.h).

There is NO CPU fallback: if the HIP library is missing or cannot be loaded the import of the
compute entry points fails loudly (GnxLibraryError).  Build it with `python __graft_entry__.py`
(or `make -C gnomix_amd/csrc`).
"""
from __future__ import annotations

import atexit
import ctypes as C
import os
import weakref

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("GNX_LIBRARY") or os.path.join(_HERE, "libgnomix_hip.so")   # GNX_LIBRARY: another build of the same ABI (kernel A/B timing, scripts/dev)

GNX_ABI_VERSION = 14
GNX_OK, GNX_EINVAL, GNX_ENOMEM, GNX_EHIP, GNX_EUNSUPPORTED, GNX_ESTATE, GNX_ESTALE = 0, -1, -2, -3, -4, -5, -6
BASE_NONE, BASE_LOGISTIC, BASE_COVRSK_SVC, BASE_FOREST, BASE_RFOREST = 0, 1, 2, 3, 4
SMOOTH_NONE, SMOOTH_XGB, SMOOTH_CRF, SMOOTH_CNN = 0, 1, 2, 3
K_BASE_LOGISTIC, K_SMOOTH_XGB, K_BASE_COVRSK, K_SMOOTH_CRF, K_GNOFIX, K_SMOOTH_ROWS, K_CALIBRATE, K_BASE_FOREST, K_SMOOTH_CNN = range(9)
KERNEL_NAMES = {K_BASE_LOGISTIC: "k_base_logistic", K_SMOOTH_XGB: "k_smooth_xgb", K_BASE_COVRSK: "k_base_covrsk",
                K_SMOOTH_CRF: "k_smooth_crf", K_GNOFIX: "k_gnofix", K_SMOOTH_ROWS: "k_smooth_rows", K_CALIBRATE: "k_calibrate",
                K_BASE_FOREST: "k_base_forest", K_SMOOTH_CNN: "k_smooth_cnn"}


class GnxLibraryError(ImportError):
    pass


class GnxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"gnomix_hip error {code}: {msg}")
        self.code = code
        self.msg = msg


class SvcWindow(C.Structure):
    _fields_ = [("xfit", C.c_void_p), ("n_fit", C.c_int32), ("width", C.c_int32), ("support", C.c_void_p),
                ("n_sv", C.c_int32), ("dual_coef", C.c_void_p), ("intercept", C.c_void_p), ("prob_a", C.c_void_p),
                ("prob_b", C.c_void_p), ("n_support", C.c_void_p), ("ms", C.c_void_p), ("n_ms", C.c_int32),
                ("kernel_kind", C.c_int32), ("poly_p", C.c_double), ("run_value", C.c_void_p)]


class ModelDesc(C.Structure):
    _fields_ = [("abi_version", C.c_int32), ("A", C.c_int32), ("C", C.c_int64), ("M", C.c_int64), ("ctx", C.c_int64),
                ("S", C.c_int32), ("base_kind", C.c_int32), ("smooth_kind", C.c_int32), ("reserved0", C.c_int32),
                ("lr_coef", C.c_void_p), ("lr_ldc", C.c_int64), ("lr_intercept", C.c_void_p),
                ("svc", C.c_void_p),
                ("n_trees", C.c_int32), ("n_nodes", C.c_int32), ("tree_off", C.c_void_p), ("left", C.c_void_p),
                ("right", C.c_void_p), ("feat", C.c_void_p), ("cond", C.c_void_p), ("tree_class", C.c_void_p),
                ("base_score", C.c_float), ("reserved2", C.c_int32),
                ("crf_state", C.c_void_p), ("crf_trans", C.c_void_p),
                ("calib_off", C.c_void_p), ("calib_x", C.c_void_p), ("calib_y", C.c_void_p),
                ("calib_is_f32", C.c_int32), ("reserved3", C.c_int32),
                ("fb_n_trees", C.c_int32), ("fb_missing", C.c_int32), ("fb_win_tree0", C.c_void_p),
                ("fb_tree_off", C.c_void_p), ("fb_left", C.c_void_p), ("fb_right", C.c_void_p), ("fb_feat", C.c_void_p),
                ("fb_cond", C.c_void_p), ("fb_default_left", C.c_void_p), ("fb_tree_class", C.c_void_p),
                ("fb_base_score", C.c_float), ("fb_n_nodes", C.c_int32),
                ("rf_n_trees", C.c_int32), ("rf_n_nodes", C.c_int32), ("rf_win_tree0", C.c_void_p), ("rf_tree_off", C.c_void_p),
                ("rf_left", C.c_void_p), ("rf_right", C.c_void_p), ("rf_feat", C.c_void_p), ("rf_thr", C.c_void_p),
                ("rf_value", C.c_void_p), ("cnn_weight", C.c_void_p), ("cnn_bias", C.c_void_p),
                ("prepared", C.c_void_p), ("prepared_bytes", C.c_int64)]


class TrainInfo(C.Structure):
    _fields_ = [("newton_iterations", C.c_int32), ("cg_iterations", C.c_int32), ("n_problems", C.c_int32), ("reserved", C.c_int32),
                ("worst_rel_gradient", C.c_double), ("objective_sum", C.c_double)]


class GbtParams(C.Structure):
    _fields_ = [("n_rounds", C.c_int32), ("max_depth", C.c_int32), ("max_bin", C.c_int32), ("tree_method", C.c_int32),
                ("eta", C.c_double), ("lam", C.c_double), ("gamma", C.c_double), ("min_child_weight", C.c_double),
                ("base_score", C.c_double)]


class CnnParams(C.Structure):
    _fields_ = [("epochs", C.c_int32), ("batch", C.c_int32), ("lr", C.c_double), ("beta1", C.c_double), ("beta2", C.c_double),
                ("eps", C.c_double), ("log_eps", C.c_double)]


class CrfParams(C.Structure):
    _fields_ = [("c1", C.c_double), ("c2", C.c_double), ("epsilon", C.c_double), ("max_iterations", C.c_int32), ("memory", C.c_int32)]


class CrfInfo(C.Structure):
    _fields_ = [("iterations", C.c_int32), ("evaluations", C.c_int32), ("objective", C.c_double), ("grad_norm", C.c_double),
                ("converged", C.c_int32), ("reserved", C.c_int32)]


class VcfInfo(C.Structure):
    _fields_ = [("n_variants", C.c_int64), ("n_samples", C.c_int64), ("ldg", C.c_int64), ("file_bytes", C.c_int64),
                ("text_bytes", C.c_int64), ("n_fast_lines", C.c_int64), ("n_general_lines", C.c_int64), ("n_overflow", C.c_int64),
                ("seconds_load", C.c_double), ("seconds_parse", C.c_double), ("seconds_alloc", C.c_double), ("seconds_merge", C.c_double),
                ("n_threads", C.c_int32), ("compression", C.c_int32), ("region_fallback", C.c_int32), ("gt2_pinned", C.c_int32)]


class ModelInfo(C.Structure):
    _fields_ = [("C", C.c_int64), ("M", C.c_int64), ("ctx", C.c_int64), ("W", C.c_int64), ("A", C.c_int32),
                ("S", C.c_int32), ("base_kind", C.c_int32), ("smooth_kind", C.c_int32), ("n_trees", C.c_int32),
                ("tree_depth", C.c_int32), ("device_bytes", C.c_int64)]


# every symbol include/gnomix_hip.h declares: (restype, argtypes)
_VP, _I, _I64 = C.c_void_p, C.c_int, C.c_int64
SYMBOLS = {
    "gnx_abi_version": (C.c_int, []),
    "gnx_build_flags": (C.c_int, []),
    "gnx_device_count": (C.c_int, []),
    "gnx_host_alloc": (_I, [_VP, C.c_size_t, C.POINTER(_VP)]),
    "gnx_host_free": (_I, [_VP, _VP]),
    "gnx_host_flags": (_I, [_VP, C.POINTER(C.c_uint)]),
    "gnx_debug_ws_devices": (_I, [_VP, _VP, C.c_int32]),
    "gnx_init": (C.c_int, [C.c_int, C.POINTER(_VP)]),
    "gnx_ctx_free": (None, [_VP]),
    "gnx_last_error": (C.c_char_p, [_VP]),
    "gnx_set_stream": (C.c_int, [_VP, _VP]),
    "gnx_reset_stream": (C.c_int, [_VP]),
    "gnx_synchronize": (C.c_int, [_VP]),
    "gnx_model_load": (C.c_int, [_VP, C.POINTER(ModelDesc), C.POINTER(_VP)]),
    "gnx_model_free": (None, [_VP]),
    "gnx_model_get_info": (C.c_int, [_VP, C.POINTER(ModelInfo)]),
    "gnx_model_set_calibrate": (C.c_int, [_VP, _I]),
    "gnx_model_export_prepared": (C.c_int, [_VP, _VP, _I64, C.POINTER(_I64)]),
    "gnx_base_predict": (C.c_int, [_VP, _VP, _I64, _I64, _VP, _VP]),
    "gnx_base_predict_dev": (C.c_int, [_VP, _VP, _I64, _I64, _VP, _VP]),
    "gnx_smooth_predict": (C.c_int, [_VP, _VP, _I, _I64, _VP, _VP, _VP]),
    "gnx_smooth_predict_dev": (C.c_int, [_VP, _VP, _I, _I64, _VP, _VP, _VP]),
    "gnx_infer": (C.c_int, [_VP, _VP, _I64, _I64, _VP, _VP, _VP]),
    "gnx_infer_dev": (C.c_int, [_VP, _VP, _I64, _I64, _VP, _VP, _VP]),
    "gnx_packed_row_bytes": (_I64, [_I64]),
    "gnx_pack_x": (C.c_int, [_VP, _I64, _I64, _I64, _VP, _I64, _I]),
    "gnx_unpack_x_dev": (C.c_int, [_VP, _VP, _I64, _I64, _I64, _VP, _I64]),
    "gnx_infer_packed": (C.c_int, [_VP, _VP, _I64, _I64, _VP, _VP, _VP]),
    "gnx_infer_packed_dev": (C.c_int, [_VP, _VP, _I64, _I64, _VP, _VP, _VP]),
    "gnx_base_predict_packed_dev": (C.c_int, [_VP, _VP, _I64, _I64, _VP, _VP]),
    "gnx_smooth_rows": (C.c_int, [_VP, _VP, _I64, _VP]),
    "gnx_calibrate_rows": (C.c_int, [_VP, _VP, _I, _I64, _VP]),
    "gnx_gnofix": (C.c_int, [_VP, _VP, _I64, _VP, _I64, C.c_int32, _VP, _VP]),
    "gnx_gnofix_dev": (C.c_int, [_VP, _VP, _I64, _VP, _I64, C.c_int32, _VP, _VP]),
    "gnx_gnofix_packed_dev": (C.c_int, [_VP, _VP, _I64, _VP, _I64, C.c_int32, _VP, _VP]),
    "gnx_train_logistic": (C.c_int, [_VP, _VP, _I64, _I64, _VP, _I64, _I64, _I64, C.c_int32, C.c_double, C.c_double, C.c_int32, _VP, _I64, _VP,
                                     C.POINTER(TrainInfo)]),
    "gnx_train_logistic_dev": (C.c_int, [_VP, _VP, _I64, _I64, _VP, _I64, _I64, _I64, C.c_int32, C.c_double, C.c_double, C.c_int32, _VP, _I64,
                                         _VP, C.POINTER(TrainInfo)]),
    "gnx_fit_isotonic_f32": (C.c_int, [_VP, _VP, _I64, _VP, _VP, _VP]),
    "gnx_train_gbt": (C.c_int, [_VP, _VP, C.c_int32, _VP, _I64, C.c_int32, C.c_int32, C.c_int32, C.POINTER(GbtParams)] + [_VP] * 8),
    "gnx_train_crf": (C.c_int, [_VP, _VP, C.c_int32, _VP, _I64, C.c_int32, C.c_int32, C.POINTER(CrfParams), _VP, _VP, C.POINTER(CrfInfo)]),
    "gnx_train_cnn": (C.c_int, [_VP, _VP, C.c_int32, _VP, _I64, C.c_int32, C.c_int32, C.c_int32, C.POINTER(CnnParams), _VP, _VP, _VP, _VP]),
    "gnx_train_gbt_dev": (C.c_int, [_VP, _VP, C.c_int32, _VP, _I64, C.c_int32, C.c_int32, C.c_int32, C.POINTER(GbtParams)] + [_VP] * 8),
    # include/gnomix_io.h: the file side
    "gnx_io_last_error": (C.c_char_p, []),
    "gnx_vcf_read": (C.c_int, [_VP, C.c_char_p, C.c_char_p, _I, C.POINTER(_VP)]),
    "gnx_vcf_free": (None, [_VP]),
    "gnx_vcf_get_info": (C.c_int, [_VP, C.POINTER(VcfInfo)]),
    "gnx_vcf_gt2": (_VP, [_VP]),
    "gnx_vcf_pos": (_VP, [_VP]),
    "gnx_vcf_qual": (_VP, [_VP]),
    "gnx_vcf_strings": (C.c_int, [_VP, _I, C.POINTER(_VP), C.POINTER(_VP), C.POINTER(_I64)]),
    "gnx_vcf_gt_int8": (C.c_int, [_VP, _VP, _I]),
    "gnx_io_inflate_raw": (C.c_int, [_VP, C.c_size_t, _VP, C.c_size_t]),
    "gnx_io_crc32": (C.c_uint32, [_VP, C.c_size_t]),
    "gnx_gt2_to_x_dev": (C.c_int, [_VP, _VP, _I64, _I64, _I64, _I64, _VP, _I64, _VP, _I64]),
    "gnx_gt2_to_p2_dev": (C.c_int, [_VP, _VP, _I64, _I64, _I64, _I64, _VP, _I64, _VP, _I64]),
    "gnx_x_to_gt2_dev": (C.c_int, [_VP, _VP, _I64, _I64, _I64, _VP, _I64, _VP, _I64]),
    "gnx_infer_gt2": (C.c_int, [_VP, _VP, _I64, _I64, _I64, _VP, _VP, _VP, _VP]),
    "gnx_phase_gt2": (C.c_int, [_VP, _VP, _I64, _I64, _I64, _VP, C.c_int32, _VP, _I64, _VP, _I64, _VP, _VP, _VP, _VP]),
    "gnx_infer_gt2_range": (C.c_int, [_VP, _VP, _I64, _I64, _I64, _I64, _VP, _VP, _VP, _VP]),
    "gnx_phase_gt2_range": (C.c_int, [_VP, _VP, _I64, _I64, _I64, _I64, _VP, C.c_int32, _VP, _I64, _VP, _I64, _VP, _VP, _VP, _VP]),
    "gnx_write_msp": (C.c_int, [C.c_char_p, C.c_char_p, _I64, _VP, _VP, _VP, _I64, _I64, _I64, _I]),
    "gnx_write_fb_dev": (C.c_int, [_VP, C.c_char_p, C.c_char_p, _I64, _VP, _VP, _VP, _I64, _I64, _I64]),
    "gnx_write_fb": (C.c_int, [C.c_char_p, C.c_char_p, _I64, _VP, _VP, _VP, _I, _I64, _I64, _I64, _I]),
    "gnx_write_vcf_gt2": (C.c_int, [C.c_char_p, C.c_char_p, _I64, _VP, _VP, _VP, _I64, _I64, _I64, _I, _I]),
    "gnx_write_phased_vcf": (C.c_int, [C.c_char_p, C.c_char_p, _I64, _VP, _VP, _I64, _VP, _VP, _VP, _VP, _VP, _I64, _I64, _I]),
    "gnx_format_floats": (C.c_int, [_VP, _I, _I64, _VP, _VP]),
    "gnx_profile_enable": (C.c_int, [_VP, _I]),
    "gnx_profile_reset": (C.c_int, [_VP]),
    "gnx_profile_get": (C.c_int, [_VP, _I, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
}

_lib = None


def io_check(rc):
    """return code of a context-free entry point of include/gnomix_io.h -> GnxError with the thread's message"""
    if rc != GNX_OK:
        raise GnxError(rc, load().gnx_io_last_error().decode(errors="replace"))


def load():
    """dlopen libgnomix_hip.so and type every entry point.  Raises GnxLibraryError when absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise GnxLibraryError(
            f"{SO_PATH} not found: the HIP extension is not built (run `python __graft_entry__.py` or "
            f"`make -C gnomix_amd/csrc`).  gnomix_amd has no CPU fallback.")
    # A context drives up to five streams (compute, copy-in, copy-out, a side stream for small grids, the caller's); HIP maps
    # streams onto 4 hardware queues unless told otherwise, and two streams sharing a queue serialise: when they were the copy-in and
    # copy-out streams of the host-pointer pipelines, H2D and D2H stopped overlapping (13.4 k -> 9.2 k individuals/s).
    # Read by the HIP runtime when it initialises: set before anything touches the GPU (gnx_init does the same for C callers).
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    if not os.environ.get("GNX_NO_TORCH"):
        try:
            # torch (when installed) ships its own libamdhip64 with the same SONAME; importing it first makes
            # this library and torch share ONE HIP runtime, so device pointers/streams are interchangeable.
            # The command line never touches torch and sets GNX_NO_TORCH (its import alone costs over a second).
            import torch  # noqa: F401
        except Exception:
            pass
    try:
        lib = C.CDLL(SO_PATH)
    except OSError as e:
        raise GnxLibraryError(f"cannot load {SO_PATH}: {e}") from e
    for name, (res, args) in SYMBOLS.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise GnxLibraryError(f"{SO_PATH} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    if lib.gnx_abi_version() != GNX_ABI_VERSION:
        raise GnxLibraryError("libgnomix_hip.so ABI version mismatch")
    _lib = lib
    return lib


class Context:
    """One gnx_ctx (one per device / per process under torch.distributed)."""

    def __init__(self, device: int = 0):
        self.lib = load()
        h = C.c_void_p()
        rc = self.lib.gnx_init(int(device), C.byref(h))
        self.h = h
        self.device = int(device)
        self._models = weakref.WeakSet()
        _live_contexts.add(self)
        if rc != GNX_OK:
            msg = self.lib.gnx_last_error(h).decode() if h else "gnx_init failed"
            if h:
                self.lib.gnx_ctx_free(h)
            self.h = None
            raise GnxError(rc, msg)

    def check(self, rc):
        if rc != GNX_OK:
            raise GnxError(rc, self.lib.gnx_last_error(self.h).decode())

    def set_stream(self, stream_ptr):
        self.check(self.lib.gnx_set_stream(self.h, C.c_void_p(stream_ptr)))

    def reset_stream(self):
        self.check(self.lib.gnx_reset_stream(self.h))

    def synchronize(self):
        self.check(self.lib.gnx_synchronize(self.h))

    def pinned_empty(self, shape, dtype):
        """numpy array over page-locked host memory (gnx_host_alloc): X / B / outputs kept in such arrays cross PCIe by DMA
        at link rate instead of through the runtime's pageable staging.  Freed when the array (its base) is collected."""
        import numpy as np
        dt = np.dtype(dtype)
        n = int(np.prod(shape)) * dt.itemsize
        p = C.c_void_p()
        self.check(self.lib.gnx_host_alloc(self.h, C.c_size_t(n), C.byref(p)))
        buf = (C.c_char * max(n, 1)).from_address(p.value)
        lib, h, addr = self.lib, self.h, p.value
        weakref.finalize(buf, lambda: lib.gnx_host_free(h, C.c_void_p(addr)))
        arr = np.frombuffer(buf, dtype=dt, count=int(np.prod(shape))).reshape(shape)
        return arr

    def workspace_devices(self):
        """device ordinal of every live device workspace of this context (gnx_debug_ws_devices): all equal to self.device"""
        import numpy as np
        out = np.full(64, -1, np.int32)
        n = self.lib.gnx_debug_ws_devices(self.h, out.ctypes.data, out.size)
        if n < 0:
            self.check(n)
        return [int(v) for v in out[:min(n, out.size)]]

    def profile_enable(self, on=True):
        self.check(self.lib.gnx_profile_enable(self.h, int(bool(on))))

    def profile_reset(self):
        self.check(self.lib.gnx_profile_reset(self.h))

    def profile_get(self, kid):
        ms, n = C.c_double(), C.c_int64()
        self.check(self.lib.gnx_profile_get(self.h, int(kid), C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def close(self):
        for m in list(getattr(self, "_models", ())):
            m.close()
        if self.h:
            self.lib.gnx_ctx_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default_ctx = {}
_live_contexts = weakref.WeakSet()


@atexit.register
def _shutdown():
    # free device state while the HIP runtime is still alive (interpreter teardown order is undefined)
    for ctx in list(_live_contexts):
        try:
            ctx.close()
        except Exception:
            pass
    _default_ctx.clear()


def default_context(device: int = 0) -> Context:
    if device not in _default_ctx:
        _default_ctx[device] = Context(device)
    return _default_ctx[device]
