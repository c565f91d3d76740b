"""ctypes binding of libdsp_hip.so (C ABI: include/dsp_hip.h) + the solver object the Bidder / Tracker call.

``HipPdlpSolver().solve(model, tee=False)`` is the drop-in for the reference's ``pyo.SolverFactory(...).solve``
(SURVEY.md 8(b)-4): it uploads the flattened LP once per model (``dsp_create``), then per call only the dense
per-scenario vectors, runs the fused HIP PDLP kernel and writes x / y / objective / status back into the model.
PyTorch is used only as the device-memory container and stream provider.  There is NO CPU fallback: a missing
library or GPU raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# DSP_LIB selects another build of the library in the package directory (development: e.g. a -DDSP_NO_DRAIN build)
_LIB_PATH = os.path.join(_HERE, os.environ.get("DSP_LIB", "libdsp_hip.so"))
_lib = None


class DspOptions(C.Structure):
    _fields_ = [("eps_rel", C.c_double), ("eps_obj", C.c_double), ("max_iter", C.c_int32), ("check_every", C.c_int32),
                ("restart_sufficient", C.c_double), ("restart_necessary", C.c_double),
                ("restart_artificial", C.c_double), ("pid_kp", C.c_double), ("max_dlog_weight", C.c_double),
                ("step_scale", C.c_double), ("weight_guard", C.c_double), ("jump_steady", C.c_double), ("jump_tol", C.c_double),
                ("jump_min", C.c_double), ("ray_jumps", C.c_int32), ("ruiz_iters", C.c_int32),
                ("waves_per_block", C.c_int32), ("kkt_every", C.c_int32), ("no_matreg", C.c_int32), ("geo_iters", C.c_int32),
                ("kkt_gate", C.c_double), ("stall_rescue", C.c_int32), ("no_simplex", C.c_int32),
                ("jump_rel", C.c_double), ("precision", C.c_int32), ("polish_patience", C.c_int32), ("no_rtc", C.c_int32), ("no_interior_point", C.c_int32),
                ("eps_infeasible", C.c_double), ("recertify_passes", C.c_int32), ("simplex_warm", C.c_int32), ("warm_patience", C.c_int32)]


class DspBatch(C.Structure):
    """dsp_batch of include/dsp_hip.h: device pointers + element strides of one call's B scenarios."""
    _fields_ = [("B", C.c_int32), ("reserved", C.c_int32),
                ("c", C.c_void_p), ("c_stride", C.c_int64),
                ("var_lb", C.c_void_p), ("var_lb_stride", C.c_int64),
                ("var_ub", C.c_void_p), ("var_ub_stride", C.c_int64),
                ("row_lb", C.c_void_p), ("row_lb_stride", C.c_int64),
                ("row_ub", C.c_void_p), ("row_ub_stride", C.c_int64),
                ("obj_offset", C.c_void_p), ("obj_offset_stride", C.c_int64),
                ("row_compliance", C.c_void_p), ("row_compliance_stride", C.c_int64),
                ("x0", C.c_void_p), ("y0", C.c_void_p), ("primal_weight", C.c_void_p),
                ("x", C.c_void_p), ("y", C.c_void_p), ("obj", C.c_void_p),
                ("status", C.c_void_p), ("iters", C.c_void_p), ("jumps", C.c_void_p), ("flags", C.c_void_p)]


class DspStats(C.Structure):
    _fields_ = [("total_iterations", C.c_int64), ("max_iterations", C.c_int32), ("n_optimal", C.c_int32),
                ("grid_blocks", C.c_int32), ("block_threads", C.c_int32), ("lds_bytes", C.c_int32),
                ("cols_per_lane", C.c_int32), ("rows_per_lane", C.c_int32), ("kernel_ms", C.c_float),
                ("matreg", C.c_int32), ("lds_conflicts_identity", C.c_int32), ("lds_conflicts_chosen", C.c_int32),
                ("simplex", C.c_int32), ("streaming", C.c_int32), ("stream_bytes_per_iteration", C.c_int64),
                ("quadratic", C.c_int32), ("precision", C.c_int32), ("rtc", C.c_int32), ("stream_form", C.c_int32), ("stream_phases", C.c_int32),
                ("ipm_solved", C.c_int32), ("reserved1", C.c_int32)]


class DspLpDesc(C.Structure):
    _fields_ = [("n", C.c_int32), ("m", C.c_int32), ("nnz", C.c_int64),
                ("A_rowptr", C.POINTER(C.c_int32)), ("A_colidx", C.POINTER(C.c_int32)),
                ("A_val", C.POINTER(C.c_double)), ("col_scale", C.POINTER(C.c_double))]


class DspWbModel(C.Structure):
    """dsp_wb_model of include/dsp_hip.h: one hourly LP of the wind + battery double loop on the device."""
    _fields_ = [("c", C.c_void_p), ("lb", C.c_void_p), ("ub", C.c_void_p), ("rlo", C.c_void_p), ("rhi", C.c_void_p),
                ("base_c", C.c_void_p), ("x", C.c_void_p),
                ("n", C.c_int32), ("m", C.c_int32), ("T", C.c_int32),
                ("soc_init", C.c_int32), ("thr_init", C.c_int32), ("soc0", C.c_int32), ("thr0", C.c_int32),
                ("wind_cols", C.c_int32 * 8), ("pt_cols", (C.c_int32 * 2) * 8), ("pda_cols", C.c_int32 * 8),
                ("track_rows", C.c_int32 * 8), ("wind_kw", C.c_double),
                ("c0", C.c_void_p), ("c0_base", C.c_double), ("waste_per_kw", C.c_double),
                ("status", C.c_void_p), ("flags", C.c_void_p)]


class DspWbState(C.Structure):
    """dsp_wb_state of include/dsp_hip.h: series, clock, realised state and accumulators of B plants."""
    _fields_ = [("B", C.c_int32), ("N", C.c_int32), ("start", C.c_void_p), ("hour", C.c_void_p),
                ("da_series", C.c_void_p), ("rt_series", C.c_void_p), ("cf_series", C.c_void_p),
                ("soc", C.c_void_p), ("thr", C.c_void_p), ("da_offer", C.c_void_p), ("da_prices", C.c_void_p),
                ("delivered", C.c_void_p), ("revenue", C.c_void_p), ("energy_mwh", C.c_void_p),
                ("bad", C.c_void_p), ("uncertified", C.c_void_p)]


class DspLoopModel(C.Structure):
    """dsp_loop_model of include/dsp_hip.h: one LP of a flowsheet's rolling loop, by descriptor."""
    _fields_ = [("c", C.c_void_p), ("lb", C.c_void_p), ("ub", C.c_void_p), ("rlo", C.c_void_p), ("rhi", C.c_void_p),
                ("base_c", C.c_void_p), ("x", C.c_void_p), ("c0", C.c_void_p),
                ("n", C.c_int32), ("m", C.c_int32), ("T", C.c_int32), ("n_state", C.c_int32),
                ("pt_cols", (C.c_int32 * 2) * 16), ("pt_coef", (C.c_double * 2) * 16), ("pt_const", C.c_double * 16),
                ("pda_cols", C.c_int32 * 16), ("track_rows", C.c_int32 * 16), ("wind_cols", C.c_int32 * 16),
                ("state_init", C.c_int32 * 2), ("state_real", C.c_int32 * 2),
                ("wind_kw", C.c_double), ("c0_base", C.c_double), ("waste_per_kw", C.c_double),
                ("status", C.c_void_p), ("flags", C.c_void_p)]


class DspLoopState(C.Structure):
    """dsp_loop_state of include/dsp_hip.h."""
    _fields_ = [("B", C.c_int32), ("N", C.c_int32), ("start", C.c_void_p), ("hour", C.c_void_p),
                ("da_series", C.c_void_p), ("rt_series", C.c_void_p), ("cf_series", C.c_void_p),
                ("state", C.c_void_p), ("state_scale", C.c_double * 2), ("da_offer", C.c_void_p), ("da_prices", C.c_void_p),
                ("delivered", C.c_void_p), ("revenue", C.c_void_p), ("energy_mwh", C.c_void_p),
                ("bad", C.c_void_p), ("uncertified", C.c_void_p)]


EXPORTED_SYMBOLS = ("dsp_default_options", "dsp_create", "dsp_solve", "dsp_spmv_step", "dsp_get_dims",
                    "dsp_get_scaling", "dsp_destroy", "dsp_strerror", "dsp_last_hip_error", "dsp_version",
                    "dsp_rtc_compile_check", "dsp_rtc_message", "dsp_wb_rolling_update", "dsp_loop_update", "dsp_bid_points", "dsp_source_hash")


ABI_VERSION = 12         # DSP_VERSION of the include/dsp_hip.h these structures mirror


BID_MAX_HOURS, BID_MAX_SCENARIOS = 64, 16384


class DspBidRequest(C.Structure):
    """include/dsp_hip.h: dsp_bid_request (ABI 10)"""
    _fields_ = [("B", C.c_int32), ("T", C.c_int32), ("ldx", C.c_int32), ("ldp", C.c_int32), ("terms", C.c_int32), ("reserved", C.c_int32),
                ("x", C.c_void_p), ("price", C.c_void_p), ("ok", C.c_void_p), ("out", C.c_void_p), ("p_min", C.c_double),
                ("col", (C.c_int32 * 2) * BID_MAX_HOURS), ("val", (C.c_double * 2) * BID_MAX_HOURS), ("constant", C.c_double * BID_MAX_HOURS)]


def source_hash(root: Optional[str] = None) -> Optional[str]:
    """What dsp_source_hash() of a library built from the sources in this tree returns: the first 16 hex digits of the SHA-256 over
    csrc/*.hip, csrc/*.hpp and include/dsp_hip.h (name order, each preceded by its base name).  None when the sources are not
    next to the package (an installed binary: nothing to compare with)."""
    import glob
    import hashlib
    root = root or os.path.dirname(_HERE)
    csrc = os.path.join(root, "dispatches_amd", "csrc")
    files = sorted(glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.hpp")), key=os.path.basename)
    hdr = os.path.join(root, "include", "dsp_hip.h")
    if not files or not os.path.exists(hdr):
        return None
    h = hashlib.sha256()
    for f in files + [hdr]:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def load_library(path: Optional[str] = None):
    """Load libdsp_hip.so (after torch, so both share one HIP runtime).  Raises if it has not been built."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    path = path or _LIB_PATH
    if not os.path.exists(path):
        raise RuntimeError(f"{path} not found: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()')")
    try:
        import torch  # noqa: F401  (loads libamdhip64 first)
    except Exception:
        pass
    lib = C.CDLL(path)
    # version and source checks FIRST: an older library lacks newer entry points, and touching one of those below would raise a bare
    # AttributeError instead of the instruction to rebuild
    lib.dsp_version.restype = C.c_int
    if not hasattr(lib, "dsp_source_hash"):
        raise RuntimeError(f"{path} predates ABI 9 (no dsp_source_hash): rebuild (python -c 'import __graft_entry__ as g; g.build(force=True)')")
    lib.dsp_source_hash.restype = C.c_char_p
    if lib.dsp_version() != ABI_VERSION:
        # the ctypes structures below mirror ONE header version; a stale library would read them with another layout
        raise RuntimeError(f"{path} has ABI version {lib.dsp_version()}, this binding mirrors include/dsp_hip.h version "
                           f"{ABI_VERSION}: rebuild (python -c 'import __graft_entry__ as g; g.build(force=True)')")
    # a binary that travelled with the tree (gpurun ships built .so files) must be the build of THESE sources: the driver's bench once
    # timed a prebuilt library that no hash tied to the code next to it.  DSP_LIB builds (development variants) are exempt.
    want, have = source_hash(), lib.dsp_source_hash().decode()
    if want is not None and have != want and "DSP_LIB" not in os.environ and os.environ.get("DSP_ALLOW_STALE_LIB") != "1":
        raise RuntimeError(f"{path} was built from other sources (library {have}, tree {want}): rebuild "
                           f"(python -c 'import __graft_entry__ as g; g.build()')")
    missing = [name for name in EXPORTED_SYMBOLS if not hasattr(lib, name)]
    if missing:
        raise RuntimeError(f"{path} lacks {missing}: rebuild (python -c 'import __graft_entry__ as g; g.build(force=True)')")
    vp, i32, i64, dp = C.c_void_p, C.c_int32, C.c_int64, C.c_void_p
    lib.dsp_default_options.argtypes = [C.POINTER(DspOptions)]
    lib.dsp_default_options.restype = None
    lib.dsp_create.argtypes = [C.POINTER(DspLpDesc), C.c_int, C.POINTER(DspOptions), C.POINTER(vp)]
    lib.dsp_create.restype = C.c_int
    lib.dsp_solve.argtypes = [vp, C.POINTER(DspBatch), C.POINTER(DspOptions), C.POINTER(DspStats), C.c_int, vp]
    lib.dsp_solve.restype = C.c_int
    lib.dsp_spmv_step.argtypes = [vp, i32, dp, dp, dp, dp, vp]
    lib.dsp_spmv_step.restype = C.c_int
    lib.dsp_get_dims.argtypes = [vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(i64)]
    lib.dsp_get_dims.restype = C.c_int
    lib.dsp_get_scaling.argtypes = [vp, dp, dp, C.POINTER(C.c_double)]
    lib.dsp_get_scaling.restype = C.c_int
    lib.dsp_destroy.argtypes = [vp]
    lib.dsp_destroy.restype = C.c_int
    lib.dsp_strerror.argtypes = [C.c_int]
    lib.dsp_strerror.restype = C.c_char_p
    lib.dsp_rtc_compile_check.argtypes = [C.c_int, C.c_int, C.c_int, C.c_uint, C.c_uint, C.c_int, C.c_char_p, C.c_int]
    lib.dsp_rtc_compile_check.restype = C.c_int
    lib.dsp_rtc_message.argtypes = [vp]
    lib.dsp_rtc_message.restype = C.c_char_p
    lib.dsp_wb_rolling_update.argtypes = [C.POINTER(DspWbState), C.POINTER(DspWbModel), C.POINTER(DspWbModel), i32, i32, vp]
    lib.dsp_wb_rolling_update.restype = C.c_int
    lib.dsp_loop_update.argtypes = [C.POINTER(DspLoopState), C.POINTER(DspLoopModel), C.POINTER(DspLoopModel), i32, i32, vp]
    lib.dsp_loop_update.restype = C.c_int
    lib.dsp_bid_points.argtypes = [C.POINTER(DspBidRequest), vp]
    lib.dsp_bid_points.restype = C.c_int
    lib.dsp_last_hip_error.restype = C.c_int
    if path == _LIB_PATH:
        _lib = lib
    return lib


class DspError(RuntimeError):
    pass


def _check(lib, rc, what):
    if rc != 0:
        raise DspError(f"{what} failed: {lib.dsp_strerror(rc).decode()} (code {rc}, hip error {lib.dsp_last_hip_error()})")


def default_options(**overrides) -> DspOptions:
    lib = load_library()
    o = DspOptions()
    lib.dsp_default_options(C.byref(o))
    # tuning knob: DSP_OPTIONS="check_every=32,kkt_every=2" overrides the library defaults (explicit arguments win)
    env = {}
    for item in filter(None, os.environ.get("DSP_OPTIONS", "").split(",")):
        k, _, v = item.partition("=")
        env[k.strip()] = type(getattr(o, k.strip(), 0.0))(float(v))
    for k, v in {**env, **overrides}.items():
        if not hasattr(o, k):
            raise TypeError(f"unknown solver option {k!r}")
        setattr(o, k, v)
    return o


class DeviceLP:
    """Device-resident shared data of one flattened LP (wraps a dsp_handle)."""

    def __init__(self, lp, device: int = 0, options: Optional[DspOptions] = None):
        import torch

        if not torch.cuda.is_available():
            raise DspError("no MI355X visible: the dispatch solver has no CPU path")
        self.lib = load_library()
        self.lp = lp
        self.device = device
        self._rowptr = np.ascontiguousarray(lp.indptr, np.int32)
        self._colidx = np.ascontiguousarray(lp.indices, np.int32)
        self._val = np.ascontiguousarray(lp.data, np.float64)
        # variable scaling factors of the model family (StandardFormLP.col_scale: typical column magnitudes), if it has any
        cs = getattr(lp, "col_scale", None)
        self._col_scale = None if cs is None else np.ascontiguousarray(cs, np.float64)
        desc = DspLpDesc(lp.n, lp.m, int(self._rowptr[-1]),
                         self._rowptr.ctypes.data_as(C.POINTER(C.c_int32)),
                         self._colidx.ctypes.data_as(C.POINTER(C.c_int32)),
                         self._val.ctypes.data_as(C.POINTER(C.c_double)),
                         None if self._col_scale is None else self._col_scale.ctypes.data_as(C.POINTER(C.c_double)))
        h = C.c_void_p()
        torch.cuda.set_device(device)
        _check(self.lib, self.lib.dsp_create(C.byref(desc), device, C.byref(options) if options else None, C.byref(h)),
               "dsp_create")
        self.handle = h
        self.options = options
        self.last_stats: Optional[DspStats] = None

    def close(self):
        if getattr(self, "handle", None):
            self.lib.dsp_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def scaling(self):
        dr = np.zeros(self.lp.m)
        dc = np.zeros(self.lp.n)
        eta = C.c_double()
        _check(self.lib, self.lib.dsp_get_scaling(self.handle, dr.ctypes.data, dc.ctypes.data, C.byref(eta)), "dsp_get_scaling")
        return dr, dc, eta.value

    @staticmethod
    def _ptr_stride(t, width):
        """(device pointer, scenario stride) of a [B,width] (dense) or [width] (broadcast) tensor, or (None, 0)."""
        if t is None:
            return None, 0
        assert t.is_cuda and t.dtype.is_floating_point and t.element_size() == 8 and t.is_contiguous()
        if t.dim() == 1:
            assert t.shape[0] == width
            return t.data_ptr(), 0
        assert t.shape[1] == width
        return t.data_ptr(), width

    def solve(self, B, c, lb=None, ub=None, rlo=None, rhi=None, x0=None, y0=None, options=None, out=None,
              sync_stats=True, obj_offset=None, primal_weight=None, row_compliance=None):
        """All arguments are CUDA(HIP) float64 torch tensors; returns dict of output tensors (+ stats).
        obj_offset: [B] objective constants (scale of the eps_obj tests); primal_weight: [B] in/out;
        row_compliance: [m] or [B, m] compliances of the soft rows (convex QP, dsp_batch::row_compliance)."""
        import torch

        n, m = self.lp.n, self.lp.m
        dev = torch.device("cuda", self.device)
        if out is None:
            out = dict(x=torch.empty((B, n), dtype=torch.float64, device=dev),
                       y=torch.empty((B, max(m, 1)), dtype=torch.float64, device=dev),
                       obj=torch.empty(B, dtype=torch.float64, device=dev),
                       status=torch.empty(B, dtype=torch.int32, device=dev),
                       iters=torch.empty(B, dtype=torch.int32, device=dev),
                       jumps=torch.empty(B, dtype=torch.int32, device=dev),
                       flags=torch.zeros(B, dtype=torch.int32, device=dev))
        bt = DspBatch()
        bt.B = B
        bt.c, bt.c_stride = self._ptr_stride(c, n)
        bt.var_lb, bt.var_lb_stride = self._ptr_stride(lb, n)
        bt.var_ub, bt.var_ub_stride = self._ptr_stride(ub, n)
        bt.row_lb, bt.row_lb_stride = self._ptr_stride(rlo, m)
        bt.row_ub, bt.row_ub_stride = self._ptr_stride(rhi, m)
        if obj_offset is not None:
            assert obj_offset.is_cuda and obj_offset.element_size() == 8 and obj_offset.numel() in (1, B)
            bt.obj_offset, bt.obj_offset_stride = obj_offset.data_ptr(), (1 if obj_offset.numel() == B and B > 1 else 0)
        if row_compliance is not None:
            bt.row_compliance, bt.row_compliance_stride = self._ptr_stride(row_compliance, m)
        bt.x0 = x0.data_ptr() if x0 is not None else None
        bt.y0 = y0.data_ptr() if y0 is not None else None
        if primal_weight is not None:
            assert primal_weight.is_cuda and primal_weight.element_size() == 8 and primal_weight.numel() == B
            bt.primal_weight = primal_weight.data_ptr()
        bt.x, bt.y, bt.obj = out["x"].data_ptr(), out["y"].data_ptr(), out["obj"].data_ptr()
        bt.status, bt.iters = out["status"].data_ptr(), out["iters"].data_ptr()
        bt.jumps = out["jumps"].data_ptr() if "jumps" in out else None
        bt.flags = out["flags"].data_ptr() if "flags" in out else None
        stats = DspStats()
        stream = torch.cuda.current_stream(dev).cuda_stream
        rc = self.lib.dsp_solve(self.handle, C.byref(bt), C.byref(options) if options is not None else None,
                                C.byref(stats), 1 if sync_stats else 0, C.c_void_p(stream))
        _check(self.lib, rc, "dsp_solve")
        self.last_stats = stats
        out["stats"] = stats
        return out

    def spmv_step(self, X, Y, AX=None, ATY=None):
        import torch

        B = X.shape[0]
        AX = torch.empty((B, self.lp.m), dtype=torch.float64, device=X.device) if AX is None else AX
        ATY = torch.empty((B, self.lp.n), dtype=torch.float64, device=X.device) if ATY is None else ATY
        stream = torch.cuda.current_stream(X.device).cuda_stream
        _check(self.lib, self.lib.dsp_spmv_step(self.handle, B, X.data_ptr(), Y.data_ptr(), AX.data_ptr(),
                                                ATY.data_ptr(), C.c_void_p(stream)), "dsp_spmv_step")
        return AX, ATY


def period_shift_maps(lp, shift: int):
    """Index maps for a rolling-horizon warm start: column / row "name[t]" takes its start value from "name[t+shift]"
    of the previous solve (the last `shift` periods keep their own old value).  Cached on the LP object."""
    import re

    cache = lp.__dict__.setdefault("_shift_maps", {})
    if shift not in cache:
        def build(names):
            pat = re.compile(r"^(.*)\[(\d+)\]$")
            where = {}
            parsed = []
            for k, nm in enumerate(names):
                mt = pat.match(nm)
                parsed.append((mt.group(1), int(mt.group(2))) if mt else None)
                if mt:
                    where[(mt.group(1), int(mt.group(2)))] = k
            return np.array([where.get((p[0], p[1] + shift), k) if p else k for k, p in enumerate(parsed)], np.int64)
        cache[shift] = (build(lp.col_names), build(lp.row_names))
    return cache[shift]


FLAG_OBJ_WAIVED = 1          # DSP_FLAG_OBJ_WAIVED of include/dsp_hip.h
STATUS_PRIMAL_INFEASIBLE, STATUS_DUAL_INFEASIBLE = 2, 3      # DSP_STATUS_* of include/dsp_hip.h
STATUS_UNCERTIFIED = 5       # host-side report code (Bidder.failed_scenarios, Tracker): the kernel said OPTIMAL with
                             # DSP_FLAG_OBJ_WAIVED set and no re-solve certified the objective accuracy either


def uncertified(status, flags):
    """Mask of the scenarios whose objective accuracy is NOT certified although their status is OPTIMAL."""
    status = np.asarray(status)
    if flags is None:
        return np.zeros(status.shape, bool)
    return (status == 0) & ((np.asarray(flags) & FLAG_OBJ_WAIVED) != 0)


class DeviceSolution:
    """x / y of a solve, still on the device (ScenarioBatchModel.store_solution(lazy=...)): a Bidder reads T of n columns, the rolling
    loop nothing at all - downloading 4096 x (n + m) doubles per call was a third of the host time of compute_day_ahead_bids."""

    def __init__(self, out, m, device, price_windows=()):
        self.x, self.y, self.m, self.device = out["x"], out["y"], m, device
        self.price_windows = price_windows            # ((host array, device tensor), ...) of the solve's objective recipe

    def fetch(self):
        import torch
        hx = torch.empty(self.x.shape, dtype=self.x.dtype, pin_memory=True)
        hy = torch.empty(self.y.shape, dtype=self.y.dtype, pin_memory=True)
        hx.copy_(self.x, non_blocking=True)
        hy.copy_(self.y, non_blocking=True)
        torch.cuda.current_stream(self.device).synchronize()
        return hx.numpy(), hy.numpy()[:, :self.m]

    def columns(self, cols):
        import torch
        idx = torch.as_tensor(np.ascontiguousarray(cols, np.int64), device=self.x.device)
        return self.x.index_select(1, idx).cpu().numpy()

    def rows(self, lo, hi):
        return self.x[lo:hi].cpu().numpy()

    def device_prices(self, prices):
        """Device copy of a [B, T] price array: the very window the solve uploaded for its objective when it is that array."""
        import torch
        for host, dev in self.price_windows:
            if host is prices:
                return dev
        return torch.as_tensor(np.ascontiguousarray(prices, np.float64), device=self.x.device)

    def bid_points(self, cols, vals, const, terms, prices, p_min, ok=None):
        """dsp_bid_points (csrc/dsp_bids.hip) on the solution where it lies: per hour the distinct offered powers with the highest price
        at each, in integer cents.  cols / vals: [T, 2]; const: [T]; prices: [B, >= T] host array (or device tensor).
        Returns (counts [T], power_cents [T, W], price_cents [T, W]) as numpy arrays (W = B), or None when the batch is beyond the
        kernel's limits (the caller keeps its tensor path)."""
        import torch
        B, T = self.x.shape[0], len(const)
        if B > BID_MAX_SCENARIOS or T > BID_MAX_HOURS or B < 1 or T < 1:
            return None
        lib = load_library()
        dev = self.x.device
        pd = prices if isinstance(prices, torch.Tensor) else self.device_prices(prices)
        okd = None if ok is None else torch.as_tensor(np.ascontiguousarray(ok, np.uint8), device=dev)
        out = torch.empty((T, B + 1), dtype=torch.int64, device=dev)
        rq = DspBidRequest()
        rq.B, rq.T, rq.ldx, rq.ldp, rq.terms = B, T, self.x.stride(0), pd.stride(0), int(terms)
        rq.x, rq.price, rq.ok, rq.out, rq.p_min = self.x.data_ptr(), pd.data_ptr(), (okd.data_ptr() if okd is not None else None), out.data_ptr(), float(p_min)
        C.memmove(rq.col, np.ascontiguousarray(cols, np.int32).ctypes.data, 8 * T)
        C.memmove(rq.val, np.ascontiguousarray(vals, np.float64).ctypes.data, 16 * T)
        C.memmove(rq.constant, np.ascontiguousarray(const, np.float64).ctypes.data, 8 * T)
        stream = torch.cuda.current_stream(dev)
        _check(lib, lib.dsp_bid_points(C.byref(rq), C.c_void_p(stream.cuda_stream)), "dsp_bid_points")
        host = torch.empty((T, B + 1), dtype=torch.int64, pin_memory=True)
        host.copy_(out, non_blocking=True)
        stream.synchronize()
        keys = host.numpy()
        halves = keys[:, 1:].view(np.int32)                      # little endian: price cents, power cents, price cents, ...
        return keys[:, 0].astype(np.int64), halves[:, 1::2], halves[:, 0::2]


class HipPdlpSolver:
    """Solver object for Bidder / SelfScheduler / Tracker: `solver.solve(model, tee=False)`.

    Options (keyword arguments) are the fields of ``dsp_options`` (include/dsp_hip.h), e.g. ``eps_rel=1e-9``.
    """

    supports_warm_start = True       # Bidder / Tracker pass warm_start / shift when the solver advertises this

    # Scenarios the kernel ACCEPTED WITHOUT CERTIFYING the objective accuracy (DSP_FLAG_OBJ_WAIVED: feasible to eps_rel, the
    # objective-error bound stagnating within 10 eps_obj) are solved again, alone, under other controller settings: the far
    # tail of the iteration is chaotic in every parameter (the slow scenarios under different gains are nearly
    # independent sets), so a scenario that stalls under one setting almost always certifies under another.  What is still
    # flagged after the last attempt keeps its flag, and Bidder / Tracker treat it as NOT solved (status
    # STATUS_UNCERTIFIED in their reports): nothing outside the 1e-6 contract becomes a bid silently.
    RECERTIFY_VARIANTS = (dict(pid_kp=0.45, restart_artificial=0.3), dict(pid_kp=0.8, restart_artificial=0.15),
                          dict(pid_kp=0.3, restart_artificial=0.5, check_every=12))

    def __init__(self, device: int = 0, recertify: int = 3, lazy_solution: bool = True, **options):
        self.device = device
        # lazy_solution: models that support it (ScenarioBatchModel) get x / y on first access instead of with every solve
        self.lazy_solution = bool(lazy_solution)
        self.recertify = max(0, min(int(recertify), len(self.RECERTIFY_VARIANTS)))
        self._option_overrides = options
        self.options = None
        self.last_stats = None
        self.last_recertified = 0        # scenarios of the last solve that were flagged by the first pass and certified by a re-solve

    def available(self, exception_flag=False):
        try:
            import torch
            load_library()
            ok = torch.cuda.is_available()
        except Exception:
            ok = False
        if not ok and exception_flag:
            raise DspError("HIP dispatch solver unavailable (library not built or no GPU)")
        return ok

    def _device_lp(self, model) -> DeviceLP:
        if self.options is None:
            self.options = default_options(**self._option_overrides)
        h = model.solve_handle
        if h is None or h.lp is not model.lp:
            # per-model-family preconditioner hints (e.g. geo_iters for tracking LPs), overridable by the user
            hints = dict(getattr(model, "solver_hints", None) or {})
            hints.update(self._option_overrides)
            h = DeviceLP(model.lp, self.device, default_options(**hints))
            model.solve_handle = h
        return h

    def _recertify(self, dlp, inputs, out, host, idx):
        """Re-solve the scenarios `idx` (flagged DSP_FLAG_OBJ_WAIVED by the first pass) under RECERTIFY_VARIANTS, one sub-batch
        per attempt, and overwrite their entries of the result arrays - the host copies AND the device outputs `out` of the first
        pass, which a sharded solve all-gathers from - with the first certified solve.  Returns how many were certified.
        Inputs: the device tensors of the first pass ([B, n] / [B, m] dense or [n] / [m] broadcast)."""
        import torch

        dev = torch.device("cuda", self.device)
        certified = 0
        todo = np.asarray(idx)
        for variant in self.RECERTIFY_VARIANTS[:self.recertify]:
            if not len(todo):
                break
            opts = DspOptions.from_buffer_copy(dlp.options)
            for k, v in variant.items():
                setattr(opts, k, v)
            sel = torch.as_tensor(todo, dtype=torch.int64, device=dev)
            pick = lambda t: None if t is None else (t.index_select(0, sel).contiguous() if t.dim() == 2 else t)
            c0 = inputs["c0"]
            sub = dlp.solve(len(todo), pick(inputs["c"]), pick(inputs["lb"]), pick(inputs["ub"]), pick(inputs["rlo"]),
                            pick(inputs["rhi"]), options=opts, obj_offset=c0.index_select(0, sel).contiguous(),
                            row_compliance=pick(inputs["kappa"]))
            st, fl = sub["status"].cpu().numpy(), sub["flags"].cpu().numpy()
            good = (st == 0) & ((fl & FLAG_OBJ_WAIVED) == 0)
            host["iters"].numpy()[todo] += sub["iters"].cpu().numpy()             # the work spent on the scenario, all attempts
            out["iters"].index_add_(0, sel, sub["iters"])
            if good.any():
                at = todo[good]
                gsel = torch.as_tensor(np.nonzero(good)[0], dtype=torch.int64, device=dev)
                if "x" in host:                                                    # (lazy solutions are read from `out` later)
                    host["x"].numpy()[at] = sub["x"].index_select(0, gsel).cpu().numpy()
                    host["y"].numpy()[at] = sub["y"].index_select(0, gsel).cpu().numpy()
                host["obj"].numpy()[at] = sub["obj"].index_select(0, gsel).cpu().numpy()
                host["jumps"].numpy()[at] = sub["jumps"].index_select(0, gsel).cpu().numpy()
                host["flags"].numpy()[at] = fl[good]
                dsel = torch.as_tensor(at, dtype=torch.int64, device=dev)
                for key in ("x", "y", "obj", "jumps", "flags"):
                    out[key].index_copy_(0, dsel, sub[key].index_select(0, gsel))
                certified += int(good.sum())
            todo = todo[~good]
        return certified

    def solve(self, model, tee=False, warm_start=False, shift=0):
        """warm_start: start from the model's previous (x, y, primal weight); shift: the previous solve was `shift`
        periods earlier in a rolling horizon, so period t starts from the old period t + shift."""
        import torch

        from .workflow.batch_model import SolveResults

        dlp = self._device_lp(model)
        dev = torch.device("cuda", self.device)
        B = model.n_scenario
        lb, ub, rlo, rhi = model.scenario_bounds()
        # convex QP: quadratic objective terms arrive as soft rows (LinearBlock.quadratic -> StandardFormLP.row_compliance)
        kappa = getattr(model.lp, "row_compliance", None)
        # uploads go through per-handle pinned staging buffers (a pageable 6 MB numpy array takes ~4 ms to reach the
        # device, a pinned one ~0.3 ms; the previous solve has synchronised, so the buffers are free to overwrite)
        stage = dlp.__dict__.setdefault("_staging", {})

        def up(key, a):
            a = np.asarray(a, dtype=np.float64)
            buf = stage.get(key)
            if buf is None or buf[0].shape != a.shape:
                buf = stage[key] = (torch.empty(a.shape, dtype=torch.float64, pin_memory=True),
                                    torch.empty(a.shape, dtype=torch.float64, device=dev))
            buf[0].numpy()[...] = a
            buf[1].copy_(buf[0], non_blocking=True)
            return buf[1]
        x0 = y0 = None
        pw = torch.zeros(B, dtype=torch.float64, device=dev)          # in: 0 = automatic; out: final primal weights
        if warm_start and getattr(model, "has_solution", model.x is not None) and model.x is not None and model.y is not None and model.x.shape == (B, model.lp.n):
            xs, ys = model.x, model.y
            if shift:
                cmap, rmap = period_shift_maps(model.lp, int(shift))
                xs, ys = xs[:, cmap], ys[:, rmap]
            x0, y0 = up("x0", xs), up("y0", ys)
            prev = getattr(model, "primal_weight", None)
            if prev is not None and len(prev) == B:
                pw = up("primal_weight", prev)
        # objective vectors: a recipe (Bidder: base vector + price windows, formed on the device) or a dense [B, n] array
        recipe = getattr(model, "c_recipe", None)
        c_dev = recipe.device(torch, dev, up, dlp.__dict__.setdefault("_recipe_cache", {})) if recipe is not None and getattr(model, "_c", 0) is None \
            else up("c", model.c)
        out = dlp.solve(B, c_dev, up("lb", lb), up("ub", ub), up("rlo", rlo) if model.lp.m else None,
                        up("rhi", rhi) if model.lp.m else None, x0=x0, y0=y0, options=dlp.options,
                        obj_offset=up("c0", np.broadcast_to(np.asarray(model.c0, np.float64), (B,))), primal_weight=pw,
                        row_compliance=(up("kappa", kappa) if kappa is not None and np.any(kappa) else None))
        st = out["stats"]
        self.last_stats = st
        # device outputs of this solve (x, y, obj = c.x without the model constant, status, iters): a sharded solve
        # all-gathers these directly over RCCL (dispatches_amd.distributed.solve_sharded)
        self.last_device_out = out
        # downloads into fresh pinned host tensors (torch caches freed pinned blocks, so this is cheap after the first
        # call); the numpy views handed to the model keep their tensors alive
        def down(t):
            h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
            h.copy_(t, non_blocking=True)
            return h

        lazy = self.lazy_solution and getattr(model, "supports_lazy_solution", False)
        host = {k: down(out[k]) for k in ("obj", "status", "iters", "jumps", "flags") + (() if lazy else ("x", "y"))}
        host["pw"] = down(pw)
        torch.cuda.current_stream(dev).synchronize()
        status = host["status"].numpy()
        self.last_recertified = 0
        waived = np.nonzero((status == 0) & ((host["flags"].numpy() & FLAG_OBJ_WAIVED) != 0))[0]
        if len(waived) and self.recertify:
            inputs = dict(c=c_dev, lb=stage["lb"][1], ub=stage["ub"][1],
                          rlo=stage["rlo"][1] if model.lp.m else None, rhi=stage["rhi"][1] if model.lp.m else None,
                          c0=stage["c0"][1], kappa=(stage["kappa"][1] if kappa is not None and np.any(kappa) else None))
            self.last_recertified = self._recertify(dlp, inputs, out, host, waived)
        if lazy:
            model.store_solution(None, None, host["obj"].numpy() + model.c0, status, host["iters"].numpy(),
                                 lazy=DeviceSolution(out, model.lp.m, dev, getattr(recipe, "device_windows", None) or ()))
        else:
            model.store_solution(host["x"].numpy(), host["y"].numpy()[:, :model.lp.m],
                                 host["obj"].numpy() + model.c0, status, host["iters"].numpy())
        model.jumps = host["jumps"].numpy()
        model.flags = host["flags"].numpy()      # DSP_FLAG_* bits AFTER the re-solves (bit 1 left = objective accuracy not certified)
        model.uncertified = uncertified(status, model.flags)
        model.primal_weight = host["pw"].numpy()
        if tee:
            print(f"[dsp_hip] B={B} n={model.lp.n} m={model.lp.m} nnz={model.lp.nnz} optimal={st.n_optimal}/{B} "
                  f"iters(sum/max)={st.total_iterations}/{st.max_iterations} kernel={st.kernel_ms:.3f} ms "
                  f"grid={st.grid_blocks}x{st.block_threads} lds={st.lds_bytes}B matreg={st.matreg} simplex={st.simplex} streaming={st.streaming} "
                  f"recertified={self.last_recertified} uncertified={int(model.uncertified.sum())}")
        all_ok = bool((status == 0).all())
        # termination condition in Pyomo's vocabulary, the worst scenario deciding (what the reference's callers branch on:
        # case_studies/renewables_case/solar_battery_hydrogen.py:451-458): statuses 2 / 3 come with a certificate (include/dsp_hip.h)
        term = ("optimal" if all_ok else "infeasible" if (status == STATUS_PRIMAL_INFEASIBLE).any() else
                "unbounded" if (status == STATUS_DUAL_INFEASIBLE).any() else "maxIterations")
        return SolveResults("ok" if all_ok else "warning", term,
                            iterations=int(st.total_iterations), kernel_ms=float(st.kernel_ms))
