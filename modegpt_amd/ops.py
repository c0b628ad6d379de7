"""torch-tensor front ends of the C ABI: pointer/stride extraction, workspace allocation, stream hand-off.

PyTorch is plumbing here (device memory + the current HIP stream); all arithmetic happens in
libmodegpt_hip.so.  Every function requires CUDA(HIP) tensors and raises otherwise.
"""
from __future__ import annotations

import contextlib
import ctypes as C
import os
import threading
from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import check

_DT = {torch.bfloat16: _lib.MDG_BF16, torch.float16: _lib.MDG_F16, torch.float32: _lib.MDG_F32,
       torch.float64: _lib.MDG_F64}


def _stream(t: torch.Tensor) -> int:
    return torch.cuda.current_stream(t.device).cuda_stream


def _need_gpu(*ts: torch.Tensor) -> None:
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("modegpt_amd ops run on the GPU only (got a CPU tensor); there is no CPU fallback")


def _ws(nbytes: int, device) -> Tuple[Optional[torch.Tensor], int]:
    if nbytes == 0:
        return None, 0
    t = torch.empty(nbytes, dtype=torch.uint8, device=device)
    return t, t.data_ptr()


def _p(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _as_weight(W: torch.Tensor) -> torch.Tensor:
    """Weights enter the fp64 GEMMs either as bf16 (converted exactly on load) or as fp64; fp16 / fp32 checkpoints are
    widened to fp64 first, exactly, instead of being squeezed through bf16."""
    W = W.detach()
    if W.dtype not in (torch.bfloat16, torch.float64):
        W = W.to(torch.float64)
    return W if W.stride(-1) == 1 else W.contiguous()


# ------------------------------------------------------------------ deferred status of a decomposition chain
class DeferredStatus:
    """`with ops.DeferredStatus(device) as st: <ridge_scores / nystrom_down / vo_compress / potrf_lower ...>` -- inside the block
    the decomposition entry points do not wait for the host to read their status word (Cholesky pivot, Jacobi convergence):
    they merge it into a device-side int[2] and return at once (mdg_deferred_status_begin / _end).  `st.check()` -- any time
    later -- copies the two ints to the host (the one synchronisation of the chain) and raises what the synchronising call
    would have raised: torch.linalg.LinAlgError for a matrix that is not positive definite, RuntimeError otherwise."""

    def __init__(self, device):
        self.device = torch.device(device)
        self.status = torch.empty(2, dtype=torch.int32, device=self.device)
        self.checked = False

    def __enter__(self):
        with torch.cuda.device(self.device):
            check(_lib.load().mdg_deferred_status_begin(self.status.data_ptr(), _stream(self.status)), "mdg_deferred_status_begin")
        return self

    def __exit__(self, *exc):
        _lib.load().mdg_deferred_status_end()
        return False

    def check(self) -> None:
        if self.checked:
            return
        self.checked = True
        host = (C.c_int * 2)(*self.status.cpu().tolist())
        check(_lib.load().mdg_deferred_status_decode(host), "decomposition chain")


# ------------------------------------------------------------------ covariance
def cov_accum(sigma: torch.Tensor, x: torch.Tensor, n_heads: int = 1, relu: bool = False) -> None:
    """sigma (lower triangle) += X^T X in fp64.  x: [..., n_heads*feat] (bf16/f16/f32/f64, last dim
    contiguous, viewed as [tokens, n_heads*feat]); sigma: [feat, feat] or [n_heads, feat, feat] fp64.
    Only the lower triangle is valid until cov_finalize()."""
    _need_gpu(sigma, x)
    lib = _lib.load()
    if sigma.dtype != torch.float64 or not sigma.is_contiguous():
        raise ValueError("sigma must be a contiguous float64 tensor")
    x2 = x.detach().reshape(-1, x.shape[-1])
    if x2.stride(-1) != 1:
        x2 = x2.contiguous()
    feat = sigma.shape[-1]
    if sigma.shape[-2] != feat or x2.shape[1] != n_heads * feat:
        raise ValueError(f"shape mismatch: x {tuple(x.shape)} vs sigma {tuple(sigma.shape)} with n_heads={n_heads}")
    if sigma.dim() == 3 and sigma.shape[0] != n_heads:
        raise ValueError("sigma batch dimension must equal n_heads")
    n_tok = x2.shape[0]
    nbytes = lib.mdg_cov_accum_ws_bytes(n_tok, feat, n_heads)
    ws, wsp = _ws(nbytes, x.device)
    with torch.cuda.device(x.device):
        check(lib.mdg_cov_accum(x2.data_ptr(), _DT[x2.dtype], n_tok, feat, n_heads, x2.stride(0), int(relu),
                                sigma.data_ptr(), feat, feat * feat, wsp, nbytes, _stream(x)), "mdg_cov_accum")


def cov_accum_i8(sigma: torch.Tensor, x: torch.Tensor, events=None, mfma_stats: Optional[dict] = None,
                 report: bool = True, route_info: Optional[dict] = None, tolerance: Optional[float] = None) -> Optional[int]:
    """sigma (lower triangle) += X^T X for one bf16 matrix through the int8 digit-plane kernel (csrc/cov_i8.hip): error-free
    split into digit planes, truncated plane-pair product.  The route -- five planes or six, which columns leave the int8 path
    for the fp64 column kernel (at most 32), or the fp64 kernel for the whole statistic -- is derived on the device from a
    per-call error bound (guaranteed <= 1.1e-11 of sqrt(sigma_ii sigma_jj) entry-wise, typically 1e-13; include/modegpt_hip.h);
    the result is valid either way.  tolerance: the factor on the route's thresholds for THIS call (None: i8_tolerance(), the
    calling thread's default).  report=True (tests, measurements) returns the route of this
    call (5, 6, or 0 for the fp64 kernel) at the price of one stream synchronisation and books it in I8_STATS;
    report=False (the hooks: cov_accum_multi) only enqueues and returns None -- the per-device counters behind
    i8_route_counts() are updated by the kernels themselves in both modes.  Feature count must be a multiple of 128.
    route_info: optional dict, filled with {"planes", "columns" (those the fp64 column kernel computed), "sq", "x" (the two parts
    of the bound for the columns that stayed), "bound" (their sum), "exact" (the call ran the exact route: no plane pair dropped,
    the bound is the rounded-element term + fp64 rounding), "remainder" ("tiles" / "wide": which implementation of the exact
    route's remainder products the device picked; None off the exact route)} -- implies report.
    events: optional pair of torch.cuda.Event(enable_timing=True), each recorded once already, re-recorded around the product
    launches alone.  mfma_stats: optional dict; its "executed" entry is increased by the number of v_mfma instructions the
    product kernel issued (it skips digit planes that are all-zero over a tile panel) and "dense" by what a kernel without
    that skipping issues -- costs a stream synchronisation, for measurement only (implies report)."""
    _need_gpu(sigma, x)
    lib = _lib.load()
    if sigma.dtype != torch.float64 or not sigma.is_contiguous() or sigma.dim() != 2:
        raise ValueError("sigma must be a contiguous 2-D float64 tensor")
    if x.dtype != torch.bfloat16:
        raise ValueError("cov_accum_i8 takes bf16 activations")
    x2 = x.detach().reshape(-1, x.shape[-1])
    if x2.stride(-1) != 1:
        x2 = x2.contiguous()
    n = sigma.shape[0]
    if sigma.shape[1] != n or x2.shape[1] != n:
        raise ValueError(f"shape mismatch: sigma {tuple(sigma.shape)}, x {tuple(x2.shape)}")
    nbytes = lib.mdg_cov_accum_i8_ws_bytes(x2.shape[0], n)
    ws, wsp = _ws(nbytes, x.device)
    report = report or mfma_stats is not None or route_info is not None
    used = C.c_int(0)
    with torch.cuda.device(x.device):
        check(lib.mdg_cov_accum_i8(x2.data_ptr(), x2.shape[0], n, x2.stride(0), sigma.data_ptr(), sigma.stride(0), wsp, nbytes,
                                   i8_tolerance() if tolerance is None else float(tolerance), _i8_flags(), C.byref(used) if report else None, _route_counters(x.device).data_ptr(),
                                   None if events is None else events[0].cuda_event,
                                   None if events is None else events[1].cuda_event, _stream(x)), "mdg_cov_accum_i8")
        info = None
        if route_info is not None or mfma_stats is not None:
            arr = (_lib.CovProblem * 1)(_lib.CovProblem(x2.data_ptr(), x2.shape[0], n, 1, x2.stride(0), sigma.data_ptr(),
                                                        sigma.stride(0), 0))
            info = _read_route(lib, 1, arr, 0, wsp, _stream(x))
        if mfma_stats is not None and used.value in (5, 6):
            done = C.c_ulonglong(0)
            check(lib.mdg_cov_accum_i8_stats(wsp, x2.shape[0], n, C.byref(done), _stream(x)), "mdg_cov_accum_i8_stats")
            ran = 3 if info["exact"] else used.value           # (the exact route: the three-plane product launch, all nine pairs)
            mfma_stats["executed"] = mfma_stats.get("executed", 0) + done.value
            mfma_stats["dense"] = mfma_stats.get("dense", 0) + i8_dense_mfma_count(x2.shape[0], n, ran)
            mfma_stats["planes_run"] = ran
        if route_info is not None:
            route_info.update(info)
    if not report:
        return None
    I8_STATS[{5: "i8_5", 6: "i8_6"}.get(used.value, "fallback_f64")] += 1
    return used.value


def _read_route(lib, count, arr, stat, wsp, stream) -> dict:
    planes, ncol, exact = C.c_int(0), C.c_int(0), C.c_int(0)
    cols = (C.c_int * 32)()
    bound = (C.c_double * 2)()
    check(lib.mdg_cov_accum_i8_route(count, arr, stat, wsp, C.byref(planes), C.byref(ncol), cols, bound, C.byref(exact), stream),
          "mdg_cov_accum_i8_route")
    return {"planes": planes.value, "columns": [cols[i] for i in range(ncol.value)], "sq": bound[0], "x": bound[1],
            "bound": bound[0] + bound[1], "exact": bool(exact.value),
            "remainder": {0: None, 1: "tiles", 2: "wide"}.get(exact.value)}


# The exact route of the int8 covariance (include/modegpt_hip.h, "THE EXACT ROUTE"): "auto" (default) takes it where it is the
# faster product -- launches of the six-plane class (the MLP statistic of a gated model) and five-plane launches of a statistic of
# 4096 features and more; "always" (True) wherever the remainder lists fit (fp64-rounding accuracy for every int8 statistic);
# "never" (False) keeps every call on the truncated five- / six-plane product with its bound.  MODEGPT_I8_EXACT=auto|1|0.
I8_EXACT = {"1": True, "always": True, "0": False, "never": False}.get(os.environ.get("MODEGPT_I8_EXACT", "auto").lower(), "auto")


def _i8_flags() -> int:
    return {True: _lib.MDG_I8_EXACT_ALWAYS, False: _lib.MDG_I8_NO_EXACT}.get(I8_EXACT, 0)


def cov_accum_i8_multi(items, events=None, mfma_stats: Optional[dict] = None, report: bool = False,
                       route_info: Optional[list] = None, tolerance: Optional[float] = None) -> Optional[int]:
    """Several statistics of ONE calibration batch through the int8 digit-plane kernels in one persistent product launch
    (mdg_cov_accum_i8_multi): items = sequence of (sigma, x, n_heads), largest first, at most 4, all bf16 with the same token
    count.  n_heads == 1: sigma [n, n], n a multiple of 128; n_heads > 1: per-head Grams, sigma [n_heads, 128, 128] of an
    activation [tokens, n_heads * 128].  The tiles of all statistics share one tile schedule -- the small ones fill what the
    large one's last round leaves idle -- and one route: the deepest any column of any of them asks for (more planes are
    never less exact); a statistic too heavy-tailed for six planes leaves the launch alone (fp64 kernel).  events / mfma_stats /
    report as in cov_accum_i8 (report: the planes of the statistics that stayed, 0 if none did; the executed / dense counts cover
    all statistics and assume none fell back).  route_info: optional list, extended by one dict per statistic (see cov_accum_i8)."""
    lib = _lib.load()
    items = list(items)
    arr = (_lib.CovProblem * len(items))()
    dense_shapes = []
    keep = []
    for i, (sigma, x, n_heads) in enumerate(items):
        _need_gpu(sigma, x)
        if sigma.dtype != torch.float64 or not sigma.is_contiguous():
            raise ValueError("sigma must be a contiguous float64 tensor")
        if x.dtype != torch.bfloat16:
            raise ValueError("cov_accum_i8_multi takes bf16 activations")
        x2 = x.detach().reshape(-1, x.shape[-1])
        if x2.stride(-1) != 1:
            x2 = x2.contiguous()
        feat = sigma.shape[-1]
        if sigma.shape[-2] != feat or x2.shape[1] != n_heads * feat or (n_heads > 1 and (sigma.dim() != 3 or sigma.shape[0] != n_heads)):
            raise ValueError(f"shape mismatch: x {tuple(x.shape)} vs sigma {tuple(sigma.shape)} with n_heads={n_heads}")
        keep.append(x2)
        arr[i] = _lib.CovProblem(x2.data_ptr(), x2.shape[0], feat, n_heads, x2.stride(0), sigma.data_ptr(), feat, feat * feat)
        dense_shapes.append((x2.shape[0], feat, n_heads))
    dev = keep[0].device
    nbytes = lib.mdg_cov_accum_i8_multi_ws_bytes(len(items), arr)
    if nbytes == 0 and keep[0].shape[0] > 0:
        raise ValueError("these statistics cannot share an int8 launch (token counts differ, widths not multiples of 128, or "
                         "per-head statistics with head_dim != 128)")
    ws, wsp = _ws(nbytes, dev)
    report = report or mfma_stats is not None or route_info is not None
    used = C.c_int(0)
    with torch.cuda.device(dev):
        check(lib.mdg_cov_accum_i8_multi(len(items), arr, wsp, nbytes, i8_tolerance() if tolerance is None else float(tolerance),
                                         _i8_flags(), C.byref(used) if report else None,
                                         _route_counters(dev).data_ptr(), None if events is None else events[0].cuda_event,
                                         None if events is None else events[1].cuda_event, _stream(keep[0])),
              "mdg_cov_accum_i8_multi")
        infos = None
        if route_info is not None or mfma_stats is not None:
            infos = [_read_route(lib, len(items), arr, i, wsp, _stream(keep[0])) for i in range(len(items))]
        if mfma_stats is not None and used.value in (5, 6):
            done = C.c_ulonglong(0)
            check(lib.mdg_cov_accum_i8_stats(wsp, 0, 0, C.byref(done), _stream(keep[0])), "mdg_cov_accum_i8_stats")
            ran = 3 if any(i_["exact"] for i_ in infos) else used.value
            mfma_stats["executed"] = mfma_stats.get("executed", 0) + done.value
            mfma_stats["dense"] = mfma_stats.get("dense", 0) + sum(i8_dense_mfma_count(t, f, ran, h) for t, f, h in dense_shapes)
            mfma_stats["planes_run"] = ran
        if route_info is not None:
            route_info.extend(infos)
    if not report:
        return None
    I8_STATS[{5: "i8_5", 6: "i8_6"}.get(used.value, "fallback_f64")] += len(items)
    return used.value


_ROUTE_COUNTERS = {}


def _route_counters(device) -> torch.Tensor:
    """Per-device int32[5] the kernels bump: [five planes, six planes, fp64 fallback of a whole statistic, columns handed to the
    fp64 column kernel, statistics on the exact route] (mdg_cov_accum_i8 route_counts)."""
    dev = torch.device(device)
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    if key not in _ROUTE_COUNTERS:
        _ROUTE_COUNTERS[key] = torch.zeros(5, dtype=torch.int32, device=torch.device("cuda", key))
    return _ROUTE_COUNTERS[key]


# The accuracy / speed dial of the int8 covariance route is an ARGUMENT of every call (mdg_cov_accum_i8's `tolerance`, ABI 9); the
# library keeps no accuracy state.  What is kept here, on the Python side, is only the default a call without an explicit
# `tolerance=` uses: a process default (set_i8_tolerance; initialised from the environment variable MODEGPT_I8_TOLERANCE when this
# module is imported) that a thread can override for itself (i8_tolerance_scope) without touching any other caller's arithmetic.
def _checked_tolerance(factor) -> float:
    f = float(factor)
    if not (1.0 <= f <= 1e6):
        raise ValueError(f"int8 route tolerance factor {f!r} outside [1, 1e6] (1 = guaranteed <= 1.1e-11)")
    return f


_I8_TOLERANCE_DEFAULT = _checked_tolerance(os.environ.get("MODEGPT_I8_TOLERANCE", "1") or "1")
_I8_TOLERANCE_LOCAL = threading.local()


def i8_tolerance() -> float:
    """The tolerance factor a cov_accum_i8 / cov_accum_i8_multi / cov_accum_multi call made by THIS thread uses when it is not
    given one: the innermost i8_tolerance_scope of the thread, else the process default."""
    stack = getattr(_I8_TOLERANCE_LOCAL, "stack", None)
    return stack[-1] if stack else _I8_TOLERANCE_DEFAULT


def set_i8_tolerance(factor: float) -> float:
    """Sets the process DEFAULT of the int8 route's tolerance factor (`factor` >= 1 on both thresholds of the per-call error bound;
    1 = guaranteed <= 1.1e-11 of sqrt(sigma_ii sigma_jj)) and returns the previous default.  A Python-side default only: every
    library call carries its factor as an argument."""
    global _I8_TOLERANCE_DEFAULT
    prev, _I8_TOLERANCE_DEFAULT = _I8_TOLERANCE_DEFAULT, _checked_tolerance(factor)
    return prev


@contextlib.contextmanager
def i8_tolerance_scope(factor: float):
    """`with ops.i8_tolerance_scope(64): ...` -- the calling THREAD's default inside the block (hooks enqueued from it included);
    other threads keep theirs."""
    stack = getattr(_I8_TOLERANCE_LOCAL, "stack", None)
    if stack is None:
        stack = _I8_TOLERANCE_LOCAL.stack = []
    stack.append(_checked_tolerance(factor))
    try:
        yield
    finally:
        stack.pop()


def i8_route_counts(device=None, reset: bool = False) -> dict:
    """How the int8-route requests on `device` (default: the current one) were served so far, counted on the device by the
    kernels that ran: {"i8_5", "i8_6", "fallback_f64"} count statistics (by the class the route kernel gave them), "fp64_columns"
    the single columns the route handed to the fp64 column kernel, "exact" how many of the i8_5 / i8_6 statistics ran the exact
    route (nine plane pairs + the fp64 remainder products) instead of the truncated product.  One small device -> host copy;
    calibration reads it once, at the end."""
    key = torch.cuda.current_device() if device is None else (torch.device(device).index or 0)
    t = _route_counters(torch.device("cuda", key))
    v = t.cpu().tolist()
    if reset:
        t.zero_()
    return {"i8_5": v[0], "i8_6": v[1], "fallback_f64": v[2], "fp64_columns": v[3], "exact": v[4]}


def i8_dense_mfma_count(n_tokens: int, n: int, planes: int, n_heads: int = 1) -> int:
    """v_mfma_i32_32x32x32_i8 instructions of the digit-plane product without zero-plane skipping: every 32 x 32 block of
    the tiles covering the lower triangle (128 x 128 tiles for five planes, 128 x 64 for six; the diagonal tiles whole), per
    k-step of 32 tokens, planes (planes + 1) / 2 plane pairs.  n_heads > 1: per-head statistics of width n = 128 each -- one
    diagonal 128 x 128 tile (16 blocks) per head."""
    rb, nk = n // 128, -(-n_tokens // 32)
    if n_heads > 1:
        blocks = n_heads * 16
    else:
        blocks = rb * (rb + 1) * 8 if planes == 6 else rb * (rb + 1) // 2 * 16
    return blocks * nk * {3: 9, 5: 15, 6: 21}[planes]       # (3: the exact route's product -- three planes, all nine pairs)


# Which matrix cores accumulate the large covariances of a layer: "f64" (v_mfma_f64, the accumulation order of the
# reference's fp64 matmul) or "i8" (error-free digit-plane split, truncated product on the int8 cores, csrc/cov_i8.hip; what it
# cannot take stays on the fp64 kernel, and every call derives its route -- planes, columns for the fp64 column kernel, or the
# fp64 kernel -- from its own error bound).
COV_MODE = os.environ.get("MODEGPT_COV_MODE", "i8")
I8_MIN_FEATURES = 2048
I8_STATS = {"i8_5": 0, "i8_6": 0, "fallback_f64": 0}      # routes of the REPORTING cov_accum_i8 calls (tests, bench); all calls: i8_route_counts()


def cov_accum_multi(items, mode: Optional[str] = None) -> None:
    """One launch for several covariance problems of the same calibration batch.  items: sequence of
    (sigma, x, n_heads), largest problem first.  Falls back to one cov_accum call per item when the fused kernel's
    preconditions do not hold (mixed dtypes, feature count not a multiple of 128, unaligned rows).
    mode (default ops.COV_MODE): "i8" sends every bf16 statistic the int8 digit-plane kernels can take -- single matrices of
    at least I8_MIN_FEATURES features (a multiple of 128) and, beside one of those, per-head statistics of head_dim 128 --
    through them: the largest in a launch of its own (cov_accum_i8), the others together in one (cov_accum_i8_multi); only the
    rest goes through the fp64 kernel."""
    items = [(s_, x_, h_) for (s_, x_, h_) in items if x_.numel() > 0]
    if not items:
        return
    mode = mode or COV_MODE
    if mode not in ("f64", "i8"):
        raise ValueError(f"covariance mode must be 'f64' or 'i8', got {mode!r}")
    if mode == "i8":
        rest, planes, heads = [], [], []
        for sigma, x, n_heads in items:
            # below ~2048 features the 128 x 128 tiles do not fill the 256 CUs and the fp64 kernel is the faster one
            # (scripts/probes/i8_small_n.py: 1536 features 1.35 vs 1.23 ms, 2048 features 1.40 vs 2.14 ms)
            if n_heads == 1 and x.dtype == torch.bfloat16 and sigma.dim() == 2 and sigma.shape[-1] % 128 == 0 \
                    and sigma.shape[-1] >= I8_MIN_FEATURES:
                planes.append((sigma, x, 1))
            elif n_heads > 1 and x.dtype == torch.bfloat16 and sigma.dim() == 3 and sigma.shape[-1] == 128:
                heads.append((sigma, x, n_heads))      # per-head statistics of head_dim 128: diagonal tiles of the same launch
            else:
                rest.append((sigma, x, n_heads))
        tokens = {x.reshape(-1, x.shape[-1]).shape[0] for _, x, _ in planes + heads}
        group = ((planes[1:] if len(planes) > 1 else planes) + heads) if planes else []
        if planes and I8_FUSE and len(tokens) == 1 and len(group) <= 4 and _fusable_device(planes[0][1].device):
            # The largest statistic (sigma_mlp) keeps a launch and a route of its own -- on a real gated MLP it is the heavy-tailed
            # one (six planes) while the others take five, and a shared launch would drag them along (measured: -2 % on SiLU-gated
            # data).  Everything else -- sigma_x and the per-head sigma_q / sigma_k tiles -- shares ONE persistent int8 launch: one
            # tile schedule, one k-split last round, no fp64 launch for the heads.
            side_rest = rest and COV_OVERLAP_SMALL   # (per-head statistics of another head_dim: fp64 kernel, on a side stream)
            if side_rest:
                dev = planes[0][1].device
                main, side = torch.cuda.current_stream(dev), _side_stream(dev)
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    _cov_accum_fused(rest)
            if len(planes) > 1:
                cov_accum_i8(planes[0][0], planes[0][1], report=False)
            if len(group) > 1:
                cov_accum_i8_multi(group, report=False)
            else:
                cov_accum_i8(group[0][0], group[0][1], report=False)
            if side_rest:
                main.wait_stream(side)
            elif rest:
                _cov_accum_fused(rest)
            return
        rest = heads + rest
        if planes and rest and COV_OVERLAP_SMALL:
            # the small fp64 problems (per-head sigma_q / sigma_k) run on a side stream next to the LAST -- smallest -- int8
            # problem, whose few hundred tiles leave CUs idle in their final round; the large problem keeps the chip to itself
            for sigma, x, _ in planes[:-1]:
                cov_accum_i8(sigma, x, report=False)
            dev = planes[-1][1].device
            main, side = torch.cuda.current_stream(dev), _side_stream(dev)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                _cov_accum_fused(rest)
            cov_accum_i8(planes[-1][0], planes[-1][1], report=False)
            main.wait_stream(side)       # later work on the caller's stream (and any reuse of these buffers) is ordered after both
            return
        for sigma, x, _ in planes:
            cov_accum_i8(sigma, x, report=False)
        items = rest
        if not items:
            return
    _cov_accum_fused(items)


_SIDE_STREAMS = {}
I8_FUSE = os.environ.get("MODEGPT_I8_FUSE", "1") != "0"      # one int8 launch for all eligible statistics of a batch (cov_accum_i8_multi)


def _fusable_device(device) -> bool:
    """The fused int8 launch is a persistent launch over a tile schedule cut for 8 XCDs x 32 CUs."""
    return torch.cuda.get_device_properties(device).multi_processor_count == 256


COV_OVERLAP_SMALL = os.environ.get("MODEGPT_COV_OVERLAP", "1") != "0"
NYSTROM_OVERLAP = os.environ.get("MODEGPT_NYSTROM_OVERLAP", "1") != "0"   # cross product of the Nystrom refit beside the factorisation of C_kk


def _side_stream(device, purpose: str = "cov", beside=None) -> "torch.cuda.Stream":
    """One helper stream per (device, purpose[, the stream it runs beside]): two layers' chains on two streams of the caller's
    get a helper each and stay independent of one another."""
    dev = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    key = (dev, purpose, None if beside is None else beside.cuda_stream)
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(device=dev)
    return _SIDE_STREAMS[key]


def _cov_accum_fused(items) -> None:
    """The fp64 part of cov_accum_multi: one fused launch when the preconditions hold, else one cov_accum per item."""
    lib = _lib.load()
    prepared = []
    dtype = items[0][1].dtype
    fusable = len(items) <= 4
    for sigma, x, n_heads in items:
        _need_gpu(sigma, x)
        if sigma.dtype != torch.float64 or not sigma.is_contiguous():
            raise ValueError("sigma must be a contiguous float64 tensor")
        x2 = x.detach().reshape(-1, x.shape[-1])
        if x2.stride(-1) != 1:
            x2 = x2.contiguous()
        feat = sigma.shape[-1]
        if x2.shape[1] != n_heads * feat:
            raise ValueError(f"shape mismatch: x {tuple(x.shape)} vs sigma {tuple(sigma.shape)} with n_heads={n_heads}")
        esz = x2.element_size()
        fusable = fusable and x2.dtype == dtype and feat % 128 == 0 and x2.data_ptr() % 16 == 0 and \
            (x2.stride(0) * esz) % 16 == 0 and x2.device == items[0][1].device
        prepared.append((sigma, x2, n_heads, feat))
    if not fusable:
        for sigma, x2, n_heads, _ in prepared:
            cov_accum(sigma, x2, n_heads=n_heads)
        return
    arr = (_lib.CovProblem * len(prepared))()
    for i, (sigma, x2, n_heads, feat) in enumerate(prepared):
        arr[i] = _lib.CovProblem(x2.data_ptr(), x2.shape[0], feat, n_heads, x2.stride(0), sigma.data_ptr(), feat, feat * feat)
    dev = prepared[0][1].device
    nbytes = lib.mdg_cov_accum_multi_ws_bytes(len(prepared), arr, _DT[dtype])
    ws, wsp = _ws(nbytes, dev)
    with torch.cuda.device(dev):
        check(lib.mdg_cov_accum_multi(len(prepared), arr, _DT[dtype], wsp, nbytes, _stream(prepared[0][1])),
              "mdg_cov_accum_multi")


def cov_finalize(sigma: torch.Tensor, scale: float) -> None:
    """sigma <- scale * sigma (lower) mirrored into the upper triangle."""
    _need_gpu(sigma)
    lib = _lib.load()
    n = sigma.shape[-1]
    batch = 1 if sigma.dim() == 2 else sigma.shape[0]
    with torch.cuda.device(sigma.device):
        check(lib.mdg_cov_finalize(sigma.data_ptr(), n, batch, n, n * n, float(scale), _stream(sigma)),
              "mdg_cov_finalize")


def bi_accum(out: torch.Tensor, x_in: torch.Tensor, x_out: torch.Tensor) -> None:
    """out[0] += sum_tokens (1 - cos(x_in, x_out)); out: 1-element fp64 device tensor."""
    _need_gpu(out, x_in, x_out)
    lib = _lib.load()
    a = x_in.detach().reshape(-1, x_in.shape[-1])
    b = x_out.detach().reshape(-1, x_out.shape[-1])
    if a.dtype != b.dtype or a.shape != b.shape:
        raise ValueError("x_in / x_out must match in dtype and shape")
    a = a if a.is_contiguous() else a.contiguous()
    b = b if b.is_contiguous() else b.contiguous()
    nbytes = lib.mdg_bi_ws_bytes(a.shape[0])
    ws, wsp = _ws(nbytes, a.device)
    with torch.cuda.device(a.device):
        check(lib.mdg_bi_accum(a.data_ptr(), b.data_ptr(), _DT[a.dtype], a.shape[0], a.shape[1], a.stride(0),
                               out.data_ptr(), wsp, nbytes, _stream(a)), "mdg_bi_accum")


# ------------------------------------------------------------------ dense blocks
def gemm(A: torch.Tensor, B: torch.Tensor, C_out: torch.Tensor, alpha: float = 1.0, beta: float = 0.0,
         trans_a: bool = False, trans_b: bool = False, a_rows: Optional[torch.Tensor] = None, flags: int = 0) -> None:
    """C_out = alpha * op(A) @ op(B) + beta * C_out for 2-D row-major tensors (f64 or bf16)."""
    _need_gpu(A, B, C_out)
    lib = _lib.load()
    M, N = C_out.shape
    K = A.shape[0] if trans_a else A.shape[1]
    sa_i, sa_k = (A.stride(1), A.stride(0)) if trans_a else (A.stride(0), A.stride(1))
    sb_k, sb_j = (B.stride(1), B.stride(0)) if trans_b else (B.stride(0), B.stride(1))
    with torch.cuda.device(A.device):
        check(lib.mdg_gemm_f64(M, N, K, alpha, A.data_ptr(), _DT[A.dtype], sa_i, sa_k, _p(a_rows), B.data_ptr(),
                               _DT[B.dtype], sb_k, sb_j, beta, C_out.data_ptr(), _DT[C_out.dtype], C_out.stride(0),
                               1, 0, 0, 0, flags, _stream(A)), "mdg_gemm_f64")


def potrf_lower(A: torch.Tensor) -> torch.Tensor:
    """In-place lower Cholesky of the square fp64 matrix A; returns the inverted-diagonal-block buffer."""
    _need_gpu(A)
    lib = _lib.load()
    n = A.shape[0]
    inv = torch.empty(lib.mdg_potrf_inv_diag_elems(n), dtype=torch.float64, device=A.device)
    with torch.cuda.device(A.device):
        check(lib.mdg_potrf_lower(A.data_ptr(), n, A.stride(0), inv.data_ptr(), _stream(A)), "mdg_potrf_lower")
    return inv


def potrs_lower(L: torch.Tensor, inv: torch.Tensor, X: torch.Tensor) -> None:
    """X <- (L L^T)^-1 X in place."""
    _need_gpu(L, inv, X)
    lib = _lib.load()
    nbytes = lib.mdg_potrs_lower_ws_bytes(L.shape[0], X.shape[1])
    ws, wsp = _ws(nbytes, L.device)
    with torch.cuda.device(L.device):
        check(lib.mdg_potrs_lower(L.data_ptr(), L.shape[0], L.stride(0), inv.data_ptr(), X.data_ptr(), X.shape[1],
                                  X.stride(0), wsp, nbytes, _stream(L)), "mdg_potrs_lower")


def syevj(A: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """Eigen-decomposition of a batch of symmetric matrices [b, n, n] (n <= 128, even).
    Returns (evals descending [b, n], evecs [b, n, n], eigenvector j in column j).  A is not modified."""
    _need_gpu(A)
    lib = _lib.load()
    A3 = A.reshape(-1, A.shape[-2], A.shape[-1]).to(torch.float64).clone()
    b, n, _ = A3.shape
    evals = torch.empty(b, n, dtype=torch.float64, device=A.device)
    evecs = torch.empty(b, n, n, dtype=torch.float64, device=A.device)
    with torch.cuda.device(A.device):
        check(lib.mdg_syevj_batched(A3.data_ptr(), n, b, evals.data_ptr(), evecs.data_ptr(), _stream(A)),
              "mdg_syevj_batched")
    return evals, evecs


def sqrt_psd_small(M: torch.Tensor, ridge: float, scaled: bool, want_inverse: bool):
    """Batched sqrt_M for n <= 128: returns (root, inv_root or None, evals descending pre-ridge)."""
    _need_gpu(M)
    lib = _lib.load()
    M3 = M.reshape(-1, M.shape[-2], M.shape[-1]).to(torch.float64).contiguous()
    b, n, _ = M3.shape
    root = torch.empty_like(M3)
    inv_root = torch.empty_like(M3) if want_inverse else None
    evals = torch.empty(b, n, dtype=torch.float64, device=M.device)
    nbytes = lib.mdg_sqrt_psd_small_ws_bytes(n, b)
    ws, wsp = _ws(nbytes, M.device)
    with torch.cuda.device(M.device):
        check(lib.mdg_sqrt_psd_small(M3.data_ptr(), n, b, float(ridge), int(scaled), root.data_ptr(), _p(inv_root),
                                     evals.data_ptr(), wsp, nbytes, _stream(M)), "mdg_sqrt_psd_small")
    return root.reshape(M.shape), (None if inv_root is None else inv_root.reshape(M.shape)), evals


def sqrt_psd_large(M: torch.Tensor, ridge: float, scaled: bool, want_inverse: bool, want_evals: bool = True):
    """sqrt_M for one symmetric matrix of any size: returns (root, inv_root or None, evals unsorted or None).
    With want_evals=False and scaled=False the library takes the GEMM-only Newton-Schulz route for sqrt(M + ridge I)
    (falling back to block Jacobi by itself when the input is not positive definite); eigenvalues need block Jacobi."""
    _need_gpu(M)
    lib = _lib.load()
    if M.dim() != 2 or M.shape[0] != M.shape[1]:
        raise ValueError("sqrt_psd_large expects one square matrix")
    M2 = M if (M.dtype == torch.float64 and M.stride(1) == 1) else M.to(torch.float64).contiguous()
    n = M2.shape[0]
    root = torch.empty(n, n, dtype=torch.float64, device=M.device)
    inv_root = torch.empty(n, n, dtype=torch.float64, device=M.device) if want_inverse else None
    evals = torch.empty(n, dtype=torch.float64, device=M.device) if want_evals else None
    nbytes = lib.mdg_sqrt_psd_large_ws_bytes(n)
    ws, wsp = _ws(nbytes, M.device)
    with torch.cuda.device(M.device):
        check(lib.mdg_sqrt_psd_large(M2.data_ptr(), n, M2.stride(0), float(ridge), int(scaled), root.data_ptr(),
                                     _p(inv_root), _p(evals), wsp, nbytes, _stream(M)), "mdg_sqrt_psd_large")
    return root, inv_root, evals


# ------------------------------------------------------------------ MLP
def ridge_scores(Cm: torch.Tensor, ridge: float, want_sens: bool = False):
    """diag((C + ridge I)^-1) for a symmetric PD fp64 matrix.  want_sens: also the first-order sensitivities `sens` of the scores to
    an entry-wise relative perturbation of C (|delta score_j| <= eps sens_j for |E_ab| <= eps sqrt(c_aa c_bb); mdg_ridge_scores)
    -> (scores, sens)."""
    _need_gpu(Cm)
    lib = _lib.load()
    if Cm.dtype != torch.float64 or Cm.stride(1) != 1:
        raise ValueError("C must be float64 with unit column stride")
    n = Cm.shape[0]
    scores = torch.empty(n, dtype=torch.float64, device=Cm.device)
    sens = torch.empty(n, dtype=torch.float64, device=Cm.device) if want_sens else None
    nbytes = lib.mdg_ridge_scores_ws_bytes(n)
    ws, wsp = _ws(nbytes, Cm.device)
    with torch.cuda.device(Cm.device):
        check(lib.mdg_ridge_scores(Cm.data_ptr(), n, Cm.stride(0), float(ridge), scores.data_ptr(), _p(sens), wsp, nbytes,
                                   _stream(Cm)), "mdg_ridge_scores")
    return (scores, sens) if want_sens else scores


def select_smallest_sorted(scores: torch.Tensor, k: int) -> torch.Tensor:
    """Indices of the k smallest scores in ascending index order (topk(largest=False) + sort)."""
    _need_gpu(scores)
    lib = _lib.load()
    s = scores.to(torch.float64).contiguous()
    idx = torch.empty(k, dtype=torch.int64, device=s.device)
    with torch.cuda.device(s.device):
        check(lib.mdg_select_smallest_sorted(s.data_ptr(), s.numel(), k, idx.data_ptr(), _stream(s)),
              "mdg_select_smallest_sorted")
    return idx


MARGIN_FIELDS = ("s_selected_max", "s_unselected_min", "selected_upper", "unselected_lower", "sens_selected_max", "sens_unselected_max",
                 "scores_at_risk", "certified")


def select_margin(scores: torch.Tensor, sens: torch.Tensor, idx: torch.Tensor, eps: float) -> torch.Tensor:
    """The certificate of a k-smallest selection (mdg_select_margin): 8 fp64 numbers ON THE DEVICE (MARGIN_FIELDS), enqueued only --
    read them when the stream is waited for anyway (decode_margin)."""
    _need_gpu(scores, sens, idx)
    lib = _lib.load()
    s = scores if (scores.dtype == torch.float64 and scores.is_contiguous()) else scores.to(torch.float64).contiguous()
    b = sens if (sens.dtype == torch.float64 and sens.is_contiguous()) else sens.to(torch.float64).contiguous()
    idx = idx.to(torch.int64).contiguous()
    out = torch.empty(8, dtype=torch.float64, device=s.device)
    with torch.cuda.device(s.device):
        check(lib.mdg_select_margin(s.data_ptr(), b.data_ptr(), idx.data_ptr() if idx.numel() else None, s.numel(), idx.numel(),
                                    float(eps), out.data_ptr(), _stream(s)), "mdg_select_margin")
    return out


def decode_margin(out8, eps: float) -> dict:
    """Host-side reading of select_margin's 8 numbers (a list / CPU tensor): the relative margin of the selection threshold, the
    bound on what a perturbation of relative size eps can do to a score there, and whether the selected set is certified."""
    v = [float(x) for x in out8]
    s_k, s_k1 = v[0], v[1]
    finite = s_k > float("-inf") and s_k1 < float("inf")
    margin = (s_k1 - s_k) / abs(s_k) if finite and s_k != 0 else float("inf")
    spread = v[4] + v[5]
    return {"margin": margin, "score_bound": eps * max(v[4], v[5]) / abs(s_k) if finite and s_k != 0 else 0.0,
            "eps": eps, "eps_certifiable": (s_k1 - s_k) / spread if finite and spread > 0 else float("inf"),
            "scores_at_risk": int(v[6]), "certified": bool(v[7] == 1.0), **{k: x for k, x in zip(MARGIN_FIELDS[:4], v[:4])}}


def gather_rows(W: torch.Tensor, rows: torch.Tensor) -> torch.Tensor:
    """W[rows, :] for a 2-byte dtype (bf16/f16) row-major matrix."""
    _need_gpu(W, rows)
    lib = _lib.load()
    if W.element_size() != 2 or W.stride(1) != 1:
        raise ValueError("gather_rows needs a 2-byte dtype with unit column stride")
    rows = rows.to(torch.int64).contiguous()
    out = torch.empty(rows.numel(), W.shape[1], dtype=W.dtype, device=W.device)
    with torch.cuda.device(W.device):
        check(lib.mdg_gather_rows_16(W.data_ptr(), W.stride(0), rows.data_ptr(), rows.numel(), W.shape[1],
                                     out.data_ptr(), out.stride(0), _stream(W)), "mdg_gather_rows_16")
    return out


def nystrom_down(Cm: torch.Tensor, idx: torch.Tensor, W_down: torch.Tensor, eps: float = 1e-6,
                 want_f64: bool = False):
    """down' [d, r] bf16 = ((C[idx,idx] + eps I)^-1 C[idx,:] W_down^T)^T;  W_down: [d, n], bf16 as is, any other
    dtype widened exactly to fp64 (what the reference's .to(float64) does; bf16 would lose bits of an fp16 weight)."""
    _need_gpu(Cm, idx, W_down)
    lib = _lib.load()
    W_down = _as_weight(W_down)
    n, r, d = Cm.shape[0], idx.numel(), W_down.shape[0]
    idx = idx.to(torch.int64).contiguous()
    out = torch.empty(d, r, dtype=torch.bfloat16, device=Cm.device)
    f64 = torch.empty(r, d, dtype=torch.float64, device=Cm.device) if want_f64 else None
    nbytes = lib.mdg_nystrom_down_ws_bytes(n, r, d)
    ws, wsp = _ws(nbytes, Cm.device)
    with torch.cuda.device(Cm.device):
        if NYSTROM_OVERLAP:
            # the gathered cross product on a side stream beside the factorisation of C_kk (independent; the factorisation's
            # 128-column steps leave most of the chip idle between their GEMMs): 36 -> 27 ms for the two at Llama-3-8B shapes
            main = torch.cuda.current_stream(Cm.device)
            side = _side_stream(Cm.device, "nystrom", beside=main)
            side.wait_stream(main)                       # (inputs and workspace were produced / allocated on `main`)
            fork, join = torch.cuda.Event(), torch.cuda.Event()
            fork.record(main)                            # materialise the HIP events; the library re-records them
            join.record(main)
            check(lib.mdg_nystrom_down_overlapped(Cm.data_ptr(), n, Cm.stride(0), idx.data_ptr(), r, W_down.data_ptr(), d,
                                                  W_down.stride(0), _DT[W_down.dtype], float(eps), out.data_ptr(), out.stride(0),
                                                  _p(f64), wsp, nbytes, side.cuda_stream, fork.cuda_event, join.cuda_event,
                                                  main.cuda_stream), "mdg_nystrom_down_overlapped")
            for t in (Cm, W_down, idx, ws):
                t.record_stream(side)                    # read (workspace: written) there
        else:
            check(lib.mdg_nystrom_down(Cm.data_ptr(), n, Cm.stride(0), idx.data_ptr(), r, W_down.data_ptr(), d,
                                       W_down.stride(0), _DT[W_down.dtype], float(eps), out.data_ptr(), out.stride(0), _p(f64), wsp, nbytes,
                                       _stream(Cm)), "mdg_nystrom_down")
    return (out, f64) if want_f64 else out


# ------------------------------------------------------------------ QK / VO
def qk_select(cov_q: torch.Tensor, cov_k: torch.Tensor, rank: int, mode: int, ridge_q: float, ridge_k: float):
    """Returns (mask [n_kv, rank] int64, q_rows [n_heads*rank], k_rows [n_kv*rank])."""
    _need_gpu(cov_q, cov_k)
    lib = _lib.load()
    cq = cov_q.to(torch.float64).contiguous()
    ck = cov_k.to(torch.float64).contiguous()
    n_heads, hd, _ = cq.shape
    n_kv = ck.shape[0]
    dev = cq.device
    mask = torch.empty(n_kv, rank, dtype=torch.int64, device=dev)
    q_rows = torch.empty(n_heads * rank, dtype=torch.int64, device=dev)
    k_rows = torch.empty(n_kv * rank, dtype=torch.int64, device=dev)
    with torch.cuda.device(dev):
        check(lib.mdg_qk_select(cq.data_ptr(), ck.data_ptr(), n_heads, n_kv, hd, float(ridge_q), float(ridge_k), rank,
                                mode, mask.data_ptr(), q_rows.data_ptr(), k_rows.data_ptr(), _stream(cq)),
              "mdg_qk_select")
    return mask, q_rows, k_rows


def vo_compress(cov_x: torch.Tensor, W_v: torch.Tensor, W_o: torch.Tensor, n_heads: int, n_kv: int, hd: int, rank: int,
                ridge: float, want_f64: bool = False):
    """Returns (v_proj [n_kv*rank, d] bf16, o_proj [d, n_heads*rank] bf16[, v_f64, o_f64])."""
    _need_gpu(cov_x, W_v, W_o)
    lib = _lib.load()
    Wv, Wo = _as_weight(W_v), _as_weight(W_o)
    if Wv.dtype != Wo.dtype:
        Wv, Wo = Wv.to(torch.float64), Wo.to(torch.float64)
    Cx = cov_x if (cov_x.dtype == torch.float64 and cov_x.stride(1) == 1) else cov_x.to(torch.float64).contiguous()
    d = Cx.shape[0]
    dev = Cx.device
    v_out = torch.empty(n_kv * rank, d, dtype=torch.bfloat16, device=dev)
    o_out = torch.empty(d, n_heads * rank, dtype=torch.bfloat16, device=dev)
    v64 = torch.empty(n_kv * rank, d, dtype=torch.float64, device=dev) if want_f64 else None
    o64 = torch.empty(d, n_heads * rank, dtype=torch.float64, device=dev) if want_f64 else None
    nbytes = lib.mdg_vo_compress_ws_bytes(d, n_heads, n_kv, hd)
    ws, wsp = _ws(nbytes, dev)
    with torch.cuda.device(dev):
        check(lib.mdg_vo_compress(Cx.data_ptr(), d, Cx.stride(0), Wv.data_ptr(), Wv.stride(0), Wo.data_ptr(),
                                  Wo.stride(0), _DT[Wv.dtype], n_heads, n_kv, hd, rank, float(ridge), v_out.data_ptr(),
                                  v_out.stride(0), o_out.data_ptr(), o_out.stride(0), _p(v64), _p(o64), wsp, nbytes,
                                  _stream(Cx)), "mdg_vo_compress")
    return (v_out, o_out, v64, o64) if want_f64 else (v_out, o_out)


def rope_gather(x: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor, mask: Optional[torch.Tensor], n_heads: int,
                n_kv: int, head_dim: int, norm_weight: Optional[torch.Tensor] = None, eps: float = 1e-6) -> torch.Tensor:
    """Rotary embedding of a compressed q / k projection.  x: [B, T, n_heads*r] (bf16 / f16 / f32, last dim
    contiguous); cos, sin: [B or 1, T, head_dim]; mask: int64 [n_kv, r] or None; returns [B, n_heads, T, r].
    With `norm_weight` ([head_dim]) the Qwen3 masked RMSNorm runs first (DenseQwenRebuild.py:262-286)."""
    _need_gpu(x, cos, sin)
    lib = _lib.load()
    B, T, width = x.shape
    r = width // n_heads
    if r * n_heads != width:
        raise ValueError(f"rope_gather: projection width {width} is not a multiple of n_heads={n_heads}")
    if x.stride(2) != 1 or x.stride(0) != T * x.stride(1):
        x = x.contiguous()
    cos = cos.to(x.dtype).contiguous()
    sin = sin.to(x.dtype).contiguous()
    if cos.shape != sin.shape or cos.shape[-2:] != (T, head_dim) or cos.shape[0] not in (1, B):
        raise ValueError(f"rope_gather: cos/sin {tuple(cos.shape)} / {tuple(sin.shape)} do not fit [B or 1, {T}, {head_dim}]")
    if mask is not None:
        if mask.dtype != torch.int64 or tuple(mask.shape) != (n_kv, r):
            raise ValueError(f"rope_gather: mask must be int64 [{n_kv}, {r}], got {mask.dtype} {tuple(mask.shape)}")
        mask = mask.contiguous()
    if norm_weight is not None:
        norm_weight = norm_weight.detach().to(x.dtype).contiguous()
    out = torch.empty(B, n_heads, T, r, dtype=x.dtype, device=x.device)
    with torch.cuda.device(x.device):
        check(lib.mdg_rope_gather(x.data_ptr(), _DT[x.dtype], x.stride(1), B, T, n_heads, n_kv, r, head_dim,
                                  cos.data_ptr(), sin.data_ptr(), 0 if cos.shape[0] == 1 else T * head_dim, _p(mask),
                                  _p(norm_weight), float(eps), out.data_ptr(), _stream(x)), "mdg_rope_gather")
    return out


def cast_transpose(x: torch.Tensor) -> torch.Tensor:
    """bf16(x^T) for an fp64 matrix, with torch's double->float->bf16 rounding."""
    _need_gpu(x)
    lib = _lib.load()
    out = torch.empty(x.shape[1], x.shape[0], dtype=torch.bfloat16, device=x.device)
    with torch.cuda.device(x.device):
        check(lib.mdg_cast_transpose_f64_bf16(x.data_ptr(), x.shape[0], x.shape[1], x.stride(0), out.data_ptr(),
                                              out.stride(0), _stream(x)), "mdg_cast_transpose_f64_bf16")
    return out


def probe_mfma_f64(iters: int = 4096) -> float:
    """Measured fp64-MFMA issue rate of this device in TFLOP/s (register-resident operands)."""
    lib = _lib.load()
    out = C.c_double(0.0)
    check(lib.mdg_probe_mfma_f64(iters, C.byref(out), torch.cuda.current_stream().cuda_stream), "mdg_probe_mfma_f64")
    return out.value


def probe_mfma_i8(iters: int = 200000, random_operands: bool = True) -> float:
    """Measured int8-MFMA rate of this device in TOP/s from register-resident operands that change every MFMA: all zero (full
    clock) or random bytes (power-capped clock) -- the ceiling of any int8 kernel on such data."""
    lib = _lib.load()
    out = C.c_double(0.0)
    check(lib.mdg_probe_mfma_i8(iters, int(random_operands), C.byref(out), torch.cuda.current_stream().cuda_stream),
          "mdg_probe_mfma_i8")
    return out.value


def device_info(device: int = 0):
    lib = _lib.load()
    name = C.create_string_buffer(64)
    n_cu = C.c_int(0)
    hbm = C.c_int64(0)
    check(lib.mdg_device_info(device, name, 64, C.byref(n_cu), C.byref(hbm)), "mdg_device_info")
    return name.value.decode(), n_cu.value, hbm.value
