"""ctypes bindings of libsfhip.so (the C ABI declared in include/sfhip.h).

Every wrapper takes torch tensors that already live on the GPU, checks what the kernels
assume about them (device, dtype, contiguity, shape) and enqueues the kernel on torch's
current HIP stream.  Nothing here falls back to another implementation: if the library is
missing or a tensor is not on the GPU the call raises.
"""
from __future__ import annotations

import ctypes
import os
from typing import List, Optional, Sequence

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# SF_LIB: another build of the same library (diagnostic / A-B timing builds under tools/diag)
LIB_PATH = os.environ.get("SF_LIB") or os.path.join(_HERE, "libsfhip.so")

F32, BF16 = 0, 1
OUT_NHWC, OUT_JI, OUT_IJ, OUT_OIHW = 0, 1, 2, 3
ACT_NONE, ACT_RELU, ACT_LRELU = 0, 1, 2

_c_int, _c_ll, _c_f, _c_d, _c_vp = ctypes.c_int, ctypes.c_longlong, ctypes.c_float, ctypes.c_double, ctypes.c_void_p


class ConvGeom(ctypes.Structure):
    _fields_ = [("N", _c_int), ("IH", _c_int), ("IW", _c_int), ("C", _c_int),
                ("MH", _c_int), ("MW", _c_int), ("sh", _c_int), ("sw", _c_int),
                ("OH", _c_int), ("OW", _c_int),
                ("osh", _c_int), ("osw", _c_int), ("ooy", _c_int), ("oox", _c_int),
                ("ntaps", _c_int),
                ("dy", _c_int * 16), ("dx", _c_int * 16), ("tap_rs", _c_int * 16),
                ("cin_real", _c_int), ("RS", _c_int)]


class GemmDesc(ctypes.Structure):
    _fields_ = [("dtype", _c_int), ("a_kmajor", _c_int), ("b_kmajor", _c_int), ("b_gather", _c_int),
                ("I", _c_ll), ("J", _c_ll), ("R", _c_ll),
                ("lda", _c_ll), ("ldb", _c_ll), ("ldo", _c_ll),
                ("nsplit", _c_int), ("out_mode", _c_int), ("out_dtype", _c_int), ("act", _c_int),
                ("aux_mode", _c_int), ("g", ConvGeom), ("split_stride", _c_ll)]


_lib = None


def lib():
    """The loaded shared library; raises if it has not been built (no fallback path exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: build it with __graft_entry__.build() "
                               f"(make -C {os.path.join(_HERE, 'csrc')})")
        L = ctypes.CDLL(LIB_PATH)
        L.sf_last_error.restype = ctypes.c_char_p
        L.sf_transformer_save_floats.restype = ctypes.c_size_t
        L.sf_lstm_save_floats.restype = ctypes.c_size_t
        L.sf_wgrad3x3_workspace_bytes.restype = ctypes.c_size_t
        L.sf_conv3x3_staged_weight_bytes.restype = ctypes.c_size_t
        L.sf_conv4x4s2_staged_weight_bytes.restype = ctypes.c_size_t
        L.sf_wgrad4x4s2_workspace_bytes.restype = ctypes.c_size_t
        L.sf_conv_k1_fwd_workspace_bytes.restype = ctypes.c_size_t
        L.sf_enc1_bwd_workspace_bytes.restype = ctypes.c_size_t
        L.sf_enc1_codes_bytes.restype = ctypes.c_size_t
        L.sf_enc1_gram_workspace_bytes.restype = ctypes.c_size_t
        L.sf_dconv1_bwd_weight_workspace_bytes.restype = ctypes.c_size_t
        L.sf_date_image_metrics_workspace_bytes.restype = ctypes.c_size_t
        L.sf_gru_save_floats.restype = ctypes.c_size_t
        L.sf_attn_save_floats.restype = ctypes.c_size_t
        _lib = L
    return _lib


EXPORTS = ["sf_version", "sf_last_error", "sf_gemm", "sf_gemm_batch", "sf_conv3x3", "sf_conv3x3_dot", "sf_conv3x3_dot_ok", "sf_conv3x3_tile_shape", "sf_conv3x3_staged_weight_bytes", "sf_conv3x3_stage_weights", "sf_conv4x4s2", "sf_conv4x4s2_staged_weight_bytes", "sf_conv4x4s2_stage_weights", "sf_wgrad3x3", "sf_set_wgrad_scheduling", "sf_wgrad3x3_workspace_bytes", "sf_wgrad4x4s2", "sf_wgrad4x4s2_workspace_bytes", "sf_nchw_to_nhwc", "sf_nchw_to_nhwc_checked", "sf_nchw_bf16_to_nhwc", "sf_nhwc_to_nchw", "sf_pack_conv_weight",
           "sf_cast", "sf_bn_stats", "sf_bn_finalize", "sf_bn_relu_maxpool_fwd", "sf_bn_relu_avgpool_fwd",
           "sf_bn_lrelu_fwd", "sf_bn_bwd_reduce", "sf_bn_bwd_apply", "sf_bn_relu_avgpool_fwd_moments", "sf_bn_bwd_reduce_pooled", "sf_channel_sum", "sf_colsum_f32",
           "sf_bilinear_fwd", "sf_bilinear_bwd", "sf_upconv3_fwd", "sf_upconv3_bwd", "sf_enc1_stats", "sf_enc1_codes_bytes", "sf_enc1_fwd", "sf_enc1_bwd", "sf_enc1_gram", "sf_enc1_gram_workspace_bytes", "sf_enc1_stats_gram", "sf_enc1_bwd_workspace_bytes", "sf_dconv1_fwd", "sf_dconv1_bwd_data", "sf_dconv1_bwd_weight", "sf_dconv1_bwd_weight_ws", "sf_dconv1_bwd_weight_workspace_bytes",
           "sf_conv_k1_fwd", "sf_conv_k1_fwd_ws", "sf_conv_k1_fwd_workspace_bytes", "sf_conv_k1_bwd_data", "sf_conv_k1_bwd_weight", "sf_conv_k1_bwd_fused", "sf_transformer_save_floats",
           "sf_transformer_save_offset", "sf_transformer_layer_fwd", "sf_transformer_layer_bwd", "sf_add_pe", "sf_mean_t_fwd", "sf_mean_t_bwd",
           "sf_lstm_save_floats", "sf_lstm_fwd", "sf_lstm_bwd", "sf_lonlat_to_pixel", "sf_image_station_loss", "sf_image_station_loss_fwd", "sf_image_station_loss_bwd", "sf_image_station_loss_static",
           "sf_bce_logits", "sf_adam", "sf_adam_begin", "sf_adam_guarded", "sf_nan_inf_flag", "sf_dataset_plane_stats", "sf_dataset_transform",
           "sf_gather_frames", "sf_date_image_metrics", "sf_date_image_metrics_workspace_bytes", "sf_bn_relu_fwd",
           "sf_relu_dropout_fwd", "sf_relu_dropout_bwd", "sf_gru_save_floats", "sf_gru_fwd", "sf_gru_bwd",
           "sf_attn_save_floats", "sf_attn_fwd", "sf_attn_bwd", "sf_token_outer4"]


def _check(rc: int, who: str):
    if rc != 0:
        raise RuntimeError(f"{who} failed ({rc}): {lib().sf_last_error().decode()}")


def _stream():
    return _c_vp(torch.cuda.current_stream().cuda_stream)


def _p(t: Optional[torch.Tensor]):
    return _c_vp(t.data_ptr()) if t is not None else _c_vp(0)


def code_of(dtype: torch.dtype) -> int:
    if dtype == torch.float32:
        return F32
    if dtype == torch.bfloat16:
        return BF16
    raise TypeError(f"unsupported dtype {dtype}")


def torch_dtype(code: int) -> torch.dtype:
    return torch.bfloat16 if code == BF16 else torch.float32


def epc(code: int) -> int:
    return 8 if code == BF16 else 4


def _dev(*ts, dtypes=None):
    for i, t in enumerate(ts):
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("streamflow HIP ops need GPU tensors (no CPU path exists)")
        if not t.is_contiguous():
            raise RuntimeError("streamflow HIP ops need contiguous tensors")
        if dtypes is not None and dtypes[i] is not None and t.dtype != dtypes[i]:
            raise TypeError(f"argument {i}: expected {dtypes[i]}, got {t.dtype}")


def roundup(v: int, m: int) -> int:
    return (v + m - 1) // m * m


# ------------------------------------------------------------------------------ gemm / conv
# Debug switch: route bf16 3x3 convolutions through the generic implicit-GEMM kernel instead of
# the LDS-patch kernels (used by tests to cross-check the two implementations).
FORCE_GENERIC_CONV = False

# Launch timing for the benchmark's roofline line: tags listed in TIMED get a HIP event pair
# recorded on the launch stream around each sf_gemm call carrying that tag.
LOSS_WS_DOUBLES = 4096  # SF_LOSS_WS_DOUBLES of include/sfhip.h
BN_WS_PARTS = 1024      # SF_BN_WS_PARTS


def bn_ws(C: int, device) -> torch.Tensor:
    """Workspace of the BatchNorm / channel-sum reductions: SF_BN_WS_DOUBLES(C) doubles; the folded
    sums end up in the first 2*C."""
    return torch.empty(2 * C * (1 + BN_WS_PARTS), device=device, dtype=torch.float64)
TIMED: dict = {}


def timed_ms(tag: str):
    """(number of launches, mean milliseconds per launch) for a timed tag; synchronises."""
    ev = TIMED.get(tag, [])
    if not ev:
        return 0, 0.0
    torch.cuda.synchronize()
    return len(ev), sum(r[0].elapsed_time(r[1]) for r in ev) / len(ev)


def timed_bytes(tag: str):
    """Mean algorithmic HBM bytes per launch of a timed tag (input + output + packed weights, each
    touched once), 0.0 when the tag's records carry no byte count."""
    ev = [r for r in TIMED.get(tag, []) if len(r) > 3]
    return sum(r[3] for r in ev) / len(ev) if ev else 0.0


def timed_tflops(tag: str):
    """(number of launches, mean ms per launch, TFLOP/s = recorded flops / recorded time)."""
    ev = TIMED.get(tag, [])
    if not ev:
        return 0, 0.0, 0.0
    torch.cuda.synchronize()
    ms = sum(r[0].elapsed_time(r[1]) for r in ev)
    fl = sum(r[2] for r in ev if len(r) > 2)
    return len(ev), ms / len(ev), (fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0)


def gemm(desc: GemmDesc, A, B, out, bias=None, aux=None, tag: Optional[str] = None):
    _dev(A, B, out, bias, aux)
    rec = TIMED.get(tag) if tag is not None else None
    if rec is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    _check(lib().sf_gemm(ctypes.byref(desc), _p(A), _p(B), _p(bias), _p(aux), _p(out), _stream()), "sf_gemm")
    if rec is not None:
        e1.record()
        rec.append((e0, e1))


def conv_geom(N, IH, IW, C, MH, MW, sh, sw, OH, OW, taps, osh=1, osw=1, ooy=0, oox=0, tap_rs=None,
              cin_real=0, RS=0) -> ConvGeom:
    g = ConvGeom()
    g.N, g.IH, g.IW, g.C, g.MH, g.MW, g.sh, g.sw, g.OH, g.OW = N, IH, IW, C, MH, MW, sh, sw, OH, OW
    g.osh, g.osw, g.ooy, g.oox = osh, osw, ooy, oox
    g.ntaps = len(taps)
    for t, (dy, dx) in enumerate(taps):
        g.dy[t], g.dx[t] = dy, dx
    if tap_rs is not None:
        for t, v in enumerate(tap_rs):
            g.tap_rs[t] = v
    g.cin_real, g.RS = cin_real, RS
    return g


def conv_fprop(x, wp, out, geom: ConvGeom, K: int, bias=None, act=ACT_NONE, aux=None, aux_mode=0, tag=None):
    """out[NHWC] = act(conv(x) + bias) [* mask(aux)]; x NHWC, wp packed [K][taps][C]."""
    d = GemmDesc()
    d.dtype = code_of(x.dtype)
    d.a_kmajor = d.b_kmajor = 0
    d.b_gather = 1
    d.I, d.J, d.R = K, geom.N * geom.MH * geom.MW, geom.ntaps * geom.C
    d.lda, d.ldb, d.ldo = geom.ntaps * geom.C, geom.C, out.shape[-1]
    d.nsplit, d.out_mode, d.out_dtype, d.act, d.aux_mode = 1, OUT_NHWC, code_of(out.dtype), act, aux_mode
    d.g = geom
    gemm(d, wp, x, out, bias, aux, tag)


def conv_fprop_batch(problems):
    """Up to four conv_fprop calls (x, wp, out, geom, K, aux, aux_mode) in one launch."""
    n = len(problems)
    descs = (GemmDesc * n)()
    A, B, O, X = (ctypes.c_void_p * n)(), (ctypes.c_void_p * n)(), (ctypes.c_void_p * n)(), (ctypes.c_void_p * n)()
    any_aux = False
    for i, (x, wp, out, geom, K, aux, aux_mode) in enumerate(problems):
        _dev(x, wp, out, aux)
        d = descs[i]
        d.dtype = code_of(x.dtype)
        d.a_kmajor, d.b_kmajor, d.b_gather = 0, 0, 1
        d.I, d.J, d.R = K, geom.N * geom.MH * geom.MW, geom.ntaps * geom.C
        d.lda, d.ldb, d.ldo = geom.ntaps * geom.C, geom.C, out.shape[-1]
        d.nsplit, d.out_mode, d.out_dtype, d.act, d.aux_mode = 1, OUT_NHWC, code_of(out.dtype), ACT_NONE, aux_mode
        d.g = geom
        A[i], B[i], O[i] = wp.data_ptr(), x.data_ptr(), out.data_ptr()
        X[i] = aux.data_ptr() if aux is not None else None
        any_aux = any_aux or aux is not None
    _check(lib().sf_gemm_batch(n, descs, A, B, None, X if any_aux else None, O, _stream()), "sf_gemm_batch")


def conv3x3_dot_ok(x, K: int, out) -> bool:
    """Whether conv3x3 can also emit the 1x1 projection of its output to one channel (`dot=`): the
    shape has to run on the kernel that keeps its weights resident in LDS."""
    return (x.dtype == torch.bfloat16 and not FORCE_GENERIC_CONV and FUSE_OUTPUT_CONV and
            bool(lib().sf_conv3x3_dot_ok(x.shape[-1], K, out.shape[-1], code_of(out.dtype))))


def conv3x3(x, wp, out, K: int, dgrad: bool = False, bias=None, act=ACT_NONE, aux=None, aux_mode=0, tag=None,
            flop_c: Optional[int] = None, flop_k: Optional[int] = None, dot=None):
    """3x3 stride-1 pad-1 conv (dgrad=False) or its data gradient (dgrad=True; wp packed
    transposed).  bf16 tensors with C % 16 == 0 take the LDS-patch kernel, everything else the
    generic implicit GEMM.  dot = (w f32 [K], b f32 [1], y f32 [N,1,H,W]): y = b + sum_k w[k] *
    out[..., k] from the epilogue (only where conv3x3_dot_ok says so)."""
    N, H, W, C = x.shape
    if dot is not None and not conv3x3_dot_ok(x, K, out):
        raise RuntimeError("conv3x3: this shape cannot fuse the output projection")
    if dgrad:
        taps = [(1 - r, 1 - s) for r in range(3) for s in range(3)]
    else:
        taps = [(r - 1, s - 1) for r in range(3) for s in range(3)]
    if x.dtype == torch.bfloat16 and C % 16 == 0 and not FORCE_GENERIC_CONV:
        _dev(x, wp, out, bias, aux)
        rec = TIMED.get(tag) if tag is not None else None
        if rec is None:
            # every launch of the persistent kernel, by variant (same rule as sf_conv3x3's dispatch)
            wide = K <= 1024 and (K > 64 or C % 32 != 0)
            bf16_out = out.dtype == torch.bfloat16
            if wide:      # plain bf16 output in whole lines, or the run-time epilogue
                plain = (bias is None and act == ACT_NONE and aux_mode == 0 and bf16_out and K % 128 == 0
                         and out.shape[-1] % 64 == 0 and out.data_ptr() % 128 == 0)
                name = "conv3_dma_wide_kernel" if plain else "conv3_dma_wide_any_kernel"
            elif C == 64:
                lines = K == 64 and out.shape[-1] == 64 and bf16_out and out.data_ptr() % 128 == 0
                if lines and bias is not None and act == ACT_RELU and aux_mode == 0 and dot is not None:
                    name = "conv3_dma_res_kernel"
                elif lines and bias is None and act == ACT_NONE and aux_mode == 1 and dot is None:
                    name = "conv3_dma_res_mask_kernel"
                else:
                    name = "conv3_dma_res_any_kernel"
            else:
                name = "conv3_dma_kernel"
            rec = TIMED.get(name) if (wide or C % 32 == 0) else None
        if rec is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        dy = (_c_int * 9)(*[t[0] for t in taps])
        dx = (_c_int * 9)(*[t[1] for t in taps])
        staged = getattr(wp, "_sf_staged", False)
        if dot is not None:
            dw_, db_, dout_ = dot
            _dev(dw_, db_, dout_, dtypes=[torch.float32, torch.float32, torch.float32])
            if dw_.numel() != K or dout_.numel() != N * H * W:
                raise RuntimeError("conv3x3: projection operands do not match the conv's shape")
            _check(lib().sf_conv3x3_dot(_p(x), _p(wp), _p(bias), _p(aux), _p(out), N, H, W, C, K, out.shape[-1], dy, dx,
                                        act, aux_mode, code_of(out.dtype), _p(dw_), _p(db_), _p(dout_), _stream()),
                   "sf_conv3x3_dot")
        else:
            _check(lib().sf_conv3x3(_p(x), _p(wp), _p(bias), _p(aux), _p(out), N, H, W, C, K, out.shape[-1], dy, dx, act,
                                    aux_mode, code_of(out.dtype), int(staged), _stream()), "sf_conv3x3")
        if rec is not None:
            e1.record()
            # algorithmic flops: real (unpadded) channel counts when the caller states them
            rec.append((e0, e1, 2.0 * N * H * W * (flop_c or C) * (flop_k or K) * 9,
                        x.numel() * x.element_size() + out.numel() * out.element_size() + wp.numel() * wp.element_size()))
        return
    geom = conv_geom(N, H, W, C, H, W, 1, 1, H, W, taps, tap_rs=list(range(9)))
    conv_fprop(x, wp, out, geom, K, bias=bias, act=act, aux=aux, aux_mode=aux_mode, tag=tag)


def s2_tap_order(dgrad: bool) -> List[int]:
    """Filter positions r*4+s in the order sf_conv4x4s2 expects its 16 packed taps: class
    (py, px) = parity of the pixel read (forward) / written (data gradient), then (a, b)."""
    out = []
    for py in (0, 1):
        for px in (0, 1):
            for a in (0, 1):
                for b in (0, 1):
                    out.append((3 - py - 2 * a) * 4 + (3 - px - 2 * b) if dgrad else (2 * a + py) * 4 + (2 * b + px))
    return out


def conv4x4s2_ok(x: torch.Tensor, K: int, ldo: int) -> bool:
    """bf16 NHWC with a multiple of 16 reduction channels and 8-channel output groups."""
    return (x.dtype == torch.bfloat16 and x.shape[-1] % 16 == 0 and K % 8 == 0 and ldo % 8 == 0 and K <= 1024
            and x.shape[1] * x.shape[2] * x.shape[3] * 2 < 2 ** 31 - 1 and not FORCE_GENERIC_CONV)


def conv4x4s2_weights(w: torch.Tensor, dgrad: bool) -> torch.Tensor:
    """f32 OIHW 4x4 weights -> the staged bf16 operand of conv4x4s2 (forward or data gradient)."""
    def build():
        O, I = w.shape[0], w.shape[1]
        rows, cols = (I, O) if dgrad else (O, I)
        wp = pack_conv_weight(w.detach().contiguous(), torch.bfloat16, s2_tap_order(dgrad), dgrad, rows, cols)
        nbytes = int(lib().sf_conv4x4s2_staged_weight_bytes(cols, rows))
        if nbytes == 0:
            raise RuntimeError("conv4x4s2_weights: the reduction channel count must be a multiple of 16")
        ws = torch.empty(nbytes // 2, device=w.device, dtype=torch.bfloat16)
        _check(lib().sf_conv4x4s2_stage_weights(_p(wp), _p(ws), cols, rows, _stream()), "sf_conv4x4s2_stage_weights")
        return ws

    return derived(w, ("s2", bool(dgrad)), build)


def conv4x4s2(x, ws, out, K: int, dgrad: bool = False, bias=None, act=ACT_NONE, aux=None, aux_mode=0):
    """4x4 stride-2 pad-1 conv (x = layer input) or its data gradient (x = gradient of the conv
    output, out = gradient of the conv input); ws from conv4x4s2_weights(w, dgrad)."""
    _dev(x, ws, out, bias, aux, dtypes=[torch.bfloat16, torch.bfloat16, torch.bfloat16, torch.float32, torch.bfloat16])
    N, IH, IW, C = x.shape
    No, OH, OW, ldo = out.shape
    if No != N or ws.numel() * 2 != int(lib().sf_conv4x4s2_staged_weight_bytes(C, K)):
        raise RuntimeError("conv4x4s2: operand shapes do not match")
    if aux is not None and aux.shape != out.shape:
        raise RuntimeError("conv4x4s2: the mask operand must have the output's shape")
    rec = TIMED.get("conv_s2_dgrad_kernel" if dgrad else "conv_s2_fwd_kernel")
    if rec is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    _check(lib().sf_conv4x4s2(_p(x), _p(ws), _p(bias), _p(aux), _p(out), N, IH, IW, C, OH, OW, K, ldo, int(dgrad), act,
                              aux_mode, _stream()), "sf_conv4x4s2")
    if rec is not None:
        e1.record()
        pix = N * (IH * IW if dgrad else OH * OW)
        rec.append((e0, e1, 2.0 * pix * C * K * 16))


def conv3x3_stage_weights(wp: torch.Tensor, C: int, K: int) -> torch.Tensor:
    """Packed [K][9][C] bf16 weights -> the staged form sf_conv3x3 copies in contiguous pieces, when
    the shape runs on the kernel that takes it; otherwise wp itself.  The result carries a marker
    that conv3x3 reads."""
    if wp.dtype != torch.bfloat16 or FORCE_GENERIC_CONV or not STAGE_CONV_WEIGHTS:
        return wp
    nbytes = int(lib().sf_conv3x3_staged_weight_bytes(C, K))
    if nbytes == 0:
        return wp
    _dev(wp)
    out = torch.empty(nbytes // 2, device=wp.device, dtype=torch.bfloat16)
    _check(lib().sf_conv3x3_stage_weights(_p(wp), _p(out), C, K, _stream()), "sf_conv3x3_stage_weights")
    out._sf_staged = True
    return out


STAGE_CONV_WEIGHTS = os.environ.get("SF_STAGE_CONV_WEIGHTS", "1") != "0"   # debug switch for A/B timing
FUSE_OUTPUT_CONV = os.environ.get("SF_FUSE_OUTPUT_CONV", "1") != "0"       # debug switch for A/B timing


_UPLOAD_STREAMS: dict = {}


def upload(host: torch.Tensor, device) -> torch.Tensor:
    """A small host array -> device memory without making the host wait.  A host-to-device copy
    queued on the compute stream can block the calling thread until the kernels queued before it
    have run (seen with pageable and with page-locked sources once a second stream is active), which
    costs the step the host's lead over the GPU.  So the array is staged in page-locked memory and
    copied on a stream of its own, which the compute stream then waits for."""
    dev = torch.device(device)
    if dev.type != "cuda":
        return host.to(dev)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    up = _UPLOAD_STREAMS.get(idx)
    if up is None:
        up = _UPLOAD_STREAMS[idx] = torch.cuda.Stream(device=idx)
    if not host.is_pinned():
        host = torch.empty(host.shape, dtype=host.dtype, pin_memory=True).copy_(host)
    cur = torch.cuda.current_stream(idx)
    with torch.cuda.stream(up):
        d = host.to(dev, non_blocking=True)
    cur.wait_stream(up)
    d.record_stream(cur)
    return d


def set_wgrad_scheduling(dynamic):
    """See sf_set_wgrad_scheduling: True / 1 when other kernels (RCCL's) share the GPU's CUs, 2 for the
    self-test in which every second workgroup leaves its tiles to the others."""
    _check(lib().sf_set_wgrad_scheduling(int(dynamic)), "sf_set_wgrad_scheduling")


def wgrad3x3(x, dout, dw, tag=None):
    """dw[OIHW f32] = weight gradient of a 3x3 stride-1 pad-1 conv; x / dout NHWC.  bf16 tensors
    take the LDS-patch kernel (deterministic slab reduction); f32 the generic GEMM with atomics."""
    N, H, W, Cx = x.shape
    K, Cin = dw.shape[0], dw.shape[1]
    taps = [(r - 1, s - 1) for r in range(3) for s in range(3)]
    if x.dtype == torch.bfloat16 and not FORCE_GENERIC_CONV:
        _dev(x, dout, dw)
        nbytes = int(lib().sf_wgrad3x3_workspace_bytes(N, H, W, K, Cin))
        ws = torch.empty(nbytes // 4, device=x.device, dtype=torch.float32)
        # same rule as sf_wgrad3x3's choice of block shape
        rec = TIMED.get(tag if tag is not None else ("wgrad_wide_kernel" if K > 64 else "wgrad3_dma_kernel"))
        if rec is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        _check(lib().sf_wgrad3x3(_p(x), _p(dout), _p(dw), _p(ws), ctypes.c_size_t(nbytes), N, H, W, Cx, dout.shape[-1],
                                 K, Cin, _stream()), "sf_wgrad3x3")
        if rec is not None:
            e1.record()
            rec.append((e0, e1, 2.0 * N * H * W * Cin * K * 9))
        return
    dw.zero_()
    geom = conv_geom(N, H, W, Cx, H, W, 1, 1, H, W, taps, tap_rs=list(range(9)), cin_real=Cin, RS=9)
    conv_wgrad(x, dout, dw, geom, dout.shape[-1], tag=tag)


def wgrad4x4s2(x, dout, dw, tag=None):
    """dw[OIHW f32] = weight gradient of a 4x4 stride-2 pad-1 conv; x [N,IH,IW,C], dout [N,OH,OW,K]."""
    N, IH, IW, Cx = x.shape
    OH, OW = dout.shape[1], dout.shape[2]
    K, Cin = dw.shape[0], dw.shape[1]
    if x.dtype == torch.bfloat16 and not FORCE_GENERIC_CONV:
        _dev(x, dout, dw)
        nbytes = int(lib().sf_wgrad4x4s2_workspace_bytes(N, IH, IW, K, Cin))
        ws = torch.empty(nbytes // 4, device=x.device, dtype=torch.float32)
        _check(lib().sf_wgrad4x4s2(_p(x), _p(dout), _p(dw), _p(ws), ctypes.c_size_t(nbytes), N, IH, IW, Cx,
                                   dout.shape[-1], K, Cin, _stream()), "sf_wgrad4x4s2")
        return
    dw.zero_()
    taps = [(r - 1, s - 1) for r in range(4) for s in range(4)]
    geom = conv_geom(N, IH, IW, Cx, OH, OW, 2, 2, OH, OW, taps, tap_rs=list(range(16)), cin_real=Cin, RS=16)
    conv_wgrad(x, dout, dw, geom, dout.shape[-1], tag=tag)


def conv_wgrad(x, dout, dw, geom: ConvGeom, K: int, nsplit: Optional[int] = None, tag=None):
    """dw[OIHW f32] += sum_pixels dout[pixel][k] * x[gather(pixel, tap)][c]; dw must be zeroed."""
    d = GemmDesc()
    d.dtype = code_of(x.dtype)
    d.a_kmajor = d.b_kmajor = 1
    d.b_gather = 1
    M = geom.N * geom.MH * geom.MW
    d.I, d.J, d.R = K, geom.ntaps * geom.C, M
    d.lda, d.ldb, d.ldo = dout.shape[-1], geom.C, dw.shape[0]
    if nsplit is None:
        tiles = ((K + 127) // 128) * ((geom.ntaps * geom.C + 127) // 128)
        nsplit = max(1, min(M // 256, (2048 + tiles - 1) // tiles))
    d.nsplit, d.out_mode, d.out_dtype = max(2, nsplit), OUT_OIHW, F32
    d.g = geom
    gemm(d, dout, x, dw, tag=tag)


def linear_gemm(A, B, out, I, J, R, lda, ldb, ldo, a_km, b_km, out_mode, nsplit=1, bias=None, split_stride=0):
    # (out may be bf16 for the JI layout without a split reduction)
    """split_stride > 0 (with nsplit > 1): `out` is a slab of nsplit partial results split_stride
    floats apart (plain stores, to be summed by the caller); 0: atomic accumulation into a zeroed out."""
    d = GemmDesc()
    d.split_stride = split_stride
    d.dtype = code_of(A.dtype)
    d.a_kmajor, d.b_kmajor, d.b_gather = int(a_km), int(b_km), 0
    d.I, d.J, d.R, d.lda, d.ldb, d.ldo = I, J, R, lda, ldb, ldo
    d.nsplit, d.out_mode, d.out_dtype = nsplit, out_mode, code_of(out.dtype)
    gemm(d, A, B, out, bias)


# ------------------------------------------------------------------------------ layout / packing
def nchw_to_nhwc(x: torch.Tensor, dtype: torch.dtype, cpad: int, flag: Optional[torch.Tensor] = None) -> torch.Tensor:
    """flag (int32 [1], optional): set to 1 when x holds Inf or NaN (sf_nchw_to_nhwc_checked)."""
    N, C, H, W = x.shape
    if flag is not None:
        x = x.float()
        _dev(x, flag, dtypes=[torch.float32, torch.int32])
        out = torch.empty((N, H, W, cpad), device=x.device, dtype=dtype)
        _check(lib().sf_nchw_to_nhwc_checked(_p(x), _p(out), code_of(dtype), N, C, H, W, cpad, _p(flag), _stream()),
               "sf_nchw_to_nhwc_checked")
        return out
    if x.dtype == torch.bfloat16 and dtype == torch.bfloat16 and cpad >= 64:
        _dev(x)
        out = torch.empty((N, H, W, cpad), device=x.device, dtype=dtype)
        _check(lib().sf_nchw_bf16_to_nhwc(_p(x), _p(out), N, C, H, W, cpad, _stream()), "sf_nchw_bf16_to_nhwc")
        return out
    x = x.float()
    _dev(x, dtypes=[torch.float32])
    out = torch.empty((N, H, W, cpad), device=x.device, dtype=dtype)
    _check(lib().sf_nchw_to_nhwc(_p(x), _p(out), code_of(dtype), N, C, H, W, cpad, _stream()), "sf_nchw_to_nhwc")
    return out


def nhwc_to_nchw(x: torch.Tensor, C: int) -> torch.Tensor:
    _dev(x)
    N, H, W, cpad = x.shape
    out = torch.empty((N, C, H, W), device=x.device, dtype=torch.float32)
    _check(lib().sf_nhwc_to_nchw(_p(x), code_of(x.dtype), _p(out), N, C, H, W, cpad, _stream()), "sf_nhwc_to_nchw")
    return out


def pack_conv_weight(w: torch.Tensor, dtype: torch.dtype, tap_rs: Sequence[int], transpose: bool,
                     rows_pad: int, cols_pad: int) -> torch.Tensor:
    _dev(w, dtypes=[torch.float32])
    O, I, R, S = w.shape
    out = torch.empty((rows_pad, len(tap_rs), cols_pad), device=w.device, dtype=dtype)
    arr = (_c_int * len(tap_rs))(*tap_rs)
    _check(lib().sf_pack_conv_weight(_p(w), _p(out), code_of(dtype), O, I, R * S, int(transpose), len(tap_rs), arr,
                                     rows_pad, cols_pad, _stream()), "sf_pack_conv_weight")
    return out


def cast(x: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    _dev(x, dtypes=[torch.float32])
    if dtype == torch.float32:
        return x
    out = torch.empty(x.shape, device=x.device, dtype=dtype)
    _check(lib().sf_cast(_p(x), _p(out), code_of(dtype), _c_ll(x.numel()), _stream()), "sf_cast")
    return out


# bf16 shadow copies of big f32 parameters (the 262144-wide Linear weights): written by the fused
# Adam kernel together with the parameter, so the bf16 kernels need no 134 MB cast per call.  A
# shadow counts as current while the parameter has not been written through torch since (its
# version counter) and still lives at the same address.
WEIGHT_EPOCH = 0     # bumped when parameters were rewritten behind torch's back and nobody kept track of which


def bump_weight_epoch():
    """Everything derived from any parameter is stale (after a write torch cannot see, e.g. through
    `.data`, when the caller does not know or care which parameters it touched)."""
    global WEIGHT_EPOCH
    WEIGHT_EPOCH += 1


def touch(params):
    """These parameters were rewritten behind torch's back (an optimiser kernel): whatever is derived
    from them is stale, the copies derived from other parameters stay valid."""
    for p in params:
        p.__dict__["_sf_epoch"] = p.__dict__.get("_sf_epoch", 0) + 1


def derived(owner: torch.Tensor, key, build):
    """A tensor computed from parameter `owner` alone (a packed / staged copy of a conv weight),
    kept on the parameter object until the parameter changes: in place through torch (its version
    counter), behind torch's back by an optimiser kernel (touch / bump_weight_epoch), or by moving
    elsewhere."""
    ver = (WEIGHT_EPOCH, owner.__dict__.get("_sf_epoch", 0), owner._version, owner.data_ptr())
    cache = owner.__dict__.setdefault("_sf_derived", {})
    hit = cache.get(key)
    if hit is not None and hit[0] == ver:
        return hit[1]
    t = build()
    cache[key] = (ver, t)
    return t


def _shadow_key(p: torch.Tensor):
    # the same staleness rules as derived(): writes through torch, writes announced by touch() /
    # bump_weight_epoch(), a move to another address
    return (WEIGHT_EPOCH, p.__dict__.get("_sf_epoch", 0), p._version, p.data_ptr(), tuple(p.shape))


def shadow_if_current(p: torch.Tensor) -> Optional[torch.Tensor]:
    sh = getattr(p, "_sf_shadow", None)
    return sh if sh is not None and getattr(p, "_sf_shadow_key", None) == _shadow_key(p) else None


def weight_shadow(p: torch.Tensor) -> torch.Tensor:
    """The current bf16 shadow of parameter p, (re)built by a cast when missing or stale."""
    sh = shadow_if_current(p)
    if sh is None:
        sh = cast(p.detach().contiguous(), torch.bfloat16)
        p._sf_shadow = sh
        p._sf_shadow_key = _shadow_key(p)
    return sh


def mark_shadow_current(p: torch.Tensor):
    """The shadow was just rewritten together with the parameter (sf_adam_guarded): call this AFTER
    the touch() that announces the parameter write, so that the key holds the new epoch."""
    p._sf_shadow_key = _shadow_key(p)


# ------------------------------------------------------------------------------ batch norm family
def bn_stats(x: torch.Tensor, M: int, C: int, ws: torch.Tensor):
    _dev(x, ws)
    if ws.dtype != torch.float64 or ws.numel() < 2 * C * (1 + BN_WS_PARTS):
        raise RuntimeError("bn_stats: the workspace must come from ops.bn_ws(C, device)")
    _check(lib().sf_bn_stats(_p(x), code_of(x.dtype), _c_ll(M), C, _p(ws), _stream()), "sf_bn_stats")


def bn_finalize(ws, M, C, Creal, gamma, beta, rm, rv, training, momentum=0.1, eps=1e-5):
    dev = gamma.device
    out = torch.empty((4, C), device=dev, dtype=torch.float32)
    _dev(gamma, beta, rm, rv, dtypes=[torch.float32] * 4)
    _check(lib().sf_bn_finalize(_p(ws), _c_ll(M), C, Creal, _p(gamma), _p(beta), _p(rm), _p(rv), _c_f(momentum),
                                _c_f(eps), int(training), _p(out[0]), _p(out[1]), _p(out[2]), _p(out[3]),
                                _stream()), "sf_bn_finalize")
    return out  # rows: scale, shift, mean, invstd


def bn_relu_maxpool_fwd(x, st):
    N, H, W, C = x.shape
    y = torch.empty((N, H // 2, W // 2, C), device=x.device, dtype=x.dtype)
    _check(lib().sf_bn_relu_maxpool_fwd(_p(x), code_of(x.dtype), _p(st[0]), _p(st[1]), _p(y), N, H, W, C, _stream()),
           "sf_bn_relu_maxpool_fwd")
    return y


def bn_relu_avgpool_fwd(x, st, Creal, OH, OW, out_dtype):
    N, H, W, C = x.shape
    y = torch.empty((N, Creal, OH, OW), device=x.device, dtype=out_dtype)
    _check(lib().sf_bn_relu_avgpool_fwd(_p(x), code_of(x.dtype), _p(st[0]), _p(st[1]), _p(y), code_of(out_dtype), N, H,
                                        W, C, Creal, OH, OW, _stream()), "sf_bn_relu_avgpool_fwd")
    return y


def avgpool_moments_ok(H: int, W: int, OH: int, OW: int) -> bool:
    """Shapes sf_bn_relu_avgpool_fwd_moments takes (LDS tile of the NCHW transpose, 16-bit counts)."""
    return OW <= 63 and OH <= 256 and (-(-H // OH) + 1) * (-(-W // OW) + 1) <= 65535


def bn_relu_avgpool_fwd_moments(x, st, Creal, OH, OW, out_dtype):
    """(y NCHW, pos u16 NHWC, xh NHWC): the pooled activation and, per bin, the count of positive
    pixels and the bin mean of [y > 0] * xhat, from which bn_backward forms its channel sums."""
    _dev(x, st)
    N, H, W, C = x.shape
    y = torch.empty((N, Creal, OH, OW), device=x.device, dtype=out_dtype)
    pos = torch.empty((N, OH, OW, C), device=x.device, dtype=torch.uint16)
    xh = torch.empty((N, OH, OW, C), device=x.device, dtype=x.dtype)
    _check(lib().sf_bn_relu_avgpool_fwd_moments(_p(x), code_of(x.dtype), _p(st[0]), _p(st[1]), _p(st[2]), _p(st[3]),
                                                _p(y), code_of(out_dtype), _p(pos), _p(xh), N, H, W, C, Creal, OH, OW,
                                                _stream()), "sf_bn_relu_avgpool_fwd_moments")
    return y, pos, xh


def enc1_ok(x: torch.Tensor, K: int) -> bool:
    """The fused first-layer kernels take bf16 NHWC input with 16 (padded) channels."""
    N, H, W, _ = x.shape
    return (x.dtype == torch.bfloat16 and x.shape[-1] == 16 and K in (128, 256) and H >= 2 and W >= 2
            and x.numel() * 2 < 2 ** 31 - 1 and N * (H // 2) * (W // 2) * K * 2 < 2 ** 31 - 1)


def enc1_stats(x, wp, K, ws):
    _dev(x, wp, ws)
    N, H, W, _ = x.shape
    _check(lib().sf_enc1_stats(_p(x), _p(wp), _p(ws), N, H, W, K, _stream()), "sf_enc1_stats")


# False (or SF_ENC1_ROUTED=0 in the environment): the backward recomputes conv + arg-max instead of
# using the forward's routing codes (A/B timing, tests)
ENC1_ROUTED = os.environ.get("SF_ENC1_ROUTED", "1") != "0"


def enc1_fwd(x, wp, st, K, want_codes: bool = False):
    """Pooled activation of the fused first encoder layer; with want_codes also the routing codes
    (uint8 buffer of sf_enc1_codes_bytes bytes) that sf_enc1_bwd can use, else None."""
    _dev(x, wp, st)
    N, H, W, _ = x.shape
    y = torch.empty((N, H // 2, W // 2, K), device=x.device, dtype=torch.bfloat16)
    codes = None
    if want_codes and ENC1_ROUTED:
        nbytes = int(lib().sf_enc1_codes_bytes(N, H, W, K))
        if nbytes < 2 ** 31 - 1:
            codes = torch.empty(nbytes, device=x.device, dtype=torch.uint8)
    _check(lib().sf_enc1_fwd(_p(x), _p(wp), _p(st[0]), _p(st[1]), _p(y), _p(codes), N, H, W, K, _stream()), "sf_enc1_fwd")
    return (y, codes) if want_codes else y


ENC1_GRAM_DOUBLES = 6400  # SF_ENC1_GRAM_DOUBLES of include/sfhip.h


def enc1_gram(x, Creal):
    """Gram record of the layer's input (sf_enc1_gram): f64 [ENC1_GRAM_DOUBLES]; Creal <= 10."""
    _dev(x)
    N, H, W, _ = x.shape
    rec = torch.empty(ENC1_GRAM_DOUBLES, device=x.device, dtype=torch.float64)
    nbytes = int(lib().sf_enc1_gram_workspace_bytes(N, H, W, Creal))
    ws = torch.empty((nbytes + 3) // 4, device=x.device, dtype=torch.float32)
    _check(lib().sf_enc1_gram(_p(x), N, H, W, Creal, _p(rec), _p(ws), ctypes.c_size_t(nbytes), _stream()), "sf_enc1_gram")
    return rec


def enc1_stats_gram(wp, rec, K, Creal, ws):
    """BatchNorm sums of the conv output from the Gram record (no pass over the conv output)."""
    _dev(wp, rec, ws, dtypes=[torch.bfloat16, torch.float64, torch.float64])
    _check(lib().sf_enc1_stats_gram(_p(wp), _p(rec), K, Creal, _p(ws), _stream()), "sf_enc1_stats_gram")


def enc1_bwd(x, wp, dy, st, training, K, Creal, codes=None, gram=None):
    """(dw [K,Creal,3,3], dgamma, dbeta) of the fused first encoder layer; st = bn_finalize rows;
    codes = what enc1_fwd(..., want_codes=True) returned for the same inputs, or None; gram = the
    record of enc1_gram for the same x, or None (then the routed backward accumulates it itself)."""
    _dev(x, wp, dy, st, codes, gram)
    N, H, W, _ = x.shape
    nbytes = int(lib().sf_enc1_bwd_workspace_bytes(N, H, W, K, Creal))
    ws = torch.empty((nbytes + 7) // 8, device=x.device, dtype=torch.float64)      # 8-byte aligned
    dw = torch.empty((K, Creal, 3, 3), device=x.device, dtype=torch.float32)
    dgamma = torch.empty(K, device=x.device, dtype=torch.float32)
    dbeta = torch.empty(K, device=x.device, dtype=torch.float32)
    _check(lib().sf_enc1_bwd(_p(x), _p(wp), _p(dy), _p(codes), _p(gram), _p(st[0]), _p(st[1]), _p(st[2]), _p(st[3]),
                             int(training), _p(dw), _p(dgamma), _p(dbeta), _p(ws), ctypes.c_size_t(nbytes), N, H, W, K, Creal,
                             _stream()), "sf_enc1_bwd")
    return dw, dgamma, dbeta


def bn_lrelu_fwd(x, st):
    y = torch.empty_like(x)
    C = x.shape[-1]
    _check(lib().sf_bn_lrelu_fwd(_p(x), code_of(x.dtype), _p(st[0]), _p(st[1]), _p(y), _c_ll(x.numel() // C), C,
                                 _stream()), "sf_bn_lrelu_fwd")
    return y


def bn_relu_fwd(x, st):
    y = torch.empty_like(x)
    C = x.shape[-1]
    _check(lib().sf_bn_relu_fwd(_p(x), code_of(x.dtype), _p(st[0]), _p(st[1]), _p(y), _c_ll(x.numel() // C), C,
                                _stream()), "sf_bn_relu_fwd")
    return y


def relu_dropout_fwd(x: torch.Tensor, p: float, seed: int) -> torch.Tensor:
    _dev(x, dtypes=[torch.float32])
    y = torch.empty_like(x)
    _check(lib().sf_relu_dropout_fwd(_p(x), _p(y), _c_ll(x.numel()), _c_f(p), ctypes.c_ulonglong(seed), _stream()),
           "sf_relu_dropout_fwd")
    return y


def relu_dropout_bwd(y: torch.Tensor, dy: torch.Tensor, p: float) -> torch.Tensor:
    _dev(y, dy, dtypes=[torch.float32, torch.float32])
    dx = torch.empty_like(y)
    _check(lib().sf_relu_dropout_bwd(_p(y), _p(dy), _p(dx), _c_ll(y.numel()), _c_f(p), _stream()), "sf_relu_dropout_bwd")
    return dx


def bn_backward(mode, x, dy, st, gamma, Creal, training, OH=0, OW=0, moments=None):
    """Returns (dx, dgamma, dbeta); see sf_bn_bwd_reduce / sf_bn_bwd_apply.  moments = (pos, xh) of
    bn_relu_avgpool_fwd_moments (mode 1): the channel sums come from the pooled maps instead of x."""
    _dev(x, dy, st, gamma)
    N, H, W, C = x.shape
    ws = bn_ws(C, x.device)
    c = code_of(x.dtype)
    if moments is not None:
        if mode != 1 or dy.dtype != x.dtype:
            raise RuntimeError("bn_backward: bin moments belong to the avgpool mode, in x's dtype")
        pos, xh = moments
        _dev(pos, xh, dtypes=[torch.uint16, x.dtype])
        if tuple(pos.shape) != (N, OH, OW, C) or tuple(xh.shape) != (N, OH, OW, C) or tuple(dy.shape) != (N, OH, OW, C):
            raise RuntimeError("bn_backward: bin moments / pooled gradient must be [N, OH, OW, C]")
        _check(lib().sf_bn_bwd_reduce_pooled(_p(dy), c, _p(pos), _p(xh), _p(ws), N, H, W, C, OH, OW, _stream()),
               "sf_bn_bwd_reduce_pooled")
    else:
        _check(lib().sf_bn_bwd_reduce(mode, _p(x), c, _p(dy), code_of(dy.dtype), _p(st[0]), _p(st[1]), _p(st[2]),
                                      _p(st[3]), _p(ws), N, H, W, C, Creal, OH, OW, _stream()), "sf_bn_bwd_reduce")
    dx = torch.empty_like(x)
    dgamma = torch.empty(Creal, device=x.device, dtype=torch.float32)
    dbeta = torch.empty(Creal, device=x.device, dtype=torch.float32)
    _check(lib().sf_bn_bwd_apply(mode, _p(x), c, _p(dy), code_of(dy.dtype), _p(st[0]), _p(st[1]), _p(st[2]), _p(st[3]),
                                 _p(gamma), _p(ws), int(training), _p(dx), _p(dgamma), _p(dbeta), N, H, W, C, Creal,
                                 OH, OW, _stream()), "sf_bn_bwd_apply")
    return dx, dgamma, dbeta


def channel_sum(x: torch.Tensor, Creal: int) -> torch.Tensor:
    _dev(x)
    C = x.shape[-1]
    out = torch.empty(Creal, device=x.device, dtype=torch.float32)
    ws = bn_ws(C, x.device)
    _check(lib().sf_channel_sum(_p(x), code_of(x.dtype), _c_ll(x.numel() // C), C, Creal, _p(out), _p(ws), _stream()),
           "sf_channel_sum")
    return out


def colsum_f32(x: torch.Tensor) -> torch.Tensor:
    _dev(x, dtypes=[torch.float32])
    M, C = x.shape
    out = torch.empty(C, device=x.device, dtype=torch.float32)
    _check(lib().sf_colsum_f32(_p(x), _c_ll(M), _c_ll(C), _p(out), _stream()), "sf_colsum_f32")
    return out


# ------------------------------------------------------------------------------ bilinear
def bilinear_fwd(src_nhwc: torch.Tensor, OH: int, OW: int, dtype: torch.dtype) -> torch.Tensor:
    _dev(src_nhwc, dtypes=[torch.float32])
    B, h, w, C = src_nhwc.shape
    out = torch.empty((B, OH, OW, C), device=src_nhwc.device, dtype=dtype)
    _check(lib().sf_bilinear_fwd(_p(src_nhwc), F32, _p(out), code_of(dtype), B, C, h, w, OH, OW, C, _stream()),
           "sf_bilinear_fwd")
    return out


def bilinear_bwd(ddst: torch.Tensor, h: int, w: int) -> torch.Tensor:
    _dev(ddst)
    B, OH, OW, C = ddst.shape
    tmp = torch.empty((B, OH, w, C), device=ddst.device, dtype=torch.float32)
    out = torch.empty((B, h, w, C), device=ddst.device, dtype=torch.float32)
    _check(lib().sf_bilinear_bwd(_p(ddst), code_of(ddst.dtype), _p(out), _p(tmp), B, C, h, w, OH, OW, C, _stream()),
           "sf_bilinear_bwd")
    return out


# ------------------------------------------------------------------------------ small convs
def upconv3_fwd(y9: torch.Tensor, bias: Optional[torch.Tensor], OH: int, OW: int, dtype: torch.dtype,
                relu: bool) -> torch.Tensor:
    """act(bias + sum over the 9 taps of the resized, shifted tap maps); y9 f32 [B,h,w,9,K]."""
    _dev(y9, bias, dtypes=[torch.float32, torch.float32])
    B, h, w, nine, K = y9.shape
    assert nine == 9
    out = torch.empty((B, OH, OW, K), device=y9.device, dtype=dtype)
    _check(lib().sf_upconv3_fwd(_p(y9), _p(bias), _p(out), code_of(dtype), B, h, w, OH, OW, K,
                                ACT_RELU if relu else ACT_NONE, _stream()), "sf_upconv3_fwd")
    return out


def upconv3_bwd(dz: torch.Tensor, h: int, w: int, g_dtype: torch.dtype) -> torch.Tensor:
    """Adjoint of upconv3_fwd's gather: dz NHWC [B,OH,OW,K] -> g [B,h,w,9,K]."""
    _dev(dz)
    B, OH, OW, K = dz.shape
    tmp = torch.empty(B * OH * 3 * w * K, device=dz.device, dtype=torch.float32)
    g = torch.empty((B, h, w, 9, K), device=dz.device, dtype=g_dtype)
    _check(lib().sf_upconv3_bwd(_p(dz), code_of(dz.dtype), _p(tmp), _p(g), code_of(g_dtype), B, h, w, OH, OW, K,
                                _stream()), "sf_upconv3_bwd")
    return g


def dconv1_fwd(x, w, b, dtype):
    _dev(x, w, b, dtypes=[torch.float32] * 3)
    N, _, H, W = x.shape
    K = w.shape[0]
    OH, OW = (H - 2) // 2 + 1, (W - 2) // 2 + 1
    y = torch.empty((N, OH, OW, K), device=x.device, dtype=dtype)
    _check(lib().sf_dconv1_fwd(_p(x), _p(w), _p(b), _p(y), code_of(dtype), N, H, W, K, _stream()), "sf_dconv1_fwd")
    return y


def dconv1_bwd_data(dz, w, H, W):
    _dev(dz, w)
    N, K = dz.shape[0], dz.shape[-1]
    dx = torch.empty((N, 1, H, W), device=dz.device, dtype=torch.float32)
    _check(lib().sf_dconv1_bwd_data(_p(dz), code_of(dz.dtype), _p(w), _p(dx), N, H, W, K, _stream()),
           "sf_dconv1_bwd_data")
    return dx


def dconv1_bwd_weight(x, dz):
    _dev(x, dz)
    N, _, H, W = x.shape
    K = dz.shape[-1]
    dw = torch.empty((K, 1, 4, 4), device=x.device, dtype=torch.float32)
    db = torch.empty(K, device=x.device, dtype=torch.float32)
    nbytes = int(lib().sf_dconv1_bwd_weight_workspace_bytes(N, H, W, K))
    ws = torch.empty(max(nbytes // 4, 1), device=x.device, dtype=torch.float32)
    _check(lib().sf_dconv1_bwd_weight_ws(_p(x), _p(dz), code_of(dz.dtype), _p(dw), _p(db), _p(ws) if nbytes else None,
                                         ctypes.c_size_t(nbytes), N, H, W, K, _stream()), "sf_dconv1_bwd_weight_ws")
    return dw, db


def conv_k1_fwd(x, w, b, R, pad):
    _dev(x, w, b)
    N, H, W, C = x.shape
    Creal = w.shape[1]
    OH, OW = H + 2 * pad - R + 1, W + 2 * pad - R + 1
    y = torch.empty((N, 1, OH, OW), device=x.device, dtype=torch.float32)
    nbytes = int(lib().sf_conv_k1_fwd_workspace_bytes(code_of(x.dtype), N, H, W, C, R))
    if nbytes:
        ws = torch.empty(nbytes // 4, device=x.device, dtype=torch.float32)
        _check(lib().sf_conv_k1_fwd_ws(_p(x), code_of(x.dtype), _p(w), _p(b), _p(y), _p(ws), ctypes.c_size_t(nbytes), N, H, W,
                                       C, Creal, R, pad, _stream()), "sf_conv_k1_fwd_ws")
        return y
    _check(lib().sf_conv_k1_fwd(_p(x), code_of(x.dtype), _p(w), _p(b), _p(y), N, H, W, C, Creal, R, pad, _stream()),
           "sf_conv_k1_fwd")
    return y


def conv_k1_bwd_data(dy, w, like: torch.Tensor, R, pad, relu_mask: bool):
    """dx (same shape/dtype as `like`, the forward input); relu_mask multiplies by (like > 0)."""
    _dev(dy, w, like)
    N, H, W, C = like.shape
    dx = torch.empty_like(like)
    _check(lib().sf_conv_k1_bwd_data(_p(dy), _p(w), _p(like) if relu_mask else _c_vp(0), _p(dx), code_of(like.dtype),
                                     N, H, W, C, w.shape[1], R, pad, _stream()), "sf_conv_k1_bwd_data")
    return dx


def conv_k1_bwd_weight(x, dy, Creal, R, pad):
    _dev(x, dy)
    N, H, W, C = x.shape
    dw = torch.empty((1, Creal, R, R), device=x.device, dtype=torch.float32)
    db = torch.empty(1, device=x.device, dtype=torch.float32)
    _check(lib().sf_conv_k1_bwd_weight(_p(x), code_of(x.dtype), _p(dy), _p(dw), _p(db), N, H, W, C, Creal, R, pad,
                                       _stream()), "sf_conv_k1_bwd_weight")
    return dw, db


def conv_k1_bwd_fused_ok(x: torch.Tensor) -> bool:
    g = x.shape[-1] // epc(code_of(x.dtype))
    return 1 <= g <= 64 and (g & (g - 1)) == 0 and x.numel() // x.shape[-1] < 2 ** 31 - 1


def conv_k1_bwd_fused(x, dy, w, Creal):
    """Backward of ReLU -> 1x1 conv to one channel in one pass: (dx, dw [1,Creal,1,1], db [1],
    dsum [Creal] = channel sums of dx)."""
    _dev(x, dy, w)
    C = x.shape[-1]
    M = x.numel() // C
    dx = torch.empty_like(x)
    dw = torch.empty((1, Creal, 1, 1), device=x.device, dtype=torch.float32)
    db = torch.empty(1, device=x.device, dtype=torch.float32)
    dsum = torch.empty(Creal, device=x.device, dtype=torch.float32)
    _check(lib().sf_conv_k1_bwd_fused(_p(x), code_of(x.dtype), _p(dy), _p(w), _p(dx), _p(dw), _p(db), _p(dsum), _c_ll(M),
                                      C, Creal, _stream()), "sf_conv_k1_bwd_fused")
    return dx, dw, db, dsum


# ------------------------------------------------------------------------------ temporal
def _ptr_array(ts: Sequence[torch.Tensor]):
    return (_c_vp * len(ts))(*[t.data_ptr() for t in ts])


def transformer_save_floats(T, D, F) -> int:
    return int(lib().sf_transformer_save_floats(T, D, F))


def transformer_layer_fwd(x, ws: List[torch.Tensor], nhead, p_drop, seed, save: Optional[torch.Tensor]):
    _dev(x, *ws, dtypes=[torch.float32] * (1 + len(ws)))
    B, T, D = x.shape
    Fd = ws[4].shape[0]
    y = torch.empty_like(x)
    _check(lib().sf_transformer_layer_fwd(_p(x), _p(y), _ptr_array(ws), _p(save), B, T, D, Fd, nhead, _c_f(p_drop),
                                          ctypes.c_ulonglong(seed), _stream()), "sf_transformer_layer_fwd")
    return y


def transformer_save_offset(T, D, F, which) -> int:
    return int(lib().sf_transformer_save_offset(T, D, F, which))


def transformer_layer_bwd(x, dy, ws, dws, save, nhead, p_drop, seed):
    """Backward of one encoder layer.  dws: 12 gradient tensors; the vector entries must be zeroed
    by the caller (accumulated with atomics), the four matrices are overwritten by GEMMs over the
    B*T tokens (A = per-token gradients, B = saved activations, both reduction-major, exact f32
    MFMA)."""
    _dev(x, dy, save, *ws, *dws)
    B, T, D = x.shape
    Fd = ws[4].shape[0]
    n = B * T
    dx = torch.empty_like(x)
    gtok = torch.empty(n * (5 * D + Fd), device=x.device, dtype=torch.float32)
    _check(lib().sf_transformer_layer_bwd(_p(x), _p(dy), _p(dx), _ptr_array(ws), _ptr_array(dws), _p(save), _p(gtok),
                                          B, T, D, Fd, nhead, _c_f(p_drop), ctypes.c_ulonglong(seed), _stream()),
           "sf_transformer_layer_bwd")

    def saved(which, width):
        o = B * transformer_save_offset(T, D, Fd, which)
        return save[o:o + n * width]

    g_qkv, g_sa = gtok[: n * 3 * D], gtok[n * 3 * D: n * 4 * D]
    g_z, g_ff = gtok[n * 4 * D: n * 4 * D + n * Fd], gtok[n * (4 * D + Fd):]
    xin = x.reshape(-1)
    if D % 64 == 0 and Fd % 64 == 0:
        gs, acts, outs = [g_qkv, g_sa, g_z, g_ff], [xin, saved(2, D), saved(5, D), saved(6, Fd)], [dws[0], dws[2], dws[4], dws[6]]
        _check(lib().sf_token_outer4(_ptr_array(gs), _ptr_array(acts), _ptr_array(outs), (_c_int * 4)(3 * D, D, Fd, D),
                                     (_c_int * 4)(D, D, D, Fd), n, _stream()), "sf_token_outer4")
        return dx
    linear_gemm(g_qkv, xin, dws[0], 3 * D, D, n, 3 * D, D, D, True, True, OUT_IJ)
    linear_gemm(g_sa, saved(2, D), dws[2], D, D, n, D, D, D, True, True, OUT_IJ)
    linear_gemm(g_z, saved(5, D), dws[4], Fd, D, n, Fd, D, D, True, True, OUT_IJ)
    linear_gemm(g_ff, saved(6, Fd), dws[6], D, Fd, n, D, Fd, Fd, True, True, OUT_IJ)
    return dx


def add_pe(x, pe):
    _dev(x, pe, dtypes=[torch.float32, torch.float32])
    B, T, D = x.shape
    y = torch.empty_like(x)
    _check(lib().sf_add_pe(_p(x), _p(pe), _p(y), B, T, D, _stream()), "sf_add_pe")
    return y


def mean_t_fwd(x):
    B, T, D = x.shape
    y = torch.empty((B, D), device=x.device, dtype=torch.float32)
    _check(lib().sf_mean_t_fwd(_p(x), _p(y), B, T, D, _stream()), "sf_mean_t_fwd")
    return y


def mean_t_bwd(dy, T):
    _dev(dy, dtypes=[torch.float32])
    B, D = dy.shape
    dx = torch.empty((B, T, D), device=dy.device, dtype=torch.float32)
    _check(lib().sf_mean_t_bwd(_p(dy), _p(dx), B, T, D, _stream()), "sf_mean_t_bwd")
    return dx


def lstm_save_floats(T, H, L) -> int:
    return int(lib().sf_lstm_save_floats(T, H, L))


def lstm_fwd(x, ws, L, p_drop, seed, save):
    _dev(x, *ws, dtypes=[torch.float32] * (1 + len(ws)))
    B, T, H = x.shape
    y = torch.empty((B, H), device=x.device, dtype=torch.float32)
    _check(lib().sf_lstm_fwd(_p(x), _p(y), _ptr_array(ws), _p(save), B, T, H, L, _c_f(p_drop), ctypes.c_ulonglong(seed),
                             _stream()), "sf_lstm_fwd")
    return y


def lstm_bwd(x, dy, ws, dws, save, L, p_drop, seed):
    _dev(x, dy, save, *ws, *dws)
    B, T, H = x.shape
    dx = torch.empty_like(x)
    _check(lib().sf_lstm_bwd(_p(x), _p(dy), _p(dx), _ptr_array(ws), _ptr_array(dws), _p(save), B, T, H, L,
                             _c_f(p_drop), ctypes.c_ulonglong(seed), _stream()), "sf_lstm_bwd")
    return dx


def gru_fwd(x, ws, L, p_drop, seed, need_save: bool):
    """x f32 [B,T,H] -> (output sequence [B,T,H], save or None)."""
    _dev(x, *ws, dtypes=[torch.float32] * (1 + len(ws)))
    B, T, H = x.shape
    y = torch.empty_like(x)
    save = torch.empty(B * int(lib().sf_gru_save_floats(T, H, L)), device=x.device, dtype=torch.float32) if need_save else None
    _check(lib().sf_gru_fwd(_p(x), _p(y), _ptr_array(ws), _p(save), B, T, H, L, _c_f(p_drop), ctypes.c_ulonglong(seed),
                            _stream()), "sf_gru_fwd")
    return y, save


def gru_bwd(x, dy, ws, dws, save, L, p_drop, seed):
    _dev(x, dy, save, *ws, *dws)
    B, T, H = x.shape
    dx = torch.empty_like(x)
    _check(lib().sf_gru_bwd(_p(x), _p(dy), _p(dx), _ptr_array(ws), _ptr_array(dws), _p(save), B, T, H, L,
                            _c_f(p_drop), ctypes.c_ulonglong(seed), _stream()), "sf_gru_bwd")
    return dx


def attn_fwd(g, ws, need_save: bool):
    """g f32 [B,T,H], ws = [Wq, bq, Wk, bk, Wv, bv] -> (context [B,H], save or None)."""
    _dev(g, *ws, dtypes=[torch.float32] * 7)
    B, T, H = g.shape
    out = torch.empty((B, H), device=g.device, dtype=torch.float32)
    save = torch.empty(B * int(lib().sf_attn_save_floats(T, H)), device=g.device, dtype=torch.float32) if need_save else None
    _check(lib().sf_attn_fwd(_p(g), _p(out), _ptr_array(ws), _p(save), B, T, H, _stream()), "sf_attn_fwd")
    return out, save


def attn_bwd(g, dout, ws, dws, save):
    _dev(g, dout, save, *ws, *dws)
    B, T, H = g.shape
    dg = torch.empty_like(g)
    _check(lib().sf_attn_bwd(_p(g), _p(dout), _p(dg), _ptr_array(ws), _ptr_array(dws), _p(save), B, T, H, _stream()),
           "sf_attn_bwd")
    return dg


# ------------------------------------------------------------------------------ losses / optimiser
def lonlat_to_pixel(lon: torch.Tensor, lat: torch.Tensor, region, W: int, H: int):
    _dev(lon, lat, dtypes=[torch.float64, torch.float64])
    n = lon.numel()
    px = torch.empty(n, device=lon.device, dtype=torch.int32)
    py = torch.empty(n, device=lon.device, dtype=torch.int32)
    _check(lib().sf_lonlat_to_pixel(_p(lon), _p(lat), n, _c_d(region[0]), _c_d(region[1]), _c_d(region[2]),
                                    _c_d(region[3]), W, H, _p(px), _p(py), _stream()), "sf_lonlat_to_pixel")
    return px, py


_LOSS_WS: dict = {}


def _loss_workspace(device: torch.device) -> torch.Tensor:
    """The loss workspace of (device, current stream): allocated once, zero (the arrival counter in
    its last double has to start at zero and every forward call leaves it so); forward and backward
    of one loss call are both issued on that stream, so its calls never overlap."""
    key = (device.index if device.index is not None else torch.cuda.current_device(), _stream().value)
    ws = _LOSS_WS.get(key)
    if ws is None:
        ws = _LOSS_WS[key] = torch.zeros(LOSS_WS_DOUBLES, device=device, dtype=torch.float64)
    return ws


def _loss_args(pred, target, pos, runoff, offsets):
    _dev(pred, target, pos, runoff, offsets,
         dtypes=[torch.float32, torch.float32, torch.int32, torch.float32, torch.int32])
    B, _, H, W = pred.shape
    return B, H, W, target.shape[-2], target.shape[-1]


def image_station_loss_fwd(pred, target, pos, runoff, offsets, S, image_metric, nonzero_weight, zero_threshold, ksize,
                           station_metric, image_weight, station_weight, nan_flag=None, keep_for_backward=True):
    """One launch: (losses f32 [5] = total, image, station, 1 / weight sum, 1 / stations counted;
    station differences [S] for image_station_loss_bwd, or None).  nan_flag: int32 [1] that is OR-ed
    with 1 when pred holds Inf / NaN."""
    B, H, W, TH, TW = _loss_args(pred, target, pos, runoff, offsets)
    _dev(nan_flag, dtypes=[torch.int32])
    losses = torch.empty(5, device=pred.device, dtype=torch.float32)
    sdiff = torch.empty(max(S, 1), device=pred.device, dtype=torch.float32) if keep_for_backward else None
    ws = _loss_workspace(pred.device)
    _check(lib().sf_image_station_loss_fwd(_p(pred), _p(target), _p(pos), _p(runoff), _p(offsets), S, B, H, W, TH, TW,
                                           image_metric, _c_f(nonzero_weight), _c_f(zero_threshold), ksize,
                                           station_metric, _c_f(image_weight), _c_f(station_weight), _p(losses),
                                           _p(sdiff), _p(nan_flag), _p(ws), _stream()), "sf_image_station_loss_fwd")
    return losses, sdiff


def image_station_loss_bwd(pred, target, pos, runoff, offsets, S, image_metric, nonzero_weight, zero_threshold, ksize,
                           station_metric, image_weight, station_weight, fwd_losses, sdiff=None, upstream=None):
    """One launch: upstream * d total / d pred, complete.  fwd_losses: the [5] tensor of the forward
    call; sdiff: its station differences (None: gathered again); upstream: f32 device scalar or
    None = 1."""
    B, H, W, TH, TW = _loss_args(pred, target, pos, runoff, offsets)
    _dev(upstream, sdiff, fwd_losses, dtypes=[torch.float32, torch.float32, torch.float32])
    if fwd_losses.numel() < 5:
        raise RuntimeError("image_station_loss_bwd: fwd_losses must be the 5 floats of image_station_loss_fwd")
    dpred = torch.empty_like(pred)
    _check(lib().sf_image_station_loss_bwd(_p(pred), _p(target), _p(pos), _p(runoff), _p(offsets), S, B, H, W, TH, TW,
                                           image_metric, _c_f(nonzero_weight), _c_f(zero_threshold), ksize,
                                           station_metric, _c_f(image_weight), _c_f(station_weight), _p(upstream),
                                           _p(sdiff), _p(fwd_losses), _p(dpred), _stream()), "sf_image_station_loss_bwd")
    return dpred


def image_station_loss(pred, target, pos, runoff, offsets, S, image_metric, nonzero_weight, zero_threshold, ksize,
                       station_metric, image_weight, station_weight):
    """sf_image_station_loss (forward and backward back to back, upstream gradient 1): returns
    (losses f32 [4] = total, image, station, 1; dpred = d total / d pred)."""
    B, H, W, TH, TW = _loss_args(pred, target, pos, runoff, offsets)
    losses = torch.empty(4, device=pred.device, dtype=torch.float32)
    dpred = torch.empty_like(pred)
    ws = torch.empty(LOSS_WS_DOUBLES, device=pred.device, dtype=torch.float64)
    _check(lib().sf_image_station_loss(_p(pred), _p(target), _p(pos), _p(runoff), _p(offsets), S, B, H, W, TH, TW,
                                       image_metric, _c_f(nonzero_weight), _c_f(zero_threshold), ksize, station_metric,
                                       _c_f(image_weight), _c_f(station_weight), _p(losses), _p(dpred), _p(ws),
                                       _stream()), "sf_image_station_loss")
    return losses, dpred


def image_station_loss_static(pred, target, pos, runoff, offsets, S, image_metric, nonzero_weight, zero_threshold, ksize,
                              station_metric, image_weight, station_weight, losses, dpred):
    """sf_image_station_loss_static: as image_station_loss, but the caller passes the outputs
    (losses f32 [4], dpred like pred) and the partial sums live in the library's own static device
    array -- the call allocates nothing, here or in the library."""
    _dev(pred, target, pos, runoff, offsets, losses, dpred,
         dtypes=[torch.float32, torch.float32, torch.int32, torch.float32, torch.int32, torch.float32, torch.float32])
    B, _, H, W = pred.shape
    TH, TW = target.shape[-2], target.shape[-1]
    if losses.numel() < 4 or dpred.shape != pred.shape:
        raise RuntimeError("image_station_loss_static: losses needs 4 floats and dpred the shape of pred")
    _check(lib().sf_image_station_loss_static(_p(pred), _p(target), _p(pos), _p(runoff), _p(offsets), S, B, H, W, TH, TW,
                                              image_metric, _c_f(nonzero_weight), _c_f(zero_threshold), ksize,
                                              station_metric, _c_f(image_weight), _c_f(station_weight), _p(losses),
                                              _p(dpred), _stream()), "sf_image_station_loss_static")


def bce_logits(x: torch.Tensor, label: float, labels: Optional[torch.Tensor] = None, need_grad: bool = True):
    _dev(x, labels, dtypes=[torch.float32, torch.float32])
    loss = torch.empty(1, device=x.device, dtype=torch.float32)
    dx = torch.empty_like(x) if need_grad else None
    ws = torch.empty(1, device=x.device, dtype=torch.float64)
    _check(lib().sf_bce_logits(_p(x), _c_ll(x.numel()), _c_f(label), _p(labels), _p(loss), _p(dx), _p(ws), _stream()),
           "sf_bce_logits")
    return loss, dx


def adam(p, g, m, v, lr, beta1, beta2, eps, step, grad_scale=1.0):
    _dev(p, g, m, v, dtypes=[torch.float32] * 4)
    bc1, bc2 = 1.0 - beta1 ** step, 1.0 - beta2 ** step
    _check(lib().sf_adam(_p(p), _p(g), _p(m), _p(v), _c_ll(p.numel()), _c_f(lr), ctypes.c_double(beta1), ctypes.c_double(beta2), _c_f(eps),
                         _c_f(bc1), _c_f(bc2), _c_f(grad_scale), _stream()), "sf_adam")


def adam_begin(state: torch.Tensor, found_inf: Optional[torch.Tensor]):
    """state int32 [2] (steps taken, steps skipped): one of the two advances, by found_inf."""
    _dev(state, found_inf, dtypes=[torch.int32, torch.int32])
    _check(lib().sf_adam_begin(_p(state), _p(found_inf), _stream()), "sf_adam_begin")


def adam_guarded(p, g, m, v, lr, beta1, beta2, eps, state, found_inf=None, grad_scale=1.0, shadow=None, step_offset=0):
    """Adam update of a flat range with the step count read from `state` on the device; a no-op
    when found_inf[0] != 0.  shadow: optional bf16 tensor of p's size that receives the new values."""
    _dev(p, g, m, v, state, found_inf, shadow,
         dtypes=[torch.float32] * 4 + [torch.int32, torch.int32, torch.bfloat16])
    if shadow is not None and shadow.numel() != p.numel():
        raise RuntimeError("adam_guarded: the bf16 shadow must have the parameter range's size")
    _check(lib().sf_adam_guarded(_p(p), _p(g), _p(m), _p(v), _p(shadow), _c_ll(p.numel()), _c_f(lr), ctypes.c_double(beta1),
                                 ctypes.c_double(beta2), _c_f(eps), _c_f(grad_scale), _p(state), _p(found_inf),
                                 int(step_offset), _stream()), "sf_adam_guarded")


def nan_inf_flag(x: torch.Tensor, flag: torch.Tensor):
    _dev(x, flag, dtypes=[torch.float32, torch.int32])
    _check(lib().sf_nan_inf_flag(_p(x), _c_ll(x.numel()), _p(flag), _stream()), "sf_nan_inf_flag")


# ------------------------------------------------------------------------------ resident dataset
def dataset_plane_stats(frames: torch.Tensor, masks: Optional[torch.Tensor], mask_row: Optional[torch.Tensor]):
    """frames f32 [D,C,H,W] -> (stats f64 [D,C,4] = sum, sumsq, min, max; count i64 [D]; bad0 i32 [D])."""
    _dev(frames, masks, mask_row, dtypes=[torch.float32, torch.float32, torch.int32])
    D, C, H, W = frames.shape
    out = torch.empty((D, C, 4), device=frames.device, dtype=torch.float64)
    count = torch.empty(D, device=frames.device, dtype=torch.int64)
    bad0 = torch.empty(D, device=frames.device, dtype=torch.int32)
    _check(lib().sf_dataset_plane_stats(_p(frames), _p(masks), _p(mask_row), D, C, ctypes.c_longlong(H * W), _p(out),
                                        _p(count), _p(bad0), _stream()), "sf_dataset_plane_stats")
    return out, count, bad0


def dataset_transform(frames: torch.Tensor, target: Optional[torch.Tensor], masks: Optional[torch.Tensor],
                      mask_row: Optional[torch.Tensor], a: Optional[torch.Tensor], b: Optional[torch.Tensor],
                      flow_threshold: float):
    """In place on frames f32 [D,C,H,W]; target (optional) f32 [D,1,H,W]."""
    _dev(frames, target, masks, mask_row, a, b,
         dtypes=[torch.float32, torch.float32, torch.float32, torch.int32, torch.float64, torch.float64])
    D, C, H, W = frames.shape
    _check(lib().sf_dataset_transform(_p(frames), _p(target), _p(masks), _p(mask_row), _p(a), _p(b),
                                      1 if a is not None else 0, _c_f(flow_threshold), D, C, ctypes.c_longlong(H * W),
                                      _stream()), "sf_dataset_transform")


def gather_frames(src: torch.Tensor, src_stride: int, n_src: int, rows: torch.Tensor, n_elems: int, out: torch.Tensor):
    """out[i].flat = src.flat[rows[i] * src_stride : ... + n_elems] (rows: device int64)."""
    _dev(src, rows, out, dtypes=[torch.float32, torch.int64, torch.float32])
    n = rows.numel()
    assert out.numel() == n * n_elems and src.numel() >= (n_src - 1) * src_stride + n_elems
    _check(lib().sf_gather_frames(_p(src), ctypes.c_longlong(src_stride), ctypes.c_longlong(n_src), _p(rows), n,
                                  ctypes.c_longlong(n_elems), _p(out), _stream()), "sf_gather_frames")
    return out


# ------------------------------------------------------------------------------ evaluation metrics
def date_image_metrics(pred: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """pred, target f32 [B,H,W] (or [B,1,H,W]) -> f64 [B,13] (see sf_date_image_metrics)."""
    _dev(pred, target, dtypes=[torch.float32, torch.float32])
    if pred.shape != target.shape:
        raise RuntimeError("date_image_metrics: prediction and target shapes differ")
    B, H, W = pred.shape[0], pred.shape[-2], pred.shape[-1]
    if pred.numel() != B * H * W:
        raise RuntimeError("date_image_metrics: single-channel maps expected")
    nbytes = int(lib().sf_date_image_metrics_workspace_bytes(B, H, W))
    ws = torch.empty(nbytes // 8, device=pred.device, dtype=torch.float64)
    out = torch.empty((B, 13), device=pred.device, dtype=torch.float64)
    _check(lib().sf_date_image_metrics(_p(pred), _p(target), B, H, W, _p(out), _p(ws), ctypes.c_size_t(nbytes),
                                       _stream()), "sf_date_image_metrics")
    return out
