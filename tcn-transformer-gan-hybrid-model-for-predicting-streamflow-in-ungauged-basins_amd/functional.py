"""torch.autograd Functions that chain the HIP kernels of one module (forward and backward).

Tensors crossing a module boundary are plain f32 NCHW torch tensors, exactly as in the
reference; inside a Function activations are NHWC in the compute dtype (f32 or bf16) with
channel counts padded to a multiple of 8.
"""
from __future__ import annotations

import warnings
from typing import List, Optional, Sequence

import torch

from . import ops

_FORCED_DTYPE: Optional[torch.dtype] = None


def set_compute_dtype(dtype: Optional[torch.dtype]):
    """Force the kernels' compute dtype (torch.float32 / torch.bfloat16); None = follow autocast."""
    global _FORCED_DTYPE
    if dtype not in (None, torch.float32, torch.bfloat16):
        raise ValueError("compute dtype must be float32, bfloat16 or None")
    _FORCED_DTYPE = dtype


_WARNED_FP16 = False


def compute_dtype() -> torch.dtype:
    """bf16 inside torch.autocast('cuda') (the reference's AMP switch, src/train.py:192), else f32.
    The kernels have no fp16 form: an autocast region that asks for float16 (torch's default for
    'cuda', which is what the reference's `amp.autocast(device_type='cuda')` gets) is computed in
    bf16 with f32 accumulation, and says so once per process."""
    global _WARNED_FP16
    if _FORCED_DTYPE is not None:
        return _FORCED_DTYPE
    if not torch.is_autocast_enabled("cuda"):
        return torch.float32
    if not _WARNED_FP16 and torch.get_autocast_dtype("cuda") == torch.float16:
        _WARNED_FP16 = True
        warnings.warn("torch.autocast('cuda') asks for float16; the HIP kernels compute in bfloat16 with float32 "
                      "accumulation instead (no loss scaling is needed; a GradScaler around them stays at its scale)",
                      UserWarning, stacklevel=2)
    return torch.bfloat16


def new_seed() -> int:
    """A dropout seed drawn from torch's CPU generator (honours torch.manual_seed, no GPU sync)."""
    return int(torch.empty((), dtype=torch.int64).random_().item()) & 0x7FFFFFFFFFFFFFFF


def _taps3():
    return [(r - 1, s - 1) for r in range(3) for s in range(3)]


def _taps3_dgrad():
    return [(1 - r, 1 - s) for r in range(3) for s in range(3)]


def conv3x3_geom(N, H, W, C, dgrad=False):
    return ops.conv_geom(N, H, W, C, H, W, 1, 1, H, W, _taps3_dgrad() if dgrad else _taps3(),
                         tap_rs=list(range(9)))


def conv3x3_wgeom(N, H, W, C, cin_real):
    return ops.conv_geom(N, H, W, C, H, W, 1, 1, H, W, _taps3(), tap_rs=list(range(9)), cin_real=cin_real, RS=9)


# ------------------------------------------------------------------------------ encoder
class EncoderFn(torch.autograd.Function):
    """CNNEncoder.forward (src/model/cnn_encoder.py:71-93): per layer conv3x3 -> BatchNorm ->
    ReLU -> maxpool2 (all but last) ; adaptive average pool to 32x32 after the last layer."""

    @staticmethod
    def forward(ctx, x, training, cdtype, target, nan_flag, *params):
        # nan_flag: None, or an int32 [1] device flag that the input layout pass sets when x holds
        # Inf / NaN (the trainer's input check, src/train.py:193, without a pass of its own)
        L = len(params) // 5
        N, Cin, H, W = x.shape
        e = ops.epc(ops.code_of(cdtype))
        a = ops.nchw_to_nhwc(x.contiguous(), cdtype, ops.roundup(Cin, 16 if cdtype == torch.bfloat16 else e), flag=nan_flag)
        saved = []
        h, w = H, W
        out = None
        for l in range(L):
            wt, gamma, beta, rm, rv = params[5 * l: 5 * l + 5]
            K = wt.shape[0]
            Kpad = ops.roundup(K, 8)
            C = a.shape[-1]
            wp = ops.derived(wt, ("pack3", cdtype, False, Kpad, C),
                             lambda: ops.pack_conv_weight(wt.detach().contiguous(), cdtype, list(range(9)), False, Kpad, C))
            M = N * h * w
            if (l == 0 and L > 1 and not ctx.needs_input_grad[0] and K == Kpad and wt.shape[1] <= 14
                    and ops.enc1_ok(a, K) and not ops.FORCE_GENERIC_CONV):
                # first layer: the conv output (the model's largest tensor) is never stored; the
                # statistics, the pooled activation and later the gradients each recompute it
                ws, gram = None, None
                routed = ops.ENC1_ROUTED and wt.shape[1] <= 10
                if training:
                    ws = torch.empty(2 * K, device=x.device, dtype=torch.float64)
                    if routed:
                        # the batch statistics of z = W xp from the Gram matrix of the input patches
                        # (sum z = W . sum xp, sum z^2 = W G W^T): one cheap pass over the 16-channel
                        # input instead of a conv pass; the backward needs the same matrix again
                        gram = ops.enc1_gram(a, wt.shape[1])
                        ops.enc1_stats_gram(wp, gram, K, wt.shape[1], ws)
                    else:
                        ops.enc1_stats(a, wp, K, ws)
                st = ops.bn_finalize(ws, M, K, K, gamma.detach(), beta.detach(), rm, rv, training)
                # with gradients wanted the forward also leaves each pooling window's arg-max and ReLU
                # bit, so that the backward routes the pooled gradient without recomputing the conv
                want = routed and any(ctx.needs_input_grad[5 + 5 * l: 8 + 5 * l])
                y0, codes = ops.enc1_fwd(a, wp, st, K, want_codes=True) if want else (ops.enc1_fwd(a, wp, st, K), None)
                saved.append((a, None, st, (h, w), (wp, codes, gram)))
                a = y0
                h, w = h // 2, w // 2
                continue
            z = torch.empty((N, h, w, Kpad), device=x.device, dtype=cdtype)
            wst = ops.derived(wt, ("stage3", cdtype, False, Kpad, C), lambda: ops.conv3x3_stage_weights(wp, C, Kpad))
            ops.conv3x3(a, wst, z, Kpad, flop_c=wt.shape[1], flop_k=K)
            ws = None
            if training:
                ws = ops.bn_ws(Kpad, x.device)
                ops.bn_stats(z, M, Kpad, ws)
            st = ops.bn_finalize(ws, M, Kpad, K, gamma.detach(), beta.detach(), rm, rv, training)
            if l < L - 1:
                saved.append((a, z, st, (h, w), None))
                a = ops.bn_relu_maxpool_fwd(z, st)
                h, w = h // 2, w // 2
            else:
                oh, ow = (target, target) if (h != target or w != target) else (h, w)
                # bf16 mode: the feature map leaves in bf16 (what the Linear layer behind it rounds to anyway)
                odt = cdtype if cdtype == torch.bfloat16 else torch.float32
                if training and ops.avgpool_moments_ok(h, w, oh, ow):
                    # the pooling pass also leaves what the backward's channel sums need (one read of z less)
                    out, pos, xh = ops.bn_relu_avgpool_fwd_moments(z, st, K, oh, ow, odt)
                    saved.append((a, z, st, (h, w), (pos, xh)))
                else:
                    out = ops.bn_relu_avgpool_fwd(z, st, K, oh, ow, odt)
                    saved.append((a, z, st, (h, w), None))
        ctx.saved = saved
        ctx.params = params
        ctx.training = training
        ctx.cdtype = cdtype
        ctx.in_shape = (N, Cin, H, W)
        return out

    @staticmethod
    def backward(ctx, dout):
        params, cdtype, training = ctx.params, ctx.cdtype, ctx.training
        L = len(params) // 5
        N, Cin, H, W = ctx.in_shape
        grads: List[Optional[torch.Tensor]] = [None] * len(params)
        K_last = params[5 * (L - 1)].shape[0]
        dy = ops.nchw_to_nhwc(dout.contiguous(), cdtype, ops.roundup(K_last, 8))
        oh, ow = dout.shape[-2], dout.shape[-1]
        dx = None
        for l in range(L - 1, -1, -1):
            wt, gamma = params[5 * l], params[5 * l + 1]
            a, z, st, (h, w), wp = ctx.saved[l]
            cin_real = wt.shape[1]
            if z is None:   # fused first layer: weight / BatchNorm gradients straight from the input
                wp, codes, gram = wp
                dw, dgamma, dbeta = ops.enc1_bwd(a, wp, dy, st, training, wt.shape[0], cin_real, codes=codes, gram=gram)
                grads[5 * l], grads[5 * l + 1], grads[5 * l + 2] = dw, dgamma, dbeta
                continue
            K, Kpad, C = wt.shape[0], z.shape[-1], a.shape[-1]
            mode = 1 if l == L - 1 else 0
            dz, dgamma, dbeta = ops.bn_backward(mode, z, dy, st, gamma.detach(), K, training, oh, ow,
                                                moments=wp if mode == 1 else None)
            dw = torch.empty_like(wt)
            ops.wgrad3x3(a, dz, dw)
            grads[5 * l], grads[5 * l + 1], grads[5 * l + 2] = dw, dgamma, dbeta
            if l > 0 or ctx.needs_input_grad[0]:
                wT = ops.derived(wt, ("stage3", cdtype, True, C, Kpad), lambda: ops.conv3x3_stage_weights(
                    ops.pack_conv_weight(wt.detach().contiguous(), cdtype, list(range(9)), True, C, Kpad), Kpad, C))
                da = torch.empty((N, h, w, C), device=dz.device, dtype=cdtype)
                ops.conv3x3(dz, wT, da, C, dgrad=True, flop_c=K, flop_k=cin_real)
                if l > 0:
                    dy = da
                else:
                    dx = ops.nhwc_to_nchw(da, Cin)
        ctx.saved = None
        return (dx, None, None, None, None, *grads)


def _tap_chunks(ntaps: int):
    """The gathered GEMM takes 16 filter taps per launch; larger filters run as a sum over tap groups."""
    return [list(range(i, min(i + 16, ntaps))) for i in range(0, ntaps, 16)]


def _conv_by_tap_groups(x, wt, out_shape, cdtype, transpose, rows_pad, cols_pad, geom_of):
    """conv of NHWC `x` with OIHW weight `wt` as the sum over groups of at most 16 taps (conv is
    linear in the filter): one group writes the output directly, several are summed in f32 and
    rounded once.  geom_of(group) is the gathered-GEMM geometry of the group's taps."""
    nt = wt.shape[2] * wt.shape[3]
    groups = _tap_chunks(nt)
    if len(groups) == 1:
        out = torch.empty(out_shape, device=x.device, dtype=cdtype)
        ops.conv_fprop(x, ops.pack_conv_weight(wt.detach().contiguous(), cdtype, groups[0], transpose, rows_pad, cols_pad), out,
                       geom_of(groups[0]), rows_pad)
        return out
    acc = None
    for grp in groups:
        part = torch.empty(out_shape, device=x.device, dtype=torch.float32)
        ops.conv_fprop(x, ops.pack_conv_weight(wt.detach().contiguous(), cdtype, grp, transpose, rows_pad, cols_pad), part,
                       geom_of(grp), rows_pad)
        acc = part if acc is None else acc.add_(part)
    return acc if cdtype == torch.float32 else ops.cast(acc, cdtype)


class EncoderGeneralFn(torch.autograd.Function):
    """CNNEncoder.forward for the constructor options the model class itself never uses
    (src/model/cnn_encoder.py:16-18,44-61): kernel_size up to 8 (more than 16 taps: summed over
    tap groups), any padding the tap offsets allow, and
    use_batchnorm=False (convs with bias, no normalisation).  Every conv runs on the gathered GEMM
    (forward, data gradient, weight gradient); a layer without BatchNorm goes through the
    BatchNorm passes with scale 1, shift = bias and frozen statistics, whose shift gradient is the
    bias gradient.  layers: per layer the number of parameters it owns (5: w, gamma, beta,
    running_mean, running_var; 2: w, bias)."""

    @staticmethod
    def forward(ctx, x, training, cdtype, target, ksize, pad, layers, *params):
        N, Cin, H, W = x.shape
        a = ops.nchw_to_nhwc(x.contiguous(), cdtype, ops.roundup(Cin, 8))
        taps = [(r - pad, s - pad) for r in range(ksize) for s in range(ksize)]
        saved, h, w, pi, out = [], H, W, 0, None
        L = len(layers)
        for l, np_ in enumerate(layers):
            ps = params[pi: pi + np_]
            pi += np_
            wt = ps[0]
            K, Kpad, C = wt.shape[0], ops.roundup(wt.shape[0], 8), a.shape[-1]
            oh, ow = h + 2 * pad - ksize + 1, w + 2 * pad - ksize + 1
            if oh < 1 or ow < 1:
                raise ValueError("CNNEncoder: the feature map vanished (kernel_size / padding against the image size)")
            z = _conv_by_tap_groups(a, wt, (N, oh, ow, Kpad), cdtype, False, Kpad, C,
                                    lambda g, h=h, w=w, C=C, oh=oh, ow=ow: ops.conv_geom(N, h, w, C, oh, ow, 1, 1, oh, ow,
                                                                                         [taps[t] for t in g]))
            M = N * oh * ow
            if np_ == 5:
                ws = None
                if training:
                    ws = ops.bn_ws(Kpad, x.device)
                    ops.bn_stats(z, M, Kpad, ws)
                st = ops.bn_finalize(ws, M, Kpad, K, ps[1].detach(), ps[2].detach(), ps[3], ps[4], training)
            else:
                st = torch.zeros((4, Kpad), device=x.device, dtype=torch.float32)
                st[0, :K] = 1.0
                st[1, :K] = ps[1].detach()
                st[3, :K] = 1.0
            saved.append((a, z, st, (h, w)))
            if l < L - 1:
                a = ops.bn_relu_maxpool_fwd(z, st)
                h, w = oh // 2, ow // 2
            else:
                th, tw = (target, target) if (oh != target or ow != target) else (oh, ow)
                out = ops.bn_relu_avgpool_fwd(z, st, K, th, tw, torch.float32)
        ctx.saved, ctx.params, ctx.layers = saved, params, layers
        ctx.training, ctx.cdtype, ctx.geom = training, cdtype, (ksize, pad, Cin)
        return out

    @staticmethod
    def backward(ctx, dout):
        params, layers, cdtype = ctx.params, ctx.layers, ctx.cdtype
        ksize, pad, Cin = ctx.geom
        taps = [(r - pad, s - pad) for r in range(ksize) for s in range(ksize)]
        taps_T = [(pad - r, pad - s) for r in range(ksize) for s in range(ksize)]
        grads: List[Optional[torch.Tensor]] = [None] * len(params)
        L = len(layers)
        K_last = params[sum(layers[:-1])].shape[0]
        dy = ops.nchw_to_nhwc(dout.contiguous().float(), cdtype, ops.roundup(K_last, 8))
        th, tw = dout.shape[-2], dout.shape[-1]
        dx = None
        for l in range(L - 1, -1, -1):
            pi = sum(layers[:l])
            ps = params[pi: pi + layers[l]]
            wt = ps[0]
            a, z, st, (h, w) = ctx.saved[l]
            N, oh, ow, Kpad = z.shape
            K, C = wt.shape[0], a.shape[-1]
            bn = layers[l] == 5
            gamma = ps[1].detach() if bn else torch.ones(K, device=z.device)
            dz, dgamma, dbeta = ops.bn_backward(1 if l == L - 1 else 0, z, dy, st, gamma, K, ctx.training and bn, th, tw)
            dw = torch.zeros_like(wt)
            for grp in _tap_chunks(ksize * ksize):      # each group adds its own taps of dw
                ops.conv_wgrad(a, dz, dw, ops.conv_geom(N, h, w, C, oh, ow, 1, 1, oh, ow, [taps[t] for t in grp], tap_rs=grp,
                                                        cin_real=wt.shape[1], RS=ksize * ksize), Kpad)
            grads[pi] = dw
            if bn:
                grads[pi + 1], grads[pi + 2] = dgamma, dbeta
            else:
                grads[pi + 1] = dbeta
            if l > 0 or ctx.needs_input_grad[0]:
                da = _conv_by_tap_groups(dz, wt, (N, h, w, C), cdtype, True, C, Kpad,
                                         lambda g, h=h, w=w, oh=oh, ow=ow, Kpad=Kpad: ops.conv_geom(
                                             N, oh, ow, Kpad, h, w, 1, 1, h, w, [taps_T[t] for t in g]))
                if l > 0:
                    dy = da
                else:
                    dx = ops.nhwc_to_nchw(da, Cin)
        ctx.saved = None
        return (dx, None, None, None, None, None, None, *grads)


# ------------------------------------------------------------------------------ 262144-wide Linear
def _split_gemm(A, B, I, J, R, lda, ldb, ldo, a_km, nsplit, bias, out_dtype=torch.float32):
    """out f32 [J][I] = B[J][R] . A^T (+ bias) with the reduction split over nsplit workgroup
    slices: every slice stores its partial result to a row of its own and sf_colsum_f32 adds the
    rows (fixed order, no atomics: with 256 slices the atomic epilogues of this shape cost more
    than the products)."""
    if nsplit <= 1:
        out = torch.empty((J, I), device=A.device, dtype=out_dtype)
        ops.linear_gemm(A, B, out, I, J, R, lda, ldb, ldo, a_km, False, ops.OUT_JI, 1, bias)
        return out
    slab = torch.empty((nsplit, J * I), device=A.device, dtype=torch.float32)
    ops.linear_gemm(A, B, slab, I, J, R, lda, ldb, ldo, a_km, False, ops.OUT_JI, nsplit, bias, split_stride=J * I)
    out = ops.colsum_f32(slab).view(J, I)
    return out if out_dtype == torch.float32 else out.to(out_dtype)


class LinearFn(torch.autograd.Function):
    """y = x W^T + b for the two skinny, weight-bandwidth-bound Linear layers
    (src/model/transformer_module.py:73 / src/model/lstm_module.py:36, src/model/__init__.py:92-95)."""

    @staticmethod
    def forward(ctx, x, W, b, cdtype, w_shadow=None):
        # w_shadow: a current bf16 copy of W (ops.shadow_if_current) that saves the cast
        M, K = x.shape
        Nf = W.shape[0]
        if K % 8 or Nf % 8:
            raise NotImplementedError(f"Linear {K}->{Nf}: both sizes must be multiples of 8 on the HIP path")
        # an input that already has the compute dtype (the encoder's bf16 output) is taken as it is,
        # and its gradient goes back in that dtype
        xt = x.contiguous() if x.dtype == cdtype else ops.cast(x.contiguous().float(), cdtype)
        ctx.x_dtype = x.dtype
        if w_shadow is not None and cdtype == torch.bfloat16 and w_shadow.shape == W.shape:
            Wt = w_shadow
        else:
            Wt = ops.cast(W.detach().contiguous(), cdtype)
        bk = 32 if cdtype == torch.bfloat16 else 16
        nsplit = max(1, min(256, K // (bk * 16)))
        y = _split_gemm(Wt, xt, Nf, M, K, K, K, Nf, False, nsplit, b.detach().contiguous() if b is not None else None)
        ctx.save_for_backward(xt, Wt)
        ctx.dims = (M, K, Nf, bk)
        ctx.has_bias = b is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        xt, Wt = ctx.saved_tensors
        M, K, Nf, bk = ctx.dims
        dy = dy.contiguous().float()
        dyt = ops.cast(dy, xt.dtype)
        dx = dW = db = None
        if ctx.needs_input_grad[0]:
            nsplit = max(1, min(256, Nf // (bk * 16)))
            dx = _split_gemm(Wt, dyt, K, M, Nf, K, Nf, K, True, nsplit, None,
                             out_dtype=ctx.x_dtype if ctx.x_dtype == torch.bfloat16 else torch.float32)
        if ctx.needs_input_grad[1]:
            dW = torch.empty((Nf, K), device=dy.device, dtype=torch.float32)
            ops.linear_gemm(dyt, xt, dW, Nf, K, M, Nf, K, K, True, True, ops.OUT_IJ, 1)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = ops.colsum_f32(dy)
        return dx, dW, db, None, None


def linear(x, lin, cdtype):
    """LinearFn on an nn.Linear's parameters, reading the weight's bf16 shadow when it is current."""
    sh = ops.shadow_if_current(lin.weight) if cdtype == torch.bfloat16 else None
    return LinearFn.apply(x, lin.weight, lin.bias, cdtype, sh)


# ------------------------------------------------------------------------------ temporal modules
class TransformerStackFn(torch.autograd.Function):
    """+positional encoding, num_layers post-LN encoder layers, mean over time
    (src/model/transformer_module.py:112-118)."""

    @staticmethod
    def forward(ctx, x, pe, nhead, p_drop, seed, *ws):
        L = len(ws) // 12
        B, T, D = x.shape
        Fd = ws[4].shape[0]
        wsd = [w.detach().contiguous() for w in ws]
        h = ops.add_pe(x.contiguous().float(), pe[:T].contiguous())
        need_save = any(ctx.needs_input_grad)
        inputs, saves = [], []
        nsave = ops.transformer_save_floats(T, D, Fd)
        for l in range(L):
            save = torch.empty(B * nsave, device=x.device, dtype=torch.float32) if need_save else None
            y = ops.transformer_layer_fwd(h, wsd[12 * l: 12 * l + 12], nhead, p_drop, seed + 7919 * l, save)
            inputs.append(h)
            saves.append(save)
            h = y
        out = ops.mean_t_fwd(h)
        ctx.inputs, ctx.saves, ctx.ws = inputs, saves, wsd
        ctx.cfg = (nhead, p_drop, seed, T)
        return out

    @staticmethod
    def backward(ctx, dout):
        nhead, p_drop, seed, T = ctx.cfg
        L = len(ctx.ws) // 12
        g = ops.mean_t_bwd(dout.contiguous().float(), T)
        # vector gradients are accumulated with atomics: one zero fill for all of them
        vecs = [w for w in ctx.ws if w.dim() == 1]
        zbuf = torch.zeros(sum((w.numel() + 3) // 4 * 4 for w in vecs), device=g.device, dtype=torch.float32)
        dws, pos = [], 0
        for w in ctx.ws:
            if w.dim() == 1:
                dws.append(zbuf[pos:pos + w.numel()])
                pos += (w.numel() + 3) // 4 * 4
            else:
                dws.append(torch.empty_like(w))
        for l in range(L - 1, -1, -1):
            g = ops.transformer_layer_bwd(ctx.inputs[l], g, ctx.ws[12 * l: 12 * l + 12], dws[12 * l: 12 * l + 12],
                                          ctx.saves[l], nhead, p_drop, seed + 7919 * l)
        ctx.inputs = ctx.saves = None
        return (g, None, None, None, None, *dws)


class LstmFn(torch.autograd.Function):
    """nn.LSTM(H, H, L, batch_first) -> h_n[-1] (src/model/lstm_module.py:66-75)."""

    @staticmethod
    def forward(ctx, x, L, p_drop, seed, *ws):
        B, T, H = x.shape
        wsd = [w.detach().contiguous() for w in ws]
        x = x.contiguous().float()
        need_save = any(ctx.needs_input_grad)
        save = torch.empty(B * ops.lstm_save_floats(T, H, L), device=x.device, dtype=torch.float32) if need_save else None
        y = ops.lstm_fwd(x, wsd, L, p_drop, seed, save)
        ctx.x, ctx.save, ctx.ws, ctx.cfg = x, save, wsd, (L, p_drop, seed)
        return y

    @staticmethod
    def backward(ctx, dy):
        L, p_drop, seed = ctx.cfg
        dws = [torch.zeros_like(w) for w in ctx.ws]
        dx = ops.lstm_bwd(ctx.x, dy.contiguous().float(), ctx.ws, dws, ctx.save, L, p_drop, seed)
        ctx.x = ctx.save = None
        return (dx, None, None, None, *dws)


# ------------------------------------------------------------------------------ decoder
def _pix_geom(B, h, w, C):
    """Geometry of a per-pixel (1x1) GEMM over an NHWC map."""
    return ops.conv_geom(B, h, w, C, h, w, 1, 1, h, w, [(0, 0)], tap_rs=[0])


class DecoderFn(torch.autograd.Function):
    """SimpleUpsamplingDecoder.forward (src/model/decoder.py:154-177).

    The bilinear resize and conv1 are both linear and the resize only touches pixels, so conv1's
    channel mixing is applied to the small source map first (a GEMM giving 9*K values per source
    pixel, one K-vector per filter tap) and the 16x larger resized map is never built:
    ops.upconv3_fwd sums, per output pixel, the interpolated tap maps; the backward goes through
    the adjoint map g (ops.upconv3_bwd) and three small GEMMs."""

    @staticmethod
    def forward(ctx, x, OH, OW, cdtype, w1, b1, w2, b2, w3, b3):
        B, C, h, w = x.shape
        if C % 8:
            raise NotImplementedError("decoder input channels must be a multiple of 8 on the HIP path")
        Hc = w1.shape[0]
        Hp = ops.roundup(Hc, 8)
        src = ops.nchw_to_nhwc(x.contiguous().float(), cdtype, C)
        # rows (tap, k): w9[t * Hp + k][c] = w1[k][c][t]; padded output channels are zero rows
        def tap_major():
            if Hp == Hc:
                w9 = w1.detach().reshape(Hc, C, 9).permute(2, 0, 1).contiguous()
            else:
                w9 = torch.zeros((9, Hp, C), device=x.device, dtype=torch.float32)
                w9[:, :Hc] = w1.detach().reshape(Hc, C, 9).permute(2, 0, 1)
            return w9.view(9 * Hp, C, 1, 1)

        w9p = ops.derived(w1, ("w9p", cdtype, Hp), lambda: ops.pack_conv_weight(tap_major(), cdtype, [0], False, 9 * Hp, C))
        w9T = None
        if ctx.needs_input_grad[0]:
            w9T = ops.derived(w1, ("w9T", cdtype, Hp), lambda: ops.pack_conv_weight(tap_major(), cdtype, [0], True, C, 9 * Hp))
        y9 = torch.empty((B, h, w, 9 * Hp), device=x.device, dtype=torch.float32)
        ops.conv_fprop(src, w9p, y9, _pix_geom(B, h, w, C), 9 * Hp)
        bp1 = _pad_vec(b1.detach(), Hp)
        h1 = ops.upconv3_fwd(y9.view(B, h, w, 9, Hp), bp1, OH, OW, cdtype, relu=True)
        w2p = ops.derived(w2, ("pack3", cdtype, False, Hp, Hp),
                          lambda: ops.pack_conv_weight(w2.detach().contiguous(), cdtype, list(range(9)), False, Hp, Hp))
        bp2 = _pad_vec(b2.detach(), Hp)
        h2 = torch.empty((B, OH, OW, Hp), device=x.device, dtype=cdtype)
        w3c = w3.detach().contiguous()
        if ops.conv3x3_dot_ok(h1, Hp, h2):
            # the 1x1 output conv (src/model/decoder.py:152,175) rides in conv2's epilogue: h2 is written
            # for the backward but not read back
            y = torch.empty((B, 1, OH, OW), device=x.device, dtype=torch.float32)
            ops.conv3x3(h1, w2p, h2, Hp, bias=bp2, act=ops.ACT_RELU, flop_c=Hc, flop_k=Hc,
                        dot=(_pad_vec(w3c.reshape(-1), Hp), b3.detach().contiguous(), y))
        else:
            ops.conv3x3(h1, w2p, h2, Hp, bias=bp2, act=ops.ACT_RELU, flop_c=Hc, flop_k=Hc)
            y = ops.conv_k1_fwd(h2, w3c, b3.detach().contiguous(), 1, 0)
        ctx.save_for_backward(src, h1, h2, w3c, w9T)
        ctx.weights = (w1, w2)
        ctx.dims = (B, C, h, w, OH, OW, Hc, Hp, cdtype)
        return y

    @staticmethod
    def backward(ctx, dy):
        src, h1, h2, w3c, w9T = ctx.saved_tensors
        w1, w2 = ctx.weights
        B, C, h, w, OH, OW, Hc, Hp, cdtype = ctx.dims
        dy = dy.contiguous().float()
        if ops.conv_k1_bwd_fused_ok(h2):
            dz2, dw3, db3, db2 = ops.conv_k1_bwd_fused(h2, dy, w3c, Hc)
        else:
            dw3, db3 = ops.conv_k1_bwd_weight(h2, dy, Hc, 1, 0)
            dz2 = ops.conv_k1_bwd_data(dy, w3c, h2, 1, 0, relu_mask=True)
            db2 = ops.channel_sum(dz2, Hc)
        dw2 = torch.empty_like(w2)
        ops.wgrad3x3(h1, dz2, dw2)
        w2T = ops.derived(w2, ("pack3", cdtype, True, Hp, Hp),
                          lambda: ops.pack_conv_weight(w2.detach().contiguous(), cdtype, list(range(9)), True, Hp, Hp))
        dz1 = torch.empty_like(h1)
        ops.conv3x3(dz2, w2T, dz1, Hp, dgrad=True, aux=h1, aux_mode=1, flop_c=Hc, flop_k=Hc)
        g = ops.upconv3_bwd(dz1, h, w, cdtype).view(B, h, w, 9 * Hp)
        # dW9[(t,k)][c] = sum_pixels g[pixel][(t,k)] * src[pixel][c]
        dw9 = torch.zeros((9 * Hp, C), device=dy.device, dtype=torch.float32)
        M = B * h * w
        ops.linear_gemm(g.view(M, 9 * Hp), src.view(M, C), dw9, 9 * Hp, C, M, 9 * Hp, C, C, True, True, ops.OUT_IJ,
                        max(1, min(32, M // 512)))
        dw1 = dw9.view(9, Hp, C)[:, :Hc].permute(1, 2, 0).reshape(Hc, C, 3, 3)
        # the interpolation weights of a pixel sum to one: the centre tap of g sums to sum_p dz1[p]
        db1 = g.view(-1, 9, Hp)[:, 4, :Hc].float().sum(dim=0)
        dx = None
        if ctx.needs_input_grad[0]:
            dsrc = torch.empty((B, h, w, C), device=dy.device, dtype=torch.float32)
            ops.conv_fprop(g, w9T, dsrc, _pix_geom(B, h, w, 9 * Hp), C)
            dx = ops.nhwc_to_nchw(dsrc, C)
        return dx, None, None, None, dw1, db1, dw2, db2, dw3, db3


def _pad_vec(v: torch.Tensor, n: int) -> torch.Tensor:
    v = v.contiguous().float()
    if v.numel() == n:
        return v
    out = torch.zeros(n, device=v.device, dtype=torch.float32)
    out[: v.numel()] = v
    return out


# ------------------------------------------------------------------------------ discriminator
def _s2_geom(N, IH, IW, C, OH, OW):
    taps = [(r - 1, s - 1) for r in range(4) for s in range(4)]
    return ops.conv_geom(N, IH, IW, C, OH, OW, 2, 2, OH, OW, taps, tap_rs=list(range(16)))


def _s2_wgeom(N, IH, IW, C, OH, OW, cin_real):
    taps = [(r - 1, s - 1) for r in range(4) for s in range(4)]
    return ops.conv_geom(N, IH, IW, C, OH, OW, 2, 2, OH, OW, taps, tap_rs=list(range(16)), cin_real=cin_real, RS=16)


def _s2_dgrad_classes(IH, IW):
    """Transposed 4x4 stride-2 pad-1 conv as four stride-1 2x2 convs, one per output parity."""
    out = []
    for py in (0, 1):
        for px in (0, 1):
            rows = [(r, (py + 1 - r) // 2) for r in range(4) if (py + 1 - r) % 2 == 0]
            cols = [(s, (px + 1 - s) // 2) for s in range(4) if (px + 1 - s) % 2 == 0]
            taps = [(dy, dx) for (_, dy) in rows for (_, dx) in cols]
            rs = [r * 4 + s for (r, _) in rows for (s, _) in cols]
            mh, mw = (IH - py + 1) // 2, (IW - px + 1) // 2
            if mh > 0 and mw > 0:
                out.append((py, px, taps, rs, mh, mw))
    return out


class DiscriminatorFn(torch.autograd.Function):
    """PatchDiscriminator.forward (src/model/gan.py:36-64); BatchNorm always uses batch
    statistics and updates its running buffers, as the reference never puts D in eval mode."""

    @staticmethod
    def forward(ctx, x, cdtype, training, reuse, record, w1, b1, *rest):
        # rest = 3 x (w, b, gamma, beta, rm, rv) + (w5, b5)
        # `record`: None, or a dict that receives the activations of this forward.
        # `reuse`: the record of an earlier training-mode forward of the SAME input with the SAME
        # parameters: nothing is recomputed, only the BatchNorm running statistics are updated
        # again, as a second forward call would.
        # training=False: BatchNorm normalises with its running statistics and updates nothing.
        if reuse is not None:
            if tuple(reuse["x"].shape) != tuple(x.shape):
                raise ValueError("DiscriminatorFn: the reused forward record belongs to another input shape")
            for i in range(3):
                wt, b, gamma, beta, rm, rv = rest[6 * i: 6 * i + 6]
                K = wt.shape[0]
                z = reuse["zs"][i]
                ops.bn_finalize(reuse["wss"][i], z.shape[0] * z.shape[1] * z.shape[2], K, K, gamma.detach(), beta.detach(),
                                rm, rv, True)
            ctx.x, ctx.acts, ctx.zs, ctx.sts = reuse["x"], reuse["acts"], reuse["zs"], reuse["sts"]
            ctx.xin = reuse.get("xin")
            ctx.params = (w1, b1, *rest)
            ctx.cdtype = cdtype
            ctx.training = True
            return reuse["out"].clone()
        N, _, H, W = x.shape
        x = x.contiguous().float()
        w1c, b1c = w1.detach().contiguous(), b1.detach().contiguous()
        K1 = w1.shape[0]
        if K1 % 8:
            raise NotImplementedError("discriminator width must be a multiple of 8 on the HIP path")
        Cin = w1.shape[1]
        xin = None
        if Cin == 1:
            y = ops.dconv1_fwd(x, w1c, b1c, cdtype)
        else:
            # several input channels (src/model/gan.py:10,25): the first layer as a gathered GEMM
            Cp = ops.roundup(Cin, 8)
            xin = ops.nchw_to_nhwc(x, cdtype, Cp)
            oh1, ow1 = (H - 2) // 2 + 1, (W - 2) // 2 + 1
            if oh1 < 1 or ow1 < 1:
                raise ValueError("input too small for the PatchGAN discriminator")
            y = torch.empty((N, oh1, ow1, K1), device=x.device, dtype=cdtype)
            ops.conv_fprop(xin, ops.pack_conv_weight(w1c, cdtype, list(range(16)), False, K1, Cp), y,
                           _s2_geom(N, H, W, Cp, oh1, ow1), K1, bias=b1c, act=ops.ACT_LRELU)
        acts = [y]
        zs, sts, wss = [], [], []
        for i in range(3):
            wt, b, gamma, beta, rm, rv = rest[6 * i: 6 * i + 6]
            K, C = wt.shape[0], y.shape[-1]
            ih, iw = y.shape[1], y.shape[2]
            oh, ow = (ih - 2) // 2 + 1, (iw - 2) // 2 + 1
            if oh < 1 or ow < 1:
                raise ValueError("input too small for the PatchGAN discriminator")
            z = torch.empty((N, oh, ow, K), device=x.device, dtype=cdtype)
            if ops.conv4x4s2_ok(y, K, K):
                ops.conv4x4s2(y, ops.conv4x4s2_weights(wt, False), z, K, bias=b.detach().contiguous())
            else:
                wp = ops.pack_conv_weight(wt.detach().contiguous(), cdtype, list(range(16)), False, K, C)
                ops.conv_fprop(y, wp, z, _s2_geom(N, ih, iw, C, oh, ow), K, bias=b.detach().contiguous())
            M = N * oh * ow
            ws = None
            if training:
                ws = ops.bn_ws(K, x.device)
                ops.bn_stats(z, M, K, ws)
            st = ops.bn_finalize(ws, M, K, K, gamma.detach(), beta.detach(), rm, rv, training)
            y = ops.bn_lrelu_fwd(z, st)
            zs.append(z)
            wss.append(ws)
            sts.append(st)
            acts.append(y)
        w5, b5 = rest[18], rest[19]
        w5c = w5.detach().contiguous()
        out = ops.conv_k1_fwd(y, w5c, b5.detach().contiguous(), 4, 1)
        ctx.x, ctx.acts, ctx.zs, ctx.sts, ctx.xin = x, acts, zs, sts, xin
        ctx.params = (w1, b1, *rest)
        ctx.cdtype = cdtype
        ctx.training = training
        if record is not None:      # the caller asked for the activations (to pass them back as `reuse`)
            record.update({"x": x, "acts": acts, "zs": zs, "sts": sts, "wss": wss, "out": out, "xin": xin})
        return out

    @staticmethod
    def backward(ctx, dout):
        x, acts, zs, sts, cdtype = ctx.x, ctx.acts, ctx.zs, ctx.sts, ctx.cdtype
        w1, b1, *rest = ctx.params
        N = x.shape[0]
        dout = dout.contiguous().float()
        w5 = rest[18]
        grads: List[Optional[torch.Tensor]] = [None] * (2 + len(rest))
        y4 = acts[3]
        # parameter gradients are skipped as a whole when no parameter asks for one (the trainer
        # turns them off for batches whose discriminator gradients would be zeroed unused)
        need_w = any(ctx.needs_input_grad[5:])
        if need_w:
            dw5, db5 = ops.conv_k1_bwd_weight(y4, dout, w5.shape[1], 4, 1)
            grads[2 + 18], grads[2 + 19] = dw5, db5
        dy = ops.conv_k1_bwd_data(dout, w5.detach().contiguous(), y4, 4, 1, relu_mask=False)
        for i in (2, 1, 0):
            wt, b, gamma = rest[6 * i], rest[6 * i + 1], rest[6 * i + 2]
            z, st, yin = zs[i], sts[i], acts[i]
            K, C = wt.shape[0], yin.shape[-1]
            ih, iw, oh, ow = yin.shape[1], yin.shape[2], z.shape[1], z.shape[2]
            dz, dgamma, dbeta = ops.bn_backward(2, z, dy, st, gamma.detach(), K, ctx.training)
            if need_w:
                dw = torch.empty_like(wt)
                ops.wgrad4x4s2(yin, dz, dw)
                db = ops.channel_sum(dz, K)
                grads[2 + 6 * i], grads[2 + 6 * i + 1] = dw, db
                grads[2 + 6 * i + 2], grads[2 + 6 * i + 3] = dgamma, dbeta
            # data gradient: four parity classes of the transposed stride-2 conv
            dyin = torch.empty_like(yin)
            if ops.conv4x4s2_ok(dz, C, C):
                # layer 0 sits on a conv without BatchNorm: fold its LeakyReLU derivative in
                ops.conv4x4s2(dz, ops.conv4x4s2_weights(wt, True), dyin, C, dgrad=True, aux=yin if i == 0 else None,
                              aux_mode=2 if i == 0 else 0)
                dy = dyin
                continue
            problems = []
            for (py, px, taps, rs, mh, mw) in _s2_dgrad_classes(ih, iw):
                wT = ops.pack_conv_weight(wt.detach().contiguous(), cdtype, rs, True, C, K)
                geom = ops.conv_geom(N, oh, ow, K, mh, mw, 1, 1, ih, iw, taps, osh=2, osw=2, ooy=py, oox=px)
                # layer 0 sits on a conv without BatchNorm: fold its LeakyReLU derivative in
                problems.append((dz, wT, dyin, geom, C, yin if i == 0 else None, 2 if i == 0 else 0))
            ops.conv_fprop_batch(problems)   # the four parity classes in one launch
            dy = dyin
        dx = None
        if ctx.xin is None:
            if need_w:
                dw1, db1 = ops.dconv1_bwd_weight(x, dy)
                grads[0], grads[1] = dw1, db1
            if ctx.needs_input_grad[0]:
                dx = ops.dconv1_bwd_data(dy, w1.detach().contiguous(), x.shape[2], x.shape[3])
        else:
            xin, (K1, Cin) = ctx.xin, w1.shape[:2]
            H, W, Cp, oh1, ow1 = xin.shape[1], xin.shape[2], xin.shape[3], dy.shape[1], dy.shape[2]
            if need_w:
                dw1 = torch.zeros_like(w1)
                ops.conv_wgrad(xin, dy, dw1, _s2_wgeom(N, H, W, Cp, oh1, ow1, Cin), K1)
                grads[0], grads[1] = dw1, ops.channel_sum(dy, K1)
            if ctx.needs_input_grad[0]:
                dxin = torch.zeros_like(xin)
                problems = []
                for (py, px, taps, rs, mh, mw) in _s2_dgrad_classes(H, W):
                    wT = ops.pack_conv_weight(w1.detach().contiguous(), cdtype, rs, True, Cp, K1)
                    geom = ops.conv_geom(N, oh1, ow1, K1, mh, mw, 1, 1, H, W, taps, osh=2, osw=2, ooy=py, oox=px)
                    problems.append((dy, wT, dxin, geom, Cp, None, 0))
                ops.conv_fprop_batch(problems)
                dx = ops.nhwc_to_nchw(dxin, Cin)
        ctx.x = ctx.acts = ctx.zs = ctx.sts = ctx.xin = None
        return (dx, None, None, None, None, *grads)


# ------------------------------------------------------------------------------ losses
class PackedStations:
    """Station rows of one batch in the layout sf_image_station_loss reads: int32 [S,2] positions
    (x, y), f32 [S] runoffs, int32 [B+1] sample offsets, all on the device, and the host-side
    total S.  A loss module given one of these (as positions, runoffs or both) skips packing."""

    def __init__(self, pos: torch.Tensor, run: torch.Tensor, offs: torch.Tensor, S: int):
        self.pos, self.run, self.offs, self.S = pos, run, offs, int(S)

    @classmethod
    def from_host(cls, pos, run, counts, device) -> "PackedStations":
        """pos int [S,2], run float [S], counts int [B] (numpy) -> one host-to-device copy."""
        import numpy as np
        S, B = int(np.sum(counts)), len(counts)
        n = max(S, 1)
        buf = np.zeros(3 * n + B + 1, dtype=np.int32)
        if S:
            buf[: 2 * S] = np.asarray(pos, dtype=np.int32).reshape(-1)
            buf[2 * n: 2 * n + S] = np.asarray(run, dtype=np.float32).reshape(-1).view(np.int32)
        buf[3 * n + 1:] = np.cumsum(counts)
        d = ops.upload(torch.from_numpy(buf), device)
        return cls(d[: 2 * n].view(-1, 2), d[2 * n: 3 * n].view(torch.float32), d[3 * n:], S)


def pack_stations(positions: Sequence[torch.Tensor], runoffs: Sequence[torch.Tensor], device):
    """List-of-tensors station batch (src/train.py:181-188) -> packed device arrays with ONE
    host-to-device copy: int32 [S,2] positions, f32 [S] runoffs, int32 [B+1] offsets."""
    for packed in (positions, runoffs):
        if isinstance(packed, PackedStations):
            return packed.pos, packed.run, packed.offs, packed.S
    B = len(positions)
    counts = [int(r.numel()) for r in runoffs]
    S = sum(counts)
    offs = [0]
    for c in counts:
        offs.append(offs[-1] + c)
    buf = torch.zeros(max(S, 1) * 3 + B + 1, dtype=torch.int32, pin_memory=torch.device(device).type == "cuda")
    if S:
        pos = torch.cat([p.reshape(-1, 2).to(torch.int32).cpu() for p, c in zip(positions, counts) if c], dim=0)
        run = torch.cat([r.reshape(-1).to(torch.float32).cpu() for r, c in zip(runoffs, counts) if c], dim=0)
        buf[: 2 * S] = pos.reshape(-1)
        buf[2 * S: 3 * S] = run.view(torch.int32)
    buf[3 * max(S, 1):] = torch.tensor(offs, dtype=torch.int32)
    d = ops.upload(buf, device)
    n = max(S, 1)
    return d[: 2 * n].view(-1, 2), d[2 * n: 3 * n].view(torch.float32), d[3 * n:], S


class ImageStationLossFn(torch.autograd.Function):
    """ImageLoss + StationLoss + their weighted sum (src/loss.py:47-70,116-160,227-230): one launch
    forward (losses; optionally the Inf/NaN check of the predictions, src/train.py:198), one launch
    backward (the complete gradient, upstream factor included).  Returns a [3] tensor: total,
    image, station; only `total` carries a gradient."""

    @staticmethod
    def forward(ctx, pred, target, pos, runoff, offsets, S, cfg, nan_flag=None):
        p, t = pred.contiguous().float(), target.contiguous().float()
        need = ctx.needs_input_grad[0]
        losses, sdiff = ops.image_station_loss_fwd(p, t, pos, runoff, offsets, S, *cfg, nan_flag=nan_flag,
                                                   keep_for_backward=need)
        if need:
            ctx.save_for_backward(p, t, pos, runoff, offsets, losses, sdiff)
            ctx.S, ctx.cfg = S, cfg
        return losses[:3]

    @staticmethod
    def backward(ctx, g):
        p, t, pos, runoff, offsets, losses, sdiff = ctx.saved_tensors
        up = g if g.is_contiguous() and g.dtype == torch.float32 else g.contiguous().float()    # element 0: d / d total
        dpred = ops.image_station_loss_bwd(p, t, pos, runoff, offsets, ctx.S, *ctx.cfg, losses, sdiff=sdiff, upstream=up)
        return dpred, None, None, None, None, None, None, None


class BceLogitsFn(torch.autograd.Function):
    """nn.BCEWithLogitsLoss() (mean) against a constant or a tensor label (src/model/gan.py:82)."""

    @staticmethod
    def forward(ctx, x, label, labels):
        loss, dx = ops.bce_logits(x.contiguous().float(), float(label),
                                  labels.contiguous().float() if labels is not None else None,
                                  need_grad=ctx.needs_input_grad[0])
        ctx.dx = dx
        ctx.shape = x.shape
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        dx = ctx.dx
        ctx.dx = None
        return (dx * g).reshape(ctx.shape), None, None
