"""Baseline comparison models of the reference (src/model/baseline_models.py) on the HIP kernels:
LightweightCNNReducer, PureLSTMModel ('LSTM'), PureBPModel ('BP') and AttentionGRUModel
('Attention-GRU'), with the reference's constructor arguments, forward signature and state_dict
keys.

The reducer's stride-2 3x3 convolutions run on the generic implicit-GEMM kernel (forward, weight
gradient, and the data gradient as four parity classes in one launch); BatchNorm, the Linear
layers, the LSTM and the 3x3 / 1x1 head use the same entry points as the main model.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch
import torch.nn as nn

from . import functional as SF
from . import ops


def _s2_3x3_geom(N, IH, IW, C, OH, OW, cin_real=0):
    taps = [(r - 1, s - 1) for r in range(3) for s in range(3)]
    if cin_real:
        return ops.conv_geom(N, IH, IW, C, OH, OW, 2, 2, OH, OW, taps, tap_rs=list(range(9)), cin_real=cin_real, RS=9)
    return ops.conv_geom(N, IH, IW, C, OH, OW, 2, 2, OH, OW, taps, tap_rs=list(range(9)))


def _s2_3x3_dgrad_classes(IH, IW):
    """Transposed 3x3 stride-2 pad-1 conv as stride-1 convs, one per input-pixel parity: input
    row iy = 2m + py receives tap r from output row m + (py + 1 - r) / 2 where that is whole."""
    out = []
    for py in (0, 1):
        for px in (0, 1):
            rows = [(r, (py + 1 - r) // 2) for r in range(3) if (py + 1 - r) % 2 == 0]
            cols = [(s, (px + 1 - s) // 2) for s in range(3) if (px + 1 - s) % 2 == 0]
            taps = [(dy, dx) for (_, dy) in rows for (_, dx) in cols]
            rs = [r * 3 + s for (r, _) in rows for (s, _) in cols]
            mh, mw = (IH - py + 1) // 2, (IW - px + 1) // 2
            if mh > 0 and mw > 0:
                out.append((py, px, taps, rs, mh, mw))
    return out


class ReducerFn(torch.autograd.Function):
    """LightweightCNNReducer.forward (src/model/baseline_models.py:53-56): three times
    conv3x3 stride 2 (no bias) -> BatchNorm -> ReLU, then adaptive average pool to target x target."""

    @staticmethod
    def forward(ctx, x, training, cdtype, target, *params):
        L = len(params) // 5
        N, Cin, H, W = x.shape
        e = ops.epc(ops.code_of(cdtype))
        a = ops.nchw_to_nhwc(x.contiguous().float(), cdtype, ops.roundup(Cin, 16 if cdtype == torch.bfloat16 else e))
        saved = []
        h, w = H, W
        out = None
        for l in range(L):
            wt, gamma, beta, rm, rv = params[5 * l: 5 * l + 5]
            K, C = wt.shape[0], a.shape[-1]
            if K % 8:
                raise NotImplementedError("reducer widths must be multiples of 8 on the HIP path")
            oh, ow = (h - 1) // 2 + 1, (w - 1) // 2 + 1
            wp = ops.pack_conv_weight(wt.detach().contiguous(), cdtype, list(range(9)), False, K, C)
            z = torch.empty((N, oh, ow, K), device=x.device, dtype=cdtype)
            ops.conv_fprop(a, wp, z, _s2_3x3_geom(N, h, w, C, oh, ow), K)
            M = N * oh * ow
            ws = None
            if training:
                ws = ops.bn_ws(K, x.device)
                ops.bn_stats(z, M, K, ws)
            st = ops.bn_finalize(ws, M, K, K, gamma.detach(), beta.detach(), rm, rv, training)
            saved.append((a, z, st, (h, w)))
            if l < L - 1:
                a = ops.bn_relu_fwd(z, st)
            else:
                out = ops.bn_relu_avgpool_fwd(z, st, K, target, target, torch.float32)
            h, w = oh, ow
        ctx.saved, ctx.params, ctx.training, ctx.cdtype = saved, params, training, cdtype
        ctx.in_shape = (N, Cin, H, W)
        return out

    @staticmethod
    def backward(ctx, dout):
        params, cdtype, training = ctx.params, ctx.cdtype, ctx.training
        L = len(params) // 5
        N, Cin, H, W = ctx.in_shape
        grads: List[Optional[torch.Tensor]] = [None] * len(params)
        toh, tow = dout.shape[-2], dout.shape[-1]
        dy = ops.nchw_to_nhwc(dout.contiguous().float(), cdtype, params[5 * (L - 1)].shape[0])
        dx = None
        for l in range(L - 1, -1, -1):
            wt, gamma = params[5 * l], params[5 * l + 1]
            a, z, st, (h, w) = ctx.saved[l]
            K, C, cin_real = wt.shape[0], a.shape[-1], wt.shape[1]
            oh, ow = z.shape[1], z.shape[2]
            if l == L - 1:
                dz, dgamma, dbeta = ops.bn_backward(1, z, dy, st, gamma.detach(), K, training, toh, tow)
            else:
                dz, dgamma, dbeta = ops.bn_backward(3, z, dy, st, gamma.detach(), K, training)
            dw = torch.zeros_like(wt)
            ops.conv_wgrad(a, dz, dw, _s2_3x3_geom(N, h, w, C, oh, ow, cin_real=cin_real), K)
            grads[5 * l], grads[5 * l + 1], grads[5 * l + 2] = dw, dgamma, dbeta
            if l > 0 or ctx.needs_input_grad[0]:
                da = torch.empty((N, h, w, C), device=dz.device, dtype=cdtype)
                problems = []
                for (py, px, taps, rs, mh, mw) in _s2_3x3_dgrad_classes(h, w):
                    wT = ops.pack_conv_weight(wt.detach().contiguous(), cdtype, rs, True, C, K)
                    geom = ops.conv_geom(N, oh, ow, K, mh, mw, 1, 1, h, w, taps, osh=2, osw=2, ooy=py, oox=px)
                    problems.append((dz, wT, da, geom, C, None, 0))
                ops.conv_fprop_batch(problems)
                if l > 0:
                    dy = da
                else:
                    dx = ops.nhwc_to_nchw(da, Cin)
        ctx.saved = None
        return (dx, None, None, None, *grads)


class ReluDropoutFn(torch.autograd.Function):
    """nn.ReLU followed by nn.Dropout(p) (p = 0 outside training)."""

    @staticmethod
    def forward(ctx, x, p, seed):
        y = ops.relu_dropout_fwd(x.contiguous().float(), p, seed)
        ctx.save_for_backward(y)
        ctx.p = p
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        return ops.relu_dropout_bwd(y, dy.contiguous().float(), ctx.p), None, None


class GruFn(torch.autograd.Function):
    """nn.GRU(H, H, L, batch_first=True)(x)[0]: the output sequence of the last layer."""

    @staticmethod
    def forward(ctx, x, L, p_drop, seed, *ws):
        wsd = [w.detach().contiguous() for w in ws]
        x = x.contiguous().float()
        y, save = ops.gru_fwd(x, wsd, L, p_drop, seed, any(ctx.needs_input_grad))
        ctx.x, ctx.save, ctx.ws, ctx.cfg = x, save, wsd, (L, p_drop, seed)
        return y

    @staticmethod
    def backward(ctx, dy):
        L, p_drop, seed = ctx.cfg
        dws = [torch.zeros_like(w) for w in ctx.ws]
        dx = ops.gru_bwd(ctx.x, dy.contiguous().float(), ctx.ws, dws, ctx.save, L, p_drop, seed)
        ctx.x = ctx.save = None
        return (dx, None, None, None, *dws)


class TemporalAttentionFn(torch.autograd.Function):
    """TemporalAttention.forward (src/model/baseline_models.py:231-247)."""

    @staticmethod
    def forward(ctx, g, *ws):
        wsd = [w.detach().contiguous() for w in ws]
        g = g.contiguous().float()
        out, save = ops.attn_fwd(g, wsd, any(ctx.needs_input_grad))
        ctx.g, ctx.save, ctx.ws = g, save, wsd
        return out

    @staticmethod
    def backward(ctx, dout):
        dws = [torch.zeros_like(w) for w in ctx.ws]
        dg = ops.attn_bwd(ctx.g, dout.contiguous().float(), ctx.ws, dws, ctx.save)
        ctx.g = ctx.save = None
        return (dg, *dws)


class HeadFn(torch.autograd.Function):
    """The output head shared by the baselines (src/model/baseline_models.py:114-118,129-133):
    [B, 64*s*s] -> view [B,64,s,s] -> conv3x3 + ReLU -> conv1x1 -> bilinear resize to (OH, OW)."""

    @staticmethod
    def forward(ctx, feat, size, OH, OW, cdtype, w1, b1, w2, b2):
        B = feat.shape[0]
        Hc = w1.shape[0]
        if Hc % 8 or w1.shape[1] % 8:
            raise NotImplementedError("head channels must be multiples of 8 on the HIP path")
        src = ops.nchw_to_nhwc(feat.contiguous().float().view(B, w1.shape[1], size, size), cdtype, w1.shape[1])
        w1p = ops.pack_conv_weight(w1.detach().contiguous(), cdtype, list(range(9)), False, Hc, w1.shape[1])
        h1 = torch.empty((B, size, size, Hc), device=feat.device, dtype=cdtype)
        ops.conv3x3(src, w1p, h1, Hc, bias=b1.detach().contiguous().float(), act=ops.ACT_RELU)
        w2c = w2.detach().contiguous()
        y8 = ops.conv_k1_fwd(h1, w2c, b2.detach().contiguous(), 1, 0)            # [B,1,s,s] f32
        Bp = ops.roundup(B, 4)
        lo = torch.zeros((1, size, size, Bp), device=feat.device, dtype=torch.float32)
        lo[0, :, :, :B] = y8[:, 0].permute(1, 2, 0)
        hi = ops.bilinear_fwd(lo, OH, OW, torch.float32)                        # the batch rides in the channel axis
        ctx.save_for_backward(src, h1, w2c)
        ctx.w1 = w1
        ctx.dims = (B, Bp, size, Hc, cdtype)
        return hi[0, :, :, :B].permute(2, 0, 1).unsqueeze(1).contiguous()

    @staticmethod
    def backward(ctx, dy):
        src, h1, w2c = ctx.saved_tensors
        w1 = ctx.w1
        B, Bp, size, Hc, cdtype = ctx.dims
        OH, OW = dy.shape[-2], dy.shape[-1]
        g = torch.zeros((1, OH, OW, Bp), device=dy.device, dtype=torch.float32)
        g[0, :, :, :B] = dy.float()[:, 0].permute(1, 2, 0)
        dlo = ops.bilinear_bwd(g, size, size)                                    # [1,s,s,Bp]
        dy8 = dlo[0, :, :, :B].permute(2, 0, 1).unsqueeze(1).contiguous()
        dw2, db2 = ops.conv_k1_bwd_weight(h1, dy8, Hc, 1, 0)
        dz1 = ops.conv_k1_bwd_data(dy8, w2c, h1, 1, 0, relu_mask=True)
        db1 = ops.channel_sum(dz1, Hc)
        dw1 = torch.empty_like(w1)
        ops.wgrad3x3(src, dz1, dw1)
        dfeat = None
        if ctx.needs_input_grad[0]:
            Cin = w1.shape[1]
            w1T = ops.pack_conv_weight(w1.detach().contiguous(), cdtype, list(range(9)), True, Cin, Hc)
            dsrc = torch.empty_like(src)
            ops.conv3x3(dz1, w1T, dsrc, Cin, dgrad=True)
            dfeat = ops.nhwc_to_nchw(dsrc, Cin).view(B, -1)
        return dfeat, None, None, None, None, dw1, db1, dw2, db2


def _need_cuda(t: torch.Tensor, who: str):
    if not t.is_cuda:
        raise RuntimeError(f"{who}: the HIP path needs GPU tensors (no CPU fallback exists)")


class LightweightCNNReducer(nn.Module):
    """Mirror of src/model/baseline_models.py:21-56."""

    def __init__(self, in_channels: int = 10, out_channels: int = 16, target_size: int = 8):
        super().__init__()
        self.target_size, self.out_channels = target_size, out_channels
        self.encoder = nn.Sequential(
            nn.Conv2d(in_channels, 32, kernel_size=3, stride=2, padding=1, bias=False), nn.BatchNorm2d(32), nn.ReLU(inplace=True),
            nn.Conv2d(32, 32, kernel_size=3, stride=2, padding=1, bias=False), nn.BatchNorm2d(32), nn.ReLU(inplace=True),
            nn.Conv2d(32, out_channels, kernel_size=3, stride=2, padding=1, bias=False), nn.BatchNorm2d(out_channels),
            nn.ReLU(inplace=True))
        self.pool = nn.AdaptiveAvgPool2d((target_size, target_size))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        _need_cuda(x, "LightweightCNNReducer")
        ps = []
        for i in (0, 3, 6):
            conv, bn = self.encoder[i], self.encoder[i + 1]
            ps += [conv.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var]
            if self.training:
                bn.num_batches_tracked += 1
        return ReducerFn.apply(x, self.training, SF.compute_dtype(), self.target_size, *ps)


class _BaselineBase(nn.Module):
    """Config handling and the output head common to the three reference baselines."""

    def __init__(self, config: Dict, head_in: int):
        super().__init__()
        self.config = config
        mcfg, dcfg = config.get("model", {}), config.get("data", {})
        self.input_channels = mcfg.get("input_channels", 10)
        self.hidden_dim = mcfg.get("hidden_dim", 64)
        self.num_layers = mcfg.get("num_layers", 2)
        self.window_size = dcfg.get("window_size", 5)
        size = dcfg.get("image_size", [256, 256])
        self.output_h, self.output_w = (size, size) if isinstance(size, int) else (int(size[0]), int(size[1]))
        up = dcfg.get("upscale_factor", 1)
        self.output_h *= up
        self.output_w *= up
        self.reduce_size, self.cnn_out_ch, self.decode_size = 8, 16, 8
        self.reducer = LightweightCNNReducer(self.input_channels, self.cnn_out_ch, self.reduce_size)
        self._head_in = head_in

    def _make_head(self):
        self.head_proj = nn.Linear(self._head_in, 64 * self.decode_size * self.decode_size)
        self.decoder = nn.Sequential(nn.Conv2d(64, 64, 3, padding=1), nn.ReLU(inplace=True), nn.Conv2d(64, 1, 1))

    def _head(self, h: torch.Tensor) -> torch.Tensor:
        cd = SF.compute_dtype()
        feat = SF.LinearFn.apply(h, self.head_proj.weight, self.head_proj.bias, cd)
        return HeadFn.apply(feat, self.decode_size, self.output_h, self.output_w, cd, self.decoder[0].weight,
                            self.decoder[0].bias, self.decoder[2].weight, self.decoder[2].bias)

    def _reduce(self, images: torch.Tensor) -> torch.Tensor:
        _need_cuda(images, type(self).__name__)
        B, T, C, H, W = images.shape
        return self.reducer(images.reshape(B * T, C, H, W))                  # [B*T, 16, 8, 8]

    def get_num_parameters(self) -> int:
        return sum(p.numel() for p in self.parameters() if p.requires_grad)


class PureLSTMModel(_BaselineBase):
    """Mirror of src/model/baseline_models.py:63-136: reducer -> Linear -> LSTM -> head."""

    def __init__(self, config: Dict):
        hidden = config.get("model", {}).get("hidden_dim", 64)
        super().__init__(config, hidden)
        flat = self.cnn_out_ch * self.reduce_size * self.reduce_size
        self.input_proj = nn.Linear(flat, self.hidden_dim)
        self.lstm = nn.LSTM(input_size=self.hidden_dim, hidden_size=self.hidden_dim, num_layers=self.num_layers,
                            dropout=0.2 if self.num_layers > 1 else 0, batch_first=True)
        self._make_head()

    def forward(self, images: torch.Tensor) -> torch.Tensor:
        B, T = images.shape[0], images.shape[1]
        x = self._reduce(images).reshape(B * T, -1)
        x = SF.LinearFn.apply(x, self.input_proj.weight, self.input_proj.bias, SF.compute_dtype()).view(B, T, -1)
        p = float(self.lstm.dropout) if self.training else 0.0      # nn.LSTM's inter-layer dropout (0 for one layer)
        ws = []
        for l in range(self.num_layers):
            ws += [getattr(self.lstm, f"weight_ih_l{l}"), getattr(self.lstm, f"weight_hh_l{l}"),
                   getattr(self.lstm, f"bias_ih_l{l}"), getattr(self.lstm, f"bias_hh_l{l}")]
        h = SF.LstmFn.apply(x, self.num_layers, p, SF.new_seed() if p > 0 else 0, *ws)
        return self._head(h)


class PureBPModel(_BaselineBase):
    """Mirror of src/model/baseline_models.py:143-214: reducer -> three-layer MLP -> head."""

    def __init__(self, config: Dict):
        hidden = config.get("model", {}).get("hidden_dim", 64)
        super().__init__(config, hidden * 2)
        flat = self.window_size * self.cnn_out_ch * self.reduce_size * self.reduce_size
        h4 = self.hidden_dim * 4
        self.mlp = nn.Sequential(nn.Linear(flat, h4), nn.ReLU(inplace=True), nn.Dropout(0.2),
                                 nn.Linear(h4, h4), nn.ReLU(inplace=True), nn.Dropout(0.2),
                                 nn.Linear(h4, self.hidden_dim * 2), nn.ReLU(inplace=True))
        self._make_head()

    def forward(self, images: torch.Tensor) -> torch.Tensor:
        B = images.shape[0]
        x = self._reduce(images).reshape(B, -1)
        cd = SF.compute_dtype()
        for i, p in ((0, self.mlp[2].p), (3, self.mlp[5].p), (6, 0.0)):
            p = p if self.training else 0.0
            x = SF.LinearFn.apply(x, self.mlp[i].weight, self.mlp[i].bias, cd)
            x = ReluDropoutFn.apply(x, p, SF.new_seed() if p > 0 else 0)
        return self._head(x)


class TemporalAttention(nn.Module):
    """Mirror of src/model/baseline_models.py:221-247."""

    def __init__(self, hidden_dim: int):
        super().__init__()
        self.query = nn.Linear(hidden_dim, hidden_dim)
        self.key = nn.Linear(hidden_dim, hidden_dim)
        self.value = nn.Linear(hidden_dim, hidden_dim)
        self.scale = hidden_dim ** 0.5

    def forward(self, gru_outputs: torch.Tensor) -> torch.Tensor:
        _need_cuda(gru_outputs, "TemporalAttention")
        return TemporalAttentionFn.apply(gru_outputs, self.query.weight, self.query.bias, self.key.weight, self.key.bias,
                                         self.value.weight, self.value.bias)


class AttentionGRUModel(_BaselineBase):
    """Mirror of src/model/baseline_models.py:250-326: reducer -> Linear -> GRU -> attention -> head."""

    def __init__(self, config: Dict):
        hidden = config.get("model", {}).get("hidden_dim", 64)
        super().__init__(config, hidden)
        flat = self.cnn_out_ch * self.reduce_size * self.reduce_size
        self.input_proj = nn.Linear(flat, self.hidden_dim)
        self.gru = nn.GRU(input_size=self.hidden_dim, hidden_size=self.hidden_dim, num_layers=self.num_layers,
                          dropout=0.2 if self.num_layers > 1 else 0, batch_first=True)
        self.attention = TemporalAttention(self.hidden_dim)
        self._make_head()

    def forward(self, images: torch.Tensor) -> torch.Tensor:
        B, T = images.shape[0], images.shape[1]
        x = self._reduce(images).reshape(B * T, -1)
        x = SF.LinearFn.apply(x, self.input_proj.weight, self.input_proj.bias, SF.compute_dtype()).view(B, T, -1)
        p = float(self.gru.dropout) if self.training else 0.0
        ws = []
        for l in range(self.num_layers):
            ws += [getattr(self.gru, f"weight_ih_l{l}"), getattr(self.gru, f"weight_hh_l{l}"),
                   getattr(self.gru, f"bias_ih_l{l}"), getattr(self.gru, f"bias_hh_l{l}")]
        g = GruFn.apply(x, self.num_layers, p, SF.new_seed() if p > 0 else 0, *ws)
        return self._head(self.attention(g))


BASELINE_MODELS = {"LSTM": PureLSTMModel, "BP": PureBPModel, "Attention-GRU": AttentionGRUModel}


def build_baseline_model(name: str, config: Dict) -> nn.Module:
    """src/model/baseline_models.py:340-352."""
    if name not in BASELINE_MODELS:
        raise ValueError(f"Unknown baseline model: {name}. Choose from {list(BASELINE_MODELS)}")
    return BASELINE_MODELS[name](config)
