"""The persistent 3x3 conv / data-gradient / weight-gradient kernels at the benchmark's two full
shapes (encoder conv2: 160 frames of 100 x 100, 128 -> 256; decoder conv2: 32 maps of 400 x 400,
64 -> 64): every workgroup walks many tiles there, which the small parity cases do not exercise.
Checked against the generic implicit-GEMM kernels on the whole tensors (an independent
implementation in the same library, itself pinned by the small cases against f32 references) and
against an f32 torch conv on the first and the last image.  Tolerance 1e-2 of the tensor's max
(bf16 output rounding; both sides accumulate in f32)."""
import pytest
import torch

from conftest import close, load_pkg

pytestmark = pytest.mark.gpu
DEV = "cuda"
SHAPES = [(160, 100, 100, 128, 256), (32, 400, 400, 64, 64)]


@pytest.fixture(scope="module")
def sf():
    return load_pkg()


def _rel(a, b):
    return float((a.float() - b.float()).abs().max() / b.float().abs().max())


@pytest.mark.parametrize("shape", SHAPES)
def test_full_size_conv_forward_and_data_gradient(sf, shape):
    ops = sf.ops
    N, H, W, C, K = shape
    torch.manual_seed(9)
    x = torch.randn(N, H, W, C, device=DEV).to(torch.bfloat16)
    w = (torch.randn(K, C, 3, 3, device=DEV) / (3.0 * C ** 0.5)).to(torch.bfloat16).float()
    b = torch.randn(K, device=DEV)
    wp = ops.pack_conv_weight(w.contiguous(), torch.bfloat16, list(range(9)), False, K, C)
    out = torch.empty(N, H, W, K, device=DEV, dtype=torch.bfloat16)
    ops.conv3x3(x, wp, out, K, bias=b, act=ops.ACT_RELU)
    ref = torch.empty_like(out)
    ops.FORCE_GENERIC_CONV = True
    try:
        ops.conv3x3(x, wp, ref, K, bias=b, act=ops.ACT_RELU)
    finally:
        ops.FORCE_GENERIC_CONV = False
    assert _rel(out, ref) <= 1e-2
    for n in (0, N - 1):
        f32 = torch.relu(torch.nn.functional.conv2d(x[n:n + 1].float().permute(0, 3, 1, 2).cpu(), w.cpu(), b.cpu(), padding=1))
        close(out[n:n + 1].float().permute(0, 3, 1, 2), f32.numpy(), 1e-2)
    # data gradient with a ReLU mask
    dz = torch.randn(N, H, W, K, device=DEV).to(torch.bfloat16)
    mask = torch.randn(N, H, W, C, device=DEV).to(torch.bfloat16)
    wT = ops.pack_conv_weight(w.contiguous(), torch.bfloat16, list(range(9)), True, C, K)
    dx = torch.empty(N, H, W, C, device=DEV, dtype=torch.bfloat16)
    ops.conv3x3(dz, wT, dx, C, dgrad=True, aux=mask, aux_mode=1)
    dref = torch.empty_like(dx)
    ops.FORCE_GENERIC_CONV = True
    try:
        ops.conv3x3(dz, wT, dref, C, dgrad=True, aux=mask, aux_mode=1)
    finally:
        ops.FORCE_GENERIC_CONV = False
    assert _rel(dx, dref) <= 1e-2
    n = N - 1
    f32 = torch.nn.grad.conv2d_input((1, C, H, W), w.cpu(), dz[n:n + 1].float().permute(0, 3, 1, 2).cpu(), padding=1)
    f32 = f32 * (mask[n:n + 1].float().permute(0, 3, 1, 2).cpu() > 0)
    close(dx[n:n + 1].float().permute(0, 3, 1, 2), f32.numpy(), 1e-2)


@pytest.mark.parametrize("shape", SHAPES)
def test_full_size_weight_gradient(sf, shape):
    """Linearity pins the full-size sum: dW(x, dz) over all images equals the sum of dW over two
    halves of the batch (f32 accumulation, 1e-4), and one half-sized call equals the generic GEMM."""
    ops = sf.ops
    N, H, W, C, K = shape
    torch.manual_seed(10)
    x = torch.randn(N, H, W, C, device=DEV).to(torch.bfloat16)
    dz = torch.randn(N, H, W, K, device=DEV).to(torch.bfloat16)
    dw = torch.empty(K, C, 3, 3, device=DEV)
    ops.wgrad3x3(x, dz, dw)
    h = N // 2
    d0, d1 = torch.empty_like(dw), torch.empty_like(dw)
    ops.wgrad3x3(x[:h].contiguous(), dz[:h].contiguous(), d0)
    ops.wgrad3x3(x[h:].contiguous(), dz[h:].contiguous(), d1)
    assert _rel(dw, d0 + d1) <= 1e-4
    q = max(1, N // 16)
    dq = torch.empty_like(dw)
    ops.wgrad3x3(x[:q].contiguous(), dz[:q].contiguous(), dq)
    dg = torch.zeros_like(dw)
    taps = [(r - 1, s - 1) for r in range(3) for s in range(3)]
    geom = ops.conv_geom(q, H, W, C, H, W, 1, 1, H, W, taps, tap_rs=list(range(9)), cin_real=C, RS=9)
    ops.conv_wgrad(x[:q].contiguous(), dz[:q].contiguous(), dg, geom, K)
    assert _rel(dq, dg) <= 1e-4


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_full_size_upconv_batch_independence(sf, dtype):
    """sf_upconv3_fwd / sf_upconv3_bwd at the decoder's full shape (32 maps, 32 x 32 -> 400 x 400,
    64 channels): an image's result does not depend on the batch it travels in, so the rows of a
    32-image call must equal, bit for bit, a 2-image call on the same data (whose values the
    small-shape test pins against interpolate + conv)."""
    ops = sf.ops
    B, h, w, OH, OW, K = 32, 32, 32, 400, 400, 64
    torch.manual_seed(12)
    y9 = torch.randn(B, h, w, 9, K, device=DEV)
    bias = torch.randn(K, device=DEV)
    out = ops.upconv3_fwd(y9, bias, OH, OW, dtype, relu=True)
    for sl in (slice(0, 2), slice(B - 2, B)):
        part = ops.upconv3_fwd(y9[sl].contiguous(), bias, OH, OW, dtype, relu=True)
        assert torch.equal(out[sl], part)
    dz = torch.randn(B, OH, OW, K, device=DEV).to(dtype)
    g = ops.upconv3_bwd(dz, h, w, dtype)
    for sl in (slice(0, 2), slice(B - 2, B)):
        part = ops.upconv3_bwd(dz[sl].contiguous(), h, w, dtype)
        assert torch.equal(g[sl], part)
    # adjoint identity <up(y9), dz> = <y9, up^T(dz)> (no bias, no ReLU), f32 only
    if dtype == torch.float32:
        lin = ops.upconv3_fwd(y9, None, OH, OW, dtype, relu=False)
        lhs = float((lin.double() * dz.double()).sum())
        rhs = float((y9.double() * g.view(B, h, w, 9, K).double()).sum())
        assert abs(lhs - rhs) <= 1e-5 * max(abs(lhs), abs(rhs), 1.0)


@pytest.mark.parametrize("layer", [(200, 64, 128), (100, 128, 256), (50, 256, 512)])
def test_full_size_patchgan_stride2_layers(sf, layer):
    """The PatchGAN's 4x4 stride-2 convs (src/model/gan.py:30-52) at the benchmark's batch of 32
    maps: forward through the implicit GEMM and weight gradient through sf_wgrad4x4s2.  The GEMM
    flattens the batch into one pixel axis, so the full call must reproduce a two-image call bit
    for bit on those images; the first image is checked against an f32 conv (1e-2: bf16 output);
    the weight gradient is additive over halves of the batch (1e-4)."""
    import importlib
    ops = sf.ops
    fn = importlib.import_module(sf.__name__ + ".functional")
    IH, C, K = layer
    N, OH = 32, IH // 2
    torch.manual_seed(13)
    x = torch.randn(N, IH, IH, C, device=DEV).to(torch.bfloat16)
    w = (torch.randn(K, C, 4, 4, device=DEV) / (4.0 * C ** 0.5)).to(torch.bfloat16).float()
    b = torch.randn(K, device=DEV)
    wp = ops.pack_conv_weight(w.contiguous(), torch.bfloat16, list(range(16)), False, K, C)
    z = torch.empty(N, OH, OH, K, device=DEV, dtype=torch.bfloat16)
    ops.conv_fprop(x, wp, z, fn._s2_geom(N, IH, IH, C, OH, OH), K, bias=b)
    for sl in (slice(0, 2), slice(N - 2, N)):
        part = torch.empty(2, OH, OH, K, device=DEV, dtype=torch.bfloat16)
        ops.conv_fprop(x[sl].contiguous(), wp, part, fn._s2_geom(2, IH, IH, C, OH, OH), K, bias=b)
        assert torch.equal(z[sl], part)
    f32 = torch.nn.functional.conv2d(x[:1].float().permute(0, 3, 1, 2).cpu(), w.cpu(), b.cpu(), stride=2, padding=1)
    close(z[:1].float().permute(0, 3, 1, 2), f32.numpy(), 1e-2)
    dz = torch.randn(N, OH, OH, K, device=DEV).to(torch.bfloat16)
    dw, d0, d1 = (torch.empty(K, C, 4, 4, device=DEV) for _ in range(3))
    ops.wgrad4x4s2(x, dz, dw)
    h = N // 2
    ops.wgrad4x4s2(x[:h].contiguous(), dz[:h].contiguous(), d0)
    ops.wgrad4x4s2(x[h:].contiguous(), dz[h:].contiguous(), d1)
    assert _rel(dw, d0 + d1) <= 1e-4


def test_full_size_encoder_batchnorm_pool_passes(sf):
    """Second encoder layer's BatchNorm -> ReLU -> adaptive pool (100 x 100 -> 32 x 32, 256 channels,
    160 frames; src/model/cnn_encoder.py:60-97) at full size: the bf16 kernels against the f32
    kernels of the same entry points on the same (bf16-representable) values -- the f32 ones are
    pinned against autograd at small shapes.  Statistics 1e-5, pooled output 1e-2, gradients 2e-2 of
    the max (bf16 rounding of the outputs; the inputs are identical)."""
    ops = sf.ops
    N, H, W, C, OH = 160, 100, 100, 256, 32
    torch.manual_seed(17)
    xb = torch.randn(N, H, W, C, device=DEV).to(torch.bfloat16)
    gamma, beta = torch.rand(C, device=DEV) + 0.5, torch.randn(C, device=DEV) * 0.3
    rm, rv = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    dyb = torch.randn(N, OH, OH, C, device=DEV).to(torch.bfloat16)
    res = {}
    for dt in (torch.bfloat16, torch.float32):
        x, dy = xb.to(dt), dyb.to(dt)
        ws = ops.bn_ws(C, DEV)
        ops.bn_stats(x, N * H * W, C, ws)
        st = ops.bn_finalize(ws[:2 * C], N * H * W, C, C, gamma, beta, rm.clone(), rv.clone(), True)
        y = ops.bn_relu_avgpool_fwd(x, st, C, OH, OH, torch.float32)
        dx, dg, db = ops.bn_backward(1, x, dy, st, gamma, C, True, OH, OH)
        res[dt] = (st.clone(), y, dx.float(), dg, db)
        del x, dy, dx
    a, b = res[torch.bfloat16], res[torch.float32]
    assert _rel(a[0], b[0]) <= 1e-5
    assert _rel(a[1], b[1]) <= 1e-2
    assert _rel(a[2], b[2]) <= 2e-2
    assert _rel(a[3], b[3]) <= 2e-2 and _rel(a[4], b[4]) <= 2e-2


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_full_size_output_conv_passes(sf, dtype):
    """Decoder output conv (ReLU'd 64-channel map -> 1 channel, 1x1; src/model/decoder.py:174-177) at
    32 x 400 x 400: sf_conv_k1_fwd and the fused backward (input gradient with the ReLU mask, weight,
    bias and channel-sum gradients in one pass) against torch expressions in f64 on the same
    operands.  Forward 1e-5 (f32 accumulation); input gradient exact up to its output rounding
    (1e-2 bf16 / 1e-6 f32); the reduced gradients 1e-4."""
    ops = sf.ops
    N, H, W, C = 32, 400, 400, 64
    torch.manual_seed(18)
    x = torch.relu(torch.randn(N, H, W, C, device=DEV)).to(dtype)        # post-ReLU activations, half of them zero
    w = torch.randn(1, C, 1, 1, device=DEV) / C ** 0.5
    b = torch.randn(1, device=DEV)
    y = ops.conv_k1_fwd(x, w, b, 1, 0)
    yr = (x.double() * w.view(C).double()).sum(-1) + b.double()
    assert _rel(y.view(N, H, W), yr) <= 1e-5
    dy = torch.randn(N, 1, H, W, device=DEV)
    assert ops.conv_k1_bwd_fused_ok(x)
    dx, dw, db, dsum = ops.conv_k1_bwd_fused(x, dy, w, C)
    dxr = dy.view(N, H, W, 1).double() * w.view(C).double() * (x > 0)
    assert _rel(dx, dxr) <= (1e-2 if dtype == torch.bfloat16 else 1e-6)
    assert _rel(dw.view(C), (dy.view(N, H, W, 1).double() * x.double()).sum((0, 1, 2))) <= 1e-4
    assert _rel(db, dy.double().sum().view(1)) <= 1e-4
    # the channel sums are those of the input gradient as stored (rounded to the storage dtype)
    assert _rel(dsum, dx.double().sum((0, 1, 2))) <= 1e-4


def test_full_size_wgrad_with_tile_stealing(sf):
    """sf_wgrad3x3 with stealing allowed (the data-parallel setting) at the two benchmark shapes, and
    with every second workgroup leaving its whole list to the others (so that tiles really are
    stolen): same result as the fixed slices up to the f32 summation order (1e-5 of the maximum), and
    repeated launches reuse the counter slots without leaving them dirty."""
    ops = sf.ops
    for (N, H, C, K) in ((160, 100, 128, 256), (32, 400, 64, 64), (3, 50, 64, 128)):
        torch.manual_seed(23)
        x = torch.randn(N, H, H, C, device=DEV).to(torch.bfloat16)
        dz = torch.randn(N, H, H, K, device=DEV).to(torch.bfloat16)
        ref = torch.empty(K, C, 3, 3, device=DEV)
        ops.wgrad3x3(x, dz, ref)
        try:
            for mode in (1, 2):
                ops.set_wgrad_scheduling(mode)
                for _ in range(70):          # more launches than counter slots
                    dw = torch.empty(K, C, 3, 3, device=DEV)
                    ops.wgrad3x3(x, dz, dw)
                assert _rel(dw, ref) <= 1e-5, (mode, N, H, C, K)
        finally:
            ops.set_wgrad_scheduling(0)
        again = torch.empty(K, C, 3, 3, device=DEV)
        ops.wgrad3x3(x, dz, again)
        assert torch.equal(again, ref)      # fixed slices: bit-reproducible


def test_full_size_patchgan_head(sf):
    """The 4x4 one-channel head at the benchmark shape (32 x 25 x 25 x 512, bf16) on its MFMA kernels
    (tap products + shifted sum; data gradient as a 16-deep product) and at a ragged shape, against
    f32 torch: forward, data gradient with and without the ReLU mask, weight and bias gradient
    (stated bf16 tolerance: 1e-2 of the tensor's maximum)."""
    ops = sf.ops
    for (N, H, W, C) in ((32, 25, 25, 512), (3, 7, 9, 64)):
        torch.manual_seed(17)
        x = torch.randn(N, H, W, C, device=DEV).to(torch.bfloat16)
        w = torch.randn(1, C, 4, 4, device=DEV) * 0.02
        b = torch.randn(1, device=DEV)
        y = ops.conv_k1_fwd(x, w, b, 4, 1)
        xg = x.float().permute(0, 3, 1, 2).requires_grad_(True)
        wg = w.clone().requires_grad_(True)
        bg = b.clone().requires_grad_(True)
        ref = torch.nn.functional.conv2d(xg, wg, bg, padding=1)
        assert _rel(y.reshape(ref.shape), ref.detach()) <= 1e-2
        dy = torch.randn_like(ref)
        ref.backward(dy)
        dx = ops.conv_k1_bwd_data(dy.reshape(y.shape).contiguous(), w, x, 4, 1, relu_mask=False)
        assert _rel(dx.float().permute(0, 3, 1, 2), xg.grad) <= 1e-2
        dxm = ops.conv_k1_bwd_data(dy.reshape(y.shape).contiguous(), w, x, 4, 1, relu_mask=True)
        assert _rel(dxm.float().permute(0, 3, 1, 2), xg.grad * (xg.detach() > 0)) <= 1e-2
        dw, db = ops.conv_k1_bwd_weight(x, dy.reshape(y.shape).contiguous(), C, 4, 1)
        assert _rel(dw, wg.grad) <= 1e-2 and _rel(db.reshape(-1), bg.grad) <= 1e-2
