"""sf_conv4x4s2: the PatchGAN's 4x4 stride-2 pad-1 convolution (src/model/gan.py:24-34), forward and
data gradient, on the persistent LDS-DMA kernel -- against f32 torch convolutions of the same
bf16-representable operands (the output is rounded to bf16 once: 6e-3 of the tensor's maximum),
at small ragged shapes (odd sizes, channel counts below one tile, masks, bias) and at the
benchmark's three layer shapes with 32 maps, where it must also agree with the generic
implicit-GEMM path on the whole tensor."""
import importlib

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import close, load_pkg

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def sf():
    return load_pkg()


def _operands(N, IH, IW, C, K, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(N, IH, IW, C, generator=g).to(torch.bfloat16)
    w = (torch.randn(K, C, 4, 4, generator=g) / (4.0 * C ** 0.5)).to(torch.bfloat16).float()
    b = torch.randn(K, generator=g)
    return x, w, b


@pytest.mark.parametrize("shape", [(2, 64, 64, 16, 32), (3, 50, 50, 32, 8), (1, 25, 37, 16, 136), (2, 11, 9, 48, 64),
                                   (1, 4, 4, 16, 8), (2, 100, 60, 64, 128)])
def test_forward_and_data_gradient_small(sf, shape):
    ops = sf.ops
    N, IH, IW, C, K = shape
    OH, OW = (IH - 2) // 2 + 1, (IW - 2) // 2 + 1
    x, w, b = _operands(N, IH, IW, C, K, seed=sum(shape))
    xd = x.to(DEV)
    wp = ops.conv4x4s2_weights(w.to(DEV), False)
    z = torch.full((N, OH, OW, K), float("nan"), device=DEV, dtype=torch.bfloat16)
    assert ops.conv4x4s2_ok(xd, K, K)
    ops.conv4x4s2(xd, wp, z, K, bias=b.to(DEV))
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), w, b, stride=2, padding=1)
    close(z.float().permute(0, 3, 1, 2), ref.numpy(), 6e-3)
    if K % 16:
        return       # the data gradient reduces over K: it needs whole 16-channel stages
    # data gradient, with and without the LeakyReLU mask operand
    g = torch.Generator().manual_seed(7)
    dz = torch.randn(N, OH, OW, K, generator=g).to(torch.bfloat16)
    wT = ops.conv4x4s2_weights(w.to(DEV), True)
    dref = F.conv_transpose2d(dz.float().permute(0, 3, 1, 2), w, stride=2, padding=1,
                              output_padding=(IH - ((OH - 1) * 2 + 2), IW - ((OW - 1) * 2 + 2)))
    assert dref.shape[-2:] == (IH, IW)
    for mode in (0, 2, 1):
        dx = torch.full((N, IH, IW, C), float("nan"), device=DEV, dtype=torch.bfloat16)
        ops.conv4x4s2(dz.to(DEV), wT, dx, C, dgrad=True, aux=xd if mode else None, aux_mode=mode)
        want = dref
        if mode:
            pos = (x.float() > 0).permute(0, 3, 1, 2)
            want = torch.where(pos, dref, dref * (0.2 if mode == 2 else 0.0))
        close(dx.float().permute(0, 3, 1, 2), want.numpy(), 6e-3)


@pytest.mark.parametrize("layer", [(200, 64, 128), (100, 128, 256), (50, 256, 512), (25, 64, 128)])
def test_full_size_layers_against_generic_path_and_f32(sf, layer):
    ops = sf.ops
    fn = importlib.import_module(sf.__name__ + ".functional")
    IH, C, K = layer
    N, OH = 32, (IH - 2) // 2 + 1
    torch.manual_seed(13)
    x = torch.randn(N, IH, IH, C, device=DEV).to(torch.bfloat16)
    w = (torch.randn(K, C, 4, 4, device=DEV) / (4.0 * C ** 0.5)).to(torch.bfloat16).float()
    b = torch.randn(K, device=DEV)
    wp = ops.conv4x4s2_weights(w, False)
    z = torch.empty(N, OH, OH, K, device=DEV, dtype=torch.bfloat16)
    ops.conv4x4s2(x, wp, z, K, bias=b)
    wg = ops.pack_conv_weight(w.contiguous(), torch.bfloat16, list(range(16)), False, K, C)
    zg = torch.empty_like(z)
    ops.conv_fprop(x, wg, zg, fn._s2_geom(N, IH, IH, C, OH, OH), K, bias=b)
    close(z.float(), zg.float().cpu().numpy(), 8e-3)       # two bf16 roundings of differently ordered f32 sums
    for i in (0, N - 1):
        f32 = F.conv2d(x[i:i + 1].float().permute(0, 3, 1, 2).cpu(), w.cpu(), b.cpu(), stride=2, padding=1)
        close(z[i:i + 1].float().permute(0, 3, 1, 2), f32.numpy(), 6e-3)
    dz = torch.randn(N, OH, OH, K, device=DEV).to(torch.bfloat16)
    wT = ops.conv4x4s2_weights(w, True)
    dx = torch.empty(N, IH, IH, C, device=DEV, dtype=torch.bfloat16)
    ops.conv4x4s2(dz, wT, dx, C, dgrad=True)
    for i in (0, N - 1):
        dref = F.conv_transpose2d(dz[i:i + 1].float().permute(0, 3, 1, 2).cpu(), w.cpu(), stride=2, padding=1,
                                  output_padding=IH - ((OH - 1) * 2 + 2))
        close(dx[i:i + 1].float().permute(0, 3, 1, 2), dref.numpy(), 6e-3)
    # linear in the batch: the call on all maps equals the calls on its halves, bit for bit
    h = N // 2
    za = torch.empty(h, OH, OH, K, device=DEV, dtype=torch.bfloat16)
    ops.conv4x4s2(x[h:].contiguous(), wp, za, K, bias=b)
    assert torch.equal(za, z[h:])
