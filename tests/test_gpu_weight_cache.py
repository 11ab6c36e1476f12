"""Packed / staged conv weights are cached on the parameter (ops.derived): the cache must follow every
way a parameter changes -- an in-place torch update (version counter), an optimiser kernel writing
behind torch's back (ops.touch, from FlatAdam.step) and load_state_dict -- and a write through `.data`, which
torch does not count, after ops.bump_weight_epoch()."""
import pytest
import torch

from conftest import load_pkg

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def sf():
    return load_pkg()


def _fresh(sf, make, x):
    """Output of a module built anew from the same state (no cache to go stale)."""
    def run(state):
        m = make()
        m.load_state_dict(state)
        m.train()
        with torch.autocast("cuda", dtype=torch.bfloat16), torch.no_grad():
            return m(x).float()
    return run


@pytest.mark.parametrize("which", ["discriminator", "encoder"])
def test_cached_weights_follow_the_parameter(sf, which):
    torch.manual_seed(3)
    if which == "discriminator":
        make = lambda: sf.PatchDiscriminator(1, 64).to(DEV)
        x = torch.randn(2, 1, 64, 64, device=DEV)
    else:
        make = lambda: sf.CNNEncoder(16, 64, 2).to(DEV)
        x = torch.randn(2, 16, 64, 64, device=DEV)
    m = make()
    m.train()
    fresh = _fresh(sf, make, x)

    def out():
        with torch.autocast("cuda", dtype=torch.bfloat16), torch.no_grad():
            return m(x).float()

    def state():
        return {k: v.clone() for k, v in m.state_dict().items()}

    convs = [p for n, p in m.named_parameters() if p.dim() == 4]
    s0 = state()
    y0 = out()
    assert torch.equal(out(), y0)                         # second call: served from the cache
    with torch.no_grad():                                 # in-place update through torch
        for p in convs:
            p.mul_(1.5)
    y1 = out()
    assert not torch.equal(y1, y0)
    assert torch.equal(y1, fresh(state()))
    for p in convs:                                       # a write torch does not see, then the bump
        p.data.mul_(0.5)
    sf.ops.bump_weight_epoch()
    y2 = out()
    assert not torch.equal(y2, y1)
    assert torch.equal(y2, fresh(state()))
    m.load_state_dict(s0)                                 # load_state_dict copies in place
    assert torch.equal(out(), fresh(s0))


def test_linear_weight_shadow_follows_the_parameter(sf):
    """The bf16 shadow of a big Linear weight (written by FlatAdam's kernel beside the f32 master)
    must go stale like every other derived copy: after a write through `.data` announced by
    ops.bump_weight_epoch(), and after ops.touch() by an optimiser that does not maintain it."""
    import importlib
    training = importlib.import_module(sf.__name__ + ".training")
    F = importlib.import_module(sf.__name__ + ".functional")
    ops = sf.ops
    torch.manual_seed(5)
    lin = torch.nn.Linear(4096, 64).to(DEV)
    x = torch.randn(8, 4096, device=DEV)

    def out():
        with torch.no_grad():
            return F.linear(x, lin, torch.bfloat16).float()

    def fresh():
        w16 = lin.weight.detach().to(torch.bfloat16).float()
        return (x.to(torch.bfloat16).float() @ w16.t() + lin.bias.detach()).float()

    def check():
        y, want = out(), fresh()
        assert float((y - want).abs().max()) <= 2e-3 * float(want.abs().max())

    opt = training.FlatAdam(lin, 1e-2, big_numel=1 << 16, shadow_big=True)
    assert [tuple(p.shape) for p in opt.big] == [(64, 4096)]
    for _ in range(2):                                    # the kernel writes parameter and shadow together
        lin.weight.grad = torch.randn_like(lin.weight)
        lin.bias.grad = torch.randn_like(lin.bias)
        opt.gather()
        opt.step()
        opt.zero_grad()
        assert ops.shadow_if_current(lin.weight) is not None
        check()
    lin.weight.data.mul_(-2.0)                            # torch does not count this write
    ops.bump_weight_epoch()
    assert ops.shadow_if_current(lin.weight) is None
    check()
    sh = ops.weight_shadow(lin.weight)                    # rebuilt by a cast: current again
    assert ops.shadow_if_current(lin.weight) is sh
    lin.weight.data.mul_(0.25)                            # an optimiser kernel without a shadow of its own
    ops.touch([lin.weight])
    assert ops.shadow_if_current(lin.weight) is None
    check()
