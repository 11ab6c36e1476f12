"""Training-step driver: the counterpart of Trainer.train_epoch (src/train.py:154-305) for the
HIP modules, one process per GPU.

It reproduces the reference's step semantics exactly (generator loss = combined loss +
gan_weight * BCE(D(pred), 1) back-propagated THROUGH an unfrozen discriminator; discriminator
loss every third batch on top; both optimisers step on (batch_idx+1) % accum == 0, the
discriminator's only when batch_idx % 3 == 0; then both gradient sets are zeroed) and adds
what the reference lacks: parameters / gradients / Adam moments live in flat buffers so that
one fused kernel launch updates a whole network, and data-parallel gradient averaging over
RCCL with the two 134 MB Linear weights reduced as soon as their gradients exist.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import torch
import torch.distributed as dist
import torch.nn as nn

from . import ops
from .modules import CombinedLoss, GANLoss, PatchDiscriminator, StreamflowPredictionModel

_ALIGN = 64  # floats: every parameter starts on a 256-byte boundary of the flat buffer


class FlatAdam:
    """All parameters of one network as views into one flat f32 buffer (same for gradients and
    Adam moments); step() is a single sf_adam launch (torch.optim.Adam defaults: eps 1e-8, no
    weight decay, src/train.py:80-91)."""

    def __init__(self, module: nn.Module, lr: float, betas=(0.9, 0.999), eps: float = 1e-8):
        self.params = [p for p in module.parameters() if p.requires_grad]
        offs, total = [], 0
        for p in self.params:
            offs.append(total)
            total += (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
        dev = self.params[0].device
        self.flat = torch.zeros(total, device=dev, dtype=torch.float32)
        self.grad = torch.zeros(total, device=dev, dtype=torch.float32)
        self.m = torch.zeros(total, device=dev, dtype=torch.float32)
        self.v = torch.zeros(total, device=dev, dtype=torch.float32)
        self.ranges = []
        for p, o in zip(self.params, offs):
            n = p.numel()
            self.flat[o:o + n].copy_(p.data.reshape(-1))
            p.data = self.flat[o:o + n].view(p.shape)
            p.grad = self.grad[o:o + n].view(p.shape)
            self.ranges.append((o, n))
        self.lr, self.betas, self.eps, self.t = lr, betas, eps, 0

    def step(self, grad_scale: float = 1.0):
        self.t += 1
        ops.adam(self.flat, self.grad, self.m, self.v, self.lr, self.betas[0], self.betas[1], self.eps, self.t,
                 grad_scale)

    def zero_grad(self):
        self.grad.zero_()

    def range_of(self, p: torch.Tensor):
        for q, r in zip(self.params, self.ranges):
            if q is p:
                return r
        raise KeyError("parameter not managed by this optimiser")


def reduce_ranges(flat_grad: torch.Tensor, skip: Sequence = ()) -> List:
    """Sum-all-reduce (async) every element of a flat gradient buffer that lies outside the
    `skip` ranges [(offset, numel), ...] -- those were already reduced from a hook -- in as few
    calls as the gaps allow.  Skipped ranges are rounded up to the buffer's alignment, like the
    parameter slots themselves.  Returns the work handles.  Device-agnostic (RCCL or gloo)."""
    total = flat_grad.numel()
    works = []
    pos = 0
    for (o, n) in sorted(skip) + [(total, 0)]:
        if o > pos:
            works.append(dist.all_reduce(flat_grad[pos:o], async_op=True))
        pos = max(pos, o + (n + _ALIGN - 1) // _ALIGN * _ALIGN)
    return works


class Trainer:
    def __init__(self, config: Dict, device: str = "cuda", distributed: Optional[bool] = None):
        self.config = config
        self.device = torch.device(device)
        mcfg, scfg = config.get("model", {}), config.get("system", {})
        self.model = StreamflowPredictionModel(config).to(self.device)
        self.loss_fn = CombinedLoss(config)
        lr = config.get("train", {}).get("lr", 1e-4)
        self.use_gan = bool(mcfg.get("use_gan", False))
        self.gan_weight = mcfg.get("gan_weight", 0.1)
        self.use_amp = bool(scfg.get("mixed_precision", True))
        self.accum = int(scfg.get("grad_accum_steps", 1))
        self.d_update_interval = 3  # src/train.py:101
        self.opt_g = FlatAdam(self.model, lr)
        if self.use_gan:
            self.discriminator = PatchDiscriminator(1, 64).to(self.device)
            self.gan_loss_fn = GANLoss(lambda_gan=self.gan_weight, lambda_pixel=1.0)
            self.opt_d = FlatAdam(self.discriminator, lr * 0.2, betas=(0.5, 0.999))
        self.batch_idx = 0
        self._d_req = True
        self.distributed = dist.is_initialized() and dist.get_world_size() > 1 if distributed is None else distributed
        self.world = dist.get_world_size() if self.distributed else 1
        self.nan_flag = torch.zeros(1, device=self.device, dtype=torch.int32)
        self._pending: List = []
        self._reduce_now = False
        if self.distributed:
            self._install_early_reduce()

    # ---- data parallel ------------------------------------------------------------------
    def _install_early_reduce(self):
        """The two 262144-wide Linear weights are 95 % of the gradient bytes; their gradients are
        complete long before the encoder's backward finishes, so their all-reduce is issued from
        a post-accumulate hook and overlaps with the remaining backward kernels."""
        big = [self.model.temporal_to_spatial.weight]
        ip = getattr(self.model.temporal_module, "input_proj", None)
        if ip is not None:
            big.append(ip.weight)
        self._big_ranges = []
        for p in big:
            o, n = self.opt_g.range_of(p)
            self._big_ranges.append((o, n))

            def hook(param, o=o, n=n):
                if self._reduce_now:
                    self._pending.append(dist.all_reduce(self.opt_g.grad[o:o + n], async_op=True))

            p.register_post_accumulate_grad_hook(hook)

    def _reduce_rest(self, opt: FlatAdam, skip: Sequence = ()):
        self._pending += reduce_ranges(opt.grad, skip)

    def _wait_reduces(self):
        for w in self._pending:
            w.wait()
        self._pending = []

    def _d_grads_consumed(self, bi: int) -> bool:
        """Discriminator gradients produced in batch `bi` are used by an optimiser step only if
        the next stepping batch (the first bs >= bi with (bs+1) % accum == 0, where both gradient
        sets are zeroed) is also a discriminator batch; otherwise they are zeroed unread, so the
        step skips computing them.  Parameters, losses and BatchNorm buffers are unaffected."""
        bs = bi + (self.accum - 1 - bi % self.accum)
        return self.use_gan and bs % self.d_update_interval == 0

    def _set_d_requires_grad(self, flag: bool):
        if self._d_req != flag:
            for p in self.discriminator.parameters():
                p.requires_grad_(flag)
            self._d_req = flag

    # ---- one batch ------------------------------------------------------------------------
    def train_step(self, images, output_image, station_positions, station_runoffs) -> Dict[str, torch.Tensor]:
        """One iteration of the reference's batch loop.  Returns loss tensors (no host sync)."""
        bi = self.batch_idx
        self.model.train()
        stepping = (bi + 1) % self.accum == 0
        d_batch = self.use_gan and bi % self.d_update_interval == 0
        self._reduce_now = self.distributed and stepping
        d_live = self._d_grads_consumed(bi)
        if self.use_gan:
            self._set_d_requires_grad(d_live)
        images = images.to(self.device, non_blocking=True)
        output_image = output_image.to(self.device, non_blocking=True)
        ops.nan_inf_flag(images.contiguous().float().view(-1), self.nan_flag)
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=self.use_amp):
            pred = self.model(images)
            ops.nan_inf_flag(pred.view(-1), self.nan_flag)
            ld = self.loss_fn(pred, output_image, station_positions, station_runoffs)
            loss_g = ld["total"]
            if self.use_gan:
                pred_fake = self.discriminator(pred)
                g_gan = self.gan_loss_fn.gan_loss(pred_fake, torch.ones(1, device=self.device))
                loss_g = loss_g + self.gan_weight * g_gan
                ld["gan_g"] = g_gan
                ld["total"] = loss_g
        (loss_g / self.accum).backward()
        if d_batch:
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=self.use_amp), torch.set_grad_enabled(d_live):
                pred_real = self.discriminator(output_image)
                pred_fake_d = self.discriminator(pred.detach())
                loss_d = self.gan_loss_fn.discriminator_loss(pred_real, pred_fake_d)
                ld["gan_d"] = loss_d
            if d_live:
                (loss_d / self.accum).backward()
        if stepping:
            scale = 1.0
            if self.distributed:
                self._reduce_rest(self.opt_g, self._big_ranges)
                if d_batch:
                    self._reduce_rest(self.opt_d)
                self._wait_reduces()
                scale = 1.0 / self.world
            self.opt_g.step(scale)
            if d_batch:
                self.opt_d.step(scale)
            self.opt_g.zero_grad()
            if self.use_gan:
                self.opt_d.zero_grad()
        self.batch_idx += 1
        return ld

    def saw_nan_or_inf(self) -> bool:
        """Host-synchronising read of the flag the three per-step checks of src/train.py:193,198,221 feed."""
        bad = bool(self.nan_flag.item())
        self.nan_flag.zero_()
        return bad

    def state_dict(self):
        return self.model.state_dict()
