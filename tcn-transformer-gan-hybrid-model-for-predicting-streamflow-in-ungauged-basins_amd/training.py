"""Training-step driver: the counterpart of Trainer.train_epoch (src/train.py:154-305) for the
HIP modules, one process per GPU.

It reproduces the reference's step semantics exactly (generator loss = combined loss +
gan_weight * BCE(D(pred), 1) back-propagated THROUGH an unfrozen discriminator; discriminator
loss every third batch on top; both optimisers step on (batch_idx+1) % accum == 0, the
discriminator's only when batch_idx % 3 == 0; then both gradient sets are zeroed) and adds
what the reference lacks: parameters and Adam moments live in flat buffers so that one fused
kernel launch updates a whole network, gradients are taken as the backward kernels produce them
(no per-parameter accumulate launches), and data-parallel gradient averaging over RCCL reduces
the two 134 MB Linear weight gradients in place as soon as they exist and everything else as
one flat buffer.
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
    """Adam (torch.optim.Adam defaults: eps 1e-8, no weight decay, src/train.py:80-91) over flat
    f32 buffers: parameters and both moments of one network live in one buffer each (parameters
    are views into it), so a whole network is updated by one sf_adam launch.

    Gradients are not views: with .grad unset, autograd hands every parameter the tensor its
    backward kernel produced instead of launching an accumulate kernel per parameter per backward
    pass.  gather() then copies the small gradients into one flat buffer (a multi-tensor copy;
    that buffer is also what data-parallel ranks all-reduce in one call), while parameters of at
    least `big_numel` elements sit at the end of the layout and are updated straight from their
    own gradient tensors (and all-reduced in place)."""

    def __init__(self, module: nn.Module, lr: float, betas=(0.9, 0.999), eps: float = 1e-8, big_numel: int = 1 << 22):
        params = [p for p in module.parameters() if p.requires_grad]
        self.small = [p for p in params if p.numel() < big_numel]
        self.big = [p for p in params if p.numel() >= big_numel]
        self.params = self.small + self.big
        offs, total = [], 0
        for p in self.params:
            offs.append(total)
            total += (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
        self.n_small = offs[len(self.small)] if self.big else total
        dev = self.params[0].device
        self.flat = torch.zeros(total, device=dev, dtype=torch.float32)
        self.m = torch.zeros(total, device=dev, dtype=torch.float32)
        self.v = torch.zeros(total, device=dev, dtype=torch.float32)
        self.gsmall = torch.zeros(self.n_small, device=dev, dtype=torch.float32)
        self.ranges = []
        for p, o in zip(self.params, offs):
            n = p.numel()
            self.flat[o:o + n].copy_(p.data.reshape(-1))
            p.data = self.flat[o:o + n].view(p.shape)
            p.grad = None
            self.ranges.append((o, n))
        self.lr, self.betas, self.eps, self.t = lr, betas, eps, 0

    def gather(self):
        """Small gradients -> self.gsmall (zeros for a parameter that received none)."""
        dst, src = [], []
        for p, (o, n) in zip(self.small, self.ranges):
            if p.grad is None:
                self.gsmall[o:o + n].zero_()
            else:
                dst.append(self.gsmall[o:o + n].view(p.shape))
                src.append(p.grad)
        if dst:
            torch._foreach_copy_(dst, src)

    def step(self, grad_scale: float = 1.0):
        """One Adam update from the gathered small gradients and the big parameters' own gradient
        tensors (a big parameter without a gradient is skipped, like torch.optim.Adam does)."""
        self.t += 1
        b1, b2 = self.betas
        S = self.n_small
        if S:
            ops.adam(self.flat[:S], self.gsmall, self.m[:S], self.v[:S], self.lr, b1, b2, self.eps, self.t, grad_scale)
        for p, (o, n) in zip(self.big, self.ranges[len(self.small):]):
            if p.grad is not None:
                ops.adam(self.flat[o:o + n], p.grad.contiguous().view(-1), self.m[o:o + n], self.v[o:o + n], self.lr,
                         b1, b2, self.eps, self.t, grad_scale)

    def zero_grad(self):
        for p in self.params:
            p.grad = None

    def state_by_name(self, module: nn.Module) -> Dict:
        """Step count and both moments keyed by the module's parameter names (host copies)."""
        names = {id(p): n for n, p in module.named_parameters()}
        out = {"t": self.t, "exp_avg": {}, "exp_avg_sq": {}}
        for p, (o, n) in zip(self.params, self.ranges):
            out["exp_avg"][names[id(p)]] = self.m[o:o + n].view(p.shape).cpu().clone()
            out["exp_avg_sq"][names[id(p)]] = self.v[o:o + n].view(p.shape).cpu().clone()
        return out

    def load_state_by_name(self, module: nn.Module, state: Dict):
        names = {id(p): n for n, p in module.named_parameters()}
        for p, (o, n) in zip(self.params, self.ranges):
            k = names[id(p)]
            if k not in state["exp_avg"] or tuple(state["exp_avg"][k].shape) != tuple(p.shape):
                raise RuntimeError(f"optimiser state for {k} is missing or has the wrong shape")
            self.m[o:o + n].copy_(state["exp_avg"][k].reshape(-1))
            self.v[o:o + n].copy_(state["exp_avg_sq"][k].reshape(-1))
        self.t = int(state["t"])

    def range_of(self, p: torch.Tensor):
        for q, r in zip(self.params, self.ranges):
            if q is p:
                return r
        raise KeyError("parameter not managed by this optimiser")


class Trainer:
    def __init__(self, config: Dict, device: str = "cuda", distributed: Optional[bool] = None,
                 model_path: Optional[str] = None):
        self.config = config
        self.device = torch.device(device)
        mcfg, scfg = config.get("model", {}), config.get("system", {})
        self.model = StreamflowPredictionModel(config).to(self.device)
        if model_path is not None:      # Trainer(config, model_path), src/train.py:49-51: a bare generator state_dict
            self.model.load_state_dict(_host_load(model_path))
        self.loss_fn = CombinedLoss(config)
        lr = config.get("train", {}).get("lr", 1e-4)
        self.use_gan = bool(mcfg.get("use_gan", False))
        self.gan_weight = mcfg.get("gan_weight", 0.1)
        self.use_amp = bool(scfg.get("mixed_precision", True))
        self.accum = int(scfg.get("grad_accum_steps", 1))
        self.d_update_interval = 3  # src/train.py:101
        big = int(scfg.get("big_grad_numel", 1 << 22))  # gradients this large skip the flat gradient buffer
        self.opt_g = FlatAdam(self.model, lr, big_numel=big)
        if self.use_gan:
            self.discriminator = PatchDiscriminator(1, 64).to(self.device)
            self.gan_loss_fn = GANLoss(lambda_gan=self.gan_weight, lambda_pixel=1.0)
            self.opt_d = FlatAdam(self.discriminator, lr * 0.2, betas=(0.5, 0.999), big_numel=big)
        self.batch_idx = 0
        self._d_req = True
        self.distributed = dist.is_initialized() and dist.get_world_size() > 1 if distributed is None else distributed
        self.world = dist.get_world_size() if self.distributed else 1
        self.nan_flag = torch.zeros(1, device=self.device, dtype=torch.int32)
        self._pending: List = []
        self._reduce_now = False
        if self.distributed:
            self._broadcast_initial_state()
            self._install_early_reduce()

    # ---- data parallel ------------------------------------------------------------------
    def _broadcast_initial_state(self):
        """Every rank starts from rank 0's parameters and buffers (what DistributedDataParallel
        does at construction), so ranks that were seeded differently or loaded different files
        still hold one model."""
        nets = [self.model] + ([self.discriminator] if self.use_gan else [])
        by_dtype: Dict[torch.dtype, List[torch.Tensor]] = {}
        for net in nets:
            for t in list(net.parameters()) + list(net.buffers()):
                by_dtype.setdefault(t.dtype, []).append(t.data)
        with torch.no_grad():
            for group in by_dtype.values():     # one collective per dtype, not one per tensor
                flat = torch.cat([t.reshape(-1) for t in group])
                torch.cuda.synchronize(self.device)
                dist.broadcast(flat, 0)
                torch.cuda.synchronize(self.device)
                off = 0
                for t in group:
                    t.copy_(flat[off:off + t.numel()].view_as(t))
                    off += t.numel()
    def _install_early_reduce(self):
        """The two 262144-wide Linear weights are 95 % of the gradient bytes; their gradients are
        complete long before the encoder's backward finishes, so each is all-reduced in place
        from a post-accumulate hook and overlaps with the remaining backward kernels."""
        for p in self.opt_g.big:
            def hook(param):
                if self._reduce_now:
                    self._pending.append(dist.all_reduce(param.grad, async_op=True))

            p.register_post_accumulate_grad_hook(hook)

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
                fake_record = self.discriminator.last_record
                g_gan = self.gan_loss_fn.gan_loss(pred_fake, torch.ones(1, device=self.device))
                loss_g = loss_g + self.gan_weight * g_gan
                ld["gan_g"] = g_gan
                ld["total"] = loss_g
        (loss_g / self.accum).backward()
        if d_batch:
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=self.use_amp), torch.set_grad_enabled(d_live):
                pred_real = self.discriminator(output_image)
                # D(pred.detach()) repeats D(pred) of the generator loss above (same input, same
                # parameters: no optimiser step in between): its activations are reused
                pred_fake_d = self.discriminator(pred.detach(), reuse=fake_record)
                loss_d = self.gan_loss_fn.discriminator_loss(pred_real, pred_fake_d)
                ld["gan_d"] = loss_d
            if d_live:
                (loss_d / self.accum).backward()
        if stepping:
            scale = 1.0
            self.opt_g.gather()
            if d_batch:
                self.opt_d.gather()
            if self.distributed:
                # big gradients are already on their way (hooks); everything else is one buffer
                self._pending.append(dist.all_reduce(self.opt_g.gsmall, async_op=True))
                if d_batch:
                    self._pending += [dist.all_reduce(self.opt_d.gsmall, async_op=True)]
                    self._pending += [dist.all_reduce(p.grad, async_op=True) for p in self.opt_d.big if p.grad is not None]
                self._wait_reduces()
                scale = 1.0 / self.world
            self.opt_g.step(scale)
            if d_batch:
                self.opt_d.step(scale)
            self.opt_g.zero_grad()
            if self.use_gan:
                self.opt_d.zero_grad()
        if self.use_gan:
            self.discriminator.last_record = None
        self.batch_idx += 1
        return ld

    def validate(self, batches) -> Dict[str, float]:
        """Trainer.validate (src/train.py:307-358): eval-mode forward over an iterable of batches
        (dicts with images, output_image, station_positions, stations), mean combined loss and
        calculate_metrics of all predictions against the targets (resized to the prediction grid
        when the shapes differ).  The loss is accumulated on the device: one host sync at the end."""
        from .evaluation import calculate_metrics, resize_bilinear
        self.model.eval()
        total = torch.zeros((), device=self.device)
        preds, tgts, n = [], [], 0
        with torch.no_grad():
            for b in batches:
                images = b["images"].to(self.device, non_blocking=True)
                target = b["output_image"].to(self.device, non_blocking=True)
                pred = self.model(images)
                total += self.loss_fn(pred, target, b["station_positions"], b["stations"])["total"].float()
                n += 1
                pc = pred.float().cpu()
                tc = target.float().cpu()
                if tc.shape != pc.shape:
                    tc = resize_bilinear(tc, pc.shape[2], pc.shape[3], self.device)
                preds.append(pc)
                tgts.append(tc)
        metrics = calculate_metrics(torch.cat(preds), torch.cat(tgts))
        metrics["loss"] = float(total.item()) / max(n, 1)
        return metrics

    def saw_nan_or_inf(self) -> bool:
        """Host-synchronising read of the flag the three per-step checks of src/train.py:193,198,221 feed."""
        bad = bool(self.nan_flag.item())
        self.nan_flag.zero_()
        return bad

    def state_dict(self):
        return self.model.state_dict()

    # ---- checkpoints ----------------------------------------------------------------------
    def save_model(self, path: str):
        """The reference's checkpoint (src/train.py:472-490): the generator's bare state_dict,
        f32, reference keys.  Tensors are cloned to the host first: parameters are views into one
        flat buffer, and torch.save of a view would write the whole buffer once per parameter."""
        torch.save(_host_state(self.model), path)

    def save_checkpoint(self, path: str):
        """Resume file, a superset of the reference's: generator and discriminator state_dicts,
        both Adam states (step, moments per parameter name), the batch counter that drives the
        discriminator / accumulation schedule and the CPU generator state dropout seeds come from.
        Plain tensors, ints and strings only (loads with weights_only=True)."""
        ck = {"format": RESUME_FORMAT, "model": _host_state(self.model), "opt_g": self.opt_g.state_by_name(self.model),
              "batch_idx": int(self.batch_idx), "cpu_rng_state": torch.get_rng_state()}
        if self.use_gan:
            ck["discriminator"] = _host_state(self.discriminator)
            ck["opt_d"] = self.opt_d.state_by_name(self.discriminator)
        torch.save(ck, path)

    def load_checkpoint(self, path: str) -> str:
        """Loads either file kind; returns "resume" or "model" (a bare generator state_dict as the
        reference writes it: optimiser state, discriminator and counters keep their current values)."""
        ck = _host_load(path)
        if not (isinstance(ck, dict) and ck.get("format") == RESUME_FORMAT):
            self.model.load_state_dict(ck)
            return "model"
        self.model.load_state_dict(ck["model"])
        self.opt_g.load_state_by_name(self.model, ck["opt_g"])
        if self.use_gan:
            if "discriminator" not in ck:
                raise RuntimeError("checkpoint was written without a discriminator but use_gan is set")
            self.discriminator.load_state_dict(ck["discriminator"])
            self.opt_d.load_state_by_name(self.discriminator, ck["opt_d"])
        self.batch_idx = int(ck["batch_idx"])
        torch.set_rng_state(ck["cpu_rng_state"])
        self.opt_g.zero_grad()
        if self.use_gan:
            self.opt_d.zero_grad()
        return "resume"


RESUME_FORMAT = "sfhip-resume-1"


def _host_state(module: nn.Module) -> Dict[str, torch.Tensor]:
    return {k: v.detach().cpu().clone() for k, v in module.state_dict().items()}


def _host_load(path: str):
    """Checkpoints are read without unpickling code (tensors and plain containers only)."""
    return torch.load(path, map_location="cpu", weights_only=True)
