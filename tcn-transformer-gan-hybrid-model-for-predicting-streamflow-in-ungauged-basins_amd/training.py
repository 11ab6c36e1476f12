"""Training driver: the counterpart of Trainer (src/train.py:29-494) for the HIP modules, one
process per GPU.

train_step reproduces the reference's batch-loop semantics exactly (generator loss = combined
loss + gan_weight * BCE(D(pred), 1) back-propagated THROUGH an unfrozen discriminator;
discriminator loss every third batch on top; both optimisers step on (batch_idx+1) % accum == 0,
the discriminator's only when batch_idx % 3 == 0; then both gradient sets are zeroed);
train_epoch / train / validate keep the reference's return values, log lines and checkpoint files.
Added to what the reference has: parameters and Adam moments live in flat buffers so that one
fused kernel launch updates a whole network, gradients are taken as the backward kernels produce
them (no per-parameter accumulate launches), the Inf/NaN checks and the GradScaler-style step skip
run on the device without host synchronisation, and data-parallel gradient averaging over RCCL
reduces the two 134 MB Linear weight gradients in place as soon as they exist and everything else
as one flat buffer per network.
"""
from __future__ import annotations

import logging
import os
import time
from datetime import datetime
from pathlib import Path
from typing import Dict, Iterable, List, Optional, Tuple

import torch
import torch.distributed as dist
import torch.nn as nn

from . import ops
from .modules import CombinedLoss, GANLoss, PatchDiscriminator, StreamflowPredictionModel

_ALIGN = 64  # floats: every parameter starts on a 256-byte boundary of the flat buffer


class FlatAdam:
    """Adam (torch.optim.Adam defaults: eps 1e-8, no weight decay, src/train.py:80-91) over flat
    f32 buffers: parameters and both moments of one network live in one buffer each (parameters
    are views into it), so a whole network is updated by one sf_adam_guarded launch.

    Gradients are not views: with .grad unset, autograd hands every parameter the tensor its
    backward kernel produced instead of launching an accumulate kernel per parameter per backward
    pass.  gather() then copies the small gradients into one flat buffer (a multi-tensor copy;
    that buffer is also what data-parallel ranks all-reduce in one call), while parameters of at
    least `big_numel` elements sit at the end of the layout and are updated straight from their
    own gradient tensors (and all-reduced in place).

    A parameter without a gradient is left untouched by step(), moments included, and its own
    step count does not advance, as in torch.optim.Adam (which counts steps per parameter; the
    one case not reproduced is a step that is both skipped and missed by the parameter).
    The step count lives on the device (state[0]) so that a step can be
    skipped by a device-side flag (`found_inf`) without a host synchronisation; state[1] counts
    the skipped steps.  Big parameters get a bf16 shadow copy written by the same kernel when
    `shadow_big` is set: the 262144-wide Linear layers read it instead of casting 134 MB per call."""

    def __init__(self, module: nn.Module, lr: float, betas=(0.9, 0.999), eps: float = 1e-8, big_numel: int = 1 << 22,
                 shadow_big: bool = False):
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
        self.state = torch.zeros(2, device=dev, dtype=torch.int32)
        self.ranges = []
        for p, o in zip(self.params, offs):
            n = p.numel()
            self.flat[o:o + n].copy_(p.data.reshape(-1))
            p.data = self.flat[o:o + n].view(p.shape)
            p.grad = None
            self.ranges.append((o, n))
        self.lr, self.betas, self.eps = lr, betas, eps
        self.shadow_big = bool(shadow_big) and dev.type == "cuda"
        self._spans: List[Tuple[int, int, int]] = []
        self._calls = 0                          # step() calls so far
        self._seen = [0] * len(self.params)      # step() calls in which the parameter had a gradient
        # views made once: the usual step (every parameter has a gradient) then costs the host one
        # multi-tensor copy and a handful of launches, no per-parameter slicing
        self._small_views = [self.gsmall[o:o + n].view(p.shape) for p, (o, n) in zip(self.small, self.ranges)]
        self._big_views = [(self.flat[o:o + n], self.m[o:o + n], self.v[o:o + n]) for (o, n) in self.ranges[len(self.small):]]
        self._uniform = True                     # every parameter has taken part in every step so far

    @property
    def t(self) -> int:
        """Optimiser steps taken so far (host-synchronising read of the device counter)."""
        return int(self.state[0].item())

    @property
    def skipped(self) -> int:
        return int(self.state[1].item())

    def gather(self):
        """Small gradients -> self.gsmall (zeros for a parameter that received none, so that the
        buffer a data-parallel all-reduce sends is defined everywhere); records the spans of
        parameters that did receive one for step()."""
        if self._uniform and all(p.grad is not None for p in self.small):
            if self.small:
                torch._foreach_copy_(self._small_views, [p.grad for p in self.small])
            self._spans = [(0, self.n_small, 0)] if self.n_small else []
            return
        dst, src, spans = [], [], []
        for i, (p, (o, n)) in enumerate(zip(self.small, self.ranges)):
            if p.grad is None:
                self.gsmall[o:o + n].zero_()
                continue
            dst.append(self.gsmall[o:o + n].view(p.shape))
            src.append(p.grad)
            end = min((o + n + _ALIGN - 1) // _ALIGN * _ALIGN, self.n_small)
            off = self._seen[i] - self._calls      # minus the steps this parameter sat out
            if spans and spans[-1][1] == o and spans[-1][2] == off:
                spans[-1] = (spans[-1][0], end, off)
            else:
                spans.append((o, end, off))
        if dst:
            torch._foreach_copy_(dst, src)
        self._spans = spans

    def step(self, grad_scale: float = 1.0, found_inf: Optional[torch.Tensor] = None,
             shards: Optional[Dict[int, Tuple[int, int, torch.Tensor]]] = None):
        """One Adam update from the gathered small gradients and the big parameters' own gradient
        tensors; skipped as a whole (step count included) when found_inf[0] != 0 on the device.
        shards: {index into self.big: (lo, hi, gradient of elements lo..hi)} -- only that range of
        the parameter, its moments and its bf16 shadow is updated (a data-parallel rank that owns
        one slice of the parameter; the caller exchanges the slices afterwards)."""
        b1, b2 = self.betas
        ops.adam_begin(self.state, found_inf)
        for (a, e, off) in self._spans:     # alignment gaps inside a span hold zero gradients and zero parameters
            ops.adam_guarded(self.flat[a:e], self.gsmall[a:e], self.m[a:e], self.v[a:e], self.lr, b1, b2, self.eps,
                             self.state, found_inf, grad_scale, step_offset=off)
        ns = len(self.small)
        shadowed = []
        for i, (p, (fl, m, v)) in enumerate(zip(self.big, self._big_views)):
            if p.grad is None:
                continue
            shadow = ops.weight_shadow(p) if self.shadow_big else None
            if shards is not None and i in shards:
                lo, hi, g = shards[i]
                ops.adam_guarded(fl[lo:hi], g, m[lo:hi], v[lo:hi], self.lr, b1, b2, self.eps, self.state, found_inf,
                                 grad_scale, shadow=None if shadow is None else shadow.view(-1)[lo:hi],
                                 step_offset=self._seen[ns + i] - self._calls)
            else:
                ops.adam_guarded(fl, p.grad.contiguous().view(-1), m, v, self.lr, b1, b2, self.eps, self.state, found_inf,
                                 grad_scale, shadow=shadow, step_offset=self._seen[ns + i] - self._calls)
            if shadow is not None:
                shadowed.append(p)
        self._calls += 1
        for i, p in enumerate(self.params):
            if p.grad is not None:
                self._seen[i] += 1
            else:
                self._uniform = False
        ops.touch(p for p in self.params if p.grad is not None)   # only this network's packed weights go stale
        for p in shadowed:                  # written by the same kernel as the parameter: current again
            ops.mark_shadow_current(p)

    def grads(self) -> List[torch.Tensor]:
        """The tensors an Inf/NaN check of this step's gradients has to look at (after gather())."""
        return [self.gsmall] + [p.grad.contiguous().view(-1) for p in self.big if p.grad is not None]

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
        self.state[0] = int(state["t"])
        self._calls = int(state["t"])
        self._seen = [self._calls] * len(self.params)
        self._uniform = True

    def range_of(self, p: torch.Tensor):
        for q, r in zip(self.params, self.ranges):
            if q is p:
                return r
        raise KeyError("parameter not managed by this optimiser")


# flag slots of Trainer.nan_flags, one per check of src/train.py:193-199,221-222 (+ the gradients)
_FLAG_IMAGES, _FLAG_PRED, _FLAG_LOSS, _FLAG_GRADS = 0, 1, 2, 3
_FLAG_WARNINGS = {_FLAG_IMAGES: "NaN or Inf found in input images!",
                  _FLAG_PRED: "NaN or Inf found in model predictions!",
                  _FLAG_LOSS: "NaN or Inf found in loss!"}


class Trainer:
    def __init__(self, config: Dict, device: Optional[str] = None, distributed: Optional[bool] = None,
                 model_path: Optional[str] = None):
        self.config = config
        if device is None:
            device = config.get("train", {}).get("device", "cuda")
        self.device = torch.device(device)
        mcfg, scfg = config.get("model", {}), config.get("system", {})
        self.logger = logging.getLogger("Trainer")
        self.model = StreamflowPredictionModel(config).to(self.device)
        if model_path is not None and os.path.exists(model_path):   # src/train.py:49-51: a bare generator state_dict
            self.logger.info(f"Loading model from {model_path}")
            self.model.load_state_dict(_host_load(model_path))
        self.loss_fn = CombinedLoss(config)
        lr = config.get("train", {}).get("lr", 1e-4)
        self.use_gan = bool(mcfg.get("use_gan", False))
        self.gan_weight = mcfg.get("gan_weight", 0.1)
        self.use_amp = bool(scfg.get("mixed_precision", True))
        self.accum = self.grad_accum_steps = int(scfg.get("grad_accum_steps", 1))
        self.d_update_interval = 3  # src/train.py:101
        self.log_interval = int(scfg.get("log_interval", 10))
        big = int(scfg.get("big_grad_numel", 1 << 22))  # gradients this large skip the flat gradient buffer
        self.opt_g = FlatAdam(self.model, lr, big_numel=big, shadow_big=self.use_amp)
        if self.use_gan:
            self.discriminator = PatchDiscriminator(1, 64).to(self.device)
            self.gan_loss_fn = GANLoss(lambda_gan=self.gan_weight, lambda_pixel=1.0)
            self.opt_d = FlatAdam(self.discriminator, lr * 0.2, betas=(0.5, 0.999), big_numel=big)
        self.epoch = 0
        self.best_loss = float("inf")
        self.batch_idx = 0
        self.epoch_batches: Optional[int] = None   # batches per epoch when driven by train_epoch
        self._d_req = True
        self.distributed = dist.is_initialized() and dist.get_world_size() > 1 if distributed is None else distributed
        self.world = dist.get_world_size() if self.distributed else 1
        self.nan_flags = torch.zeros(4, device=self.device, dtype=torch.int32)
        self._one = torch.ones(1, device=self.device)      # the 'real' label of the adversarial losses
        self.found_inf = torch.zeros(2, device=self.device, dtype=torch.int32)    # generator / discriminator gradients
        self._skips_reported = (0, 0)      # skipped generator / discriminator steps already told by nan_report
        self._pending: List = []
        self._reduce_now = False
        self.comm_events: Optional[List] = None     # when a list: (before, after) event pairs around the waits
        self.comm_host_s = 0.0
        self._device_collectives = False
        if self.distributed:
            # RCCL works on device tensors and is ordered against streams; any other backend (gloo on
            # hosts without a device transport) gets the gradients staged through host memory
            self._device_collectives = dist.get_backend() == "nccl"
            self._comm_stream = torch.cuda.Stream(device=self.device) if self._device_collectives else None
            # communication kernels will hold CUs while the backward runs: the weight-gradient kernel
            # deals its tiles through a queue too (the 3x3 / 4x4 forward and data-gradient kernels
            # always do; for the weight gradient it costs the bit-reproducible summation order)
            ops.set_wgrad_scheduling(bool(scfg.get("dp_dynamic_wgrad", True)))
            self._broadcast_initial_state()
            self._install_early_reduce()
        # system.dp_shard_big: every rank owns 1 / world of each big parameter.  The gradient is
        # reduce-scattered instead of all-reduced, Adam updates the rank's slice (parameter, moments,
        # bf16 shadow) and the shadow slices are all-gathered: 1.5 x the parameter's bytes on the links
        # instead of 2 x, and 1 / world of the optimiser pass.  The f32 parameter and the moments
        # outside the own slice go stale until consolidate() (a collective) gathers them, which
        # state_dict / save_* do.  Needs mixed precision (the forward pass reads the shadow) and sizes
        # that divide by the world size; off by default.
        self.shard_big = False
        if self.distributed and bool(scfg.get("dp_shard_big", False)):
            ok = self.use_amp and self.opt_g.shadow_big and self.opt_g.big and all(p.numel() % self.world == 0 for p in self.opt_g.big)
            if not ok:
                self.logger.warning("dp_shard_big ignored: needs mixed precision and big parameters divisible by the world size")
            else:
                self.shard_big = True
                self._rank = dist.get_rank()
                self._shard_grads = [torch.empty(p.numel() // self.world, device=self.device, dtype=torch.float32)
                                     for p in self.opt_g.big]
                self._stale_master = False

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
                if self._device_collectives:
                    dist.broadcast(flat, 0)     # stream ordered: the copies below queue behind it
                else:
                    host = flat.cpu()
                    dist.broadcast(host, 0)
                    flat.copy_(host)
                off = 0
                for t in group:
                    t.copy_(flat[off:off + t.numel()].view_as(t))
                    off += t.numel()
        ops.bump_weight_epoch()

    def _all_reduce(self, t: torch.Tensor):
        """Sum `t` over the ranks, in place.  RCCL: asynchronous on the communication stream, which
        first waits for everything queued so far on the stream that produced `t` (the stream that
        is current here: inside an autograd hook that is the stream of the backward kernel whose
        output `t` is); _wait_reduces later makes the compute stream wait for the result.  Other
        backends: staged through host memory and finished before returning (the host copy
        synchronises with the producing stream), so no device tensor ever crosses the backend."""
        if self._device_collectives:
            ready = torch.cuda.Event()
            ready.record(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(self._comm_stream):
                self._comm_stream.wait_event(ready)
                work = dist.all_reduce(t, async_op=True)
            # (no record_stream: the communication stream only carries the wait, the process group
            # keeps `t` alive until its own stream is done, and _wait_reduces orders every later use)
            self._pending.append(work)
        else:
            t0 = time.perf_counter() if self.comm_events is not None else 0.0
            host = t.detach().to("cpu")
            dist.all_reduce(host)
            t.copy_(host, non_blocking=False)
            if self.comm_events is not None:       # staged: the whole exchange is exposed, and the host pays it
                self.comm_host_s += time.perf_counter() - t0

    def _install_early_reduce(self):
        """The two 262144-wide Linear weights are 95 % of the gradient bytes; their gradients are
        complete long before the encoder's backward finishes, so each is all-reduced in place
        from a post-accumulate hook and overlaps with the remaining backward kernels.  Hooks run
        in the order of the backward graph, which is the same on every rank."""
        for i, p in enumerate(self.opt_g.big):
            def hook(param, i=i):
                if self._reduce_now:
                    if self.shard_big:
                        self._reduce_scatter(param.grad, self._shard_grads[i])
                    else:
                        self._all_reduce(param.grad)

            p.register_post_accumulate_grad_hook(hook)

    def _shard_range(self, p: torch.Tensor) -> Tuple[int, int]:
        n = p.numel() // self.world
        return self._rank * n, (self._rank + 1) * n

    def _reduce_scatter(self, g: torch.Tensor, out: torch.Tensor):
        """out = this rank's slice of the sum of `g` over the ranks (ordering as in _all_reduce)."""
        flat = g.contiguous().view(-1)
        if self._device_collectives:
            ready = torch.cuda.Event()
            ready.record(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(self._comm_stream):
                self._comm_stream.wait_event(ready)
                work = dist.reduce_scatter_tensor(out, flat, async_op=True)
            self._pending.append(work)
        else:
            t0 = time.perf_counter() if self.comm_events is not None else 0.0
            host = flat.detach().to("cpu")
            dist.all_reduce(host)           # host transports: reduce everything, keep the slice
            n = out.numel()
            out.copy_(host[self._rank * n:(self._rank + 1) * n], non_blocking=False)
            if self.comm_events is not None:
                self.comm_host_s += time.perf_counter() - t0

    def _all_gather_slices(self, full: torch.Tensor):
        """Every rank's slice of the flat tensor `full` (its own slice is current) to every rank, in place."""
        n = full.numel() // self.world
        mine = full[self._rank * n:(self._rank + 1) * n]
        if self._device_collectives:
            ready = torch.cuda.Event()
            ready.record(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(self._comm_stream):
                self._comm_stream.wait_event(ready)
                work = dist.all_gather_into_tensor(full, mine, async_op=True)
            self._pending.append(work)
        else:
            t0 = time.perf_counter() if self.comm_events is not None else 0.0
            host = mine.detach().to("cpu")
            host = host.view(torch.uint8) if host.dtype == torch.bfloat16 else host     # (gloo has no bf16: as bytes)
            parts = [torch.empty_like(host) for _ in range(self.world)]
            dist.all_gather(parts, host)
            got = torch.cat(parts)
            full.copy_(got.view(torch.bfloat16) if full.dtype == torch.bfloat16 else got, non_blocking=False)
            if self.comm_events is not None:
                self.comm_host_s += time.perf_counter() - t0

    def consolidate(self):
        """Sharded big parameters: gathers the f32 parameter and both Adam moments of every rank's
        slice, so that each rank holds the whole state again.  A collective: every rank must call it
        (state_dict and the save_* methods do); a no-op otherwise."""
        if not self.shard_big or not self._stale_master:
            return
        self._wait_reduces()
        for (fl, m, v) in self.opt_g._big_views:
            for t in (fl, m, v):
                self._all_gather_slices(t)
        self._wait_reduces()
        self._stale_master = False

    def _wait_reduces(self):
        """The compute stream waits for the collectives issued so far (the host does not).  With
        `comm_events` set to a list, an event pair brackets the wait on the compute stream: the
        time between the two is what the exchange costs the step beyond what the backward hid
        (the stream was idle, waiting for the communication stream)."""
        if not self._pending:
            return
        timed = self.comm_events is not None and self._device_collectives
        if timed:
            before, after = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            before.record()
        for w in self._pending:
            w.wait()
        if timed:
            after.record()
            self.comm_events.append((before, after))
        self._pending = []

    def comm_exposed_ms(self) -> float:
        """Milliseconds the steps so far spent waiting for gradient exchange that the backward did
        not hide, since `comm_events` was last set to an empty list (host-synchronising).  RCCL:
        from the event pairs of _wait_reduces; staged transports: the host time of the exchange."""
        if self.comm_events is None:
            return 0.0
        torch.cuda.synchronize(self.device)
        return sum(a.elapsed_time(b) for a, b in self.comm_events) + self.comm_host_s * 1e3

    def _d_grads_consumed(self, bi: int) -> bool:
        """Discriminator gradients produced in batch `bi` are used by an optimiser step only if
        the next stepping batch (the first bs >= bi with (bs+1) % accum == 0, where both gradient
        sets are zeroed) is also a discriminator batch; otherwise they are zeroed unread, so the
        step skips computing them.  When the epoch ends before that batch the gradients carry over
        into the next epoch, whose batch index restarts at 0 (src/train.py:176): the consuming step
        is then batch accum - 1 of that epoch.  Parameters, losses and BatchNorm buffers are unaffected."""
        if not self.use_gan:
            return False
        bs = bi + (self.accum - 1 - bi % self.accum)
        if self.epoch_batches is not None and bs >= self.epoch_batches:
            if self.epoch_batches < self.accum:
                return True      # no stepping batch in sight: keep what the reference would keep
            bs = self.accum - 1
        return bs % self.d_update_interval == 0

    def _set_d_requires_grad(self, flag: bool):
        if self._d_req != flag:
            for p in self.discriminator.parameters():
                p.requires_grad_(flag)
            self._d_req = flag

    # ---- one batch ------------------------------------------------------------------------
    def train_step(self, images, output_image, station_positions, station_runoffs) -> Dict[str, torch.Tensor]:
        """One iteration of the reference's batch loop (src/train.py:176-268).  Returns loss
        tensors (no host sync).  The three Inf/NaN checks of src/train.py:193-199,221-222 set device
        flags (read them with nan_report()); with mixed precision an optimiser step whose
        gradients hold Inf/NaN is skipped on the device, as GradScaler.step does (src/train.py:255-260)."""
        bi = self.batch_idx
        self.model.train()
        if self.shard_big:
            self._wait_reduces()        # the shadow slices gathered after the previous step
        stepping = (bi + 1) % self.accum == 0
        d_batch = self.use_gan and bi % self.d_update_interval == 0
        self._reduce_now = self.distributed and stepping
        d_live = self._d_grads_consumed(bi)
        # The reference's generator backward also leaves gradients in the discriminator's parameters
        # (D(pred) is part of loss_G's graph, src/train.py:212-228), which its optimizer_D then applies
        # together with those of loss_D.  The discriminator's backward is linear in the gradient it is
        # fed, and D(pred.detach()) below replays the very same activations, so on a discriminator batch
        # that contribution is added to loss_D's backward as gan_weight * BCE(D(pred.detach()), 1)
        # instead: one pass of weight-gradient kernels over the fake branch, not two.
        merge_d = d_batch and d_live
        if self.use_gan:
            self._set_d_requires_grad(d_live and not merge_d)
        images = images.to(self.device, non_blocking=True)
        output_image = output_image.to(self.device, non_blocking=True)
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=self.use_amp):
            # the input check of src/train.py:193 rides on the encoder's layout pass
            pred = self.model(images, nan_flag=self.nan_flags[_FLAG_IMAGES:_FLAG_IMAGES + 1])
            # the prediction check of src/train.py:198 rides on the loss's forward pass
            ld = self.loss_fn(pred, output_image, station_positions, station_runoffs,
                              nan_flag=self.nan_flags[_FLAG_PRED:_FLAG_PRED + 1])
            loss_g = ld["total"]
            if self.use_gan:
                pred_fake = self.discriminator(pred, keep_record=d_batch)
                fake_record = self.discriminator.last_record
                g_gan = self.gan_loss_fn.gan_loss(pred_fake, self._one)
                loss_g = loss_g + self.gan_weight * g_gan
                ld["gan_g"] = g_gan
                ld["total"] = loss_g
            ops.nan_inf_flag(loss_g.detach().reshape(1), self.nan_flags[_FLAG_LOSS:_FLAG_LOSS + 1])
        (loss_g / self.accum if self.accum != 1 else loss_g).backward()      # (x / 1 is x: two launches less)
        # The generator's update needs nothing from the discriminator pass below (which works on
        # pred.detach() and never reads a generator parameter), so it is issued first: its large
        # kernels keep the GPU busy while the host works through the many small launches of the
        # discriminator pass.  Data parallel, the big gradients are already on their way (hooks) and
        # the rest of a network travels as one buffer; every rank issues the same collectives in the
        # same order for a given batch index.
        scale = 1.0 / self.world if self.distributed else 1.0
        if stepping:
            self.opt_g.gather()
            if self.distributed:
                self._all_reduce(self.opt_g.gsmall)
                self._wait_reduces()
            self._step(self.opt_g, scale, 0)
        if d_batch:
            self._set_d_requires_grad(d_live)
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=self.use_amp), torch.set_grad_enabled(d_live):
                pred_real = self.discriminator(output_image)
                # D(pred.detach()) repeats D(pred) of the generator loss above (same input, same
                # parameters: no optimiser step in between): its activations are reused
                pred_fake_d = self.discriminator(pred.detach(), reuse=fake_record)
                loss_d = self.gan_loss_fn.discriminator_loss(pred_real, pred_fake_d)
                ld["gan_d"] = loss_d
                back_d = loss_d
                if merge_d:
                    back_d = loss_d + self.gan_weight * self.gan_loss_fn.gan_loss(pred_fake_d, self._one)
            if d_live:
                (back_d / self.accum if self.accum != 1 else back_d).backward()
        if stepping:
            if d_batch:
                self.opt_d.gather()
                if self.distributed:
                    self._all_reduce(self.opt_d.gsmall)
                    for p in self.opt_d.big:
                        if p.grad is not None:
                            self._all_reduce(p.grad)
                    self._wait_reduces()
                self._step(self.opt_d, scale, 1)
            if self.use_gan:
                self.opt_d.zero_grad()
        if self.use_gan:
            self.discriminator.last_record = None
        self.batch_idx += 1
        return ld

    def _step(self, opt: FlatAdam, scale: float, which: int):
        """Adam step of one network from its gathered (and, data-parallel, exchanged) gradients.
        With mixed precision the step is skipped on the device when these gradients hold Inf/NaN:
        GradScaler.step decides per optimiser (src/train.py:255-260); the check runs after the
        exchange, so every rank takes the same decision."""
        found = None
        sharded = self.shard_big and opt is self.opt_g
        if self.use_amp:
            found = self.found_inf[which:which + 1]
            found.zero_()
            for g in ([opt.gsmall] + self._shard_grads if sharded else opt.grads()):
                ops.nan_inf_flag(g, found)
            if sharded:     # a rank sees its own slices only: the decision is the OR over the ranks
                if self._device_collectives:
                    dist.all_reduce(found, op=dist.ReduceOp.MAX)
                else:
                    host = found.cpu()
                    dist.all_reduce(host, op=dist.ReduceOp.MAX)
                    found.copy_(host)
        if sharded:
            shards = {i: self._shard_range(p) + (self._shard_grads[i],) for i, p in enumerate(opt.big) if p.grad is not None}
            opt.step(scale, found, shards=shards)
            for i, p in enumerate(opt.big):
                if i in shards:
                    self._all_gather_slices(ops.weight_shadow(p).view(-1))
            self._stale_master = True
        else:
            opt.step(scale, found)
        opt.zero_grad()

    # ---- epochs ---------------------------------------------------------------------------
    def train_epoch(self, dataloader) -> Dict[str, float]:
        """Trainer.train_epoch (src/train.py:154-305): one pass over `dataloader` (batches are
        dicts with images, output_image, station_positions, stations), the batch index restarting
        at 0; returns {'loss', 'image_loss', 'station_loss'} (means over the batches) and logs the
        reference's running-mean line every log_interval batches.  The per-batch losses are summed
        on the device; the host reads them only for a log line and at the end of the epoch."""
        self.model.train()
        self.batch_idx = 0
        try:
            self.epoch_batches = len(dataloader)
        except TypeError:
            self.epoch_batches = None
        sums = torch.zeros(3, device=self.device, dtype=torch.float64)
        num_batches = 0
        for batch_idx, batch in enumerate(dataloader):
            ld = self.train_step(batch["images"], batch["output_image"], batch["station_positions"], batch["stations"])
            sums += torch.stack([ld["total"].detach().double().reshape(()), ld["image"].double().reshape(()),
                                 ld["station"].double().reshape(())])
            num_batches += 1
            if (batch_idx + 1) % self.log_interval == 0:
                s = (sums / num_batches).tolist()
                msg = f"Epoch {self.epoch} Batch {batch_idx + 1}: Loss={s[0]:.6f}, Img={s[1]:.6f}, Stn={s[2]:.6f}"
                if self.use_gan and "gan_d" in ld:
                    msg += f", D_Loss={float(ld['gan_d'].detach()):.6f}"
                self.logger.info(msg)
                self._warn_nan()
        self.epoch_batches = None
        self._warn_nan()
        if num_batches == 0:
            raise ZeroDivisionError("train_epoch: the data loader produced no batch")
        s = (sums / num_batches).tolist()
        return {"loss": s[0], "image_loss": s[1], "station_loss": s[2]}

    def _warn_nan(self):
        """The reference's three warnings (src/train.py:193-199,221-222), emitted from the device
        flags at the points where the host reads losses anyway."""
        rep = self.nan_report()
        for slot, text in _FLAG_WARNINGS.items():
            if rep["flags"][slot]:
                self.logger.warning(text)

    def train(self, train_dataset, num_epochs: int = 100, val_dataset=None) -> None:
        """Trainer.train (src/train.py:360-494): epochs of train_epoch (+ validate), the
        reference's checkpoint files: best_model.pt on a validation improvement,
        checkpoint_epoch_{e}.pt every 10th epoch, gan/gan_epoch_{e}.pt every epoch with GAN; all of
        them the generator's bare state_dict."""
        batch_size = self.config.get("train", {}).get("batch_size", 4)
        output_dir = Path(self.config.get("data", {}).get("output_dir", "./output"))
        output_dir.mkdir(parents=True, exist_ok=True)
        self._attach_file_log(output_dir)
        rank = dist.get_rank() if self.distributed else 0
        # map-style datasets: worker processes, page-locked batches (src/train.py:374-376, default 4 workers)
        nw = int(self.config.get("train", {}).get("num_workers", 4))
        pin = self.device.type == "cuda"
        self._train_loaders, self._val_loaders = {}, {}
        for epoch in range(num_epochs):
            self.epoch = epoch
            train_metrics = self.train_epoch(_epoch_loader(train_dataset, batch_size, True, epoch, rank, self.world, nw, pin,
                                                           self._train_loaders))
            self.logger.info(f"Epoch {epoch} - Train Loss: {train_metrics['loss']:.6f}")
            self.consolidate()      # sliced big parameters: every rank, before rank 0 alone decides what to save
            if val_dataset is not None:
                val_metrics = self.validate(_epoch_loader(val_dataset, batch_size, False, epoch, 0, 1, nw, pin, self._val_loaders))
                self.logger.info(f"Epoch {epoch} - Val Loss: {val_metrics['loss']:.6f}, "
                                 f"RMSE: {val_metrics['RMSE']:.4f}, NSE: {val_metrics['NSE']:.4f}")
                if val_metrics["loss"] < self.best_loss:
                    self.best_loss = val_metrics["loss"]
                    if rank == 0:
                        self.save_model(str(output_dir / "best_model.pt"))
                        self.logger.info(f"Model saved to {output_dir / 'best_model.pt'}")
            if rank == 0 and (epoch + 1) % 10 == 0:
                self.save_model(str(output_dir / f"checkpoint_epoch_{epoch}.pt"))
                self.logger.info(f"Checkpoint saved to {output_dir / f'checkpoint_epoch_{epoch}.pt'}")
            if rank == 0 and self.use_gan:
                gan_dir = output_dir / "gan"
                gan_dir.mkdir(parents=True, exist_ok=True)
                self.save_model(str(gan_dir / f"gan_epoch_{epoch}.pt"))
                self.logger.info(f"GAN Checkpoint saved to {gan_dir / f'gan_epoch_{epoch}.pt'}")
        self.logger.info("Training complete!")

    def _attach_file_log(self, log_dir: Path):
        """training_<timestamp>.log in the output directory (src/train.py:113-131), one handler per
        process however many Trainers are built."""
        if any(getattr(h, "_sf_train_log", False) for h in self.logger.handlers):
            return
        self.logger.setLevel(logging.DEBUG if self.config.get("system", {}).get("debug", False) else logging.INFO)
        h = logging.FileHandler(log_dir / f"training_{datetime.now().strftime('%Y%m%d_%H%M%S')}.log")
        h.setFormatter(logging.Formatter("%(asctime)s - %(name)s - %(levelname)s - %(message)s"))
        h._sf_train_log = True
        self.logger.addHandler(h)

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

    def nan_report(self) -> Dict:
        """Host-synchronising read (and reset) of the device flags: 'flags' = [input images,
        predictions, loss, gradients] saw Inf/NaN since the last report (src/train.py:193,198,221;
        the fourth is what GradScaler checks: an optimiser step of either network was skipped
        since the last report), 'skipped_steps' = generator steps skipped so far,
        'skipped_steps_d' = discriminator steps skipped so far (both cumulative)."""
        vals = self.nan_flags.tolist()
        self.nan_flags.zero_()
        skipped_g = self.opt_g.skipped
        skipped_d = self.opt_d.skipped if self.use_gan else 0
        seen_g, seen_d = self._skips_reported
        if skipped_g > seen_g or skipped_d > seen_d:
            vals[_FLAG_GRADS] = 1
        self._skips_reported = (skipped_g, skipped_d)
        return {"flags": [bool(v) for v in vals], "skipped_steps": skipped_g, "skipped_steps_d": skipped_d}

    def saw_nan_or_inf(self) -> bool:
        """True when any of the checks fired since the last read (host-synchronising)."""
        return any(self.nan_report()["flags"])

    def state_dict(self):
        self.consolidate()
        return self.model.state_dict()

    # ---- checkpoints ----------------------------------------------------------------------
    def save_model(self, path: str):
        """The reference's checkpoint (src/train.py:472-490): the generator's bare state_dict,
        f32, reference keys.  Tensors are cloned to the host first: parameters are views into one
        flat buffer, and torch.save of a view would write the whole buffer once per parameter."""
        self.consolidate()
        torch.save(_host_state(self.model), path)

    def save_checkpoint(self, path: str):
        """Resume file, a superset of the reference's: generator and discriminator state_dicts,
        both Adam states (step, moments per parameter name), the batch counter that drives the
        discriminator / accumulation schedule and the CPU generator state dropout seeds come from.
        Plain tensors, ints and strings only (loads with weights_only=True)."""
        self.consolidate()
        ck = {"format": RESUME_FORMAT, "model": _host_state(self.model), "opt_g": self.opt_g.state_by_name(self.model),
              "batch_idx": int(self.batch_idx), "epoch": int(self.epoch), "cpu_rng_state": torch.get_rng_state()}
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
        self.epoch = int(ck.get("epoch", 0))
        torch.set_rng_state(ck["cpu_rng_state"])
        self.opt_g.zero_grad()
        if self.use_gan:
            self.opt_d.zero_grad()
        return "resume"


RESUME_FORMAT = "sfhip-resume-1"


def _collate(batch):
    """The collate function of src/train.py:379-390."""
    return {"images": torch.stack([b["images"] for b in batch], dim=0),
            "output_image": torch.stack([b["output_image"] for b in batch], dim=0),
            "stations": [b["stations"] for b in batch],
            "station_positions": [b["station_positions"] for b in batch]}


class _ListLoader:
    """A sized, re-iterable view of a batch generator (train_epoch asks for len())."""

    def __init__(self, make, n):
        self._make, self._n = make, n

    def __len__(self):
        return self._n

    def __iter__(self):
        return iter(self._make())


def _epoch_loader(dataset, batch_size: int, shuffle: bool, epoch: int, rank: int, world: int, num_workers: int = 0,
                  pin_memory: bool = False, cache: Optional[Dict] = None):
    """Batches of one epoch.  A dataset with a loader() method (ResidentStreamflowDataset) serves
    them from HBM, sharded over the ranks; any other map-style dataset goes through
    torch.utils.data.DataLoader as in src/train.py:374-401: the reference's collate function,
    `num_workers` worker processes that stay alive between epochs (the loader is kept in `cache`),
    two batches prefetched per worker, page-locked batches when the trainer sits on a GPU."""
    if hasattr(dataset, "loader"):
        n = len(dataset) // world
        nb = (n + batch_size - 1) // batch_size
        return _ListLoader(lambda: dataset.loader(batch_size, shuffle=shuffle, seed=0, epoch=epoch, rank=rank, world=world), nb)
    from torch.utils.data import DataLoader
    key = (id(dataset), batch_size, shuffle, rank, world, num_workers, pin_memory)
    if cache is not None and key in cache:
        loader, sampler = cache[key]
    else:
        sampler = None
        if world > 1:
            from torch.utils.data.distributed import DistributedSampler
            sampler = DistributedSampler(dataset, num_replicas=world, rank=rank, shuffle=shuffle, seed=0)
        loader = DataLoader(dataset, batch_size=batch_size, shuffle=shuffle and sampler is None, sampler=sampler,
                            num_workers=num_workers, pin_memory=pin_memory, collate_fn=_collate,
                            persistent_workers=num_workers > 0, prefetch_factor=2 if num_workers > 0 else None)
        if cache is not None:
            cache.clear()           # one dataset pair at a time: workers of loaders no longer used are released
            cache[key] = (loader, sampler)
    if sampler is not None:
        sampler.set_epoch(epoch)
    return loader


def _host_state(module: nn.Module) -> Dict[str, torch.Tensor]:
    return {k: v.detach().cpu().clone() for k, v in module.state_dict().items()}


def _host_load(path: str):
    """Checkpoints are read without unpickling code (tensors and plain containers only)."""
    return torch.load(path, map_location="cpu", weights_only=True)
