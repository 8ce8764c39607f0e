"""Fused AdamW + global-norm gradient clipping on the packed fp32 arenas of the HIP UNet.

Reference: `torch.optim.AdamW(unet.parameters(), lr, betas, weight_decay, eps)` (train_diffute_v1.py:721-727),
`accelerator.clip_grad_norm_(unet.parameters(), args.max_grad_norm)` + `optimizer.step()` (:927-930).  Same arithmetic
(torch's single-tensor AdamW formulas, fp32), but one pass over the gradient arena the HIP backward already filled:
no per-parameter gradient export, no 686-tensor optimizer loop, no re-pack of the weights (the kernel writes the bf16 /
fp32 compute copies in place).  The torch Parameters are refreshed from the master arena on demand (`sync_to_model`,
done automatically by `state_dict()` / `save_pretrained()`).

`ema_decay` adds the reference's `--use_ema` shadow copy (`EMAModel(ema_unet.parameters(), ...)` + `ema_unet.step(...)` after
every optimizer step, train_diffute_v1.py:642-646,934-935) as a fourth fp32 arena updated inside the same kernel pass;
the decay schedule is diffusers' `EMAModel.get_decay`."""
import ctypes

import torch

from . import _cabi


class FusedAdamW(torch.optim.Optimizer):
    """Drop-in for `torch.optim.AdamW(unet.parameters(), ...)` in the reference's loop (train_diffute_v1.py:721-727,
    :925-933): a torch Optimizer (so `get_scheduler(..., optimizer=optimizer)` / any LRScheduler can drive
    `param_groups[0]["lr"]`, :745-750), `step()`, `zero_grad()`, `state_dict()` / `load_state_dict()`.  Gradient
    accumulation (`accelerator.accumulate(unet)`, :873) works: every backward before the next `step()` / `zero_grad()`
    ADDS to the gradient arena."""

    def __init__(self, unet, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, max_grad_norm=1.0,
                 ema_decay=None, ema_min_decay=0.0, ema_update_after_step=0, ema_use_warmup=False, ema_inv_gamma=1.0, ema_power=2.0 / 3.0):
        unet._ensure_packed()
        super().__init__(unet._param_list(), dict(lr=float(lr), betas=tuple(betas), eps=float(eps), weight_decay=float(weight_decay)))
        self.unet = unet
        self.ema_decay, self.ema_min_decay, self.ema_update_after_step = ema_decay, float(ema_min_decay), int(ema_update_after_step)
        self.ema_use_warmup, self.ema_inv_gamma, self.ema_power = bool(ema_use_warmup), float(ema_inv_gamma), float(ema_power)
        self.max_grad_norm = float(max_grad_norm or 0.0)
        self.t = 0
        lib = unet._lib                        # the build the model computes in writes its 16-bit weights (bf16 / fp16)
        dev = unet.device
        n = lib.dmx_unet_grad_bytes(unet._h) // 4
        self.masters = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=dev)
        self._import_masters()
        self.ema = self.masters.clone() if ema_decay is not None else None      # shadow parameters start as a copy of the model
        st = _cabi.current_stream()
        self.nchunks = lib.dmx_unet_optim_chunks(unet._h)
        self.table = torch.empty(lib.dmx_unet_optim_table_bytes(unet._h), dtype=torch.uint8, device=dev)
        _cabi.check(lib.dmx_unet_optim_table(unet._h, _cabi.ptr(self.table), self.table.numel(), st), "optim_table")
        self.scalars = torch.zeros(3, dtype=torch.float32, device=dev)          # (|g| before clipping, clip coefficient [x 1 / loss scale], found_inf)
        self.found_inf = False                  # the last step(grad_scale=...) met an inf / NaN gradient and changed nothing
        self.ws = torch.empty(self.nchunks, dtype=torch.float32, device=dev)
        self.dirty = False
        self._pending = 0                       # backward passes since the last step() / zero_grad()
        self._acc = None                        # stash of the accumulated gradient while a further backward overwrites the arena
        unet._fused = self                      # the backward stops exporting per-parameter gradients; the arena is authoritative

    # hyper-parameters live in param_groups[0] like in any torch optimizer (LR schedulers write "lr" there)
    @property
    def lr(self):
        return float(self.param_groups[0]["lr"])

    @property
    def betas(self):
        return tuple(self.param_groups[0]["betas"])

    @property
    def eps(self):
        return float(self.param_groups[0]["eps"])

    @property
    def weight_decay(self):
        return float(self.param_groups[0]["weight_decay"])

    def _import_masters(self):
        u = self.unet
        lib = u._lib
        st = _cabi.current_stream()
        for k, p in zip(u._keys, u._param_list()):
            src = p.detach().to(torch.float32).contiguous()
            _cabi.check(lib.dmx_unet_master_import(u._h, _cabi.ptr(self.masters), k.encode(), _cabi.ptr(src), st), f"master_import({k})")

    def reimport_masters(self):
        """the torch Parameters were changed from outside (load_state_dict, broadcast_parameters): they replace the master
        copy; the Adam moments are kept (what torch.optim.AdamW does when parameters are overwritten in place).  Called by
        UNet2DConditionModel._ensure_packed, which has just re-packed the weights arena from the same Parameters."""
        self._import_masters()
        self.dirty = False

    @property
    def grad_norm(self):
        return self.scalars[0]

    def ema_decay_at(self, optimization_step):
        """diffusers EMAModel.get_decay (the value used by the `optimization_step`-th call of EMAModel.step)"""
        step = max(0, optimization_step - self.ema_update_after_step - 1)
        if step <= 0:
            return 0.0
        cur = 1.0 - (1.0 + step / self.ema_inv_gamma) ** -self.ema_power if self.ema_use_warmup else (1.0 + step) / (10.0 + step)
        return max(min(cur, float(self.ema_decay)), self.ema_min_decay)

    # ---- gradient accumulation: the HIP backward WRITES the arena, so a backward that follows another one without a
    # step() in between first stashes the arena and adds the stash back afterwards (UNet2DConditionModel._train_backward)
    def before_backward(self, grads):
        if self._pending > 0:
            if self._acc is None:
                self._acc = torch.empty_like(grads)
            self._acc.copy_(grads)

    def after_backward(self, grads):
        if self._pending > 0:
            grads.add_(self._acc)
        self._pending += 1

    def zero_grad(self, set_to_none=True):
        """drops the accumulated gradient: the next backward starts from zero"""
        self._pending = 0
        sync = getattr(self.unet, "_sync", None)
        if sync is not None:
            sync["acc"].exchanged()                 # ... and so does an open no_sync() / accumulate_steps window

    @torch.no_grad()
    def step(self, closure=None, grad_scale=None):
        """grad_scale: the loss scale S the backward ran under (fp16 mixed precision; diffute_amd.GradScaler passes it): the arena
        holds g * S, the step unscales inside its pass.  A gradient with an inf / NaN skips the step as a whole (`found_inf`), like
        `GradScaler.step` - one host read of a device flag per step, the same synchronisation torch's scaler makes."""
        if closure is not None:
            raise NotImplementedError("FusedAdamW.step: closures are not supported")
        if self._pending == 0:
            raise RuntimeError("FusedAdamW.step() without a backward pass since the last step() / zero_grad()")
        u = self.unet
        lib = u._lib
        tb = u._tb
        self.t += 1
        st = _cabi.current_stream()
        head = (u._h, _cabi.ptr(self.table), self.nchunks, _cabi.ptr(self.masters), _cabi.ptr(self.exp_avg),
                _cabi.ptr(self.exp_avg_sq), _cabi.ptr(tb["grads"]), self.lr, self.betas[0], self.betas[1], self.eps,
                self.weight_decay, self.t, self.max_grad_norm, _cabi.ptr(self.scalars),
                _cabi.ptr(self.ws), self.ws.numel() * 4,
                _cabi.ptr(self.ema) if self.ema is not None else None,
                self.ema_decay_at(self.t) if self.ema is not None else 0.0)
        self.found_inf = False
        if grad_scale is None:
            _cabi.check(lib.dmx_unet_adamw_step(*head, st), "adamw_step")
        else:
            _cabi.check(lib.dmx_unet_adamw_step_scaled(*head, 1.0 / float(grad_scale), st), "adamw_step_scaled")
            self.found_inf = bool(self.scalars[2].item() != 0.0)
            if self.found_inf:                                       # nothing was written: the step does not count
                self.t -= 1
                self._pending = 0
                return
        _cabi.check(lib.dmx_unet_refresh_derived(u._h, st), "refresh_derived")
        u._arena_version = getattr(u, "_arena_version", 0) + 1      # transposed weights are refreshed by the next training forward
        for sl in u._slots.values():
            sl["ctx_key"] = None                                     # cached context K/V were projected with the old weights
        self.dirty = True
        self._pending = 0

    # ---- checkpointing (accelerator.save_state / load_state, train_diffute_v1.py:664-690,955): packed arenas + hyper-parameters
    def state_dict(self):
        st = dict(step=self.t, masters=self.masters.clone(), exp_avg=self.exp_avg.clone(), exp_avg_sq=self.exp_avg_sq.clone())
        if self.ema is not None:
            st["ema"] = self.ema.clone()
        groups = [{k: v for k, v in g.items() if k != "params"} for g in self.param_groups]
        return dict(state=st, param_groups=groups, layout="diffute_amd packed fp32 arenas (dmx_unet_grad_range)")

    def load_state_dict(self, sd):
        st = sd["state"]
        if st["masters"].numel() != self.masters.numel():
            raise ValueError("FusedAdamW.load_state_dict: arena size mismatch (different UNet config)")
        self.t = int(st["step"])
        self.masters.copy_(st["masters"]); self.exp_avg.copy_(st["exp_avg"]); self.exp_avg_sq.copy_(st["exp_avg_sq"])
        if self.ema is not None:
            if "ema" not in st:
                raise ValueError("FusedAdamW.load_state_dict: this optimizer keeps an EMA shadow (ema_decay=...) but the checkpoint "
                                 "has no 'ema' entry; load it into an optimizer built without ema_decay, or re-create the shadow")
            self.ema.copy_(st["ema"])
        for g, src in zip(self.param_groups, sd["param_groups"]):
            g.update({k: v for k, v in src.items() if k != "params"})
        # the loaded masters become the weights: Parameters first, then the packed arena + derived copies from them
        self.dirty = True
        self.sync_to_model()
        self.unet._packed_sig = None
        fused, self.unet._fused = self.unet._fused, None        # re-pack without re-importing the masters we just loaded
        try:
            self.unet._ensure_packed()
        finally:
            self.unet._fused = fused
        self.unet._arena_version = getattr(self.unet, "_arena_version", 0) + 1   # transposed weights follow at the next training forward
        self._pending = 0

    def ema_state_dict(self):
        """the EMA shadow parameters as fp32 tensors in torch layouts (what `ema_unet.save_pretrained` would store)"""
        if self.ema is None:
            raise RuntimeError("FusedAdamW was built without ema_decay")
        u = self.unet
        lib = u._lib
        st = _cabi.current_stream()
        out = {}
        for k, p in zip(u._keys, u._param_list()):
            dst = torch.empty(p.shape, dtype=torch.float32, device=p.device)
            _cabi.check(lib.dmx_unet_grad_export(u._h, _cabi.ptr(self.ema), k.encode(), _cabi.ptr(dst), st), "ema_export")
            out[k] = dst
        return out

    def sync_to_model(self):
        """master arena -> the torch Parameters (fp32, torch layouts)"""
        if not self.dirty:
            return
        u = self.unet
        lib = u._lib
        st = _cabi.current_stream()
        with torch.no_grad():
            for k, p in zip(u._keys, u._param_list()):
                dst = p.data if (p.dtype == torch.float32 and p.is_contiguous()) else torch.empty(p.shape, dtype=torch.float32, device=p.device)
                _cabi.check(lib.dmx_unet_grad_export(u._h, _cabi.ptr(self.masters), k.encode(), _cabi.ptr(dst), st), "master_export")
                if dst is not p.data:
                    p.data.copy_(dst)
        self.dirty = False


class GradScaler:
    """Dynamic loss scaling with `torch.cuda.amp.GradScaler`'s contract and defaults - what `Accelerator(mixed_precision="fp16")`
    (train_diffute_v1.py:267,583) puts around the reference's `accelerator.backward(loss)` / `clip_grad_norm_` / `optimizer.step()`
    (:925-930): the fp16 build stores activation gradients in fp16, so the backward runs on loss * S, the optimizer sees g / S, a step
    whose gradient overflowed is skipped and S halves; after `growth_interval` clean steps S doubles.

        scaler.scale(loss).backward(); scaler.unscale_(opt); clip_grad_norm_(...); scaler.step(opt); scaler.update()

    Works with FusedAdamW (unscale, inf check, clipping and update are one HIP pass over the arena: `unscale_` is then a marker)
    and with any torch optimizer over `unet.parameters()` (the exported `.grad` tensors are unscaled and checked with torch ops)."""

    def __init__(self, init_scale=2.0 ** 16, growth_factor=2.0, backoff_factor=0.5, growth_interval=2000, enabled=True):
        if growth_factor <= 1.0 or not 0.0 < backoff_factor < 1.0:
            raise ValueError("GradScaler: growth_factor must be > 1 and backoff_factor in (0, 1)")
        self._scale, self._growth_factor, self._backoff_factor = float(init_scale), float(growth_factor), float(backoff_factor)
        self._growth_interval, self._enabled = int(growth_interval), bool(enabled)
        self._growth_tracker = 0
        self._found = None                      # found_inf of the step since the last update(): None = no step yet
        self._unscaled = set()                  # ids of the optimizers whose gradients were unscaled since the last update()

    def is_enabled(self):
        return self._enabled

    def get_scale(self):
        return self._scale if self._enabled else 1.0

    def scale(self, loss):
        return loss * self._scale if self._enabled else loss

    def unscale_(self, optimizer):
        if not self._enabled:
            return
        if id(optimizer) in self._unscaled:
            raise RuntimeError("unscale_() has already been called on this optimizer since the last update().")
        self._unscaled.add(id(optimizer))
        if hasattr(optimizer, "masters"):       # FusedAdamW: unscaled inside step()
            return
        grads = [p.grad for g in optimizer.param_groups for p in g["params"] if p.grad is not None]
        found = False
        if grads:
            dev = grads[0].device
            found_t = torch.zeros(1, dtype=torch.float32, device=dev)
            inv = torch.full((1,), 1.0 / self._scale, dtype=torch.float32, device=dev)
            torch._amp_foreach_non_finite_check_and_unscale_(grads, found_t, inv)
            found = bool(found_t.item() != 0.0)
        self._found = bool(self._found) or found

    def step(self, optimizer, *args, **kwargs):
        if not self._enabled:
            return optimizer.step(*args, **kwargs)
        if hasattr(optimizer, "masters"):
            self._unscaled.add(id(optimizer))
            ret = optimizer.step(*args, grad_scale=self._scale, **kwargs)
            self._found = bool(self._found) or optimizer.found_inf
            return ret
        if id(optimizer) not in self._unscaled:
            self.unscale_(optimizer)
        if self._found:
            return None                         # skipped (GradScaler.step)
        return optimizer.step(*args, **kwargs)

    def update(self, new_scale=None):
        if not self._enabled:
            return
        if new_scale is not None:
            self._scale = float(new_scale)
        else:
            if self._found is None:
                raise RuntimeError("No inf checks were recorded prior to update.")
            if self._found:
                self._scale *= self._backoff_factor
                self._growth_tracker = 0
            else:
                self._growth_tracker += 1
                if self._growth_tracker == self._growth_interval:
                    self._scale *= self._growth_factor
                    self._growth_tracker = 0
        self._found = None
        self._unscaled.clear()

    def state_dict(self):
        return dict(scale=self._scale, growth_factor=self._growth_factor, backoff_factor=self._backoff_factor,
                    growth_interval=self._growth_interval, _growth_tracker=self._growth_tracker) if self._enabled else {}

    def load_state_dict(self, sd):
        if not self._enabled:
            return
        self._scale = float(sd["scale"]); self._growth_factor = float(sd["growth_factor"]); self._backoff_factor = float(sd["backoff_factor"])
        self._growth_interval = int(sd["growth_interval"]); self._growth_tracker = int(sd["_growth_tracker"])
