"""DDPMScheduler / DDIMScheduler with the surface the reference uses:

  from_pretrained(path, subfolder="scheduler")   train_diffute_v1.py:628, app.ipynb:545
  .num_train_timesteps / .config.prediction_type  train_diffute_v1.py:892,904
  .add_noise / .get_velocity                      train_diffute_v1.py:897,907
  .init_noise_sigma / .set_timesteps / .timesteps app.ipynb:800,803-804
  .scale_model_input / .step(...).prev_sample     app.ipynb:810,816

Host logic (tables, integer timestep grids, the scalar coefficients of a step) is Python/torch,
written with the same tensor expressions as diffusers >=0.15 so it rounds identically on the same
machine; the elementwise update over the latents is a gfx950 kernel behind the C-ABI.
"""
import json
import os
from types import SimpleNamespace

import numpy as np
import torch

from . import _cabi

SD2_SCHEDULER_CONFIG = dict(
    num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
    prediction_type="epsilon", clip_sample=False, steps_offset=1, set_alpha_to_one=False,
    variance_type="fixed_small", timestep_spacing="leading")


class SchedulerOutput(SimpleNamespace):
    """`.prev_sample` (app.ipynb:816)."""


class _Config(SimpleNamespace):
    def __getitem__(self, k):
        return getattr(self, k)


class _SchedulerBase:
    order = 1

    def __init__(self, **config):
        cfg = dict(SD2_SCHEDULER_CONFIG); cfg.update(config)
        # options of the diffusers schedulers whose arithmetic is NOT implemented by the step kernels: refuse them instead of
        # silently computing something else (DDPMScheduler's own default is clip_sample=True; SD2's scheduler config sets False)
        if cfg.get("clip_sample"):
            raise NotImplementedError("clip_sample=True is not implemented (the SD2-inpainting scheduler config uses clip_sample=false)")
        if cfg.get("thresholding"):
            raise NotImplementedError("thresholding=True is not implemented")
        if cfg.get("variance_type", "fixed_small") != "fixed_small":
            raise NotImplementedError(f"variance_type={cfg['variance_type']!r} is not implemented (only 'fixed_small')")
        if cfg.get("timestep_spacing", "leading") != "leading":
            raise NotImplementedError(f"timestep_spacing={cfg['timestep_spacing']!r} is not implemented (only 'leading')")
        if cfg["prediction_type"] not in ("epsilon", "v_prediction"):
            raise NotImplementedError(f"prediction_type={cfg['prediction_type']!r} is not implemented")
        self.config = _Config(**cfg)
        N = cfg["num_train_timesteps"]
        self.num_train_timesteps = N                      # read directly at train_diffute_v1.py:892
        if cfg["beta_schedule"] == "scaled_linear":
            self.betas = torch.linspace(cfg["beta_start"] ** 0.5, cfg["beta_end"] ** 0.5, N, dtype=torch.float32) ** 2
        elif cfg["beta_schedule"] == "linear":
            self.betas = torch.linspace(cfg["beta_start"], cfg["beta_end"], N, dtype=torch.float32)
        else:
            raise NotImplementedError(f"{cfg['beta_schedule']} is not implemented for {self.__class__}")
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.one = torch.tensor(1.0)
        self.init_noise_sigma = 1.0                       # app.ipynb:800
        self.num_inference_steps = None
        self.timesteps = torch.from_numpy(np.arange(0, N)[::-1].copy().astype(np.int64))
        self._dev_tables = {}

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, subfolder=None, **kw):
        d = pretrained_model_name_or_path if subfolder is None else os.path.join(pretrained_model_name_or_path, subfolder)
        with open(os.path.join(d, "scheduler_config.json")) as f:
            cfg = json.load(f)
        known = set(SD2_SCHEDULER_CONFIG) | {"thresholding"}
        return cls(**{k: v for k, v in cfg.items() if k in known})

    def save_pretrained(self, save_directory):
        os.makedirs(save_directory, exist_ok=True)
        with open(os.path.join(save_directory, "scheduler_config.json"), "w") as f:
            json.dump(dict(vars(self.config), _class_name=type(self).__name__), f, indent=2)

    def __len__(self):
        return self.config.num_train_timesteps

    def scale_model_input(self, sample, timestep=None):
        """Identity for DDPM/DDIM (app.ipynb:810)."""
        return sample

    def _grid(self, num_inference_steps):
        N = self.config.num_train_timesteps
        if num_inference_steps > N:
            raise ValueError(f"`num_inference_steps`: {num_inference_steps} cannot be larger than {N}")
        step_ratio = N // num_inference_steps                              # integer index math (bit-exact)
        return (np.arange(0, num_inference_steps) * step_ratio).round()[::-1].copy().astype(np.int64)

    def previous_timestep(self, timestep):
        return int(timestep) - self.config.num_train_timesteps // self.num_inference_steps

    # ---- training-side helpers: per-sample coefficient gather on the host tables, update on the GPU
    def _coef_tables(self, device):
        key = str(device)
        if key not in self._dev_tables:
            sa = (self.alphas_cumprod ** 0.5).to(device)
            sb = ((1 - self.alphas_cumprod) ** 0.5).to(device)
            self._dev_tables[key] = (sa, sb)
        return self._dev_tables[key]

    def _mix(self, a, b, timesteps, velocity):
        _cabi.require_cuda(a, b)
        sa_t, sb_t = self._coef_tables(a.device)
        t = timesteps.to(a.device).reshape(-1).long()
        if t.numel() != a.shape[0]:
            raise ValueError("timesteps must have one entry per sample")
        sa = sa_t[t].contiguous(); sb = sb_t[t].contiguous()
        x = a.to(torch.float32).contiguous(); n = b.to(torch.float32).contiguous()
        out = torch.empty_like(x)
        fn = _cabi.lib().dmx_sched_get_velocity if velocity else _cabi.lib().dmx_sched_add_noise
        _cabi.check(fn(_cabi.ptr(x), _cabi.ptr(n), _cabi.ptr(sa), _cabi.ptr(sb), _cabi.ptr(out), x.shape[0],
                       x.numel() // x.shape[0], _cabi.current_stream()), "add_noise/get_velocity")
        return out.to(a.dtype)

    def add_noise(self, original_samples, noise, timesteps):
        """sqrt(abar_t) x0 + sqrt(1-abar_t) noise (train_diffute_v1.py:897)."""
        return self._mix(original_samples, noise, timesteps, False)

    def get_velocity(self, sample, noise, timesteps):
        """sqrt(abar_t) noise - sqrt(1-abar_t) sample (train_diffute_v1.py:907)."""
        return self._mix(sample, noise, timesteps, True)

    @staticmethod
    def _t_int(timestep):
        return int(timestep.item()) if torch.is_tensor(timestep) else int(timestep)


class DDPMScheduler(_SchedulerBase):
    """The scheduler the reference instantiates (app.ipynb:545, train_diffute_v1.py:628).

    `steps_offset`: implemented is the behaviour of the diffusers release the reference was written against (0.15-era,
    SURVEY Appendix A.3): DDPMScheduler.set_timesteps does NOT add `steps_offset` to the leading-spaced grid (the inference
    grid ends at timestep 0), while DDIMScheduler does.  Later diffusers releases apply the offset to DDPM as well.  The SD2
    `scheduler_config.json` carries `steps_offset: 1`; the key is kept in `.config` (round trip through save_pretrained) and a
    warning says once per process that the DDPM grid ignores it."""
    _warned_offset = False

    def __init__(self, **config):
        super().__init__(**config)
        if int(self.config.steps_offset) != 0 and not DDPMScheduler._warned_offset:
            DDPMScheduler._warned_offset = True
            import warnings
            warnings.warn(f"DDPMScheduler: steps_offset={self.config.steps_offset} is kept in the config but NOT applied to the "
                          "inference grid (the behaviour of the diffusers release the reference uses; later releases add it)", stacklevel=2)

    def set_timesteps(self, num_inference_steps, device=None):
        self.num_inference_steps = int(num_inference_steps)
        self.timesteps = torch.from_numpy(self._grid(self.num_inference_steps))
        if device is not None:
            self.timesteps = self.timesteps.to(device)

    def step_coefficients(self, timestep):
        """(sqrt_beta_prod_t, sqrt_alpha_prod_t, coef_x0, coef_xt, sigma) as python floats holding fp32 values."""
        t = self._t_int(timestep)
        if self.num_inference_steps is None:
            raise ValueError("Number of inference steps is 'None', you need to run 'set_timesteps' after creating the scheduler")
        prev_t = self.previous_timestep(t)
        alpha_prod_t = self.alphas_cumprod[t]
        alpha_prod_t_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.one
        beta_prod_t = 1 - alpha_prod_t
        beta_prod_t_prev = 1 - alpha_prod_t_prev
        current_alpha_t = alpha_prod_t / alpha_prod_t_prev
        current_beta_t = 1 - current_alpha_t
        c0 = (alpha_prod_t_prev ** (0.5) * current_beta_t) / beta_prod_t
        c1 = current_alpha_t ** (0.5) * beta_prod_t_prev / beta_prod_t
        sigma = torch.tensor(0.0)
        if t > 0:
            variance = (1 - alpha_prod_t_prev) / (1 - alpha_prod_t) * current_beta_t
            variance = torch.clamp(variance, min=1e-20)                 # fixed_small
            sigma = variance ** 0.5
        return (float(beta_prod_t ** 0.5), float(alpha_prod_t ** 0.5), float(c0), float(c1), float(sigma))

    def step(self, model_output, timestep, sample, generator=None, variance_noise=None, return_dict=True):
        _cabi.require_cuda(model_output, sample)
        t = self._t_int(timestep)
        sbt, sat, c0, c1, sigma = self.step_coefficients(t)
        x = sample.to(torch.float32).contiguous(); eps = model_output.to(torch.float32).contiguous()
        noise = None
        if t > 0:
            if variance_noise is None:      # the reference passes no generator: device RNG (app.ipynb:816)
                variance_noise = torch.randn(x.shape, generator=generator, device=x.device, dtype=torch.float32)
            noise = variance_noise.to(torch.float32).contiguous()
        out = torch.empty_like(x)
        _cabi.check(_cabi.lib().dmx_sched_step_ddpm(_cabi.ptr(x), _cabi.ptr(eps), _cabi.ptr(noise), _cabi.ptr(out), x.numel(),
                                                    sbt, sat, c0, c1, sigma, int(self.config.prediction_type == "v_prediction"),
                                                    _cabi.current_stream()), "sched_step_ddpm")
        out = out.to(sample.dtype)
        return SchedulerOutput(prev_sample=out) if return_dict else (out,)


class DDIMScheduler(_SchedulerBase):
    """Named by BASELINE.json's north_star (deterministic eta=0 sampler); same surface as DDPM."""

    def __init__(self, **config):
        super().__init__(**config)
        self.final_alpha_cumprod = torch.tensor(1.0) if self.config.set_alpha_to_one else self.alphas_cumprod[0]

    def set_timesteps(self, num_inference_steps, device=None):
        self.num_inference_steps = int(num_inference_steps)
        ts = self._grid(self.num_inference_steps) + np.int64(self.config.steps_offset)
        self.timesteps = torch.from_numpy(ts)
        if device is not None:
            self.timesteps = self.timesteps.to(device)

    def step_coefficients(self, timestep, eta=0.0):
        """(sqrt_beta_prod_t, sqrt_alpha_prod_t, sqrt_alpha_prod_prev, dir_coef, std_dev)."""
        t = self._t_int(timestep)
        if self.num_inference_steps is None:
            raise ValueError("Number of inference steps is 'None', you need to run 'set_timesteps' after creating the scheduler")
        prev_t = self.previous_timestep(t)
        alpha_prod_t = self.alphas_cumprod[t]
        alpha_prod_t_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
        beta_prod_t = 1 - alpha_prod_t
        beta_prod_t_prev = 1 - alpha_prod_t_prev
        variance = (beta_prod_t_prev / beta_prod_t) * (1 - alpha_prod_t / alpha_prod_t_prev)
        std_dev_t = eta * variance ** (0.5)
        dir_coef = (1 - alpha_prod_t_prev - std_dev_t ** 2) ** (0.5)
        return (float(beta_prod_t ** 0.5), float(alpha_prod_t ** 0.5), float(alpha_prod_t_prev ** 0.5),
                float(dir_coef), float(std_dev_t))

    def step(self, model_output, timestep, sample, eta=0.0, generator=None, variance_noise=None, return_dict=True):
        _cabi.require_cuda(model_output, sample)
        sbt, sat, sap, dirc, std = self.step_coefficients(timestep, eta)
        x = sample.to(torch.float32).contiguous(); eps = model_output.to(torch.float32).contiguous()
        noise = None
        if eta > 0:
            if variance_noise is None:
                variance_noise = torch.randn(x.shape, generator=generator, device=x.device, dtype=torch.float32)
            noise = variance_noise.to(torch.float32).contiguous()
        out = torch.empty_like(x)
        _cabi.check(_cabi.lib().dmx_sched_step_ddim(_cabi.ptr(x), _cabi.ptr(eps), _cabi.ptr(noise), _cabi.ptr(out), x.numel(),
                                                    sbt, sat, sap, dirc, std, int(self.config.prediction_type == "v_prediction"),
                                                    _cabi.current_stream()), "sched_step_ddim")
        out = out.to(sample.dtype)
        return SchedulerOutput(prev_sample=out) if return_dict else (out,)
