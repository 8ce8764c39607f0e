"""Oracle: DDPM / DDIM noise schedulers, numpy fp32 (index math in int64).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Restates the public diffusers
>=0.15 `DDPMScheduler` (the one the reference instantiates: app.ipynb:545,
train_diffute_v1.py:628) and `DDIMScheduler` (named by BASELINE.json's north_star)
for the SD2-inpainting scheduler config:
beta_start=0.00085, beta_end=0.012, scaled_linear, N=1000, epsilon prediction,
clip_sample=False, steps_offset=1, set_alpha_to_one=False.

Call sites mirrored: set_timesteps/timesteps app.ipynb:803-804; scale_model_input
:810; step(...).prev_sample :816; init_noise_sigma :800; add_noise
train_diffute_v1.py:897; get_velocity :907.
"""
import numpy as np

f32 = np.float32


def linspace_f32(start, end, steps):
    """torch.linspace(start, end, steps, dtype=float32) CPU semantics (Appendix A.3):
    step=(end-start)/(steps-1); first half start+step*i, second half end-step*(steps-1-i)."""
    start = f32(start); end = f32(end)
    step = f32((end - start) / f32(steps - 1))
    i = np.arange(steps)
    lo = (start + step * i.astype(f32)).astype(f32)
    hi = (end - step * (steps - 1 - i).astype(f32)).astype(f32)
    return np.where(i < steps // 2, lo, hi).astype(f32)


def make_tables(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012):
    """betas = linspace(sqrt(b0), sqrt(b1), N)**2 ; alphas_cumprod = cumprod(1-betas).
    torch's CPU cumprod accumulates float32 inputs in double and rounds each prefix."""
    b = linspace_f32(f32(beta_start) ** f32(0.5), f32(beta_end) ** f32(0.5), num_train_timesteps)
    betas = (b * b).astype(f32)
    alphas = (f32(1.0) - betas).astype(f32)
    acc = np.float64(1.0)
    ac = np.empty(num_train_timesteps, dtype=f32)
    for i in range(num_train_timesteps):
        acc = acc * np.float64(alphas[i])
        ac[i] = f32(acc)
    return betas, alphas, ac


def timesteps_ddpm(n, N=1000):
    """DDPMScheduler.set_timesteps: (arange(n) * (N//n))[::-1], int64."""
    return (np.arange(n, dtype=np.int64) * (N // n))[::-1].copy()


def timesteps_ddim(n, N=1000, steps_offset=1):
    """DDIMScheduler.set_timesteps ('leading'): same grid + steps_offset."""
    return timesteps_ddpm(n, N) + np.int64(steps_offset)


def prev_timestep(t, n, N=1000):
    return int(t) - N // n


def ddim_coefs(ac, t, n, N=1000, eta=0.0):
    """Scalar coefficients of DDIMScheduler.step: (sqrt(1-abar_t), sqrt(abar_t), sqrt(abar_prev), dir, std).
    diffusers writes `x ** 0.5` on 0-d fp32 tensors; pow(x, 0.5) and sqrt(x) differ by <= 1 ulp between
    libm implementations, so coefficients are compared at 1 ulp and the elementwise update bit-exactly."""
    p = prev_timestep(t, n, N)
    a_t = f32(ac[t]); a_p = f32(ac[p]) if p >= 0 else f32(ac[0])
    b_t = f32(1.0) - a_t
    var = (f32(1.0) - a_p) / (f32(1.0) - a_t) * (f32(1.0) - a_t / a_p)
    std = f32(f32(eta) * np.sqrt(f32(var)))
    return (f32(np.sqrt(b_t)), f32(np.sqrt(a_t)), f32(np.sqrt(a_p)), f32(np.sqrt(f32(f32(1.0) - a_p - std * std))), std)


def ddim_apply(c, eps, x, noise=None, prediction_type="epsilon"):
    """Elementwise part of DDIMScheduler.step in diffusers' op order, fp32."""
    sbt, sat, sap, dirc, std = [f32(v) for v in c]
    eps = eps.astype(f32); x = x.astype(f32)
    if prediction_type == "epsilon":
        x0 = (x - sbt * eps) / sat
        pe = eps
    elif prediction_type == "v_prediction":
        x0 = sat * x - sbt * eps
        pe = sat * eps + sbt * x
    else:
        raise ValueError(prediction_type)
    prev = sap * x0 + dirc * pe
    if noise is not None:
        prev = prev + std * noise.astype(f32)
    return prev.astype(f32)


def ddim_step(ac, eps, t, x, n, N=1000, eta=0.0, noise=None, prediction_type="epsilon"):
    """DDIMScheduler.step(...).prev_sample ; all arithmetic fp32, op order as diffusers."""
    return ddim_apply(ddim_coefs(ac, t, n, N, eta), eps, x, noise if eta > 0 else None, prediction_type)


def ddpm_coefs(ac, t, n, N=1000):
    """(sqrt(1-abar_t), sqrt(abar_t), coef_x0, coef_xt, sigma) of DDPMScheduler.step, variance_type fixed_small,
    strided prev_t (the >=0.15 formulation; SURVEY.md Appendix A.3 version caveat)."""
    p = prev_timestep(t, n, N)
    a_t = f32(ac[t]); a_p = f32(ac[p]) if p >= 0 else f32(1.0)
    b_t = f32(1.0) - a_t; b_p = f32(1.0) - a_p
    cur_a = f32(a_t / a_p); cur_b = f32(1.0) - cur_a
    c0 = f32(f32(np.sqrt(a_p) * cur_b) / b_t)
    c1 = f32(f32(np.sqrt(cur_a) * b_p) / b_t)
    sigma = f32(0.0)
    if t > 0:
        var = f32(f32(b_p / b_t) * cur_b)
        var = max(var, f32(1e-20))
        sigma = f32(np.sqrt(f32(var)))
    return (f32(np.sqrt(b_t)), f32(np.sqrt(a_t)), c0, c1, sigma)


def ddpm_apply(c, eps, x, noise=None, prediction_type="epsilon"):
    sbt, sat, c0, c1, sigma = [f32(v) for v in c]
    eps = eps.astype(f32); x = x.astype(f32)
    if prediction_type == "epsilon":
        x0 = (x - sbt * eps) / sat
    elif prediction_type == "v_prediction":
        x0 = sat * x - sbt * eps
    else:
        raise ValueError(prediction_type)
    prev = c0 * x0 + c1 * x
    if noise is not None:
        prev = prev + sigma * noise.astype(f32)
    return prev.astype(f32)


def ddpm_step(ac, eps, t, x, n, N=1000, noise=None, prediction_type="epsilon"):
    """DDPMScheduler.step(...).prev_sample; noise is added only when t > 0."""
    return ddpm_apply(ddpm_coefs(ac, t, n, N), eps, x, noise if t > 0 else None, prediction_type)


def add_noise(ac, x0, noise, t):
    """sqrt(abar_t)*x0 + sqrt(1-abar_t)*noise, per-sample t (train_diffute_v1.py:897)."""
    t = np.asarray(t, dtype=np.int64).reshape(-1)
    sa = np.sqrt(ac[t]).astype(f32).reshape(-1, *([1] * (x0.ndim - 1)))
    sb = np.sqrt((f32(1.0) - ac[t]).astype(f32)).astype(f32).reshape(-1, *([1] * (x0.ndim - 1)))
    return (sa * x0.astype(f32) + sb * noise.astype(f32)).astype(f32)


def get_velocity(ac, x0, noise, t):
    """sqrt(abar_t)*noise - sqrt(1-abar_t)*x0 (train_diffute_v1.py:907)."""
    t = np.asarray(t, dtype=np.int64).reshape(-1)
    sa = np.sqrt(ac[t]).astype(f32).reshape(-1, *([1] * (x0.ndim - 1)))
    sb = np.sqrt((f32(1.0) - ac[t]).astype(f32)).astype(f32).reshape(-1, *([1] * (x0.ndim - 1)))
    return (sa * noise.astype(f32) - sb * x0.astype(f32)).astype(f32)
