// Scheduler-step and latent glue kernels, fp32 elementwise.
// Compiled with -ffp-contract=off and written with explicit _rn ops in the op order of
// diffusers' DDIMScheduler.step / DDPMScheduler.step so results are bit-identical to the
// fp32 CPU evaluation of the same formulas (scalar coefficients are computed on the host).
#include "common.h"
#include "kernels.h"

// DDIM: x0 = (x - sqrt(1-abar_t)*eps)/sqrt(abar_t); prev = sqrt(abar_p)*x0 + dir*eps (+ std*noise)
__global__ __launch_bounds__(256) void dmx_sched_ddim_kernel(const float* x, const float* eps, const float* noise, float* out, size_t n,
                                                             float sqrt_bt, float sqrt_at, float sqrt_ap, float dir_coef, float std, int vpred) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float xv = x[i], ev = eps[i];
    float x0, pe;
    if (!vpred) {
      x0 = __fdiv_rn(__fsub_rn(xv, __fmul_rn(sqrt_bt, ev)), sqrt_at);
      pe = ev;
    } else {
      x0 = __fsub_rn(__fmul_rn(sqrt_at, xv), __fmul_rn(sqrt_bt, ev));
      pe = __fadd_rn(__fmul_rn(sqrt_at, ev), __fmul_rn(sqrt_bt, xv));
    }
    float prev = __fadd_rn(__fmul_rn(sqrt_ap, x0), __fmul_rn(dir_coef, pe));
    if (noise) prev = __fadd_rn(prev, __fmul_rn(std, noise[i]));
    out[i] = prev;
  }
}
int dmx_sched_ddim_launch(const float* x, const float* eps, const float* noise, float* out, size_t n,
                          float sqrt_bt, float sqrt_at, float sqrt_ap, float dir_coef, float std, int vpred, hipStream_t stream) {
  int blocks = (int)((n + 255) / 256); if (blocks > 2048) blocks = 2048; if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(dmx_sched_ddim_kernel, dim3(blocks), dim3(256), 0, stream, x, eps, noise, out, n, sqrt_bt, sqrt_at, sqrt_ap, dir_coef, std, vpred);
  return dmx_check_launch("dmx_sched_ddim_kernel");
}

// DDPM: prev = c0*x0 + c1*x (+ sigma*noise when t>0)
__global__ __launch_bounds__(256) void dmx_sched_ddpm_kernel(const float* x, const float* eps, const float* noise, float* out, size_t n,
                                                             float sqrt_bt, float sqrt_at, float c0, float c1, float sigma, int vpred) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float xv = x[i], ev = eps[i];
    float x0;
    if (!vpred) x0 = __fdiv_rn(__fsub_rn(xv, __fmul_rn(sqrt_bt, ev)), sqrt_at);
    else x0 = __fsub_rn(__fmul_rn(sqrt_at, xv), __fmul_rn(sqrt_bt, ev));
    float prev = __fadd_rn(__fmul_rn(c0, x0), __fmul_rn(c1, xv));
    if (noise) prev = __fadd_rn(prev, __fmul_rn(sigma, noise[i]));
    out[i] = prev;
  }
}
int dmx_sched_ddpm_launch(const float* x, const float* eps, const float* noise, float* out, size_t n,
                          float sqrt_bt, float sqrt_at, float c0, float c1, float sigma, int vpred, hipStream_t stream) {
  int blocks = (int)((n + 255) / 256); if (blocks > 2048) blocks = 2048; if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(dmx_sched_ddpm_kernel, dim3(blocks), dim3(256), 0, stream, x, eps, noise, out, n, sqrt_bt, sqrt_at, c0, c1, sigma, vpred);
  return dmx_check_launch("dmx_sched_ddpm_kernel");
}

// add_noise: sa[b]*x0 + sb[b]*noise ; velocity: sa[b]*noise - sb[b]*x0   (per-sample coefficients)
__global__ __launch_bounds__(256) void dmx_add_noise_kernel(const float* x0, const float* noise, const float* sa, const float* sb,
                                                            float* out, int B, size_t per, int velocity) {
  const size_t n = (size_t)B * per;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int b = (int)(i / per);
    const float a = sa[b], s = sb[b];
    out[i] = velocity ? __fsub_rn(__fmul_rn(a, noise[i]), __fmul_rn(s, x0[i]))
                      : __fadd_rn(__fmul_rn(a, x0[i]), __fmul_rn(s, noise[i]));
  }
}
int dmx_add_noise_launch(const float* x0, const float* noise, const float* sa, const float* sb, float* out,
                         int B, size_t per, int velocity, hipStream_t stream) {
  const size_t n = (size_t)B * per;
  int blocks = (int)((n + 255) / 256); if (blocks > 2048) blocks = 2048; if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(dmx_add_noise_kernel, dim3(blocks), dim3(256), 0, stream, x0, noise, sa, sb, out, B, per, velocity);
  return dmx_check_launch("dmx_add_noise_kernel");
}

// DiagonalGaussianDistribution.sample()/mode() on NCHW fp32 moments [B][2C][HW], then * scale
__global__ __launch_bounds__(256) void dmx_gaussian_sample_kernel(const float* moments, const float* noise, float* out, int B, int C, int HW, float scale) {
  const size_t n = (size_t)B * C * HW;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const size_t b = i / ((size_t)C * HW);
    const size_t r = i - b * (size_t)C * HW;
    const float mean = moments[b * 2 * C * HW + r];
    float v = mean;
    if (noise) {
      float lv = moments[b * 2 * C * HW + (size_t)C * HW + r];
      lv = fminf(fmaxf(lv, -30.0f), 20.0f);
      const float sd = expf(__fmul_rn(0.5f, lv));
      v = __fadd_rn(mean, __fmul_rn(sd, noise[i]));
    }
    out[i] = __fmul_rn(v, scale);
  }
}
int dmx_gaussian_sample_launch(const float* moments, const float* noise, float* out, int B, int C, int HW, float scale, hipStream_t stream) {
  const size_t n = (size_t)B * C * HW;
  int blocks = (int)((n + 255) / 256); if (blocks > 2048) blocks = 2048; if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(dmx_gaussian_sample_kernel, dim3(blocks), dim3(256), 0, stream, moments, noise, out, B, C, HW, scale);
  return dmx_check_launch("dmx_gaussian_sample_kernel");
}
