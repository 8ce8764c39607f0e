// fp32 VALIDATION instantiation of the UNet graph's kernels (north_star: "outputs match the reference diffusers CPU path
// ... within 1e-3 rel fp32"; the reference runs inference in fp32, app.ipynb:560).  Selected per call through
// dmx_unet_forward_f32: the SAME host-side graph walker (unet.hip) runs with fp32 activations, the fp32 master copy of the
// weights (the packed layout of the gradient / master arena: element of the weights arena at byte o <-> float at byte 2*o)
// and these plain fp32 kernels - so what the bf16 product path and this path share is exactly the wiring, the parameter
// packing and the fusion algebra, and what differs is the arithmetic.  Tests only; nothing here is tuned (a 64x64-tile FMA
// GEMM, exact libm erf / exp), and the product path never calls it.
#include "common.h"
#include "kernels.h"
#include <math.h>

// ------------------------------------------------------------------ implicit-GEMM conv / linear, fp32
__device__ __forceinline__ float gf32_x(const GemmF32Args& p, int m, int k) {
  if (m >= p.M || k >= p.K) return 0.f;
  if (p.direct) return k < p.cx0 ? p.x0[(size_t)m * p.ldx0 + k] : p.x1[(size_t)m * p.ldx1 + (k - p.cx0)];
  if (k >= p.Ktaps) {                                  // fused 1x1 shortcut: source has the output grid
    const int cs = k - p.Ktaps;
    return cs < p.cs0 ? p.s0[(size_t)m * p.lds0 + cs] : p.s1[(size_t)m * p.lds1 + (cs - p.cs0)];
  }
  const int tap = k / p.Cin, ci = k - tap * p.Cin;
  const int dy = tap / p.ksize, dx = tap - dy * p.ksize;
  const int ohw = p.OH * p.OW;
  const int b = m / ohw, rem = m - b * ohw, oy = rem / p.OW, ox = rem - oy * p.OW;
  const int iy = oy * p.stride - p.pad + dy, ix = ox * p.stride - p.pad + dx;
  const int eh = p.ups ? 2 * p.IH : p.IH, ew = p.ups ? 2 * p.IW : p.IW;
  if (iy < 0 || iy >= eh || ix < 0 || ix >= ew) return 0.f;
  const int sy = p.ups ? iy >> 1 : iy, sx = p.ups ? ix >> 1 : ix;
  const size_t pix = (size_t)b * p.IH * p.IW + (size_t)sy * p.IW + sx;
  return ci < p.cx0 ? p.x0[pix * p.ldx0 + ci] : p.x1[pix * p.ldx1 + (ci - p.cx0)];
}
__device__ __forceinline__ float gelu_exact(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

__global__ __launch_bounds__(256) void dmx_gemm_f32_kernel(const GemmF32Args p) {
  __shared__ float Xs[16][68], Ws[16][68], T[64][65];
  const int t = threadIdx.x, tx = t & 15, ty = t >> 4;
  const int m0 = blockIdx.x * 64, n0 = blockIdx.y * 64;
  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
  for (int k0 = 0; k0 < p.K; k0 += 16) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = t + 256 * i, r = e >> 4, kk = e & 15;
      Xs[kk][r] = gf32_x(p, m0 + r, k0 + kk);
      Ws[kk][r] = (n0 + r < p.N && k0 + kk < p.K) ? p.w[(size_t)(n0 + r) * p.ldw + k0 + kk] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      float a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { a[i] = Xs[kk][ty * 4 + i]; b[i] = Ws[kk][tx * 4 + i]; }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int m = m0 + ty * 4 + i, n = n0 + tx * 4 + j;
      float v = acc[i][j];
      if (m < p.M && n < p.N) {
        if (p.bias) v += p.bias[n];
        if (p.bias2) v += p.bias2[n];
        if (p.rowbias) v += p.rowbias[(size_t)(m / p.rows_per_group) * p.ldrb + n];
        if (p.act == 1) v = gelu_exact(v);
      }
      T[ty * 4 + i][tx * 4 + j] = v;
    }
  __syncthreads();
  if (p.geglu) {                                      // packed 64-column groups: 32 'a' columns then their 32 gate columns
    for (int e = t; e < 64 * 32; e += 256) {
      const int r = e >> 5, jj = e & 31, m = m0 + r;
      if (m < p.M && n0 + 32 + jj < p.N) p.out[(size_t)m * p.ldo + (n0 >> 1) + jj] = T[r][jj] * gelu_exact(T[r][32 + jj]);
    }
    return;
  }
  for (int e = t; e < 64 * 64; e += 256) {
    const int r = e >> 6, c = e & 63, m = m0 + r, n = n0 + c;
    if (m < p.M && n < p.N) {
      float v = T[r][c];
      if (p.res) v += p.res[(size_t)m * p.ldres + n];
      p.out[(size_t)m * p.ldo + n] = v;
    }
  }
}
int dmx_gemm_f32_launch(const GemmF32Args& a, hipStream_t stream) {
  DMX_REQUIRE(a.M > 0 && a.N > 0 && a.K > 0 && a.x0 && a.w && a.out, "gemm_f32: bad arguments");
  if (a.geglu) DMX_REQUIRE(a.N % 64 == 0 && !a.res, "gemm_f32: GEGLU needs N %% 64 == 0 and no residual");
  hipLaunchKernelGGL(dmx_gemm_f32_kernel, dim3(cdiv(a.M, 64), cdiv(a.N, 64)), dim3(256), 0, stream, a);
  return dmx_check_launch("dmx_gemm_f32_kernel");
}

// ------------------------------------------------------------------ GroupNorm (+SiLU), two-source concat, fp32: one block per (sample, group)
__global__ __launch_bounds__(256) void dmx_groupnorm_f32_kernel(const float* x0, int ldx0, int c0, const float* x1, int ldx1, int C, int groups, int HW,
                                                                const float* gamma, const float* beta, float eps, int silu, float* y, int ldy) {
  __shared__ double red[256];
  const int b = blockIdx.x / groups, g = blockIdx.x % groups, cpg = C / groups, t = threadIdx.x;
  const int n = HW * cpg;
  auto at = [&](int e) -> float {
    const int pix = e / cpg, c = g * cpg + (e - pix * cpg);
    const size_t row = (size_t)b * HW + pix;
    return c < c0 ? x0[row * ldx0 + c] : x1[row * ldx1 + (c - c0)];
  };
  double s = 0.0;
  for (int e = t; e < n; e += 256) s += (double)at(e);
  red[t] = s; __syncthreads();
  for (int k = 128; k >= 1; k >>= 1) { if (t < k) red[t] += red[t + k]; __syncthreads(); }
  const double mean = red[0] / n;
  __syncthreads();
  double q = 0.0;
  for (int e = t; e < n; e += 256) { const double d = (double)at(e) - mean; q += d * d; }
  red[t] = q; __syncthreads();
  for (int k = 128; k >= 1; k >>= 1) { if (t < k) red[t] += red[t + k]; __syncthreads(); }
  const float rstd = (float)(1.0 / sqrt(red[0] / n + (double)eps)), fm = (float)mean;
  for (int e = t; e < n; e += 256) {
    const int pix = e / cpg, c = g * cpg + (e - pix * cpg);
    float v = (at(e) - fm) * rstd * gamma[c] + beta[c];
    if (silu) v = v / (1.0f + expf(-v));
    y[((size_t)b * HW + pix) * ldy + c] = v;
  }
}
int dmx_groupnorm_f32_launch(const float* x0, int ldx0, int c0, const float* x1, int ldx1, int C, int groups, int B, int HW,
                             const float* gamma, const float* beta, float eps, int silu, float* y, int ldy, hipStream_t stream) {
  DMX_REQUIRE(C % groups == 0, "groupnorm_f32: C=%d not divisible by groups=%d", C, groups);
  if (!x1) { x1 = x0; ldx1 = ldx0; c0 = C; }
  hipLaunchKernelGGL(dmx_groupnorm_f32_kernel, dim3(B * groups), dim3(256), 0, stream, x0, ldx0, c0, x1, ldx1, C, groups, HW, gamma, beta, eps, silu, y, ldy);
  return dmx_check_launch("dmx_groupnorm_f32_kernel");
}

// ------------------------------------------------------------------ LayerNorm, fp32: one wave per row
__global__ __launch_bounds__(256) void dmx_layernorm_f32_kernel(const float* x, int ldx, float* y, int ldy, const float* gamma, const float* beta,
                                                                int rows, int C, float eps) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + (size_t)row * ldx;
  float s = 0.f;
  for (int c = lane; c < C; c += 64) s += xr[c];
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d);
  const float mean = s / (float)C;
  float q = 0.f;
  for (int c = lane; c < C; c += 64) { const float d = xr[c] - mean; q += d * d; }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) q += __shfl_xor(q, d);
  const float rstd = 1.0f / sqrtf(q / (float)C + eps);
  for (int c = lane; c < C; c += 64) y[(size_t)row * ldy + c] = (xr[c] - mean) * rstd * gamma[c] + beta[c];
}
int dmx_layernorm_f32_launch(const float* x, int ldx, float* y, int ldy, const float* gamma, const float* beta, int rows, int C, float eps, hipStream_t stream) {
  hipLaunchKernelGGL(dmx_layernorm_f32_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, stream, x, ldx, y, ldy, gamma, beta, rows, C, eps);
  return dmx_check_launch("dmx_layernorm_f32_kernel");
}

// ------------------------------------------------------------------ attention, head dim D (64 in the UNet, 128 / 256 / 512 = the
// single head of the autoencoder's mid block), fp32, exact two-pass softmax: block = 16 queries of one (b, h)
template <int D>
__global__ __launch_bounds__(256) void dmx_attention_f32_kernel(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, int kv_rows,
                                                                float* o, int ldo, int H, int Sq, int Skv, float scale) {
  extern __shared__ float sm[];                       // [16][D] q | [16][Skv] scores
  float* qs = sm; float* sc = sm + 16 * D;
  const int t = threadIdx.x, b = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * 16;
  for (int e = t; e < 16 * D; e += 256) {
    const int r = e / D, d = e % D; int qr = q0 + r; if (qr >= Sq) qr = Sq - 1;
    qs[e] = q[((size_t)b * Sq + qr) * ldq + h * D + d];
  }
  __syncthreads();
  const int qi = t >> 4, sub = t & 15;
  for (int kj = sub; kj < Skv; kj += 16) {
    const float* kr = k + ((size_t)b * kv_rows + kj) * ldk + h * D;
    float s = 0.f;
#pragma unroll 8
    for (int d = 0; d < D; ++d) s = fmaf(qs[qi * D + d], kr[d], s);
    sc[qi * Skv + kj] = s * scale;
  }
  __syncthreads();
  float mx = -INFINITY;
  for (int kj = sub; kj < Skv; kj += 16) mx = fmaxf(mx, sc[qi * Skv + kj]);
#pragma unroll
  for (int d = 8; d >= 1; d >>= 1) mx = fmaxf(mx, __shfl_xor(mx, d));
  float sum = 0.f;
  for (int kj = sub; kj < Skv; kj += 16) { const float e = expf(sc[qi * Skv + kj] - mx); sc[qi * Skv + kj] = e; sum += e; }
#pragma unroll
  for (int d = 8; d >= 1; d >>= 1) sum += __shfl_xor(sum, d);
  __syncthreads();
  const float inv = 1.0f / sum;
  constexpr int PT = D / 16;                          // output channels per thread
  float acc[PT];
#pragma unroll
  for (int e = 0; e < PT; ++e) acc[e] = 0.f;
  for (int kj = 0; kj < Skv; ++kj) {
    const float pr = sc[qi * Skv + kj];
    const float* vr = v + ((size_t)b * kv_rows + kj) * ldv + h * D + sub * PT;
#pragma unroll
    for (int e = 0; e < PT; ++e) acc[e] = fmaf(pr, vr[e], acc[e]);
  }
  if (q0 + qi < Sq) {
    float* orow = o + ((size_t)b * Sq + q0 + qi) * ldo + h * D + sub * PT;
#pragma unroll
    for (int e = 0; e < PT; ++e) orow[e] = acc[e] * inv;
  }
}
template <int D>
static int attention_f32_launch_d(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, int kv_rows, float* o, int ldo,
                                  int B, int H, int Sq, int Skv, float scale, hipStream_t stream) {
  const size_t lds = (size_t)(16 * D + 16 * Skv) * sizeof(float);
  DMX_REQUIRE(lds <= 150 * 1024, "attention_f32: Skv=%d too long for the validation kernel (scores of 16 queries live in LDS)", Skv);
  DMX_LDS_OPT_IN((dmx_attention_f32_kernel<D>), 150 * 1024);
  hipLaunchKernelGGL((dmx_attention_f32_kernel<D>), dim3(cdiv(Sq, 16), H, B), dim3(256), lds, stream, q, ldq, k, ldk, v, ldv, kv_rows, o, ldo, H, Sq, Skv, scale);
  return dmx_check_launch("dmx_attention_f32_kernel");
}
int dmx_attention_f32_launch(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, int kv_rows, float* o, int ldo,
                             int B, int H, int Sq, int Skv, float scale, hipStream_t stream, int head_dim) {
  switch (head_dim) {
    case 64: return attention_f32_launch_d<64>(q, ldq, k, ldk, v, ldv, kv_rows, o, ldo, B, H, Sq, Skv, scale, stream);
    case 128: return attention_f32_launch_d<128>(q, ldq, k, ldk, v, ldv, kv_rows, o, ldo, B, H, Sq, Skv, scale, stream);
    case 256: return attention_f32_launch_d<256>(q, ldq, k, ldk, v, ldv, kv_rows, o, ldo, B, H, Sq, Skv, scale, stream);
    case 512: return attention_f32_launch_d<512>(q, ldq, k, ldk, v, ldv, kv_rows, o, ldo, B, H, Sq, Skv, scale, stream);
  }
  dmx_set_error("attention_f32: head dim %d not built (64, 128, 256, 512)", head_dim);
  return DMX_ERR_ARG;
}

// ------------------------------------------------------------------ small elementwise helpers
__global__ void dmx_silu_f32_kernel(float* x, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { const float v = x[i]; x[i] = v / (1.0f + expf(-v)); }
}
int dmx_silu_f32_launch(float* x, size_t n, hipStream_t stream) {
  hipLaunchKernelGGL(dmx_silu_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, x, n);
  return dmx_check_launch("dmx_silu_f32_kernel");
}
// torch.cat([f0, f1, f2], 1) of NCHW fp32 tensors -> NHWC fp32 [B*HW][C]
__global__ void dmx_concat_nchw_to_nhwc_f32_kernel(const float* f0, int c0, const float* f1, int c1, const float* f2, int c2, float* out, int B, int HW) {
  const int C = c0 + c1 + c2;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)B * HW * C) return;
  const int c = (int)(i % C); const size_t pix = i / C; const int b = (int)(pix / HW); const int p = (int)(pix - (size_t)b * HW);
  float v;
  if (c < c0) v = f0[((size_t)b * c0 + c) * HW + p];
  else if (c < c0 + c1) v = f1[((size_t)b * c1 + (c - c0)) * HW + p];
  else v = f2[((size_t)b * c2 + (c - c0 - c1)) * HW + p];
  out[i] = v;
}
int dmx_concat_nchw_to_nhwc_f32_launch(const float* f0, int c0, const float* f1, int c1, const float* f2, int c2, float* out, int B, int HW, hipStream_t stream) {
  const size_t n = (size_t)B * HW * (c0 + c1 + c2);
  hipLaunchKernelGGL(dmx_concat_nchw_to_nhwc_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, f0, c0, f1, c1, f2, c2, out, B, HW);
  return dmx_check_launch("dmx_concat_nchw_to_nhwc_f32_kernel");
}
