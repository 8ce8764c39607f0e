// Backward of GroupNorm(+SiLU), LayerNorm and GEGLU for NHWC bf16 tensors (training rows of SURVEY.md 8a: P5 over K3,
// K4, K8).  Same shape as the forward kernels: 16 B per lane, fp32 statistics, deterministic (fixed-order partials,
// no float atomics).
//
// GroupNorm, y = act(xhat*gamma + beta), xhat = (x - mean_g)*rstd_g, act = SiLU or identity:
//   du      = dy * act'(u)                     (u recomputed from x and the saved mean / rstd)
//   dbeta_c = sum du ; dgamma_c = sum du*xhat  (over batch and pixels)
//   dx      = rstd_g * (du*gamma_c - m1_g - xhat*m2_g),  m1 = mean_g(du*gamma), m2 = mean_g(du*gamma*xhat)
//   reduce   : grid (nchunk, B)   per-channel partial (sum du, sum du*xhat) of a row chunk  -> part[b][chunk][C][2]
//   finalize : grid (C/256, B)    AB[b][c] = sum over chunks ;  params: dgamma/dbeta (+)= sum over b
//   apply    : grid (rows, B)     prologue folds AB into m1/m2 per group, then streams rows and writes dx (two-source
//                                 concat splits into dx0 | dx1; an optional residual gradient is added)
#include "common.h"
#include "kernels.h"

namespace {
__device__ __forceinline__ const bf16* src2(const bf16* x0, int ld0, const bf16* x1, int ld1, int c0, size_t row, int c) {
  return (c < c0) ? (x0 + row * ld0 + c) : (x1 + row * ld1 + (c - c0));
}
__device__ __forceinline__ float dsilu_f(float u) {          // d/du [u * sigmoid(u)]
  const float s = 1.0f / (1.0f + __expf(-u));
  return s * (1.0f + u * (1.0f - s));
}

struct GnCh {           // a thread's 8 channels: xhat = x*ra - rb ; u = xhat*g + bt
  float ra[8], rb[8], g[8], bt[8];
};
__device__ __forceinline__ void gn_load_ch(const GroupNormBwdArgs& p, int b, int co, GnCh& k) {
  const int cpg = p.C / p.groups;
  const f32x4 g0 = *(const f32x4*)(p.gamma + co * 8), g1 = *(const f32x4*)(p.gamma + co * 8 + 4);
  const f32x4 b0 = *(const f32x4*)(p.beta + co * 8), b1 = *(const f32x4*)(p.beta + co * 8 + 4);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int gg = (co * 8 + i) / cpg;
    const float mean = p.stats[((size_t)b * p.groups + gg) * 2], rstd = p.stats[((size_t)b * p.groups + gg) * 2 + 1];
    k.ra[i] = rstd; k.rb[i] = mean * rstd;
    k.g[i] = i < 4 ? g0[i] : g1[i - 4]; k.bt[i] = i < 4 ? b0[i] : b1[i - 4];
  }
}

__global__ __launch_bounds__(1024) void dmx_gn_bwd_reduce_kernel(const GroupNormBwdArgs p) {
  extern __shared__ float sm[];          // [R][C] sum du, [R][C] sum du*xhat
  const int oc = p.C >> 3;
  const int R = blockDim.x / oc;
  const int t = threadIdx.x;
  const int r = t / oc, co = t - r * oc;
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int row0 = chunk * p.rows_per_chunk;
  const int row1 = min(row0 + p.rows_per_chunk, p.HW);
  GnCh k; gn_load_ch(p, b, co, k);
  float sa[8], sb[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { sa[i] = 0.f; sb[i] = 0.f; }
  const size_t base = (size_t)b * p.HW;
  for (int row = row0 + r; row < row1; row += R) {
    const u32x4 xv = *(const u32x4*)src2(p.x0, p.ldx0, p.x1, p.ldx1, p.c0, base + row, co * 8);
    const u32x4 dv = *(const u32x4*)(p.dy + (base + row) * p.lddy + co * 8);
    float x[8], d[8]; unpack_bf8(xv, x); unpack_bf8(dv, d);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float xh = x[i] * k.ra[i] - k.rb[i];
      const float du = p.silu ? d[i] * dsilu_f(xh * k.g[i] + k.bt[i]) : d[i];
      sa[i] += du; sb[i] += du * xh;
    }
  }
  float* SA = sm; float* SB = sm + R * p.C;
#pragma unroll
  for (int i = 0; i < 8; ++i) { SA[r * p.C + co * 8 + i] = sa[i]; SB[r * p.C + co * 8 + i] = sb[i]; }
  __syncthreads();
  float* o = p.part + ((size_t)b * p.nchunk + chunk) * p.C * 2;
  for (int c = t; c < p.C; c += blockDim.x) {
    float a = 0.f, q = 0.f;
    for (int j = 0; j < R; ++j) { a += SA[j * p.C + c]; q += SB[j * p.C + c]; }
    o[2 * c] = a; o[2 * c + 1] = q;
  }
}

__global__ __launch_bounds__(256) void dmx_gn_bwd_finalize_kernel(const GroupNormBwdArgs p) {      // grid (C/256, B)
  const int c = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
  if (c >= p.C) return;
  float a = 0.f, q = 0.f;
  const float* pp = p.part + (size_t)b * p.nchunk * p.C * 2 + 2 * c;
  for (int j = 0; j < p.nchunk; ++j) { a += pp[(size_t)j * p.C * 2]; q += pp[(size_t)j * p.C * 2 + 1]; }
  p.ab[((size_t)b * p.C + c) * 2] = a; p.ab[((size_t)b * p.C + c) * 2 + 1] = q;
}
__global__ __launch_bounds__(256) void dmx_gn_bwd_params_kernel(const GroupNormBwdArgs p) {        // grid (C/256)
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= p.C) return;
  float dg = 0.f, db = 0.f;
  for (int b = 0; b < p.B; ++b) { db += p.ab[((size_t)b * p.C + c) * 2]; dg += p.ab[((size_t)b * p.C + c) * 2 + 1]; }
  if (p.dgamma) p.dgamma[c] = p.accumulate ? p.dgamma[c] + dg : dg;
  if (p.dbeta) p.dbeta[c] = p.accumulate ? p.dbeta[c] + db : db;
}

__global__ __launch_bounds__(1024) void dmx_gn_bwd_apply_kernel(const GroupNormBwdArgs p, int rows_per_block) {
  __shared__ float M1[64], M2[64];
  const int oc = p.C >> 3;
  const int R = blockDim.x / oc;
  const int t = threadIdx.x;
  const int r = t / oc, co = t - r * oc;
  const int b = blockIdx.y;
  const int cpg = p.C / p.groups;
  if (t < p.groups) {
    float a = 0.f, q = 0.f;
    for (int j = 0; j < cpg; ++j) {
      const int c = t * cpg + j;
      const float g = p.gamma[c];
      a += g * p.ab[((size_t)b * p.C + c) * 2]; q += g * p.ab[((size_t)b * p.C + c) * 2 + 1];
    }
    const float inv_n = 1.0f / ((float)p.HW * (float)cpg);
    M1[t] = a * inv_n; M2[t] = q * inv_n;
  }
  __syncthreads();
  GnCh k; gn_load_ch(p, b, co, k);
  float m1[8], m2[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { const int gg = (co * 8 + i) / cpg; m1[i] = M1[gg]; m2[i] = M2[gg]; }
  const int row0 = blockIdx.x * rows_per_block;
  const int row1 = min(row0 + rows_per_block, p.HW);
  const size_t base = (size_t)b * p.HW;
  const int c = co * 8;
  for (int row = row0 + r; row < row1; row += R) {
    const u32x4 xv = *(const u32x4*)src2(p.x0, p.ldx0, p.x1, p.ldx1, p.c0, base + row, c);
    const u32x4 dv = *(const u32x4*)(p.dy + (base + row) * p.lddy + c);
    float x[8], d[8], o[8]; unpack_bf8(xv, x); unpack_bf8(dv, d);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float xh = x[i] * k.ra[i] - k.rb[i];
      const float du = p.silu ? d[i] * dsilu_f(xh * k.g[i] + k.bt[i]) : d[i];
      o[i] = k.ra[i] * (du * k.g[i] - m1[i] - xh * m2[i]);
    }
    bf16* dst = (c < p.c0) ? (p.dx0 + (base + row) * p.lddx0 + c) : (p.dx1 + (base + row) * p.lddx1 + (c - p.c0));
    const bf16* rs = nullptr;
    if (c < p.c0) { if (p.res0) rs = p.res0 + (base + row) * p.ldres0 + c; }
    else if (p.res1) rs = p.res1 + (base + row) * p.ldres1 + (c - p.c0);
    if (rs) {
      float f[8]; unpack_bf8(*(const u32x4*)rs, f);
#pragma unroll
      for (int i = 0; i < 8; ++i) o[i] += f[i];
    }
    *(u32x4*)dst = pack_bf8(o);
  }
}

// ---------------------------------------------------------------------------- LayerNorm backward
// One wave per row (the row lives in registers, C <= 2048): mean / rstd are recomputed, then
//   dx = rstd * (dy*gamma - mean(dy*gamma) - xhat * mean(dy*gamma*xhat)).
// Each block walks LNB_ROWS rows per wave and keeps per-lane column partials of dgamma (dy*xhat) / dbeta (dy) in
// registers; the 4 waves are folded through LDS -> part[block][C][2]; a second kernel sums the blocks in order.
#define LNB_ROWS 4
__global__ __launch_bounds__(256) void dmx_ln_bwd_kernel(const bf16* x, int ldx, const bf16* dy, int lddy, const float* gamma,
                                                         bf16* dx, int lddx, const bf16* res, int ldres,
                                                         float* part, int rows, int C, float eps) {
  extern __shared__ float sm[];               // [4][C][2]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int oc = C >> 3;
  const int nj = (oc + 63) >> 6;              // octets per lane (1..4)
  float ga[4][8], pg[4][8], pb[4][8];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int o = lane + 64 * j;
#pragma unroll
    for (int i = 0; i < 8; ++i) { pg[j][i] = 0.f; pb[j][i] = 0.f; ga[j][i] = (j < nj && o < oc) ? gamma[o * 8 + i] : 0.f; }
  }
  const int row_base = (blockIdx.x * 4 + wave) * LNB_ROWS;
  for (int rr = 0; rr < LNB_ROWS; ++rr) {
    const int row = row_base + rr;
    if (row >= rows) break;
    float xf[4][8], df[4][8];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int o = lane + 64 * j;
      if (j < nj && o < oc) {
        unpack_bf8(*(const u32x4*)(x + (size_t)row * ldx + o * 8), xf[j]);
        unpack_bf8(*(const u32x4*)(dy + (size_t)row * lddy + o * 8), df[j]);
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) { xf[j][i] = 0.f; df[j][i] = 0.f; }
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) s += xf[j][i];
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d);
    const float mean = s / (float)C;
    float v = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int o = lane + 64 * j;
      if (j < nj && o < oc) {
#pragma unroll
        for (int i = 0; i < 8; ++i) { const float dlt = xf[j][i] - mean; v += dlt * dlt; }
      }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    const float rstd = rsqrtf(v / (float)C + eps);
    float a = 0.f, q = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float xh = (xf[j][i] - mean) * rstd;           // 0 contribution from padded octets: df = ga = 0
        const float dg = df[j][i] * ga[j][i];
        a += dg; q += dg * xh;
        pg[j][i] += df[j][i] * xh; pb[j][i] += df[j][i];
        xf[j][i] = xh;
      }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { a += __shfl_xor(a, d); q += __shfl_xor(q, d); }
    const float m1 = a / (float)C, m2 = q / (float)C;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int o = lane + 64 * j;
      if (j < nj && o < oc) {
        float ov[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) ov[i] = rstd * (df[j][i] * ga[j][i] - m1 - xf[j][i] * m2);
        if (res) {
          float f[8]; unpack_bf8(*(const u32x4*)(res + (size_t)row * ldres + o * 8), f);
#pragma unroll
          for (int i = 0; i < 8; ++i) ov[i] += f[i];
        }
        *(u32x4*)(dx + (size_t)row * lddx + o * 8) = pack_bf8(ov);
      }
    }
  }
  // fold the 4 waves' column partials
  float* W = sm + (size_t)wave * C * 2;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int o = lane + 64 * j;
    if (j < nj && o < oc) {
#pragma unroll
      for (int i = 0; i < 8; ++i) { W[(o * 8 + i) * 2] = pg[j][i]; W[(o * 8 + i) * 2 + 1] = pb[j][i]; }
    }
  }
  __syncthreads();
  float* o = part + (size_t)blockIdx.x * C * 2;
  for (int c = threadIdx.x; c < 2 * C; c += 256) o[c] = (sm[c] + sm[2 * C + c]) + (sm[4 * C + c] + sm[6 * C + c]);
}
// part[nblk][C][2] -> dgamma / dbeta, two fixed-order levels (LNP_SLICES partial sums in parallel, then their sum)
#define LNP_SLICES 32
__global__ __launch_bounds__(256) void dmx_ln_bwd_params1_kernel(const float* part, int nblk, int C2, float* part2) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x, sl = blockIdx.y;
  if (c >= C2) return;
  const int per = (nblk + LNP_SLICES - 1) / LNP_SLICES;
  const int j0 = sl * per, j1 = min(j0 + per, nblk);
  float s = 0.f;
  for (int j = j0; j < j1; ++j) s += part[(size_t)j * C2 + c];
  part2[(size_t)sl * C2 + c] = s;
}
__global__ __launch_bounds__(256) void dmx_ln_bwd_params2_kernel(const float* part2, int C, float* dgamma, float* dbeta, int accumulate) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float g = 0.f, b = 0.f;
  for (int j = 0; j < LNP_SLICES; ++j) { g += part2[((size_t)j * C + c) * 2]; b += part2[((size_t)j * C + c) * 2 + 1]; }
  if (dgamma) dgamma[c] = accumulate ? dgamma[c] + g : g;
  if (dbeta) dbeta[c] = accumulate ? dbeta[c] + b : b;
}

// ---------------------------------------------------------------------------- GEGLU (unfused, training)
// h = [a | g] with C2 columns each: y = a * gelu(g) (exact erf GELU);  da = dy*gelu(g), dg = dy*a*(Phi(g) + g*phi(g))
__device__ __forceinline__ float gelu_exact_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float dgelu_exact_f(float x) {
  return 0.5f * (1.0f + erff(x * 0.70710678118654752440f)) + x * 0.39894228040143267794f * __expf(-0.5f * x * x);
}
__global__ __launch_bounds__(256) void dmx_geglu_fwd_kernel(const bf16* h, int ldh, bf16* y, int ldy, int rows, int C2, int packed) {
  const int c8 = C2 / 8;
  const size_t total = (size_t)rows * c8;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % c8) * 8; const size_t r = i / c8;
    const int ac = packed ? 64 * (c >> 5) + (c & 31) : c, gc = packed ? ac + 32 : C2 + c;
    float a[8], g[8]; unpack_bf8(*(const u32x4*)(h + r * ldh + ac), a); unpack_bf8(*(const u32x4*)(h + r * ldh + gc), g);
#pragma unroll
    for (int e = 0; e < 8; ++e) a[e] *= gelu_exact_f(g[e]);
    *(u32x4*)(y + r * ldy + c) = pack_bf8(a);
  }
}
__global__ __launch_bounds__(256) void dmx_geglu_bwd_kernel(const bf16* h, int ldh, const bf16* dy, int lddy, bf16* dh, int lddh, int rows, int C2, int packed) {
  const int c8 = C2 / 8;
  const size_t total = (size_t)rows * c8;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % c8) * 8; const size_t r = i / c8;
    const int ac = packed ? 64 * (c >> 5) + (c & 31) : c, gc = packed ? ac + 32 : C2 + c;
    float a[8], g[8], d[8];
    unpack_bf8(*(const u32x4*)(h + r * ldh + ac), a); unpack_bf8(*(const u32x4*)(h + r * ldh + gc), g);
    unpack_bf8(*(const u32x4*)(dy + r * lddy + c), d);
    float da[8], dg[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { da[e] = d[e] * gelu_exact_f(g[e]); dg[e] = d[e] * a[e] * dgelu_exact_f(g[e]); }
    *(u32x4*)(dh + r * lddh + ac) = pack_bf8(da);
    *(u32x4*)(dh + r * lddh + gc) = pack_bf8(dg);
  }
}
}  // namespace

// workspace: part [B][nchunk<=256][C][2] + ab [B][C][2]
static void gn_bwd_chunks(int C, int HW, int* R_out, int* rpc_out, int* nchunk_out) {
  const int oc = C / 8;
  int R = 1024 / oc; if (R > 16) R = 16; if (R < 1) R = 1;
  int rpc = 8 * R;
  if (cdiv(HW, rpc) > 256) rpc = cdiv(cdiv(HW, 256), R) * R;
  *R_out = R; *rpc_out = rpc; *nchunk_out = cdiv(HW, rpc);
}
size_t dmx_gn_bwd_workspace_bytes(int B, int HW, int C) {
  int R, rpc, nchunk; gn_bwd_chunks(C, HW, &R, &rpc, &nchunk);
  return ((size_t)B * nchunk * C * 2 + (size_t)B * C * 2) * sizeof(float);
}
int dmx_groupnorm_bwd_launch(GroupNormBwdArgs a, hipStream_t stream) {
  DMX_REQUIRE(a.C % 8 == 0 && a.C % a.groups == 0 && a.C <= 2560 && a.groups <= 64, "groupnorm_bwd: unsupported C=%d groups=%d", a.C, a.groups);
  DMX_REQUIRE(a.stats && a.part && a.dy && a.dx0, "groupnorm_bwd: null argument");
  if (a.c0 >= a.C || a.x1 == nullptr) { a.c0 = a.C; a.x1 = a.x0; a.ldx1 = a.ldx0; a.dx1 = a.dx0; a.lddx1 = a.lddx0; a.res1 = a.res0; a.ldres1 = a.ldres0; }
  DMX_REQUIRE(a.c0 % 8 == 0 && a.ldx0 % 8 == 0 && a.lddy % 8 == 0 && a.lddx0 % 8 == 0, "groupnorm_bwd: strides/splits must be multiples of 8");
  int R, rpc, nchunk; gn_bwd_chunks(a.C, a.HW, &R, &rpc, &nchunk);
  a.rows_per_chunk = rpc; a.nchunk = nchunk;
  a.ab = a.part + (size_t)a.B * nchunk * a.C * 2;
  const int threads = (a.C / 8) * R;
  const size_t lds = (size_t)2 * R * a.C * sizeof(float);
  DMX_LDS_OPT_IN((dmx_gn_bwd_reduce_kernel), 160 * 1024);
  ProfScope ps(PROF_GNORM, stream, 0.0, 10.0 * a.B * (double)a.HW * a.C, "gn_bwd");
  hipLaunchKernelGGL(dmx_gn_bwd_reduce_kernel, dim3(nchunk, a.B), dim3(threads), lds, stream, a);
  int rc = dmx_check_launch("dmx_gn_bwd_reduce_kernel");
  if (rc) return rc;
  hipLaunchKernelGGL(dmx_gn_bwd_finalize_kernel, dim3(cdiv(a.C, 256), a.B), dim3(256), 0, stream, a);
  rc = dmx_check_launch("dmx_gn_bwd_finalize_kernel");
  if (rc) return rc;
  hipLaunchKernelGGL(dmx_gn_bwd_params_kernel, dim3(cdiv(a.C, 256)), dim3(256), 0, stream, a);
  rc = dmx_check_launch("dmx_gn_bwd_params_kernel");
  if (rc) return rc;
  int rpb = 8 * R;
  while (rpb > R && (long)cdiv(a.HW, rpb) * a.B < 512) rpb -= R;
  hipLaunchKernelGGL(dmx_gn_bwd_apply_kernel, dim3(cdiv(a.HW, rpb), a.B), dim3(threads), 0, stream, a, rpb);
  return dmx_check_launch("dmx_gn_bwd_apply_kernel");
}

size_t dmx_ln_bwd_workspace_bytes(int rows, int C) { return ((size_t)cdiv(rows, 4 * LNB_ROWS) + LNP_SLICES) * C * 2 * sizeof(float); }
int dmx_layernorm_bwd_launch(const bf16* x, int ldx, const bf16* dy, int lddy, const float* gamma, bf16* dx, int lddx,
                             const bf16* res, int ldres, float* dgamma, float* dbeta, int accumulate,
                             int rows, int C, float eps, void* workspace, size_t workspace_bytes, hipStream_t stream) {
  DMX_REQUIRE(C % 8 == 0 && C <= 2048 && ldx % 8 == 0 && lddy % 8 == 0 && lddx % 8 == 0, "layernorm_bwd: C=%d must be a multiple of 8 and <= 2048", C);
  const size_t need = dmx_ln_bwd_workspace_bytes(rows, C);
  if (workspace == nullptr || workspace_bytes < need) { dmx_set_error("layernorm_bwd: needs %zu bytes of workspace, got %zu", need, workspace_bytes); return DMX_ERR_WORKSPACE; }
  const int nblk = cdiv(rows, 4 * LNB_ROWS);
  ProfScope ps(PROF_LNORM, stream, 0.0, 6.0 * rows * (double)C, "ln_bwd");
  hipLaunchKernelGGL(dmx_ln_bwd_kernel, dim3(nblk), dim3(256), (size_t)8 * C * sizeof(float), stream, x, ldx, dy, lddy, gamma, dx, lddx, res, ldres,
                     (float*)workspace, rows, C, eps);
  int rc = dmx_check_launch("dmx_ln_bwd_kernel");
  if (rc) return rc;
  float* part2 = (float*)workspace + (size_t)nblk * C * 2;
  hipLaunchKernelGGL(dmx_ln_bwd_params1_kernel, dim3(cdiv(2 * C, 256), LNP_SLICES), dim3(256), 0, stream, (const float*)workspace, nblk, 2 * C, part2);
  rc = dmx_check_launch("dmx_ln_bwd_params1_kernel");
  if (rc) return rc;
  hipLaunchKernelGGL(dmx_ln_bwd_params2_kernel, dim3(cdiv(C, 256)), dim3(256), 0, stream, (const float*)part2, C, dgamma, dbeta, accumulate);
  return dmx_check_launch("dmx_ln_bwd_params2_kernel");
}

int dmx_geglu_fwd_launch(const bf16* h, int ldh, bf16* y, int ldy, int rows, int C2, int packed, hipStream_t stream) {
  DMX_REQUIRE(C2 % 8 == 0 && ldh % 8 == 0 && ldy % 8 == 0 && (!packed || C2 % 32 == 0), "geglu: C2 %% 8 (%% 32 when packed)");
  const size_t total = (size_t)rows * (C2 / 8);
  int blocks = (int)((total + 255) / 256); if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(dmx_geglu_fwd_kernel, dim3(blocks), dim3(256), 0, stream, h, ldh, y, ldy, rows, C2, packed);
  return dmx_check_launch("dmx_geglu_fwd_kernel");
}
int dmx_geglu_bwd_launch(const bf16* h, int ldh, const bf16* dy, int lddy, bf16* dh, int lddh, int rows, int C2, int packed, hipStream_t stream) {
  DMX_REQUIRE(C2 % 8 == 0 && ldh % 8 == 0 && lddy % 8 == 0 && lddh % 8 == 0, "geglu_bwd: C2 %% 8");
  const size_t total = (size_t)rows * (C2 / 8);
  int blocks = (int)((total + 255) / 256); if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(dmx_geglu_bwd_kernel, dim3(blocks), dim3(256), 0, stream, h, ldh, dy, lddy, dh, lddh, rows, C2, packed);
  return dmx_check_launch("dmx_geglu_bwd_kernel");
}
