// GroupNorm(32 groups)[+SiLU] and LayerNorm for NHWC bf16 tensors.
//
// Both are HBM-bound: every load/store is 16 B per lane (8 bf16), statistics in fp32.
//
// GroupNorm is ONE launch when a (sample, group) slab fits the registers of a 1024-thread block (every shape of the
// 512-px pipeline; dmx_gn_slab_kernel below), otherwise two launches; both are deterministic (no float atomics):
//   stats : grid (nchunk, B); each block reduces rows_per_chunk pixels x C channels to
//           per-group (sum, sumsq) partials  -> partial[b][chunk][g][2]
//   apply : grid (row blocks, B); a parallel prologue folds the <=256 partials of its sample in a
//           fixed order into mean/rstd, each thread derives its 8 channels' affine into registers,
//           then y = x*a + b (+SiLU), bf16 out, 4 rows in flight per thread.
// The input may be the virtual channel-concat of two tensors (UNet skip connections), so
// torch.cat([h, skip], 1) is never materialised.
#include "common.h"
#include "kernels.h"
#include <stdlib.h>

__device__ __forceinline__ const bf16* gn_src(const GroupNormArgs& p, size_t row, int c) {
  return (c < p.c0) ? (p.x0 + row * p.ldx0 + c) : (p.x1 + row * p.ldx1 + (c - p.c0));
}

// stats: thread = (row lane r, channel octet co); 4 rows in flight per thread
__global__ __launch_bounds__(1024) void dmx_gn_stats_kernel(const GroupNormArgs p) {
  extern __shared__ float sm[];          // [R][C] sums, [R][C] sumsq, then [C] x2
  const int oc = p.C >> 3;               // octets per row
  const int R = blockDim.x / oc;
  const int t = threadIdx.x;
  const int r = t / oc, co = t - r * oc;
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int row0 = chunk * p.rows_per_chunk;
  const int row1 = min(row0 + p.rows_per_chunk, p.HW);
  float s[8], ss[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { s[i] = 0.f; ss[i] = 0.f; }
  const size_t base = (size_t)b * p.HW;
  int row = row0 + r;
  for (; row + 3 * R < row1; row += 4 * R) {
    u32x4 v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = *(const u32x4*)gn_src(p, base + row + j * R, co * 8);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float f[8]; unpack_bf8(v[j], f);
#pragma unroll
      for (int i = 0; i < 8; ++i) { s[i] += f[i]; ss[i] += f[i] * f[i]; }
    }
  }
  for (; row < row1; row += R) {
    const u32x4 v = *(const u32x4*)gn_src(p, base + row, co * 8);
    float f[8]; unpack_bf8(v, f);
#pragma unroll
    for (int i = 0; i < 8; ++i) { s[i] += f[i]; ss[i] += f[i] * f[i]; }
  }
  float* S = sm; float* SS = sm + R * p.C;
#pragma unroll
  for (int i = 0; i < 8; ++i) { S[r * p.C + co * 8 + i] = s[i]; SS[r * p.C + co * 8 + i] = ss[i]; }
  __syncthreads();
  float* CS = sm + 2 * R * p.C; float* CSS = CS + p.C;
  for (int c = t; c < p.C; c += blockDim.x) {
    float a = 0.f, q = 0.f;
    for (int k = 0; k < R; ++k) { a += S[k * p.C + c]; q += SS[k * p.C + c]; }
    CS[c] = a; CSS[c] = q;
  }
  __syncthreads();
  const int cpg = p.C / p.groups;
  if (t < p.groups) {
    float a = 0.f, q = 0.f;
    for (int k = 0; k < cpg; ++k) { a += CS[t * cpg + k]; q += CSS[t * cpg + k]; }
    float* o = p.partial + (((size_t)b * p.nchunk + chunk) * p.groups + t) * 2;
    o[0] = a; o[1] = q;
  }
}

// apply: prologue folds the <=256 partials of the block's sample in a fixed order (deterministic; all loads of a
// thread are independent and issued together), derives each thread's 8 channel affines a = rstd*gamma,
// b = beta - mean*a into registers, then streams rows: y = x*a + b (+SiLU), 4 rows in flight per thread.
__global__ __launch_bounds__(1024) void dmx_gn_apply_kernel(const GroupNormArgs p, int rows_per_block) {
  __shared__ float red[2][16][64];
  __shared__ float MEAN[64], RSTD[64];
  const int oc = p.C >> 3;
  const int R = blockDim.x / oc;
  const int t = threadIdx.x;
  const int r = t / oc, co = t - r * oc;
  const int b = blockIdx.y;
  const int G = p.groups;
  if (p.st0) {
    // statistics from the PRODUCER of the tensor(s): per (sample, channel) fixed-point (sum * 2^20, sumsq * 2^32) written by the
    // epilogue of the GEMM that produced x0 / x1 (gemm.hip publish_colstats) - no statistics pass over the tensor at all
    // (DmxStat records, common.h).  The group reduction and the variance run in double: the integer records are exact, and
    // E[x^2] - mean^2 in fp32 loses the variance of channels with a large DC offset (advisor, round 3)
    extern __shared__ double csum[];                   // [C] sums, [C] sums of squares
    for (int c = t; c < p.C; c += blockDim.x) {
      const long long* q = (c < p.c0) ? p.st0 + ((size_t)b * p.c0 + c) * DMX_STAT_WORDS : p.st1 + ((size_t)b * (p.C - p.c0) + (c - p.c0)) * DMX_STAT_WORDS;
      csum[c] = dmx_stat_sum(q[0]);
      csum[p.C + c] = dmx_stat_sumsq(q[1], q[2]);
    }
    __syncthreads();
    const int cpg = p.C / G;
    if (t < G) {
      double a = 0.0, q = 0.0;
      for (int k = 0; k < cpg; ++k) { a += csum[t * cpg + k]; q += csum[p.C + t * cpg + k]; }
      const double inv_n = 1.0 / ((double)p.HW * (double)cpg);
      const double mean = a * inv_n;
      double var = q * inv_n - mean * mean;
      var = var < 0.0 ? 0.0 : var;
      MEAN[t] = (float)mean; RSTD[t] = (float)(1.0 / __builtin_sqrt(var + (double)p.eps));
      if (p.stats_out && blockIdx.x == 0) { p.stats_out[((size_t)b * G + t) * 2] = MEAN[t]; p.stats_out[((size_t)b * G + t) * 2 + 1] = RSTD[t]; }
    }
    __syncthreads();
  } else {
    int slices = blockDim.x / G; if (slices > 16) slices = 16;
    const int g = t % G, sl = t / G;
    if (sl < slices) {
      float a = 0.f, q = 0.f;
      const float* pp = p.partial + ((size_t)b * p.nchunk * G + g) * 2;
      int k = sl;
      for (; k + 3 * slices < p.nchunk; k += 4 * slices) {
        float va[4], vq[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { va[j] = pp[(size_t)(k + j * slices) * G * 2]; vq[j] = pp[(size_t)(k + j * slices) * G * 2 + 1]; }
#pragma unroll
        for (int j = 0; j < 4; ++j) { a += va[j]; q += vq[j]; }
      }
      for (; k < p.nchunk; k += slices) { a += pp[(size_t)k * G * 2]; q += pp[(size_t)k * G * 2 + 1]; }
      red[0][sl][g] = a; red[1][sl][g] = q;
    }
    __syncthreads();
    const int cpg = p.C / G;
    if (t < G) {
      float a = 0.f, q = 0.f;
      for (int k2 = 0; k2 < slices; ++k2) { a += red[0][k2][t]; q += red[1][k2][t]; }
      const float inv_n = 1.0f / ((float)p.HW * (float)cpg);
      const float mean = a * inv_n;
      float var = q * inv_n - mean * mean;
      var = var < 0.f ? 0.f : var;
      MEAN[t] = mean; RSTD[t] = rsqrtf(var + p.eps);
      if (p.stats_out && blockIdx.x == 0) { p.stats_out[((size_t)b * G + t) * 2] = mean; p.stats_out[((size_t)b * G + t) * 2 + 1] = RSTD[t]; }
    }
    __syncthreads();
  }
  float A[8], Bv[8];
  {
    const int cpg = p.C / G;
    const f32x4 g0 = *(const f32x4*)(p.gamma + co * 8), g1 = *(const f32x4*)(p.gamma + co * 8 + 4);
    const f32x4 b0 = *(const f32x4*)(p.beta + co * 8), b1 = *(const f32x4*)(p.beta + co * 8 + 4);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int gg = (co * 8 + i) / cpg;
      const float gm = i < 4 ? g0[i] : g1[i - 4], bt = i < 4 ? b0[i] : b1[i - 4];
      A[i] = RSTD[gg] * gm; Bv[i] = bt - MEAN[gg] * A[i];
    }
  }
  const int row0 = blockIdx.x * rows_per_block;
  const int row1 = min(row0 + rows_per_block, p.HW);
  const size_t base = (size_t)b * p.HW;
  int row = row0 + r;
  for (; row + 3 * R < row1; row += 4 * R) {
    u32x4 v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = *(const u32x4*)gn_src(p, base + row + j * R, co * 8);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float f[8]; unpack_bf8(v[j], f);
#pragma unroll
      for (int i = 0; i < 8; ++i) { float y = f[i] * A[i] + Bv[i]; f[i] = p.silu ? silu_f(y) : y; }
      *(u32x4*)(p.y + (base + row + j * R) * p.ldy + co * 8) = pack_bf8(f);
    }
  }
  for (; row < row1; row += R) {
    const u32x4 v = *(const u32x4*)gn_src(p, base + row, co * 8);
    float f[8]; unpack_bf8(v, f);
#pragma unroll
    for (int i = 0; i < 8; ++i) { float y = f[i] * A[i] + Bv[i]; f[i] = p.silu ? silu_f(y) : y; }
    *(u32x4*)(p.y + (base + row) * p.ldy + co * 8) = pack_bf8(f);
  }
}

// ---- single-launch GroupNorm ("slab" kernel): one block owns one (sample, group) slab - HW pixels x cpg channels,
// 5..240 KB of bf16 - and keeps it in REGISTERS between the statistics and the normalisation: one read, one write,
// one launch, no partials buffer and no grid-wide dependency.  Thread = (pixel lane r, unit `within` of the pixel's
// cpg-channel segment); a unit is VEC dwords (2*VEC channels), so the per-thread channel affine sits in registers and
// the pixel stride R is a constant.  The variance is taken about the mean (second pass over the registers).
// Blocks are dealt to XCDs round-robin, so XCD x gets the groups [x*G/8, (x+1)*G/8): a contiguous channel band whose
// cache lines it shares with a neighbour only at the band edges.
template <int VEC> struct GnVec;
template <> struct GnVec<1> { typedef unsigned int T; };
template <> struct GnVec<2> { typedef u32x2 T; };
template <> struct GnVec<4> { typedef u32x4 T; };
template <int VEC> __device__ __forceinline__ unsigned int gn_dw(const typename GnVec<VEC>::T& v, int i) { return v[i]; }
template <> __device__ __forceinline__ unsigned int gn_dw<1>(const unsigned int& v, int) { return v; }
template <int VEC> __device__ __forceinline__ void gn_set(typename GnVec<VEC>::T& v, int i, unsigned int x) { v[i] = x; }
template <> __device__ __forceinline__ void gn_set<1>(unsigned int& v, int, unsigned int x) { v = x; }

__device__ __forceinline__ float gn_block_sum(float v, float* red, int t, int nw) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
  __syncthreads();                                   // `red` may still be read from the previous reduction
  if ((t & 63) == 0) red[t >> 6] = v;
  __syncthreads();
  float a = 0.f;
  for (int k = 0; k < nw; ++k) a += red[k];          // every thread folds the wave sums in the same fixed order
  return a;
}

template <int VEC, int NU, bool SILU, bool RED = false>
__global__ __launch_bounds__(1024) void dmx_gn_slab_kernel(const GroupNormArgs p, int SU, int R, int dbg) {
  typedef typename GnVec<VEC>::T V;
  typedef __attribute__((address_space(1))) V GV;   // the opaque running pointers lose their address space: say global again
  __shared__ float red[16];
  __shared__ float AF[2][128];
  const int t = threadIdx.x, nw = (blockDim.x + 63) >> 6;
  const int G = p.groups;
  int b, g;
  if ((G & 7) == 0) { const int gpx = G >> 3, j = blockIdx.x >> 3; b = j / gpx; g = (blockIdx.x & 7) * gpx + j % gpx; }
  else { b = blockIdx.x / G; g = blockIdx.x - b * G; }
  const int cpg = p.C / G;
  const int within = t % SU, r = t / SU;
  const bool active = r < R;
  const int c = g * cpg + within * 2 * VEC;          // first channel of this thread's unit
  const bf16* src; int ld;
  if (c < p.c0) { src = p.x0 + c; ld = p.ldx0; } else { src = p.x1 + (c - p.c0); ld = p.ldx1; }
  src += (size_t)b * p.HW * ld;
  V v[NU];
  float s = 0.f;
  // Branch-free on purpose: a conditional load makes the compiler carry the whole register-resident slab through a
  // phi per unit (and spill it).  Out-of-range units re-read the slab's first pixel and are masked in the arithmetic.
  // One running pointer per direction, stepped by a constant stride and made opaque to the optimiser: otherwise it
  // materialises all NU 64-bit addresses up front.
  const char* const safe = (const char*)src;
  const char* lp = (const char*)(src + (size_t)r * ld);
  const size_t lstep = (size_t)R * ld * 2;
  bool red_done = false;
  if constexpr (RED) {
    // x0 is the output of a split-K GEMM whose reduce pass was deferred to this kernel (GroupNormArgs.red_*): this thread's units of x0 are
    // SUMMED here - plane 0, + plane 1, ... in split order, + bias + row bias, + residual: the arithmetic of dmx_splitk_reduce_kernel, bit for
    // bit -, rounded, written to x0 and kept.  A group lies in x0 or in x1 as a whole (launcher), so the branch is block-uniform.
    if (p.red_partial && c < p.c0) {
      red_done = true;
      constexpr int NE = 2 * VEC;                      // channels of a unit
#pragma unroll
      for (int k = 0; k < NU; ++k) {
        const int pr = r + k * R;
        const bool ok = active && pr < p.HW;
        V o;
#pragma unroll
        for (int i = 0; i < VEC; ++i) gn_set<VEC>(o, i, 0u);
        if (ok) {
          const size_t m = (size_t)b * p.HW + pr;
          const float* q = p.red_partial + m * (size_t)p.c0 + c;
          // epilogue operands first (their latency overlaps the partial loads), then the planes FOUR at a time - all loads of a batch in flight
          // before the first add - summed in split order: the pass is latency-bound
          float bvv[NE], rbv[NE]; unsigned int rw[VEC];
#pragma unroll
          for (int e = 0; e < NE; e += 2) {
            const f32x2 t2 = p.red_bias ? *(const f32x2*)(p.red_bias + c + e) : (f32x2){0.f, 0.f};
            const f32x2 t3 = p.red_rowbias ? *(const f32x2*)(p.red_rowbias + (size_t)(m / p.red_rpg) * p.red_ldrb + c + e) : (f32x2){0.f, 0.f};
            bvv[e] = t2.x; bvv[e + 1] = t2.y; rbv[e] = t3.x; rbv[e + 1] = t3.y;
          }
#pragma unroll
          for (int i = 0; i < VEC; ++i) rw[i] = p.red_res ? *(const unsigned int*)(p.red_res + m * (size_t)p.red_ldres + c + 2 * i) : 0u;
          float a[NE];
#pragma unroll
          for (int e = 0; e < NE; e += 2) { const f32x2 t2 = *(const f32x2*)(q + e); a[e] = t2.x; a[e + 1] = t2.y; }
          int j = 1;
          for (; j + 3 < p.red_splitk; j += 4) {
            f32x2 tt[4][NE / 2];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
              for (int e = 0; e < NE / 2; ++e) tt[u][e] = *(const f32x2*)(q + (size_t)(j + u) * (size_t)p.red_mn + 2 * e);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
              for (int e = 0; e < NE / 2; ++e) { a[2 * e] += tt[u][e].x; a[2 * e + 1] += tt[u][e].y; }
          }
          for (; j < p.red_splitk; ++j) {
            const float* qj = q + (size_t)j * (size_t)p.red_mn;
#pragma unroll
            for (int e = 0; e < NE; e += 2) { const f32x2 t2 = *(const f32x2*)(qj + e); a[e] += t2.x; a[e + 1] += t2.y; }
          }
#pragma unroll
          for (int e = 0; e < NE; ++e) a[e] = a[e] + bvv[e] + rbv[e];
          if (p.red_res) {
#pragma unroll
            for (int i = 0; i < VEC; ++i) { a[2 * i] += h2f_lo(rw[i]); a[2 * i + 1] += h2f_hi(rw[i]); }
          }
#pragma unroll
          for (int i = 0; i < VEC; ++i) gn_set<VEC>(o, i, pack_bf2(a[2 * i], a[2 * i + 1]));
          *(GV*)(p.x0 + m * (size_t)p.ldx0 + c) = o;
        }
        v[k] = o;
      }
    }
  }
  if (!red_done) {
#pragma unroll
  for (int k = 0; k < NU; ++k) {
    const bool ok = active && (r + k * R < p.HW);
    asm volatile("" : "+v"(lp));
    v[k] = *(const GV*)((ok && !(dbg & 2)) ? lp : safe);
    lp += lstep;
    __builtin_amdgcn_sched_barrier(0);               // issue each load before forming the next address
  }
  }
#pragma unroll
  for (int k = 0; k < NU; ++k) {
    const bool ok = active && (r + k * R < p.HW);
    float u = 0.f;
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      const unsigned int w = gn_dw<VEC>(v[k], i);
      u += h2f_lo(w) + h2f_hi(w);
    }
    s += ok ? u : 0.f;
  }
  const float inv_n = 1.0f / ((float)p.HW * (float)cpg);
  const float mean = gn_block_sum(s, red, t, nw) * inv_n;
  float q = 0.f;
#pragma unroll
  for (int k = 0; k < NU; ++k) {
    const bool ok = active && (r + k * R < p.HW);
    float u = 0.f;
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      const unsigned int w = gn_dw<VEC>(v[k], i);
      const float d0 = h2f_lo(w) - mean, d1 = h2f_hi(w) - mean;
      u += d0 * d0 + d1 * d1;
    }
    q += ok ? u : 0.f;
  }
  const float var = gn_block_sum(q, red, t, nw) * inv_n;
  const float rstd = rsqrtf(var + p.eps);
  if (t < cpg) {
    const float a = rstd * p.gamma[g * cpg + t];
    AF[0][t] = a; AF[1][t] = p.beta[g * cpg + t] - mean * a;
  }
  if (t == 0 && p.stats_out) { p.stats_out[((size_t)b * G + g) * 2] = mean; p.stats_out[((size_t)b * G + g) * 2 + 1] = rstd; }
  __syncthreads();
  float A[2 * VEC], Bv[2 * VEC];
#pragma unroll
  for (int e = 0; e < 2 * VEC; ++e) { A[e] = AF[0][within * 2 * VEC + e]; Bv[e] = AF[1][within * 2 * VEC + e]; }
  float A2[2 * VEC], B2[2 * VEC];
#pragma unroll
  for (int e = 0; e < 2 * VEC; ++e) { A2[e] = -1.44269504088896f * A[e]; B2[e] = -1.44269504088896f * Bv[e]; }
  const size_t sstep = (size_t)R * p.ldy * 2;
  {
    char* sp = (char*)(p.y + ((size_t)b * p.HW + r) * p.ldy + c);
#pragma unroll
    for (int k = 0; k < NU; ++k) {
      if (k * R >= p.HW) break;                        // uniform: the instance may hold more units than this shape has
      const bool ok = active && (r + k * R < p.HW);
      asm volatile("" : "+v"(sp));
      V o;
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        const unsigned int w = gn_dw<VEC>(v[k], i);
        const float x0 = h2f_lo(w), x1 = h2f_hi(w);
        float y0 = __builtin_fmaf(x0, A[2 * i], Bv[2 * i]);
        float y1 = __builtin_fmaf(x1, A[2 * i + 1], Bv[2 * i + 1]);
        if (SILU) {
          // y * sigmoid(y) = y * rcp(1 + 2^(-y*log2(e))); the exponent comes from its own fma with pre-scaled affine
          // terms and v_rcp_f32 (1 ulp) replaces an IEEE division: the kernel is VALU-bound and the result is
          // rounded to bf16 anyway
          y0 *= __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(__builtin_fmaf(x0, A2[2 * i], B2[2 * i])));
          y1 *= __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(__builtin_fmaf(x1, A2[2 * i + 1], B2[2 * i + 1])));
        }
        gn_set<VEC>(o, i, pack_bf2(y0, y1));
      }
      if (ok && !(dbg & 1)) *(GV*)sp = o;
      sp += sstep;
      __builtin_amdgcn_sched_barrier(0);             // finish a unit before starting the next: bounded live ranges
    }
  }
}

// picks the slab instance for (C, groups, HW), or returns false (shape outside the register budget -> two-pass path)
static bool gn_slab_launch(const GroupNormArgs& a, hipStream_t stream, bool dry = false) {
  const int cpg = a.C / a.groups;
  if (cpg & 1 || cpg > 128 || a.B * a.groups > 65535) return false;
  const int vec = (cpg % 8 == 0) ? 4 : (cpg % 4 == 0) ? 2 : 1;
  if (a.c0 % (2 * vec) || a.ldx0 % (2 * vec) || a.ldx1 % (2 * vec) || a.ldy % (2 * vec)) return false;
  // measured (B=4, MI355X): thin segments (< 40 B per pixel: dword accesses, ~13 cache lines per wave instruction) over
  // many pixels are slower than the two coalesced streaming passes; everything else gains 3-8 us per GroupNorm
  if (cpg * 2 < 40 && a.HW > 1024) return false;
  const int SU = cpg / (2 * vec);
  int R = 1024 / SU; if (R > a.HW) R = a.HW;
  const int nu = cdiv(a.HW, R);
  const int threads = cdiv(R * SU, 64) * 64;
  const dim3 grid(a.B * a.groups), block(threads);
#ifdef DMX_PROBES      // probe builds only (results invalid): 1 = no stores, 2 = no loads
  static const int dbg = getenv("DMX_GN_SLAB_DBG") ? atoi(getenv("DMX_GN_SLAB_DBG")) : 0;
#else
  constexpr int dbg = 0;
#endif
  // (the fused split-K reduce, GroupNormArgs.red_partial: instances of up to four units, where the deep levels' slabs live)
#define GN_SLAB(V_, N_)                                                                                      \
  if (vec == V_ && nu <= N_) {                                                                              \
    if (a.red_partial) {                                                                                    \
      if constexpr (N_ <= 4 && V_ >= 2) {                                                                   \
        if (dry) return true;                                                                               \
        if (a.silu) hipLaunchKernelGGL((dmx_gn_slab_kernel<V_, N_, true, true>), grid, block, 0, stream, a, SU, R, dbg);   \
        else hipLaunchKernelGGL((dmx_gn_slab_kernel<V_, N_, false, true>), grid, block, 0, stream, a, SU, R, dbg);         \
        return true;                                                                                        \
      } else return false;                                                                                  \
    }                                                                                                       \
    if (dry) return true;                                                                                   \
    if (a.silu) hipLaunchKernelGGL((dmx_gn_slab_kernel<V_, N_, true>), grid, block, 0, stream, a, SU, R, dbg);    \
    else hipLaunchKernelGGL((dmx_gn_slab_kernel<V_, N_, false>), grid, block, 0, stream, a, SU, R, dbg);          \
    return true;                                                                                            \
  }
  // (instances whose slab would spill - 64 x 1, 44 x 2, 16 x 4 dwords per thread - are not built: those shapes take the two-launch path)
  // (an instance walks all N_ units, masked: keep the ladder tight - the 32x32-level C = 640 slabs have 6 units and ran on the
  // 16-unit instance at 11.9 us; on the 8-unit one: see DESIGN.md 5b)
  GN_SLAB(1, 8) GN_SLAB(1, 16) GN_SLAB(1, 24)
  GN_SLAB(2, 2) GN_SLAB(2, 4) GN_SLAB(2, 8) GN_SLAB(2, 16)
  GN_SLAB(4, 1) GN_SLAB(4, 2) GN_SLAB(4, 4) GN_SLAB(4, 8)
#undef GN_SLAB
  return false;
}

static bool gn_two_pass_forced() {
#ifdef DMX_PROBES
  static const bool off = getenv("DMX_GN_TWO_PASS") != nullptr;          // measurement aid: force the two-launch path
  return off;
#else
  return false;
#endif
}
// true when dmx_groupnorm_launch will take the single-launch path for this shape (1 read + 1 write of the tensor)
bool dmx_gn_single_launch(GroupNormArgs a) {
  if (a.c0 >= a.C || a.x1 == nullptr) { a.c0 = a.C; a.x1 = a.x0; a.ldx1 = a.ldx0; }
  return !gn_two_pass_forced() && a.C % a.groups == 0 && gn_slab_launch(a, nullptr, true);
}

// the slab path takes this shape with the split-K reduce of x0 fused in: a slab instance of <= 4 units with >= 4-channel units exists, every group
// lies in ONE source, x0 is dense over its c0 channels (the partial planes are [M][c0])
bool dmx_gn_red_ok(GroupNormArgs a) {
  if (!a.red_partial || a.red_splitk < 2 || a.C % a.groups) return false;
  if (a.c0 >= a.C || a.x1 == nullptr) { a.c0 = a.C; a.x1 = a.x0; a.ldx1 = a.ldx0; }
  const int cpg = a.C / a.groups;
  if (a.c0 % cpg || a.c0 % 8 || a.red_rpg <= 0) return false;
  return !gn_two_pass_forced() && gn_slab_launch(a, nullptr, true);
}

#define GN_MAX_CHUNKS 256
size_t dmx_gn_workspace_bytes(int B, int HW, int groups) {
  // partials [B][<=256 chunks][groups][2] + per-channel affine [B][C<=2560][2]
  (void)HW;
  return (size_t)B * GN_MAX_CHUNKS * groups * 2 * sizeof(float) + (size_t)B * 2560 * 2 * sizeof(float);
}

// GroupNorm whose statistics come from the producers of x0 / x1 (GroupNormArgs.st0 / st1): ONE apply-only launch over all CUs
int dmx_groupnorm_sums_launch(GroupNormArgs a, hipStream_t stream) {
  DMX_REQUIRE(a.C % 8 == 0 && a.C % a.groups == 0 && a.C <= 2560 && a.groups <= 64, "groupnorm: C=%d / groups=%d unsupported", a.C, a.groups);
  DMX_REQUIRE(a.c0 % 8 == 0 && a.ldx0 % 8 == 0 && a.ldy % 8 == 0, "groupnorm: strides/splits must be multiples of 8");
  if (a.c0 >= a.C || a.x1 == nullptr) { a.c0 = a.C; a.x1 = a.x0; a.ldx1 = a.ldx0; a.st1 = a.st0; }
  DMX_REQUIRE(a.st0 != nullptr && a.st1 != nullptr, "groupnorm: producer statistics missing");
  const int oc = a.C / 8;
  int R = 1024 / oc; if (R > 16) R = 16; if (R < 1) R = 1;
  const int threads = oc * R;
  int rpb = 8 * R;                                                  // up to 8 rows per thread
  while (rpb > R && (long)cdiv(a.HW, rpb) * a.B < 512) rpb -= R;
  a.partial = nullptr; a.nchunk = 0;
  hipLaunchKernelGGL(dmx_gn_apply_kernel, dim3(cdiv(a.HW, rpb), a.B), dim3(threads), (size_t)2 * a.C * sizeof(double), stream, a, rpb);
  return dmx_check_launch("dmx_gn_apply_kernel");
}

int dmx_groupnorm_launch(GroupNormArgs a, hipStream_t stream) {
  DMX_REQUIRE(a.C % 8 == 0 && a.C % a.groups == 0, "groupnorm: C=%d must be a multiple of 8 and of groups=%d", a.C, a.groups);
  DMX_REQUIRE(a.C <= 2560 && a.groups <= 64, "groupnorm: C=%d > 2560 or groups > 64 unsupported", a.C);
  DMX_REQUIRE(a.c0 % 8 == 0 && a.ldx0 % 8 == 0 && a.ldy % 8 == 0, "groupnorm: strides/splits must be multiples of 8");
  DMX_REQUIRE(a.partial != nullptr, "groupnorm: partial workspace is null");
  if (a.c0 >= a.C || a.x1 == nullptr) { a.c0 = a.C; a.x1 = a.x0; a.ldx1 = a.ldx0; }
  a.st0 = a.st1 = nullptr;
  if (!gn_two_pass_forced() && gn_slab_launch(a, stream)) return dmx_check_launch("dmx_gn_slab_kernel");
  DMX_REQUIRE(a.red_partial == nullptr, "groupnorm: the fused split-K reduce needs the slab path (ask dmx_gn_red_ok first)");
  // ---- two-launch path (slabs that do not fit the register budget, e.g. 1024-px images)
  // thread = (row lane r < R, channel octet); wide blocks so each thread walks only a few rows
  const int oc = a.C / 8;
  int R = 1024 / oc; if (R > 16) R = 16; if (R < 1) R = 1;
  const int threads = oc * R;
  int rpc = 4 * R;                                                  // rows per stats block: 4 per thread
  if (cdiv(a.HW, rpc) > GN_MAX_CHUNKS) rpc = cdiv(cdiv(a.HW, GN_MAX_CHUNKS), R) * R;
  a.rows_per_chunk = rpc;
  a.nchunk = cdiv(a.HW, rpc);
  a.coef = (float*)((char*)a.partial + (size_t)a.B * GN_MAX_CHUNKS * a.groups * 2 * sizeof(float));
  const size_t lds_stats = (size_t)(2 * R * a.C + 2 * a.C) * sizeof(float);
  DMX_LDS_OPT_IN((dmx_gn_stats_kernel), 128 * 1024);
  hipLaunchKernelGGL(dmx_gn_stats_kernel, dim3(a.nchunk, a.B), dim3(threads), lds_stats, stream, a);
  int rc = dmx_check_launch("dmx_gn_stats_kernel");
  if (rc) return rc;
  int rpb = 8 * R;                                                  // up to 8 rows per thread in apply
  while (rpb > R && (long)cdiv(a.HW, rpb) * a.B < 512) rpb -= R;
  hipLaunchKernelGGL(dmx_gn_apply_kernel, dim3(cdiv(a.HW, rpb), a.B), dim3(threads), 0, stream, a, rpb);
  return dmx_check_launch("dmx_gn_apply_kernel");
}

// ---------------------------------------------------------------------------- LayerNorm
// One wave per row, LN_ROWS rows per wave with all their loads issued up front (memory-level parallelism);
// a row lives in registers (<= 4 octets per lane, C <= 2048).
#define LN_ROWS 1
__global__ __launch_bounds__(256) void dmx_layernorm_kernel(const bf16* x, int ldx, bf16* y, int ldy,
                                                            const float* gamma, const float* beta,
                                                            int rows, int C, float eps) {
  const int lane = threadIdx.x & 63;
  const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * LN_ROWS;
  if (row0 >= rows) return;
  const int oc = C >> 3;
  const int nj = (oc + 63) >> 6;              // octets per lane (1..4)
  u32x4 raw[LN_ROWS][4];
#pragma unroll
  for (int r = 0; r < LN_ROWS; ++r) {
    int row = row0 + r; if (row >= rows) row = rows - 1;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int o = lane + 64 * j;
      if (j < nj && o < oc) raw[r][j] = *(const u32x4*)(x + (size_t)row * ldx + o * 8);
    }
  }
#pragma unroll
  for (int r = 0; r < LN_ROWS; ++r) {
    const int row = row0 + r;
    float f[4][8];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int o = lane + 64 * j;
      if (j < nj && o < oc) {
        unpack_bf8(raw[r][j], f[j]);
#pragma unroll
        for (int i = 0; i < 8; ++i) s += f[j][i];
      }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d);
    const float mean = s / (float)C;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int o = lane + 64 * j;
      if (j < nj && o < oc) {
#pragma unroll
        for (int i = 0; i < 8; ++i) { const float d = f[j][i] - mean; q += d * d; }
      }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) q += __shfl_xor(q, d);
    const float rstd = rsqrtf(q / (float)C + eps);
    if (row < rows) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int o = lane + 64 * j;
        if (j < nj && o < oc) {
          float g[8];
          const f32x4 g0 = *(const f32x4*)(gamma + o * 8), g1 = *(const f32x4*)(gamma + o * 8 + 4);
          const f32x4 b0 = *(const f32x4*)(beta + o * 8), b1 = *(const f32x4*)(beta + o * 8 + 4);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            g[i] = (f[j][i] - mean) * rstd * g0[i] + b0[i];
            g[4 + i] = (f[j][4 + i] - mean) * rstd * g1[i] + b1[i];
          }
          *(u32x4*)(y + (size_t)row * ldy + o * 8) = pack_bf8(g);
        }
      }
    }
  }
}

int dmx_layernorm_launch(const bf16* x, int ldx, bf16* y, int ldy, const float* gamma, const float* beta,
                         int rows, int C, float eps, hipStream_t stream) {
  DMX_REQUIRE(C % 8 == 0 && C <= 2048, "layernorm: C=%d must be a multiple of 8 and <= 2048", C);
  DMX_REQUIRE(ldx % 8 == 0 && ldy % 8 == 0, "layernorm: strides must be multiples of 8");
  hipLaunchKernelGGL(dmx_layernorm_kernel, dim3(cdiv(rows, 4 * LN_ROWS)), dim3(256), 0, stream, x, ldx, y, ldy, gamma, beta, rows, C, eps);
  return dmx_check_launch("dmx_layernorm_kernel");
}

// ---------------------------------------------------------------------------- row softmax
// fp32 scores [rows][n] -> bf16 probabilities, softmax(scale * s) along the row (VAE mid-block
// attention, single head d=512, SURVEY.md 8a K6b: "softmax in fp32").  One block per row.
__global__ __launch_bounds__(256) void dmx_softmax_rows_kernel(const float* s, int lds_, bf16* p, int ldp, int n, float scale) {
  __shared__ float red[8];
  const int row = blockIdx.x, t = threadIdx.x;
  const float* sr = s + (size_t)row * lds_;
  const float sl2 = scale * 1.4426950408889634f;
  float mx = -INFINITY;
  for (int i = t; i < n; i += 256) mx = fmaxf(mx, sr[i]);
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) mx = fmaxf(mx, __shfl_xor(mx, d));
  if ((t & 63) == 0) red[t >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float sum = 0.f;
  for (int i = t; i < n; i += 256) sum += __builtin_amdgcn_exp2f((sr[i] - mx) * sl2);
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) sum += __shfl_xor(sum, d);
  if ((t & 63) == 0) red[4 + (t >> 6)] = sum;
  __syncthreads();
  sum = (red[4] + red[5]) + (red[6] + red[7]);
  const float inv = 1.0f / sum;
  unsigned short* pr = (unsigned short*)p + (size_t)row * ldp;
  for (int i = t; i < n; i += 256) pr[i] = f2bf_bits(__builtin_amdgcn_exp2f((sr[i] - mx) * sl2) * inv);
}
int dmx_softmax_rows_launch(const float* s, int lds_, bf16* p, int ldp, int rows, int n, float scale, hipStream_t stream) {
  hipLaunchKernelGGL(dmx_softmax_rows_kernel, dim3(rows), dim3(256), 0, stream, s, lds_, p, ldp, n, scale);
  return dmx_check_launch("dmx_softmax_rows_kernel");
}
