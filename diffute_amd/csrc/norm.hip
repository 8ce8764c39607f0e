// GroupNorm(32 groups)[+SiLU] and LayerNorm for NHWC bf16 tensors.
//
// Both are HBM-bound: every load/store is 16 B per lane (8 bf16), statistics in fp32.
//
// GroupNorm is two launches and deterministic (no float atomics):
//   stats : grid (nchunk, B); each block reduces rows_per_chunk pixels x C channels to
//           per-group (sum, sumsq) partials  -> partial[b][chunk][g][2]
//   apply : grid (row blocks, B); a parallel prologue folds the <=256 partials of its sample in a
//           fixed order into mean/rstd, each thread derives its 8 channels' affine into registers,
//           then y = x*a + b (+SiLU), bf16 out, 4 rows in flight per thread.
// The input may be the virtual channel-concat of two tensors (UNet skip connections), so
// torch.cat([h, skip], 1) is never materialised.
#include "common.h"
#include "kernels.h"

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
  {
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

#define GN_MAX_CHUNKS 256
size_t dmx_gn_workspace_bytes(int B, int HW, int groups) {
  // partials [B][<=256 chunks][groups][2] + per-channel affine [B][C<=2560][2]
  (void)HW;
  return (size_t)B * GN_MAX_CHUNKS * groups * 2 * sizeof(float) + (size_t)B * 2560 * 2 * sizeof(float);
}

int dmx_groupnorm_launch(GroupNormArgs a, hipStream_t stream) {
  DMX_REQUIRE(a.C % 8 == 0 && a.C % a.groups == 0, "groupnorm: C=%d must be a multiple of 8 and of groups=%d", a.C, a.groups);
  DMX_REQUIRE(a.C <= 2560 && a.groups <= 64, "groupnorm: C=%d > 2560 or groups > 64 unsupported", a.C);
  DMX_REQUIRE(a.c0 % 8 == 0 && a.ldx0 % 8 == 0 && a.ldy % 8 == 0, "groupnorm: strides/splits must be multiples of 8");
  DMX_REQUIRE(a.partial != nullptr, "groupnorm: partial workspace is null");
  if (a.c0 >= a.C || a.x1 == nullptr) { a.c0 = a.C; a.x1 = a.x0; a.ldx1 = a.ldx0; }
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
  static bool attr = false;
  if (!attr) { (void)hipFuncSetAttribute((const void*)dmx_gn_stats_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024); attr = true; }
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
