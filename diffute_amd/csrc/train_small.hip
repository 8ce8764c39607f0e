// Small helpers of the training path: bf16 transposes (the data-gradient GEMMs read W^T),
// elementwise gradient adds, the MSE loss (train_diffute_v1.py:918) with its gradient, and the backward of the tiny
// fp32 time-embedding linears (M = batch).  Everything deterministic.
#include "common.h"
#include "kernels.h"
#include <string.h>

namespace {
// out[c][r] = in[r][c], 32x32 tiles through LDS (both sides coalesced)
__global__ __launch_bounds__(256) void dmx_transpose_bf16_kernel(const unsigned short* in, int ldin, unsigned short* out, int ldout, int R, int C) {
  __shared__ unsigned short tile[32][34];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int r = r0 + ty + 8 * j, c = c0 + tx;
    tile[ty + 8 * j][tx] = (r < R && c < C) ? in[(size_t)r * ldin + c] : (unsigned short)0;
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int c = c0 + ty + 8 * j, r = r0 + tx;
    if (c < C && r < R) out[(size_t)c * ldout + r] = tile[tx][ty + 8 * j];
  }
}

__global__ __launch_bounds__(256) void dmx_add_bf16_kernel(const bf16* a, int lda, const bf16* b, int ldb, bf16* o, int ldo, int rows, int C) {
  const int c8 = C / 8;
  const size_t total = (size_t)rows * c8;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % c8) * 8; const size_t r = i / c8;
    float x[8], y[8]; unpack_bf8(*(const u32x4*)(a + r * lda + c), x); unpack_bf8(*(const u32x4*)(b + r * ldb + c), y);
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] += y[e];
    *(u32x4*)(o + r * ldo + c) = pack_bf8(x);
  }
}

// loss = mean((p - t)^2) ; dp = 2 (p - t) / n * gscale.  Stage 1: per-block partial sums; stage 2: one block, fixed order.
__global__ __launch_bounds__(256) void dmx_mse_part_kernel(const float* p, const float* t, float* dp, float* part, size_t n, float gs) {
  __shared__ float red[256];
  float s = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float d = p[i] - t[i];
    s += d * d;
    if (dp) dp[i] = d * gs;
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int k = 128; k >= 1; k >>= 1) { if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k]; __syncthreads(); }
  if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}
__global__ __launch_bounds__(256) void dmx_mse_final_kernel(const float* part, int nb, float* loss, float inv_n) {
  __shared__ float red[256];
  float s = 0.f;
  for (int i = threadIdx.x; i < nb; i += 256) s += part[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int k = 128; k >= 1; k >>= 1) { if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k]; __syncthreads(); }
  if (threadIdx.x == 0) *loss = red[0] * inv_n;
}

__device__ __forceinline__ float silu1(float x) { return x / (1.0f + __expf(-x)); }
__device__ __forceinline__ float dsilu1(float u) { const float s = 1.0f / (1.0f + __expf(-u)); return s * (1.0f + u * (1.0f - s)); }
// y = W * act(x) + b  (act = SiLU when silu_in):  dW[n][k] (+)= sum_b dy[b][n] act(x[b][k]) ; db[n] (+)= sum_b dy[b][n]
__global__ __launch_bounds__(256) void dmx_linear_small_bwd_w_kernel(const float* x, int ldx, const float* dy, int lddy, float* dw, int lddw,
                                                                     float* db, int db_stride, int B, int N, int K, int silu_in, int accumulate) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x, n = blockIdx.y;
  if (k < K) {
    float s = 0.f;
    for (int b = 0; b < B; ++b) { const float v = x[(size_t)b * ldx + k]; s += dy[(size_t)b * lddy + n] * (silu_in ? silu1(v) : v); }
    float* o = dw + (size_t)n * lddw + k;
    *o = accumulate ? *o + s : s;
  }
  if (db && blockIdx.x == 0 && threadIdx.x == 0) {
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += dy[(size_t)b * lddy + n];
    db[(size_t)n * db_stride] = accumulate ? db[(size_t)n * db_stride] + s : s;
  }
}
// dx[b][k] = act'(x[b][k]) * sum_n dy[b][n] W[n][k].  N is the long axis here (time_emb_proj of all 22 resnets stacked:
// N ~ 20k rows of K = 1280), so a block owns only 8 columns k (and up to 8 samples) and splits N over 128 lanes of
// threads: 160 blocks at K = 1280 (32 columns per block left 40 blocks streaming the 52 MB of W: 717 us per call); W is
// read once in 16-byte row segments, the dy values are block-uniform, the 128 partial sums are folded in lane order.
__global__ __launch_bounds__(1024) void dmx_linear_small_bwd_x_kernel(const float* dy, int lddy, const bf16* w, int ldw, const float* x, int ldx,
                                                                      float* dx, int lddx, int B, int N, int K, int silu_in) {
  __shared__ float red[128][8][9];
  const int kk = threadIdx.x & 7, nl = threadIdx.x >> 3;
  const int k = blockIdx.x * 8 + kk, b0 = blockIdx.y * 8;
  const int nb = min(8, B - b0);
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
  if (k < K)
    for (int n = nl; n < N; n += 128) {
      const float wv = (float)w[(size_t)n * ldw + k];
#pragma unroll
      for (int j = 0; j < 8; ++j) if (j < nb) acc[j] += dy[(size_t)(b0 + j) * lddy + n] * wv;
    }
#pragma unroll
  for (int j = 0; j < 8; ++j) red[nl][j][kk] = acc[j];
  __syncthreads();
  if (threadIdx.x < 64) {
    const int j = threadIdx.x >> 3;
    if (j < nb && k < K) {
      float s = 0.f;
      for (int l = 0; l < 128; ++l) s += red[l][j][kk];
      if (silu_in) s *= dsilu1(x[(size_t)(b0 + j) * ldx + k]);
      dx[(size_t)(b0 + j) * lddx + k] = s;
    }
  }
}
// ---- 1x1 convolutions between tiny channel counts (VAE quant_conv 8->8, post_quant_conv 4->4): one thread per row
#define PW_MAXC 8
__global__ __launch_bounds__(256) void dmx_pointwise_small_fwd_kernel(const bf16* x, int ldx, const bf16* w, int ldw, const float* bias,
                                                                      void* y, int ldy, int M, int Cin, int Cout, int out_f32) {
  __shared__ float ws[PW_MAXC * PW_MAXC + PW_MAXC];
  if ((int)threadIdx.x < Cout * Cin) ws[threadIdx.x] = (float)w[(threadIdx.x / Cin) * ldw + threadIdx.x % Cin];
  if ((int)threadIdx.x < Cout) ws[PW_MAXC * PW_MAXC + threadIdx.x] = bias ? bias[threadIdx.x] : 0.f;
  __syncthreads();
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M) return;
  float xv[PW_MAXC];
  for (int i = 0; i < Cin; ++i) xv[i] = (float)x[(size_t)m * ldx + i];
  for (int o = 0; o < Cout; ++o) {
    float a = ws[PW_MAXC * PW_MAXC + o];
    for (int i = 0; i < Cin; ++i) a += ws[o * Cin + i] * xv[i];
    if (out_f32) ((float*)y)[(size_t)m * ldy + o] = a;
    else ((unsigned short*)y)[(size_t)m * ldy + o] = f2bf_bits(a);
  }
}
// dX[m][i] = sum_o dy[m][o] W[o][i] ; per-block partials of dW[o][i] = sum_m dy[m][o] x[m][i] and db[o] = sum_m dy[m][o]
__global__ __launch_bounds__(256) void dmx_pointwise_small_bwd_kernel(const bf16* x, int ldx, const float* dy, int lddy, const bf16* w, int ldw,
                                                                      bf16* dx, int lddx, float* part, int M, int Cin, int Cout) {
  __shared__ float ws[PW_MAXC * PW_MAXC];
  __shared__ float red[256];
  if ((int)threadIdx.x < Cout * Cin) ws[threadIdx.x] = (float)w[(threadIdx.x / Cin) * ldw + threadIdx.x % Cin];
  __syncthreads();
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  float xv[PW_MAXC], dv[PW_MAXC];
  for (int i = 0; i < PW_MAXC; ++i) { xv[i] = 0.f; dv[i] = 0.f; }
  if (m < M) {
    for (int i = 0; i < Cin; ++i) xv[i] = (float)x[(size_t)m * ldx + i];
    for (int o = 0; o < Cout; ++o) dv[o] = dy[(size_t)m * lddy + o];
    if (dx)
      for (int i = 0; i < Cin; ++i) {
        float a = 0.f;
        for (int o = 0; o < Cout; ++o) a += dv[o] * ws[o * Cin + i];
        ((unsigned short*)dx)[(size_t)m * lddx + i] = f2bf_bits(a);
      }
  }
  const int nq = Cout * Cin + Cout;
  for (int q = 0; q < nq; ++q) {                 // fixed-order tree per quantity
    red[threadIdx.x] = q < Cout * Cin ? dv[q / Cin] * xv[q % Cin] : dv[q - Cout * Cin];
    __syncthreads();
    for (int k = 128; k >= 1; k >>= 1) { if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k]; __syncthreads(); }
    if (threadIdx.x == 0) part[(size_t)blockIdx.x * nq + q] = red[0];
    __syncthreads();
  }
}
__global__ void dmx_pointwise_small_final_kernel(const float* part, int nblk, int Cin, int Cout, float* dw, int lddw, float* db) {
  const int q = threadIdx.x, nq = Cout * Cin + Cout;
  if (q >= nq) return;
  float s = 0.f;
  for (int j = 0; j < nblk; ++j) s += part[(size_t)j * nq + q];
  if (q < Cout * Cin) dw[(q / Cin) * lddw + q % Cin] = s; else db[q - Cout * Cin] = s;
}

// softmax backward over rows: dS = scale * P o (dP - rowsum(dP o P)); one block per row
__global__ __launch_bounds__(256) void dmx_softmax_bwd_rows_kernel(const bf16* P, int ldp, const float* dP, int lddp, bf16* dS, int ldds, int n, float scale) {
  __shared__ float red[256];
  const size_t r = blockIdx.x;
  float s = 0.f;
  for (int c = threadIdx.x; c < n; c += 256) s += (float)P[r * ldp + c] * dP[r * lddp + c];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int k = 128; k >= 1; k >>= 1) { if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k]; __syncthreads(); }
  const float dot = red[0];
  for (int c = threadIdx.x; c < n; c += 256)
    ((unsigned short*)dS)[r * ldds + c] = f2bf_bits(scale * (float)P[r * ldp + c] * (dP[r * lddp + c] - dot));
}
}  // namespace

int dmx_pointwise_small_fwd_launch(const bf16* x, int ldx, const bf16* w, int ldw, const float* bias, void* y, int ldy, int M, int Cin, int Cout,
                                   int out_f32, hipStream_t stream) {
  DMX_REQUIRE(Cin <= PW_MAXC && Cout <= PW_MAXC && Cin > 0 && Cout > 0, "pointwise_small: Cin=%d Cout=%d must be <= 8", Cin, Cout);
  hipLaunchKernelGGL(dmx_pointwise_small_fwd_kernel, dim3(cdiv(M, 256)), dim3(256), 0, stream, x, ldx, w, ldw, bias, y, ldy, M, Cin, Cout, out_f32);
  return dmx_check_launch("dmx_pointwise_small_fwd_kernel");
}
size_t dmx_pointwise_small_bwd_ws_bytes(int M, int Cin, int Cout) { return (size_t)cdiv(M, 256) * (Cout * Cin + Cout) * sizeof(float); }
int dmx_pointwise_small_bwd_launch(const bf16* x, int ldx, const float* dy, int lddy, const bf16* w, int ldw, bf16* dx, int lddx,
                                   float* dw, int lddw, float* db, int M, int Cin, int Cout, void* workspace, size_t workspace_bytes, hipStream_t stream) {
  DMX_REQUIRE(Cin <= PW_MAXC && Cout <= PW_MAXC && Cin > 0 && Cout > 0, "pointwise_small_bwd: Cin=%d Cout=%d must be <= 8", Cin, Cout);
  DMX_REQUIRE(workspace && workspace_bytes >= dmx_pointwise_small_bwd_ws_bytes(M, Cin, Cout), "pointwise_small_bwd: workspace too small");
  const int nblk = cdiv(M, 256);
  hipLaunchKernelGGL(dmx_pointwise_small_bwd_kernel, dim3(nblk), dim3(256), 0, stream, x, ldx, dy, lddy, w, ldw, dx, lddx, (float*)workspace, M, Cin, Cout);
  int rc = dmx_check_launch("dmx_pointwise_small_bwd_kernel");
  if (rc) return rc;
  hipLaunchKernelGGL(dmx_pointwise_small_final_kernel, dim3(1), dim3(128), 0, stream, (const float*)workspace, nblk, Cin, Cout, dw, lddw, db);
  return dmx_check_launch("dmx_pointwise_small_final_kernel");
}
int dmx_softmax_bwd_rows_launch(const bf16* P, int ldp, const float* dP, int lddp, bf16* dS, int ldds, int rows, int n, float scale, hipStream_t stream) {
  hipLaunchKernelGGL(dmx_softmax_bwd_rows_kernel, dim3(rows), dim3(256), 0, stream, P, ldp, dP, lddp, dS, ldds, n, scale);
  return dmx_check_launch("dmx_softmax_bwd_rows_kernel");
}

// one block per 64x64 tile of one job; the job of a block is found by bisection over the jobs' first-tile indices.
// 16-byte accesses on both sides where the job's strides and bases allow it (every weight of the UNet except the 4-channel
// conv_out), element accesses on the edges: 1.7 GB in + 1.7 GB out per training step.
__global__ __launch_bounds__(256) void dmx_transpose_batch_kernel(const TrJob* jobs, int njobs) {
  __shared__ unsigned short tile[64][72];              // 144-byte rows
  int lo = 0, hi = njobs - 1;
  while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (jobs[mid].first <= (int)blockIdx.x) lo = mid; else hi = mid - 1; }
  const TrJob j = jobs[lo];
  const int tix = (int)blockIdx.x - j.first;
  const int r0 = (tix / j.tiles_x) * 64, c0 = (tix % j.tiles_x) * 64;
  const int t = threadIdx.x, sub = t & 7, line = t >> 3;            // 8 threads x 8 elements per 64-element line, 32 lines per pass
  const bool vin = ((j.ldin & 7) == 0) && (((size_t)j.in & 15) == 0);
  const bool vout = ((j.ldout & 7) == 0) && (((size_t)j.out & 15) == 0);
#pragma unroll
  for (int ps = 0; ps < 2; ++ps) {
    const int rr = line + 32 * ps, r = r0 + rr, c = c0 + sub * 8;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (r < j.R) {
      const unsigned short* src = j.in + (size_t)r * j.ldin + c;
      if (vin && c + 8 <= j.C) v = *(const u32x4*)src;
      else {
        unsigned short e[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) e[q] = (c + q < j.C) ? src[q] : (unsigned short)0;
        v = (u32x4){(unsigned)e[0] | ((unsigned)e[1] << 16), (unsigned)e[2] | ((unsigned)e[3] << 16), (unsigned)e[4] | ((unsigned)e[5] << 16), (unsigned)e[6] | ((unsigned)e[7] << 16)};
      }
    }
    *(u32x4*)&tile[rr][sub * 8] = v;
  }
  __syncthreads();
#pragma unroll
  for (int ps = 0; ps < 2; ++ps) {
    const int cc = line + 32 * ps, c = c0 + cc, r = r0 + sub * 8;     // output row c, 8 consecutive output columns r ..
    if (c >= j.C || r >= j.R) continue;
    unsigned short e[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) e[q] = tile[sub * 8 + q][cc];
    unsigned short* dst = j.out + (size_t)c * j.ldout + r;
    if (vout && r + 8 <= j.R) {
      *(u32x4*)dst = (u32x4){(unsigned)e[0] | ((unsigned)e[1] << 16), (unsigned)e[2] | ((unsigned)e[3] << 16), (unsigned)e[4] | ((unsigned)e[5] << 16), (unsigned)e[6] | ((unsigned)e[7] << 16)};
    } else {
#pragma unroll
      for (int q = 0; q < 8; ++q) if (r + q < j.R) dst[q] = e[q];
    }
  }
}
static thread_local TrBatch* g_tr_batch = nullptr;
TrBatch::TrBatch() { g_tr_batch = this; }
TrBatch::~TrBatch() { if (g_tr_batch == this) g_tr_batch = nullptr; }
int TrBatch::run(void* table_dev, size_t table_bytes, std::vector<TrJob>& cache, hipStream_t stream) {
  g_tr_batch = nullptr;
  if (jobs.empty()) return DMX_OK;
  DMX_REQUIRE(table_dev && jobs.size() * sizeof(TrJob) <= table_bytes, "transpose batch: %zu jobs do not fit the table", jobs.size());
  if (cache.size() != jobs.size() || memcmp(cache.data(), jobs.data(), jobs.size() * sizeof(TrJob)) != 0) {
    DMX_HIP(hipStreamSynchronize(stream));           // (a previous batch may still be reading the old table; pointers change ~never)
    DMX_HIP(hipMemcpy(table_dev, jobs.data(), jobs.size() * sizeof(TrJob), hipMemcpyHostToDevice));
    cache = jobs;
  }
  hipLaunchKernelGGL(dmx_transpose_batch_kernel, dim3(tiles), dim3(256), 0, stream, (const TrJob*)table_dev, (int)jobs.size());
  return dmx_check_launch("dmx_transpose_batch_kernel");
}

int dmx_transpose_bf16_launch(const bf16* in, int ldin, bf16* out, int ldout, int R, int C, hipStream_t stream) {
  if (g_tr_batch) {
    TrJob j; j.in = (const unsigned short*)in; j.out = (unsigned short*)out; j.ldin = ldin; j.ldout = ldout; j.R = R; j.C = C;
    j.tiles_x = cdiv(C, 64); j.first = g_tr_batch->tiles;
    g_tr_batch->tiles += j.tiles_x * cdiv(R, 64);
    g_tr_batch->jobs.push_back(j);
    return DMX_OK;
  }
  hipLaunchKernelGGL(dmx_transpose_bf16_kernel, dim3(cdiv(C, 32), cdiv(R, 32)), dim3(256), 0, stream,
                     (const unsigned short*)in, ldin, (unsigned short*)out, ldout, R, C);
  return dmx_check_launch("dmx_transpose_bf16_kernel");
}
int dmx_add_bf16_launch(const bf16* a, int lda, const bf16* b, int ldb, bf16* o, int ldo, int rows, int C, hipStream_t stream) {
  DMX_REQUIRE(C % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0 && ldo % 8 == 0, "add_bf16: C %% 8");
  const size_t total = (size_t)rows * (C / 8);
  int blocks = (int)((total + 255) / 256); if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(dmx_add_bf16_kernel, dim3(blocks), dim3(256), 0, stream, a, lda, b, ldb, o, ldo, rows, C);
  return dmx_check_launch("dmx_add_bf16_kernel");
}
size_t dmx_mse_workspace_bytes() { return 1024 * sizeof(float); }
int dmx_mse_loss_launch(const float* pred, const float* target, size_t n, float* loss, float* dpred, float grad_scale,
                        void* workspace, size_t workspace_bytes, hipStream_t stream) {
  DMX_REQUIRE(pred && target && loss && n > 0, "mse_loss: null argument");
  DMX_REQUIRE(workspace && workspace_bytes >= dmx_mse_workspace_bytes(), "mse_loss: workspace too small");
  int nb = (int)((n + 255) / 256); if (nb > 1024) nb = 1024;
  hipLaunchKernelGGL(dmx_mse_part_kernel, dim3(nb), dim3(256), 0, stream, pred, target, dpred, (float*)workspace, n, 2.0f * grad_scale / (float)n);
  int rc = dmx_check_launch("dmx_mse_part_kernel");
  if (rc) return rc;
  hipLaunchKernelGGL(dmx_mse_final_kernel, dim3(1), dim3(256), 0, stream, (const float*)workspace, nb, loss, 1.0f / (float)n);
  return dmx_check_launch("dmx_mse_final_kernel");
}
int dmx_linear_small_bwd_launch(const float* x, int ldx, const float* dy, int lddy, const bf16* w, int ldw,
                                float* dw, int lddw, float* db, int db_stride, float* dx, int lddx,
                                int B, int N, int K, int silu_in, int accumulate, hipStream_t stream) {
  if (dw) {
    hipLaunchKernelGGL(dmx_linear_small_bwd_w_kernel, dim3(cdiv(K, 256), N), dim3(256), 0, stream, x, ldx, dy, lddy, dw, lddw, db, db_stride, B, N, K, silu_in, accumulate);
    int rc = dmx_check_launch("dmx_linear_small_bwd_w_kernel");
    if (rc) return rc;
  }
  if (dx) {
    hipLaunchKernelGGL(dmx_linear_small_bwd_x_kernel, dim3(cdiv(K, 8), cdiv(B, 8)), dim3(1024), 0, stream, dy, lddy, w, ldw, x, ldx, dx, lddx, B, N, K, silu_in);
    return dmx_check_launch("dmx_linear_small_bwd_x_kernel");
  }
  return DMX_OK;
}
