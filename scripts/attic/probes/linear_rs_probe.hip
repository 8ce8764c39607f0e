// Feasibility probe for the K = C linears of the 32x32 / 16x16 / 8x8 levels: out[M][N] = X[M][K] W[N][K]^T with
//   * a 64 x 64 output tile per 8-wave block, the K range split over the waves INSIDE the block (intra-block split-K),
//   * operands streamed from L2 straight into MFMA fragment registers - no LDS, no barrier, no DMA in the K loop: every wave runs alone,
//   * an fp32 tree reduction of the partial tiles through LDS in a fixed order, then a one-item-per-thread epilogue (bias, bf16, 16-byte stores).
// hipcc --offload-arch=gfx950 -O3 -o linear_rs_probe linear_rs_probe.hip && ./linear_rs_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)
__device__ __forceinline__ unsigned pack2(float a, float b) { bf16x2 v = {(__bf16)a, (__bf16)b}; return __builtin_bit_cast(unsigned, v); }

// KS = K slices per tile (8: every wave the whole 64 x 64 tile; 4: two wave groups of 32 rows each); PD = prefetch distance in k16 steps
template <int KS, int PD>
__global__ __launch_bounds__(512, 2) void linear_rs(const __bf16* __restrict__ X, const __bf16* __restrict__ W, const float* __restrict__ bias, __bf16* __restrict__ out, int M, int N, int K) {
  constexpr int MB = KS == 8 ? 2 : 1;                  // 32-row m blocks per wave
  __shared__ __attribute__((aligned(16))) float red[4 * 64 * 64];        // 64 KB: four partial tiles in register order / the final row-major tile
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int lr = lane & 31, lh = lane >> 5;
  const int n0 = blockIdx.x * 64, m0 = blockIdx.y * 64;
  const int ks = wave % KS, mg = wave / KS;            // K slice, m group (KS = 4: rows 32 mg ..)
  const int kslice = K / KS, kbeg = ks * kslice, nstep = kslice / 16;
  // operand pointers: A = W rows (n), B = X rows (m); lane (lr, lh) reads 16 bytes at k = kbeg + 16 s + 8 lh
  const __bf16* wp[2]; const __bf16* xp[MB];
#pragma unroll
  for (int a = 0; a < 2; ++a) wp[a] = W + (size_t)(n0 + 32 * a + lr) * K + kbeg + 8 * lh;
#pragma unroll
  for (int b = 0; b < MB; ++b) xp[b] = X + (size_t)(m0 + 32 * (MB == 2 ? b : mg) + lr) * K + kbeg + 8 * lh;
  f32x16 acc[2][MB];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < MB; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
  bf16x8 wf[PD][2], xf[PD][MB];
#pragma unroll
  for (int s = 0; s < PD; ++s) {
#pragma unroll
    for (int a = 0; a < 2; ++a) wf[s][a] = *(const bf16x8*)(wp[a] + 16 * s);
#pragma unroll
    for (int b = 0; b < MB; ++b) xf[s][b] = *(const bf16x8*)(xp[b] + 16 * s);
  }
  for (int s0 = 0; s0 < nstep; s0 += PD) {
#pragma unroll
    for (int u = 0; u < PD; ++u) {
      const int s = s0 + u;
      bf16x8 wc[2], xc[MB];
#pragma unroll
      for (int a = 0; a < 2; ++a) wc[a] = wf[u][a];
#pragma unroll
      for (int b = 0; b < MB; ++b) xc[b] = xf[u][b];
      if (s + PD < nstep) {
#pragma unroll
        for (int a = 0; a < 2; ++a) wf[u][a] = *(const bf16x8*)(wp[a] + 16 * (s + PD));
#pragma unroll
        for (int b = 0; b < MB; ++b) xf[u][b] = *(const bf16x8*)(xp[b] + 16 * (s + PD));
      }
      if (s < nstep) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < MB; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wc[a], xc[b], acc[a][b], 0, 0, 0);
      }
    }
  }
  // ---- fixed-order tree over the K slices through LDS, register order (float4 groups x lanes: conflict-free, coalesced)
  constexpr int NG = 2 * MB * 4;                       // float4 groups per wave
  auto put = [&](int slot) {
    f32x4* d = (f32x4*)red + (size_t)slot * (NG * 64);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < MB; ++b)
#pragma unroll
        for (int g = 0; g < 4; ++g) d[((a * MB + b) * 4 + g) * 64 + lane] = (f32x4){acc[a][b][4 * g], acc[a][b][4 * g + 1], acc[a][b][4 * g + 2], acc[a][b][4 * g + 3]};
  };
  auto add = [&](int slot) {
    const f32x4* d = (const f32x4*)red + (size_t)slot * (NG * 64);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < MB; ++b)
#pragma unroll
        for (int g = 0; g < 4; ++g) { const f32x4 v = d[((a * MB + b) * 4 + g) * 64 + lane]; acc[a][b][4 * g] += v[0]; acc[a][b][4 * g + 1] += v[1]; acc[a][b][4 * g + 2] += v[2]; acc[a][b][4 * g + 3] += v[3]; }
  };
  // slices ks >= KS/2 park, ks < KS/2 add; then quarter, ...: (p0 + p[KS/2]) ... slot = mg * (KS/2) + (ks mod half)
  for (int half = KS / 2; half >= 1; half >>= 1) {
    if (ks >= half && ks < 2 * half) put(mg * half + (ks - half));
    __syncthreads();
    if (ks < half) add(mg * half + ks);
    __syncthreads();
  }
  // ---- final tile to LDS row-major [64 m][64 n + 4], then one (row, octet) item per thread
  constexpr int LDT = 68;
  if (ks == 0) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < MB; ++b)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *(f32x4*)(red + (32 * (MB == 2 ? b : mg) + lr) * LDT + 32 * a + 8 * g + 4 * lh) = (f32x4){acc[a][b][4 * g], acc[a][b][4 * g + 1], acc[a][b][4 * g + 2], acc[a][b][4 * g + 3]};
  }
  __syncthreads();
  {
    const int r = t >> 3, o = t & 7;
    const f32x4 v0 = *(const f32x4*)(red + r * LDT + 8 * o), v1 = *(const f32x4*)(red + r * LDT + 8 * o + 4);
    const f32x4 b0 = *(const f32x4*)(bias + n0 + 8 * o), b1 = *(const f32x4*)(bias + n0 + 8 * o + 4);
    u32x4 pk = {pack2(v0[0] + b0[0], v0[1] + b0[1]), pack2(v0[2] + b0[2], v0[3] + b0[3]), pack2(v1[0] + b1[0], v1[1] + b1[1]), pack2(v1[2] + b1[2], v1[3] + b1[3])};
    *(u32x4*)(out + (size_t)(m0 + r) * N + n0 + 8 * o) = pk;
  }
}

static float bf(float x) { __bf16 h = (__bf16)x; return (float)h; }
template <int KS, int PD> static void run(int M, int N, int K, const char* what) {
  std::vector<__bf16> hx((size_t)M * K), hw((size_t)N * K); std::vector<float> hb(N);
  srand(1);
  for (auto& v : hx) v = (__bf16)((rand() % 2001 - 1000) * 1e-3f);
  for (auto& v : hw) v = (__bf16)((rand() % 2001 - 1000) * 1e-3f * 0.05f);
  for (auto& v : hb) v = (rand() % 2001 - 1000) * 1e-3f;
  __bf16 *dx, *dw, *dout; float* db;
  CK(hipMalloc(&dx, hx.size() * 2)); CK(hipMalloc(&dw, hw.size() * 2)); CK(hipMalloc(&dout, (size_t)M * N * 2)); CK(hipMalloc(&db, N * 4));
  CK(hipMemcpy(dx, hx.data(), hx.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dw, hw.data(), hw.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(db, hb.data(), N * 4, hipMemcpyHostToDevice));
  dim3 grid(N / 64, M / 64);
  hipStream_t s; CK(hipStreamCreate(&s));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((linear_rs<KS, PD>), grid, dim3(512), 0, s, dx, dw, db, dout, M, N, K);
  CK(hipStreamSynchronize(s));
  // timing inside a graph of 20 launches (what the pass does: back-to-back launches)
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((linear_rs<KS, PD>), grid, dim3(512), 0, s, dx, dw, db, dout, M, N, K);
  CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(ge, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<__bf16> ho((size_t)M * N); CK(hipMemcpy(ho.data(), dout, ho.size() * 2, hipMemcpyDeviceToHost));
  double maxerr = 0;
  for (int q = 0; q < 200; ++q) {
    const int m = rand() % M, n = rand() % N; double acc = hb[n];
    for (int k = 0; k < K; ++k) acc += (double)(float)hx[(size_t)m * K + k] * (double)(float)hw[(size_t)n * K + k];
    const double err = fabs((double)(float)ho[(size_t)m * N + n] - acc) / (fabs(acc) + 0.05);
    if (err > maxerr) maxerr = err;
  }
  printf("%-34s M=%d N=%d K=%d  KS=%d PD=%d: %6.2f us per launch (%.0f TF/s), max rel err of 200 samples %.2e, blocks %d\n", what, M, N, K, KS, PD, ms * 1e3 / 40, 2.0 * M * N * K / (ms * 1e-3 / 40) / 1e12, maxerr, grid.x * grid.y);
  hipFree(dx); hipFree(dw); hipFree(dout); hipFree(db);
}
int main() {
  run<8, 2>(1024, 1280, 1280, "16x16 level to_out / proj"); run<8, 5>(1024, 1280, 1280, "16x16 level to_out / proj"); run<8, 10>(1024, 1280, 1280, "16x16 level to_out / proj");
  run<4, 5>(1024, 1280, 1280, "16x16 level, 4 slices x 2 row groups");
  run<4, 5>(4096, 640, 640, "32x32 level"); run<4, 10>(4096, 640, 640, "32x32 level"); run<8, 5>(4096, 640, 640, "32x32 level");
  run<8, 5>(256, 1280, 1280, "8x8 level"); run<8, 10>(256, 1280, 1280, "8x8 level");
  run<8, 5>(64, 1280, 1280, "batch 1, 8x8 level");
  run<8, 5>(1024, 3840, 1280, "16x16 level q|k|v");
  return 0;
}
