// Timestep-embedding path: sinusoid -> Linear -> SiLU -> Linear, and the
// 22 per-resnet Linear(1280, Cout)(SiLU(emb)) projections batched into one launch.
// M = batch (tiny), so these are weight-streaming GEMVs: one wave per output feature,
// 16-byte bf16 weight loads, fp32 activations and accumulation.
#include "common.h"
#include "kernels.h"

// out[b][0:half] = cos(t_b * freq), out[b][half:] = sin(t_b * freq)  (flip_sin_to_cos=True)
__global__ void dmx_timestep_embedding_kernel(const long long* t, int t_count, const float* freq, int B, int dim, float* out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int half = dim >> 1;
  if (i >= B * half) return;
  const int b = i / half, j = i - b * half;
  const float tv = (float)t[t_count == 1 ? 0 : b];
  const float arg = tv * freq[j];
  out[(size_t)b * dim + j] = cosf(arg);
  out[(size_t)b * dim + half + j] = sinf(arg);
}
int dmx_timestep_embedding_launch(const long long* t, int t_count, const float* freq, int B, int dim, float* out, hipStream_t stream) {
  DMX_REQUIRE(t_count == 1 || t_count == B, "timestep_embedding: need 1 or B timesteps, got %d", t_count);
  hipLaunchKernelGGL(dmx_timestep_embedding_kernel, dim3(cdiv(B * dim / 2, 256)), dim3(256), 0, stream, t, t_count, freq, B, dim, out);
  return dmx_check_launch("dmx_timestep_embedding_kernel");
}

#define LS_MAXB 8
// One wave per output feature.  The (SiLU'd) activations are staged once per block in LDS (the SiLU used to be
// re-evaluated by every wave for every output row: 70 us for the 17.9k-row time_emb_proj matrix, now weight-streaming).
__global__ __launch_bounds__(256) void dmx_linear_small_kernel(const float* x, int ldx, const bf16* w, int ldw, const float* bias,
                                                               float* y, int ldy, int B, int N, int K, int silu_in) {
  extern __shared__ float xs[];                       // [min(B, LS_MAXB)][K]
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  for (int b0 = 0; b0 < B; b0 += LS_MAXB) {
    const int nb = min(LS_MAXB, B - b0);
    __syncthreads();
    for (int i = threadIdx.x; i < nb * K; i += 256) {
      const int j = i / K, k = i - j * K;
      const float v = x[(size_t)(b0 + j) * ldx + k];
      xs[i] = silu_in ? silu_f(v) : v;
    }
    __syncthreads();
    if (n >= N) continue;
    float acc[LS_MAXB];
#pragma unroll
    for (int j = 0; j < LS_MAXB; ++j) acc[j] = 0.f;
    for (int k = lane * 8; k < K; k += 64 * 8) {
      float wf[8]; unpack_bf8(*(const u32x4*)(w + (size_t)n * ldw + k), wf);
#pragma unroll
      for (int j = 0; j < LS_MAXB; ++j) {
        if (j < nb) {
          const f32x4 x0 = *(const f32x4*)(xs + j * K + k), x1 = *(const f32x4*)(xs + j * K + k + 4);
          acc[j] += x0[0] * wf[0] + x0[1] * wf[1] + x0[2] * wf[2] + x0[3] * wf[3] + x1[0] * wf[4] + x1[1] * wf[5] + x1[2] * wf[6] + x1[3] * wf[7];
        }
      }
    }
#pragma unroll
    for (int j = 0; j < LS_MAXB; ++j) {
      float s = acc[j];
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d);
      if (lane == 0 && j < nb) y[(size_t)(b0 + j) * ldy + n] = s + (bias ? bias[n] : 0.f);
    }
  }
}
int dmx_linear_small_launch(const float* x, int ldx, const bf16* w, int ldw, const float* bias, float* y, int ldy,
                            int B, int N, int K, int silu_in, hipStream_t stream) {
  DMX_REQUIRE(K % 8 == 0 && ldw % 8 == 0, "linear_small: K=%d and ldw=%d must be multiples of 8", K, ldw);
  const size_t lds = (size_t)(B < LS_MAXB ? B : LS_MAXB) * K * sizeof(float);
  DMX_REQUIRE(lds <= 64 * 1024, "linear_small: K=%d too large for the LDS activation stage", K);
  hipLaunchKernelGGL(dmx_linear_small_kernel, dim3(cdiv(N, 4)), dim3(256), lds, stream, x, ldx, w, ldw, bias, y, ldy, B, N, K, silu_in);
  return dmx_check_launch("dmx_linear_small_kernel");
}
