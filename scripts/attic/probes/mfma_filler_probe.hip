// What does one "slot" = [v_mfma_f32_32x32x16_bf16 + fillers] cost a LONE wave per SIMD on gfx950?  (round 5: the software-pipelined attention loop
// measured 118 cycles per slot whatever its VALU mix.)   hipcc --offload-arch=gfx950 -O3 -o mfma_filler_probe mfma_filler_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
#define MFMA(acc, a, b) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
// fillers as asm so that nothing is optimised away or re-packed; every filler reads/writes its own registers (independent of the MFMA)
#define EXP(x) asm volatile("v_exp_f32 %0, %0" : "+v"(x))
#define FMA(x, y) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(y))
#define ADD(x, y) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(y))
#define PKFMA(x, y) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(y))
#define PKADD(x, y) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x) : "v"(y))
#define CVT(d, x, y) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(d) : "v"(x), "v"(y))
#define DSR(d, a) asm volatile("ds_read_b128 %0, %1" : "=v"(d) : "v"(a))
template <int V>
__global__ __launch_bounds__(256) void k(long long* out, float seed) {
  __shared__ char lds[16384];
  f32x16 acc[4];
  for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) acc[j][i] = seed * i;
  u32x4 au = {1u, 2u, 3u, (unsigned)threadIdx.x}, bu = {5u, 6u, 7u, 8u};
  bf16x8 a = __builtin_bit_cast(bf16x8, au), b = __builtin_bit_cast(bf16x8, bu);
  float x0 = seed, x1 = seed + 1, x2 = seed + 2, x3 = seed + 3, y = 0.5f * seed, s0 = 0.f, s1 = 0.f;
  typedef __attribute__((ext_vector_type(2))) float f32x2;
  f32x2 p0 = {seed, seed}, p1 = {seed, seed}, p2 = {seed, seed}, p3 = {seed, seed}, py = {y, y}; float x4 = seed, x5 = seed;
  unsigned cv = 0; u32x4 dr = {0u, 0u, 0u, 0u}; unsigned la = (threadIdx.x & 63) * 16;
  lds[threadIdx.x] = 1; __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < 256; ++it) {
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      if (V != 9 && V < 10) MFMA(acc[g & (V == 8 ? 0 : 3)], a, b);
      if (V == 1 || V == 5 || V == 6 || V == 7 || V == 9) { EXP(x0); EXP(x1); }
      if (V == 2) { FMA(x0, y); FMA(x1, y); FMA(x2, y); FMA(x3, y); }
      if (V == 3) { PKFMA(p0, py); PKADD(p1, py); }
      if (V == 4) { FMA(x0, y); FMA(x1, y); FMA(x2, y); FMA(x3, y); FMA(s0, y); FMA(s1, y); FMA(x0, y); FMA(x1, y); }
      if (V == 5 || V == 6 || V == 7 || V == 9) { FMA(x2, y); FMA(x3, y); ADD(s0, x0); ADD(s1, x1); }
      if (V == 6 || V == 7 || V == 9) CVT(cv, x0, x1);
      if (V == 7 || V == 9) DSR(dr, la);
      if (V == 10) { PKFMA(p0, py); PKFMA(p1, py); PKFMA(p2, py); PKFMA(p3, py); }
      if (V == 11) { FMA(x0, y); FMA(x1, y); FMA(x2, y); FMA(x3, y); FMA(s0, y); FMA(s1, y); FMA(x4, y); FMA(x5, y); }
      if (V == 12) { PKADD(p0, py); PKADD(p1, py); PKADD(p2, py); PKADD(p3, py); }
      if (V == 13) { EXP(x0); EXP(x1); EXP(x2); EXP(x3); }
    }
    if (V == 7 || V == 9) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  float r = x4 + x5 + p2.x + p3.y + x0 + x1 + x2 + x3 + s0 + s1 + p0.x + p1.y + __builtin_bit_cast(float, cv) + __builtin_bit_cast(float, dr.x);
  for (int j = 0; j < 4; ++j) r += acc[j][0];
  if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = (long long)r; }
}
int main() {
  long long* d; hipMalloc(&d, 64); long long h[2];
  const char* names[] = {"MFMA only (4 accumulators)", "MFMA + 2 exp", "MFMA + 4 fma", "MFMA + pk_fma + pk_add", "MFMA + 8 fma", "MFMA + 2 exp + 2 fma + 2 add",
                         "MFMA + 2 exp + 2 fma + 2 add + cvt_pk", "MFMA + 2 exp + 2 fma + 2 add + cvt_pk + ds_read_b128", "MFMA only, ONE accumulator (dependent chain)", "no MFMA: 2 exp + 2 fma + 2 add + cvt + ds_read",
                         "no MFMA: 4 pk_fma (8 flop-pairs)", "no MFMA: 8 fma", "no MFMA: 4 pk_add", "no MFMA: 4 exp"};
#define RUN(V) hipLaunchKernelGGL(k<V>, dim3(1), dim3(256), 0, 0, d, 1.0f); hipMemcpy(h, d, 16, hipMemcpyDeviceToHost); printf("%-58s %6.1f cycles per slot\n", names[V], (double)h[0] / (256 * 8));
  RUN(0) RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9) RUN(10) RUN(11) RUN(12) RUN(13)
  return 0;
}
