// Common device/host helpers for the DiffUTE gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// The 16-bit storage / MFMA-operand element of this build.  The library is compiled twice from the same sources:
//   libdiffute_hip.so      bf16  (default; `bf16` = __bf16, v_mfma_f32_32x32x16_bf16)
//   libdiffute_hip_f16.so  fp16  (-DDMX_F16; `bf16` = _Float16, v_mfma_f32_32x32x16_f16) - what `.to(dtype=torch.float16)`
//                          selects (train_diffute_v1.py:789-797, BASELINE configs[4]).
// Throughout the sources the identifier `bf16` therefore means "the build's 16-bit element"; everything that depends on
// its bit layout goes through the helpers below (DMX_MFMA_*, pack_bf2 / unpack_bf8 / h2f_lo / h2f_hi / f2bf_bits / bf_bits2f).
// LDS images, fragment layouts and transpose reads are identical for both (16-bit lanes); accumulation is fp32 in both.
#ifdef DMX_F16
typedef _Float16 dmx_h16;
#define DMX_ELEM_NAME "fp16"
#define DMX_MFMA_32x32x16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)
#define DMX_MFMA_16x16x32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16((a), (b), (c), 0, 0, 0)
#else
typedef __bf16 dmx_h16;
#define DMX_ELEM_NAME "bf16"
#define DMX_MFMA_32x32x16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)
#define DMX_MFMA_16x16x32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)
#endif
typedef dmx_h16 bf16;
typedef __attribute__((ext_vector_type(8))) dmx_h16 bf16x8;
typedef __attribute__((ext_vector_type(4))) dmx_h16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define DMX_OK 0
#define DMX_ERR_ARG (-1)
#define DMX_ERR_HIP (-2)
#define DMX_ERR_UNSUPPORTED (-3)
#define DMX_ERR_WORKSPACE (-4)
#define DMX_ERR_DEVICE (-5)            // a kernel gave up on an in-kernel wait (a peer block never published): the result of that launch is invalid

void dmx_set_error(const char* fmt, ...);
int dmx_check_launch(const char* what);

// ---- device -> host error channel.  Eight ints of pinned host memory mapped into the device address space (one set per process):
// [0] code (the DmxDevKernel that raised, 0 = none), [1] claim word, [2] block, [3..5] kernel-specific detail.  A kernel that gives up on a
// bounded in-kernel wait RAISES here (first raiser wins) and goes on, so the GPU never hangs; every later launch check (dmx_check_launch),
// hipGraph replay and dmx_device_error() reads word 0 from the host side - no synchronisation - and returns DMX_ERR_DEVICE with the
// detail in dmx_last_error().  The failing launch itself has returned by then: the error surfaces at the NEXT C-ABI call or poll.
enum DmxDevKernel { DMX_DEVK_HALO_PEER = 1, DMX_DEVK_STREAMK_HELPER = 2, DMX_DEVK_SKINNY_PEER = 3, DMX_DEVK_ATTN_PEER = 4 };
int* dmx_dev_err_words();              // device-visible pointer (nullptr when the pinned allocation failed: kernels then do not raise)
int dmx_poll_device_error();           // DMX_OK, or DMX_ERR_DEVICE + dmx_set_error(...) and the words cleared
#ifdef __HIPCC__
__device__ __forceinline__ void dmx_dev_raise(int* err, int kernel, int block, int d0, int d1, int d2) {
  if (!err) return;
  int expect = 0;
  if (__hip_atomic_compare_exchange_strong(err + 1, &expect, 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) {
    err[2] = block; err[3] = d0; err[4] = d1; err[5] = d2;
    __atomic_thread_fence(__ATOMIC_RELEASE);                       // (system scope: the detail is in host memory before the code)
    __hip_atomic_store(err, kernel, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
#endif

#define DMX_REQUIRE(cond, ...)                \
  do {                                        \
    if (!(cond)) {                            \
      dmx_set_error(__VA_ARGS__);             \
      return DMX_ERR_ARG;                     \
    }                                         \
  } while (0)

#define DMX_HIP(expr)                                                        \
  do {                                                                       \
    hipError_t _e = (expr);                                                  \
    if (_e != hipSuccess) {                                                  \
      dmx_set_error("%s failed: %s", #expr, hipGetErrorString(_e));          \
      return DMX_ERR_HIP;                                                    \
    }                                                                        \
  } while (0)

// fp32 -> 16-bit element, round-to-nearest-even: native casts so hipcc emits gfx950's v_cvt_pk_bf16_f32 (one instruction
// per pair; v_cvt_f16_f32 + v_pack in the fp16 build) instead of a ~7-op integer sequence per value.
typedef __attribute__((ext_vector_type(2))) dmx_h16 bf16x2;
__device__ __forceinline__ unsigned short f2bf_bits(float f) {
  const dmx_h16 h = (dmx_h16)f;
  return __builtin_bit_cast(unsigned short, h);
}
#ifdef DMX_F16
__device__ __forceinline__ float bf_bits2f(unsigned short b) { return (float)__builtin_bit_cast(_Float16, b); }
// the two elements of a packed dword (low / high half) as floats
__device__ __forceinline__ float h2f_lo(unsigned int w) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(w & 0xffffu)); }
__device__ __forceinline__ float h2f_hi(unsigned int w) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(w >> 16)); }
#else
__device__ __forceinline__ float bf_bits2f(unsigned short b) {
  return __uint_as_float(((unsigned int)b) << 16);
}
__device__ __forceinline__ float h2f_lo(unsigned int w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float h2f_hi(unsigned int w) { return __uint_as_float(w & 0xffff0000u); }
#endif
// value of the lane N positions below within the 16-lane DPP row (0 for the first N lanes of a row)
template <int N> __device__ __forceinline__ float dpp_row_shr(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x110 + N, 0xf, 0xf, true));
}
__device__ __forceinline__ unsigned int pack_bf2(float lo, float hi) {
  const bf16x2 v = {(dmx_h16)lo, (dmx_h16)hi};
  return __builtin_bit_cast(unsigned int, v);
}
__device__ __forceinline__ void unpack_bf8(const u32x4 v, float* f) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f[2 * i] = h2f_lo(v[i]);
    f[2 * i + 1] = h2f_hi(v[i]);
  }
}
__device__ __forceinline__ u32x4 pack_bf8(const float* f) {
  u32x4 v;
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = pack_bf2(f[2 * i], f[2 * i + 1]);
  return v;
}
__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }
// exact-GELU 0.5*x*(1+erf(x/sqrt2)) with erf from Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, far below bf16
// resolution): one rcp + one exp2 + 7 fma instead of libm's branchy erff - the GEGLU epilogue is VALU-bound.
__device__ __forceinline__ float gelu_erf_f(float x) {
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, z, 1.0f));
  float poly = __builtin_fmaf(1.061405429f, t, -1.453152027f);
  poly = __builtin_fmaf(poly, t, 1.421413741f);
  poly = __builtin_fmaf(poly, t, -0.284496736f);
  poly = __builtin_fmaf(poly, t, 0.254829592f);
  poly *= t;
  const float e = __builtin_amdgcn_exp2f(-z * z * 1.4426950408889634f);
  const float erf_abs = 1.0f - poly * e;                       // erf(|x|/sqrt2)
  const float erf_v = copysignf(erf_abs, x);
  return 0.5f * x * (1.0f + erf_v);
}

// the same function on two values with gfx950's packed fp32 ops (v_pk_fma_f32 / v_pk_mul_f32): 21 instructions for two values
// against 2 x 18 - for epilogues that are VALU-bound on it (xf_chain.hip's in-kernel GEGLU)
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 gelu_erf_f2(f32x2 x) {
  const f32x2 z = __builtin_elementwise_abs(x) * 0.70710678118654752440f;
  const f32x2 d = __builtin_elementwise_fma((f32x2)(0.3275911f), z, (f32x2)(1.0f));
  const f32x2 t = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
  f32x2 poly = __builtin_elementwise_fma((f32x2)(1.061405429f), t, (f32x2)(-1.453152027f));
  poly = __builtin_elementwise_fma(poly, t, (f32x2)(1.421413741f));
  poly = __builtin_elementwise_fma(poly, t, (f32x2)(-0.284496736f));
  poly = __builtin_elementwise_fma(poly, t, (f32x2)(0.254829592f));
  poly = poly * t;
  const f32x2 a = z * z * -1.4426950408889634f;
  const f32x2 e = {__builtin_amdgcn_exp2f(a.x), __builtin_amdgcn_exp2f(a.y)};
  const f32x2 erf_abs = __builtin_elementwise_fma(-poly, e, (f32x2)(1.0f));
  const f32x2 erf_v = __builtin_elementwise_copysign(erf_abs, x);
  return x * 0.5f * (erf_v + 1.0f);
}

// ---- GroupNorm statistics records ("DmxStat"): per (sample, channel) FOUR 64-bit integers {sum * 2^20, floor(sumsq * 2^8),
// (sumsq - that) * 2^40, 0}, accumulated with integer atomics: integer addition is associative, so the totals are bit-reproducible
// whatever order the tiles finish in.  The two-word sum of squares is exact to 2^-40 per addend and cannot wrap below 2^55 = 3.6e16
// (512 x 512 pixels at an RMS of 3.7e5) - the single word scaled by 2^32 of round 3 wrapped at 2^31 (RMS ~90 at 512 x 512).
#define DMX_STAT_WORDS 4
__device__ __forceinline__ void dmx_stat_add(long long* dst, float sum, float sumsq) {
  const double s = fmin(fmax((double)sum, -4.0e12), 4.0e12), q = fmin(fmax((double)sumsq, 0.0), 3.0e16);
  const long long qh = (long long)(q * 256.0);
  const long long ql = (long long)__builtin_rint((q - (double)qh * (1.0 / 256.0)) * 1099511627776.0);
  __hip_atomic_fetch_add(dst, (long long)__builtin_rint(s * 1048576.0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_fetch_add(dst + 1, qh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_fetch_add(dst + 2, ql, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double dmx_stat_sum(long long s) { return (double)s * (1.0 / 1048576.0); }
__device__ __forceinline__ double dmx_stat_sumsq(long long qh, long long ql) { return (double)qh * (1.0 / 256.0) + (double)ql * (1.0 / 1099511627776.0); }

#ifdef __HIPCC__
// LDS-DMA as an asm statement (16 bytes per lane -> lds_addr + 16 lane).  NOT __builtin_amdgcn_global_load_lds: hipcc knows that the builtin
// writes LDS and, unable to prove that the tile it fills is not the tile being read, puts `s_waitcnt vmcnt(0)` in front of the next ds_read -
// the prefetch of the NEXT K / V tile (and, in the first iteration, the weight prefetch units that come from HBM) was waited for right
// after it had been issued, in every iteration (found in the ISA in round 5: the loop ran at half the speed its instruction mix allows).
// The asm statement is invisible to that pass; the loops wait with their own counted `s_waitcnt vmcnt(N)` + barrier before a tile is read,
// and scripts/isa_audit.py rule L checks the drain before s_endpgm.  M0 (the LDS base of the DMA) is saved and restored inside the statement.
__device__ __forceinline__ void dmx_dma16(const void* gsrc, unsigned lds_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_addr) : "memory");
}
#define DMX_LDS_ADDR(p) ((unsigned)(size_t)(__attribute__((address_space(3))) char*)(p))

#endif

// dynamic-LDS opt-in above 64 KB: a per-DEVICE function attribute, so it is set once per (kernel, device), result checked
#define DMX_LDS_OPT_IN(kernel, bytes)                                                                        \
  do {                                                                                                       \
    static bool dmx_attr_[64] = {};                                                                          \
    int dmx_dev_ = 0; DMX_HIP(hipGetDevice(&dmx_dev_));                                                      \
    if (dmx_dev_ >= 64 || !dmx_attr_[dmx_dev_ & 63]) {                                                       \
      DMX_HIP(hipFuncSetAttribute((const void*)(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes))); \
      dmx_attr_[dmx_dev_ & 63] = true;                                                                       \
    }                                                                                                        \
  } while (0)

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t a, size_t b) { return (a + b - 1) / b * b; }
