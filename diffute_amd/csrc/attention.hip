// Flash-style fused attention forward, head dim 64, bf16 MFMA (SURVEY.md 8a K5 self, K6 cross).
//
//   O[b, q, h, :] = softmax(Q K^T * scale) V        (no mask; Skv = 577 tail handled)
//
// grid (ceil(Sq/128), H, B), 256 threads = 4 waves; each wave owns 32 query rows, the
// block streams 64-key K / V^T tiles through LDS (register-staged, double buffered: the
// next tile's global loads are issued before the current tile is multiplied and written
// to LDS after it).  Scores are computed TRANSPOSED (S^T = K Q^T, K rows as the MFMA A
// operand) so that every lane holds 32 scores of ONE query: the row max / row sum are
// in-register plus one lane^32 exchange, and the bf16 P fragment that feeds P·V is
// exactly 8 consecutive accumulator registers - no LDS round trip for P.  V arrives
// transposed (V^T[d][key], produced by the projection GEMM with swapped operand roles),
// so the P·V A-operand is two ds_read_b64 per MFMA.  Online softmax in fp32 (exp2 domain).
#include "common.h"
#include "kernels.h"
#include <stdlib.h>

#define KROW 72      // K tile row stride in elements (144 B: conflict-free ds_read_b128)
#define VROW 68      // V^T tile row stride in elements (136 B: conflict-free ds_read_b64)
#define VRS 192      // row-major V tile row stride in BYTES (64 data + 32 pad elements: conflict-free tr reads)
#define KT_BYTES (64 * KROW * 2)
#define VT_BYTES (64 * VRS)          // sized for the larger of the two V layouts (V^T needs 64*136)

// ds_read_b64_tr_b16 (gfx950 LDS transpose read; semantics verified by scripts/probes/tr_probe.hip): within each
// 16-lane group, lane p supplies the address of 4 contiguous b16 = row (p>>2), columns 4*(p&3).. of a [4][16] block;
// lane q receives column q of that block, rows 0..3.  Eight reads, one asm statement (the compiler does not count
// asm LDS ops: the caller waits with s_waitcnt lgkmcnt + sched_barrier before consuming).
#define DMX_TR8(V, A, O0, O1, O2, O3, O4, O5, O6, O7)                                                     \
  asm volatile("ds_read_b64_tr_b16 %0, %8 offset:%9\n\tds_read_b64_tr_b16 %1, %8 offset:%10\n\t"          \
               "ds_read_b64_tr_b16 %2, %8 offset:%11\n\tds_read_b64_tr_b16 %3, %8 offset:%12\n\t"         \
               "ds_read_b64_tr_b16 %4, %8 offset:%13\n\tds_read_b64_tr_b16 %5, %8 offset:%14\n\t"         \
               "ds_read_b64_tr_b16 %6, %8 offset:%15\n\tds_read_b64_tr_b16 %7, %8 offset:%16"              \
               : "=&v"(V[0]), "=&v"(V[1]), "=&v"(V[2]), "=&v"(V[3]), "=&v"(V[4]), "=&v"(V[5]), "=&v"(V[6]), "=&v"(V[7]) \
               : "v"(A), "i"(O0), "i"(O1), "i"(O2), "i"(O3), "i"(O4), "i"(O5), "i"(O6), "i"(O7) : "memory")

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((address_space(3))) s16x4* lds_s16x4;

// R = 32-row query sub-blocks per wave (1 or 2), NW = waves per block: a block covers 32*R*NW query rows and streams the K/V
// tiles ONCE for all of them.  R = 2 halves the K/V traffic out of L2 per query row (the S = 4096 self-attention of the
// 64x64 level is bound by exactly that: every 128-row block re-streams the head's whole K/V), halves the LDS fragment
// reads per MFMA (a K / V fragment feeds both sub-blocks) and gives the scheduler two independent softmax / MFMA chains
// to interleave inside one wave.
template <bool VROWMAJOR, int R, int NW>
__global__ __launch_bounds__(64 * NW, R == 2 ? 2 : (NW == 4 ? 2 : 3)) void dmx_attn_d64_kernel(const AttnArgs p) {
  constexpr int NT = 64 * NW, SL = 512 / NT;          // staging loads per thread per operand per tile (64 rows x 8 pieces)
  __shared__ __attribute__((aligned(16))) char smem[2 * (KT_BYTES + VT_BYTES)];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  const int b = blockIdx.z, h = blockIdx.y;
  const int q0 = blockIdx.x * (32 * R * NW) + wave * (32 * R);
  const float sl2 = p.scale * 1.4426950408889634f;
  const float thr = 8.0f / sl2;                      // defer-max threshold in raw-score units

  // ---- Q fragments (MFMA B operand): sub-block j, query lr, d = 16*kk + 8*lh .. +8
  bf16x8 qf[R][4];
#pragma unroll
  for (int j = 0; j < R; ++j) {
    int qrow = q0 + 32 * j + lr; if (qrow >= p.Sq) qrow = p.Sq - 1;
    const bf16* qp = p.q + ((size_t)b * p.Sq + qrow) * p.ldq + h * 64 + 8 * lh;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) qf[j][kk] = *(const bf16x8*)(qp + 16 * kk);
  }

  // ---- staging assignment: chunk c = t + NT*i -> row c>>3, 16-byte piece c&7
  const int srow0 = t >> 3, spc = t & 7;
  constexpr int SROWS = NT / 8;                       // rows covered per staging pass
  const bf16* kbase = p.k + (size_t)b * p.kv_rows * p.ldk + h * 64 + spc * 8;
  const bf16* vbase = VROWMAJOR ? (p.v + (size_t)b * p.kv_rows * p.ldv + h * 64 + spc * 8)
                                : (p.vt + (size_t)(h * 64) * p.ldvt + (size_t)b * p.skv_stride + spc * 8);
  u32x4 kreg[SL], vreg[SL];
  auto load_tile = [&](int kv0) {
#pragma unroll
    for (int i = 0; i < SL; ++i) {
      int key = kv0 + srow0 + SROWS * i; if (key >= p.Skv) key = p.Skv - 1;
      kreg[i] = *(const u32x4*)(kbase + (size_t)key * p.ldk);
      if (VROWMAJOR) {
        vreg[i] = *(const u32x4*)(vbase + (size_t)key * p.ldv);          // row = key (clamped: P is 0 past Skv)
      } else {
        const int d = srow0 + SROWS * i;
        if (kv0 + spc * 8 < p.Skv) vreg[i] = *(const u32x4*)(vbase + (size_t)d * p.ldvt + kv0);
        else vreg[i] = (u32x4){0u, 0u, 0u, 0u};
      }
    }
  };
  auto write_tile = [&](int buf) {
    char* ks = smem + buf * (KT_BYTES + VT_BYTES);
    char* vs = ks + KT_BYTES;
#pragma unroll
    for (int i = 0; i < SL; ++i) {
      const int r = srow0 + SROWS * i;
      *(u32x4*)(ks + r * (KROW * 2) + spc * 16) = kreg[i];
      if (VROWMAJOR) {
        *(u32x4*)(vs + r * VRS + spc * 16) = vreg[i];
      } else {
        u32x2 lo = {vreg[i][0], vreg[i][1]}, hi = {vreg[i][2], vreg[i][3]};
        *(u32x2*)(vs + r * (VROW * 2) + spc * 16) = lo;
        *(u32x2*)(vs + r * (VROW * 2) + spc * 16 + 8) = hi;
      }
    }
  };

  f32x16 o[R][2];
  float m_run[R], l_run[R];
#pragma unroll
  for (int j = 0; j < R; ++j) {
#pragma unroll
    for (int i = 0; i < 16; ++i) { o[j][0][i] = 0.f; o[j][1][i] = 0.f; }
    m_run[j] = -INFINITY; l_run[j] = 0.f;
  }

  const int ntiles = (p.Skv + 63) / 64;
  load_tile(0);
  write_tile(0);
  // The Q fragments come from plain global loads issued before the loop.  hipcc's waitcnt pass cannot prove them complete
  // at the loop header (the back edge merges with the entry state), so it re-waits for them INSIDE the loop with
  // vmcnt(3), (2), (1), (0) in front of the QK MFMAs - and since vmcnt retires in order, that also waits for the K/V tile
  // prefetch issued at the top of the very same iteration: the prefetch was effectively synchronous (47 % of all wave cycles
  // sat in s_waitcnt).  Passing the fragments through an empty asm makes them fresh values that carry no pending load.
#pragma unroll
  for (int j = 0; j < R; ++j)
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) asm volatile("" : "+v"(qf[j][kk]));
  __syncthreads();
  for (int it = 0; it < ntiles; ++it) {
    const int kv0 = it * 64;
    if (it + 1 < ntiles) load_tile(kv0 + 64);
    const char* ks = smem + (it & 1) * (KT_BYTES + VT_BYTES);
    const char* vs = ks + KT_BYTES;

    // ---- S^T = K Q^T : s[j][kt][r] = score(query 32j + lr, key kv0 + 32kt + (r&3) + 8(r>>2) + 4lh); a K fragment feeds all sub-blocks
    f32x16 s[R][2];
    {
      const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // folds to the inline constant 0 of the first MFMA
      bf16x8 kf[2][4];
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) kf[kt][kk] = *(const bf16x8*)(ks + (32 * kt + lr) * (KROW * 2) + (2 * kk + lh) * 16);
      // kk outer, key-half inner: consecutive MFMAs go to DIFFERENT accumulators (a dependent 32x32x16 chain issues every 64 cycles, independent ones every 32)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
          for (int j = 0; j < R; ++j) s[j][kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[kt][kk], qf[j][kk], kk == 0 ? zero : s[j][kt], 0, 0, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);             // all eight fragment reads first ...
      __builtin_amdgcn_sched_group_barrier(0x008, 8 * R, 0);         // ... then the MFMAs in the order written
    }
    // ---- online softmax on the RAW scores: p = exp2(s*c - m*c) with c = scale*log2(e) folded into one fma per
    // score (no separate scaling pass); the running max is only raised - and O / l rescaled - when a score exceeds
    // it by more than 8/c (p stays <= 2^8: harmless in bf16/fp32, saves the O-wide rescale on almost every tile).
    bf16x8 pf[R][4];
#pragma unroll
    for (int j = 0; j < R; ++j) {
      float mx = -INFINITY;
      if (kv0 + 64 > p.Skv) {
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int key = kv0 + 32 * kt + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (key >= p.Skv) s[j][kt][r] = -INFINITY;
          }
      }
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[j][kt][r]);
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      if (!__all(mx <= m_run[j] + thr)) {                // wave-uniform: some query row needs a higher reference max
        const float m_new = fmaxf(m_run[j], mx);
        const float alpha = __builtin_amdgcn_exp2f((m_run[j] - m_new) * sl2);
        m_run[j] = m_new;
        l_run[j] *= alpha;
#pragma unroll
        for (int i = 0; i < 16; ++i) { o[j][0][i] *= alpha; o[j][1][i] *= alpha; }
      }
      const float mc = -m_run[j] * sl2;
      float psum = 0.f;
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          unsigned int w[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float p0 = __builtin_amdgcn_exp2f(__builtin_fmaf(s[j][kt][8 * u + 2 * e], sl2, mc));
            const float p1 = __builtin_amdgcn_exp2f(__builtin_fmaf(s[j][kt][8 * u + 2 * e + 1], sl2, mc));
            psum += p0 + p1;
            w[e] = pack_bf2(p0, p1);
          }
          u32x4 wv = {w[0], w[1], w[2], w[3]};
          pf[j][2 * kt + u] = __builtin_bit_cast(bf16x8, wv);
        }
      l_run[j] += psum;
    }

    // ---- O^T += V^T P^T : k-slot e of step s4 <-> key 16*s4 + 4lh + (e&3) + 8(e>>2); a V fragment feeds all sub-blocks
    if (VROWMAJOR) {
      // V tile is [key][d]; the MFMA A operand (rows d, k-slots = keys) comes from LDS transpose reads:
      // group g = lane>>4 covers d = 32dt + 16(g&1) + 0..15 and key-half lh = g>>1; two reads per operand.
      const int p16 = lane & 15, g = lane >> 4;
      const char* va = vs + (4 * (g >> 1) + (p16 >> 2)) * VRS + (16 * (g & 1) + 4 * (p16 & 3)) * 2;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(va + 64 * dt + (2 * s4) * 8 * VRS));
          const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(va + 64 * dt + (2 * s4 + 1) * 8 * VRS));
          const s16x8 vv = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
          for (int j = 0; j < R; ++j) o[j][dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vv), pf[j][s4], o[j][dt], 0, 0, 0);
        }
    } else {
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
          const char* vp = vs + (32 * dt + lr) * (VROW * 2) + (16 * s4 + 4 * lh) * 2;
          const u32x2 lo = *(const u32x2*)vp;
          const u32x2 hi = *(const u32x2*)(vp + 16);
          const u32x4 vv = {lo[0], lo[1], hi[0], hi[1]};
#pragma unroll
          for (int j = 0; j < R; ++j) o[j][dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vv), pf[j][s4], o[j][dt], 0, 0, 0);
        }
    }

    if (it + 1 < ntiles) write_tile((it + 1) & 1);
    __syncthreads();
  }

  // ---- normalise and store: lane holds query 32j + lr, d = 32dt + 8g + 4lh + e
#pragma unroll
  for (int j = 0; j < R; ++j) {
    const float l_tot = l_run[j] + __shfl_xor(l_run[j], 32);
    const float inv = 1.0f / l_tot;
    const int qrow = q0 + 32 * j + lr;
    // training: keep the row's log2-sum-exp of the scaled scores, P = exp2(s*scale*log2e - lse)
    if (p.lse && qrow < p.Sq && lh == 0) p.lse[((size_t)b * p.H + h) * p.Sq + qrow] = m_run[j] * sl2 + log2f(l_tot);
    if (qrow < p.Sq) {
      bf16* op = p.o + ((size_t)b * p.Sq + qrow) * p.ldo + h * 64 + 4 * lh;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          u32x2 pk = {pack_bf2(o[j][dt][4 * g] * inv, o[j][dt][4 * g + 1] * inv),
                      pack_bf2(o[j][dt][4 * g + 2] * inv, o[j][dt][4 * g + 3] * inv)};
          *(u32x2*)(op + 32 * dt + 8 * g) = pk;
        }
    }
  }
}

// block shape: 128 query rows (4 waves x 32) by default; 256 rows (4 waves x 64) once the grid still covers the chip with
// the larger blocks (the 64x64-level and 768-px self / cross attention), DMX_ATTN_ROWS=128|256 pins it (measurement aid)
static int attn_rows_per_block(const AttnArgs& a) {
  static const int pin = getenv("DMX_ATTN_ROWS") ? atoi(getenv("DMX_ATTN_ROWS")) : 0;
  if (pin == 128 || pin == 256) return pin;
  const long blocks256 = (long)cdiv(a.Sq, 256) * a.H * a.B;
  (void)blocks256;
  return 128;                                        // measured: the 256-row blocks lose on every shape of the pass (fewer, fatter waves per SIMD)
}

int dmx_attention_launch(const AttnArgs& a, hipStream_t stream) {
  DMX_REQUIRE(a.B > 0 && a.H > 0 && a.Sq > 0 && a.Skv > 0, "attention: empty problem");
  DMX_REQUIRE(a.kv_rows >= a.Skv, "attention: kv_rows=%d < Skv=%d", a.kv_rows, a.Skv);
  DMX_REQUIRE(a.ldq % 8 == 0 && a.ldk % 8 == 0 && a.ldo % 4 == 0, "attention: strides must be multiples of 8 (ldq=%d ldk=%d)", a.ldq, a.ldk);
  const int rows = attn_rows_per_block(a);
  dim3 grid(cdiv(a.Sq, rows), a.H, a.B);
  if (a.v) {
    DMX_REQUIRE(a.ldv % 8 == 0, "attention: ldv=%d must be a multiple of 8", a.ldv);
    static const int r2 = getenv("DMX_ATTN_R2") ? 1 : 0;       // measurement aid: 4 waves x 64 rows instead of 8 waves x 32 rows
    if (rows == 256 && r2) hipLaunchKernelGGL((dmx_attn_d64_kernel<true, 2, 4>), grid, dim3(256), 0, stream, a);
    else if (rows == 256) hipLaunchKernelGGL((dmx_attn_d64_kernel<true, 1, 8>), grid, dim3(512), 0, stream, a);
    else hipLaunchKernelGGL((dmx_attn_d64_kernel<true, 1, 4>), grid, dim3(256), 0, stream, a);
  } else {
    grid = dim3(cdiv(a.Sq, 128), a.H, a.B);
    DMX_REQUIRE(a.vt && a.ldvt % 8 == 0 && a.skv_stride % 8 == 0, "attention: V^T strides must be multiples of 8 (ldvt=%d skv_stride=%d)", a.ldvt, a.skv_stride);
    DMX_REQUIRE(a.skv_stride >= (a.Skv + 7) / 8 * 8, "attention: skv_stride=%d < Skv=%d rounded to 8", a.skv_stride, a.Skv);
    hipLaunchKernelGGL((dmx_attn_d64_kernel<false, 1, 4>), grid, dim3(256), 0, stream, a);
  }
  return dmx_check_launch("dmx_attn_d64_kernel");
}
