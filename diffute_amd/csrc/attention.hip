// Flash-style fused attention forward, head dim 64, bf16 MFMA (SURVEY.md 8a K5 self, K6 cross).
//
//   O[b, q, h, :] = softmax(Q K^T * scale) V        (no mask; Skv = 577 tail handled)
//
// grid (ceil(Sq/128), H, B), 256 threads = 4 waves; each wave owns 32 query rows, the
// block streams 64-key K / V tiles through LDS, double buffered.  Row-major V (the UNet path): both tiles go L2 -> LDS by
// LDS-DMA (global_load_lds, 16 B per lane, no staging registers and no address VALU in the loop: a wave-uniform tile
// base plus two per-lane constant offsets), unpadded 128-byte rows made conflict-free by XOR swizzles applied on the DMA
// SOURCE address (K: 16-byte chunk ^ ((key >> 1) & 7) for the ds_read_b128 fragment reads; V: 64-byte half ^ ((key >> 1) & 1) for
// the transpose reads).  4 % faster than staging through registers (147.7 -> 141.4 us at S = 4096, B*H = 20) and 16
// VGPRs / 11 KB of LDS lighter.  scripts/attic/attn_pieces.py (probe build) prices the pieces of the loop: exp 19 %, QK MFMAs
// 17 %, PV MFMAs 10 %, row sum 9 %, row max 5 %, and 34 % for the skeleton (24 LDS fragment reads = 16 KB per wave per
// tile, the scale fma, bf16 packing) - the costs add up almost exactly, i.e. the three waves of a SIMD hide little of each
// other: VALU issue, LDS reads and the matrix pipe are each 30-47 % busy and effectively serialised.
// V^T input (cached cross-attention operands of other callers): register-staged tiles as before.  Scores are computed TRANSPOSED (S^T = K Q^T, K rows as the MFMA A
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

// ds_read_b64_tr_b16 (gfx950 LDS transpose read; semantics verified by scripts/attic/probes/tr_probe.hip): within each
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
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// R = 32-row query sub-blocks per wave (1 or 2), NW = waves per block: a block covers 32*R*NW query rows and streams the K/V
// tiles ONCE for all of them.  R = 2 halves the K/V traffic out of L2 per query row (the S = 4096 self-attention of the
// 64x64 level is bound by exactly that: every 128-row block re-streams the head's whole K/V), halves the LDS fragment
// reads per MFMA (a K / V fragment feeds both sub-blocks) and gives the scheduler two independent softmax / MFMA chains
// to interleave inside one wave.
// PROBE (builds with -DDMX_ATTN_PROBE only; results invalid): bit 0 no exp, 1 no row sum, 2 no row max, 3 no QK MFMAs,
// 4 no PV MFMAs, 5 no K/V staging after tile 0, 6 no per-tile barrier - the marginal cost of each piece of the loop.
template <bool VROWMAJOR, int R, int NW, int PROBE = 0>
__global__ __launch_bounds__(64 * NW, R == 2 ? 2 : (NW == 4 ? 2 : 3)) void dmx_attn_d64_kernel(const AttnArgs p) {
  constexpr int NT = 64 * NW, SL = 512 / NT;          // staging loads per thread per operand per tile (64 rows x 8 pieces)
  constexpr bool DMA = VROWMAJOR;
  constexpr int KTB = DMA ? 64 * 128 : KT_BYTES, VTB = DMA ? 64 * 128 : VT_BYTES;     // DMA tiles: unpadded, swizzled
  constexpr int NBUF = 2;                             // (a third slot with tiles requested two iterations ahead measured the same: 141.3 vs 141.4 us)
  __shared__ __attribute__((aligned(16))) char smem[NBUF * (KTB + VTB)];
  __shared__ __attribute__((aligned(16))) char pf_dump[DMA ? 1024 : 16];
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int lr = lane & 31, lh = lane >> 5;
  const int b = blockIdx.z, h = blockIdx.y;
  const int q0 = blockIdx.x * (32 * R * NW) + wave * (32 * R);
  const float sl2 = p.scale * 1.4426950408889634f;

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

  // ---- LDS-DMA staging (row-major V): instruction i of this wave fills rows 8*(wave + NW*i) .. +8 of a tile; lane =
  // (row drow, physical 16-byte chunk dchk) of that 1-KB slab and fetches the source chunk the swizzle assigns to it
  const int drow = lane >> 3, dchk = lane & 7;
  const unsigned ldkb = (unsigned)p.ldk * 2u, ldvb = (unsigned)p.ldv * 2u;
  // K: chunk ^ ((row >> 1) & 7) - the key that makes the 32-row ds_read_b128 fragment reads conflict-free with 128-byte rows
  // (row & 7, rounds 1-2, left rows r and r + 8 of a lane group in one bank quad).  Slab rows are 8 (wave + NW i) + drow and
  // NW is even, so the key is a per-thread constant
  static_assert(NW % 2 == 0, "K swizzle key assumes an even number of waves");
  const unsigned kcb = (unsigned)((dchk ^ (((drow >> 1) + 4 * (wave & 1)) & 7)) << 4);
  // V: 64-byte half ^ ((row >> 1) & 1): a 16-lane group of a transpose read touches 32 bytes of four consecutive rows (128-byte
  // stride = 32 banks), so rows r and r + 2 must sit in different halves; keyed on row & 1 (rounds 1-2) rows 0 / 2 and 1 / 3 collided:
  // SQ_LDS_BANK_CONFLICT was 2.7x the LDS instruction cycles of the kernel
  const unsigned vcb = (unsigned)((((((dchk >> 2) ^ ((drow >> 1) & 1)) << 2) | (dchk & 3))) << 4);
  const unsigned koff = (unsigned)drow * ldkb + kcb, voff = (unsigned)drow * ldvb + vcb;
  const char* kdma = (const char*)(p.k + (size_t)b * p.kv_rows * p.ldk + h * 64);
  const char* vdma = DMA ? (const char*)(p.v + (size_t)b * p.kv_rows * p.ldv + h * 64) : nullptr;
  auto stage = [&](int buf, int kv0) {
    char* ksd = smem + buf * (KTB + VTB);
    char* vsd = ksd + KTB;
    if (kv0 + 64 <= p.Skv) {
#pragma unroll
      for (int i = 0; i < 8 / NW; ++i) {
        const int r0 = 8 * (wave + NW * i);
        dmx_dma16((kdma + (size_t)(kv0 + r0) * ldkb + koff), DMX_LDS_ADDR((ksd + r0 * 128)));
        dmx_dma16((vdma + (size_t)(kv0 + r0) * ldvb + voff), DMX_LDS_ADDR((vsd + r0 * 128)));
      }
    } else {                                           // last, ragged tile: keys past Skv re-read the last valid row (their P is 0)
      const int last = p.Skv - 1 - kv0;
#pragma unroll
      for (int i = 0; i < 8 / NW; ++i) {
        const int r0 = 8 * (wave + NW * i);
        const unsigned rc = (unsigned)min(r0 + drow, last);
        dmx_dma16((kdma + (size_t)kv0 * ldkb + (rc * ldkb + kcb)), DMX_LDS_ADDR((ksd + r0 * 128)));
        dmx_dma16((vdma + (size_t)kv0 * ldvb + (rc * ldvb + vcb)), DMX_LDS_ADDR((vsd + r0 * 128)));
      }
    }
  };
  const int ksw = (lh ^ ((lr >> 1) & 7)) << 4;        // K fragment: row lr, chunk (2kk + lh) ^ ((lr >> 1) & 7) == (2kk << 4) ^ ksw

  f32x16 o[R][2];
  float m_run[R], l_run[R];
#pragma unroll
  for (int j = 0; j < R; ++j) {
#pragma unroll
    for (int i = 0; i < 16; ++i) { o[j][0][i] = 0.f; o[j][1][i] = 0.f; }
    m_run[j] = -INFINITY; l_run[j] = 0.f;
  }

  const int ntiles = (p.Skv + 63) / 64;
  int npf = 0;                                         // prefetch units this wave has in flight (first iteration only)
  if (DMA) { stage(0, 0); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
  else { load_tile(0); write_tile(0); }
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
    if (it + 1 < ntiles && !(PROBE & 32)) { if (DMA) stage((it + 1) & 1, kv0 + 64); else load_tile(kv0 + 64); }
    if (DMA && it == 0) {
      // weight prefetch for the launches that follow (AttnArgs.pf): 1-KB units dealt over (block, wave), at most three per wave, LDS-DMA
      // into a dump slot.  Requested BEHIND the second K / V tile and left in flight by this iteration's counted wait: cold weights come
      // from HBM, the tiles from L2 - the first wait that covers them is the one at the end of the second iteration
      const int nblk = gridDim.x * gridDim.y * gridDim.z, blk = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int nb = p.pf_bytes[r];
        for (int u = blk * NW + wave; u * 1024 < nb && npf < 3; u += nblk * NW, ++npf) {
          int off = u * 1024 + lane * 16; if (off > nb - 16) off = nb - 16;
          dmx_dma16(((const char*)p.pf[r] + off), DMX_LDS_ADDR(pf_dump));
        }
      }
    }
    const char* ks = smem + ((PROBE & 32) ? 0 : (it & 1)) * (KTB + VTB);
    const char* vs = ks + KTB;

    // ---- S^T = K Q^T : s[j][kt][r] = score(query 32j + lr, key kv0 + 32kt + (r&3) + 8(r>>2) + 4lh); a K fragment feeds all sub-blocks
    f32x16 s[R][2];
    {
      const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // folds to the inline constant 0 of the first MFMA
      bf16x8 kf[2][4];
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
          kf[kt][kk] = DMA ? *(const bf16x8*)(ks + (32 * kt + lr) * 128 + (((2 * kk) << 4) ^ ksw))
                           : *(const bf16x8*)(ks + (32 * kt + lr) * (KROW * 2) + (2 * kk + lh) * 16);
      // kk outer, key-half inner: consecutive MFMAs go to DIFFERENT accumulators (a dependent 32x32x16 chain issues every 64 cycles, independent ones every 32)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
          for (int j = 0; j < R; ++j) {
            if (PROBE & 8) { if (kk == 0) { s[j][kt] = zero; s[j][kt][0] = __builtin_bit_cast(float, (int)kf[kt][0][0] + (int)kf[kt][1][1] + (int)kf[kt][2][2] + (int)kf[kt][3][3]) * 1e-30f; } }
            else s[j][kt] = DMX_MFMA_32x32x16(kf[kt][kk], qf[j][kk], kk == 0 ? zero : s[j][kt]);
          }
      __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);             // all eight fragment reads first ...
      __builtin_amdgcn_sched_group_barrier(0x008, 8 * R, 0);         // ... then the MFMAs in the order written
    }
    // ---- online softmax on the RAW scores: p = exp2(s*c - m*c) with c = scale*log2(e) folded into one fma per score
    // (no separate scaling pass).  OPTIMISTIC reference max: the tile is exponentiated against the running max as it
    // stands, WITHOUT looking for the tile's own max first; a lane whose 32 exponentials sum to more than 2^13 (or to
    // inf / NaN: the first tile, where the running max is -inf) proves some score sits far above the reference, and only
    // then - wave-uniform, a handful of tiles per row - the row max is taken, O and l are rescaled and the tile is
    // exponentiated again.  Otherwise every p <= 2^13: harmless in fp32 accumulators and bf16 P (relative precision),
    // and the 16 max3 + cross-lane exchange + compare per tile are gone (-5 %: 141 -> 133 us at S = 4096, B*H = 20).
    // (Also tried: scale and reference max folded into the QK MFMAs - Q pre-multiplied, the first k-step accumulating
    // onto a register splat of -M - which removes the 32 fma per tile but needs 16 more registers: 145 us at three
    // waves per SIMD with spills, 150 us at two waves without, against 137 / 143 us on the same boxes.  Dropped.)
    bf16x8 pf[R][4];
#pragma unroll
    for (int j = 0; j < R; ++j) {
      if (kv0 + 64 > p.Skv) {
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int key = kv0 + 32 * kt + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (key >= p.Skv) s[j][kt][r] = -INFINITY;
          }
      }
      auto exponentiate = [&](float mc) {
#ifndef DMX_ATTN_PACKED_SM
        // SCALAR f32 instructions (v_fma_f32 / v_add_f32; attention.o is built with -fno-slp-vectorize so that they stay scalar): packed f32
        // VALU issued on a SIMD whose matrix pipe is busy costs ~25 cycles more than the two plain instructions it replaces
        // (MI355X_MICROARCH.md cycle constants; measured here: 4096^2 B=4 147 -> 139 us, 9216^2 fp16 314 -> 303 us, bit-identical results)
        float q0 = 0.f, q1 = 0.f;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            unsigned int w[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float a0 = __builtin_fmaf(s[j][kt][8 * u + 2 * e], sl2, mc), a1 = __builtin_fmaf(s[j][kt][8 * u + 2 * e + 1], sl2, mc);
              const float p0 = __builtin_amdgcn_exp2f(a0), p1 = __builtin_amdgcn_exp2f(a1);
              q0 += p0; q1 += p1;
              w[e] = pack_bf2(p0, p1);
            }
            u32x4 wv = {w[0], w[1], w[2], w[3]};
            pf[j][2 * kt + u] = __builtin_bit_cast(bf16x8, wv);
          }
        return q0 + q1;
#else
        f32x2 ps2 = {0.f, 0.f};                       // (even, odd) scores of the pairs: one v_pk_add_f32 per pair
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            unsigned int w[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              // (one v_pk_fma_f32 per score pair: the accumulator registers of a pair are consecutive)
              const f32x2 sv = {s[j][kt][8 * u + 2 * e], s[j][kt][8 * u + 2 * e + 1]};
              const f32x2 av = __builtin_elementwise_fma(sv, (f32x2){sl2, sl2}, (f32x2){mc, mc});
              const float a0 = av.x, a1 = av.y;
              const float p0 = (PROBE & 1) ? a0 : __builtin_amdgcn_exp2f(a0);
              const float p1 = (PROBE & 1) ? a1 : __builtin_amdgcn_exp2f(a1);
              if (!(PROBE & 2)) ps2 += (f32x2){p0, p1};
              w[e] = pack_bf2(p0, p1);
            }
            u32x4 wv = {w[0], w[1], w[2], w[3]};
            pf[j][2 * kt + u] = __builtin_bit_cast(bf16x8, wv);
          }
        return ps2.x + ps2.y;
#endif
      };
      float psum = exponentiate(-m_run[j] * sl2);
      if (__any(!(psum <= 8192.0f)) && !(PROBE & 4)) {   // some score is far above the reference max (or there is none yet)
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
          for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[j][kt][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float m_new = fmaxf(m_run[j], mx);
        const float alpha = __builtin_amdgcn_exp2f((m_run[j] - m_new) * sl2);       // 0 on the first tile (m_run = -inf, m_new finite)
        m_run[j] = m_new;
        l_run[j] *= alpha;
#pragma unroll
        for (int i = 0; i < 16; ++i) { o[j][0][i] *= alpha; o[j][1][i] *= alpha; }
        psum = exponentiate(-m_run[j] * sl2);
      }
      l_run[j] += psum;
    }

    // ---- O^T += V^T P^T : k-slot e of step s4 <-> key 16*s4 + 4lh + (e&3) + 8(e>>2); a V fragment feeds all sub-blocks
    if (VROWMAJOR) {
      // V tile is [key][d]; the MFMA A operand (rows d, k-slots = keys) comes from LDS transpose reads:
      // group g = lane>>4 covers d = 32dt + 16(g&1) + 0..15 and key-half lh = g>>1; two reads per operand.
      // (all of a lane's rows have (row >> 1) & 1 == (p16 >> 3) & 1: the swizzled half of d-tile dt is a per-lane constant)
      const int p16 = lane & 15, g = lane >> 4;
      const char* va = vs + (4 * (g >> 1) + (p16 >> 2)) * 128 + (16 * (g & 1) + 4 * (p16 & 3)) * 2;
      const int vsw = (p16 >> 3) & 1;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(va + 64 * (dt ^ vsw) + (2 * s4) * 8 * 128));
          const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(va + 64 * (dt ^ vsw) + (2 * s4 + 1) * 8 * 128));
          const s16x8 vv = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
          for (int j = 0; j < R; ++j) {
            if (PROBE & 16) { o[j][dt][s4] += __builtin_bit_cast(float, (int)vv[0] + (int)vv[5]) * 1e-30f + __builtin_bit_cast(float, (int)pf[j][s4][0] + (int)pf[j][s4][7]) * 1e-30f; }
            else o[j][dt] = DMX_MFMA_32x32x16(__builtin_bit_cast(bf16x8, vv), pf[j][s4], o[j][dt]);
          }
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
          for (int j = 0; j < R; ++j) o[j][dt] = DMX_MFMA_32x32x16(__builtin_bit_cast(bf16x8, vv), pf[j][s4], o[j][dt]);
        }
    }

    if (DMA) {                                                      // tile it+1 has landed (requested a whole iteration ago)
      if (it > 0 || npf == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else if (npf == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
      else if (npf == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    }
    else if (it + 1 < ntiles && !(PROBE & 32)) write_tile((it + 1) & 1);
    if (!(PROBE & 64)) __syncthreads();
  }

  if (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (a single-tile stream leaves the prefetch units in flight behind its only wait: nothing may outlive the block)
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

#ifdef DMX_ATTN_PIPE            // the software-pipelined variant is built in probe builds only (-DDMX_ATTN_PIPE): see EXPERIMENTS.md round 5
#ifndef DMX_ATTN_DMA_SLOTS
#define DMX_ATTN_DMA_SLOTS 0    // 1: the wave's four K / V pieces are issued inside the slot stream instead of at the top of the iteration
#endif
#ifndef DMX_ATTN_PD
#define DMX_ATTN_PD 4           // operand prefetch distance of the slot loop, in slots
#endif
#ifndef DMX_ATTN_ABL
#define DMX_ATTN_ABL 0          // probe builds: 1 no exp2, 2 no MFMAs, 4 no operand reads in the slots (results invalid)
#endif
// ---- software-pipelined variant (round 5; row-major V, LDS-DMA tiles, 4 waves x 32 query rows).
// The loop above runs QK^T -> softmax -> P V tile by tile inside each wave: while the wave's 80 softmax instructions (32 of them
// quarter-rate exponentials) issue, the matrix pipe idles, and while its 16 MFMAs run nothing else of the wave does - and the other
// waves of the SIMD hide little of it (the pieces of the loop add up: header).  Here ONE iteration issues, slot by slot,
//     MFMA of  S(it+1) = K(it+1) Q^T      (8 slots)        |
//     MFMA of  O += V(it-1) P(it-1)       (8 slots)        |  one softmax pair (pk_fma, 2 exp2, pk_add, cvt_pk) of tile `it` per slot
// so that every MFMA has ~5 independent VALU instructions of the SAME wave behind it: the matrix pipe works under the softmax.  Same
// arithmetic in the same order as the loop above (bit-identical results).  Costs: S of two tiles and P of two tiles live at once
// (~200 VGPRs: two waves per SIMD instead of three), a third V buffer (V(it-1) is read while V(it) waits and V(it+1) lands), one
// wasted QK^T at the end of the key stream.  The rescale of the optimistic reference max stays a rare wave-uniform branch BEHIND the
// slots: O already holds P(it-1) V(it-1) by then, so O *= alpha is still exact.
__global__ __launch_bounds__(256, 2) void dmx_attn_d64_pipe_kernel(const AttnArgs p) {
  constexpr int NW = 4, KTB = 64 * 128, VTB = 64 * 128;
  __shared__ __attribute__((aligned(16))) char smem[2 * KTB + 3 * VTB];
  __shared__ __attribute__((aligned(16))) char pf_dump[1024];
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int lr = lane & 31, lh = lane >> 5;
  const int b = blockIdx.z, h = blockIdx.y;
  const int q0 = blockIdx.x * (32 * NW) + wave * 32;
  const float sl2 = p.scale * 1.4426950408889634f;

  bf16x8 qf[4];
  {
    int qrow = q0 + lr; if (qrow >= p.Sq) qrow = p.Sq - 1;
    const bf16* qp = p.q + ((size_t)b * p.Sq + qrow) * p.ldq + h * 64 + 8 * lh;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) qf[kk] = *(const bf16x8*)(qp + 16 * kk);
  }
  // LDS-DMA geometry: as in dmx_attn_d64_kernel (K: 16-byte chunk ^ ((row >> 1) & 7); V: 64-byte half ^ ((row >> 1) & 1), on the SOURCE address)
  const int drow = lane >> 3, dchk = lane & 7;
  const unsigned ldkb = (unsigned)p.ldk * 2u, ldvb = (unsigned)p.ldv * 2u;
  const unsigned kcb = (unsigned)((dchk ^ (((drow >> 1) + 4 * (wave & 1)) & 7)) << 4);
  const unsigned vcb = (unsigned)((((((dchk >> 2) ^ ((drow >> 1) & 1)) << 2) | (dchk & 3))) << 4);
  const unsigned koff = (unsigned)drow * ldkb + kcb, voff = (unsigned)drow * ldvb + vcb;
  const char* kdma = (const char*)(p.k + (size_t)b * p.kv_rows * p.ldk + h * 64);
  const char* vdma = (const char*)(p.v + (size_t)b * p.kv_rows * p.ldv + h * 64);
  auto stage_one = [&](const char* src, unsigned ldb, unsigned off, unsigned cb, char* dst, int kv0) {
    if (kv0 + 64 <= p.Skv) {
#pragma unroll
      for (int i = 0; i < 8 / NW; ++i) {
        const int r0 = 8 * (wave + NW * i);
        dmx_dma16((src + (size_t)(kv0 + r0) * ldb + off), DMX_LDS_ADDR((dst + r0 * 128)));
      }
    } else {                                           // ragged last tile: keys past Skv re-read the last valid row (their P is 0)
      const int last = p.Skv - 1 - kv0;
#pragma unroll
      for (int i = 0; i < 8 / NW; ++i) {
        const int r0 = 8 * (wave + NW * i);
        const unsigned rc = (unsigned)min(r0 + drow, last);
        dmx_dma16((src + (size_t)kv0 * ldb + (rc * ldb + cb)), DMX_LDS_ADDR((dst + r0 * 128)));
      }
    }
  };
  auto stageK = [&](int buf, int kv0) { stage_one(kdma, ldkb, koff, kcb, smem + buf * KTB, kv0); };
  auto stageV = [&](int buf, int kv0) { stage_one(vdma, ldvb, voff, vcb, smem + 2 * KTB + buf * VTB, kv0); };
  const int ksw = (lh ^ ((lr >> 1) & 7)) << 4;
  const int p16 = lane & 15, lg = lane >> 4;
  const int va_off = (4 * (lg >> 1) + (p16 >> 2)) * 128 + (16 * (lg & 1) + 4 * (p16 & 3)) * 2;
  const int vsw = (p16 >> 3) & 1;

  f32x16 o[2];
#pragma unroll
  for (int i = 0; i < 16; ++i) { o[0][i] = 0.f; o[1][i] = 0.f; }
  float m_run = -INFINITY, l_run = 0.f;
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  const int ntiles = (p.Skv + 63) / 64;
  stageK(0, 0); stageV(0, 0);
  if (ntiles > 1) stageK(1, 64);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) asm volatile("" : "+v"(qf[kk]));      // (fresh values: no pending load for hipcc to re-wait on inside the loop - see above)
  __syncthreads();

  auto rdK = [&](const char* kb, const int g) {      // operand of QK slot g = (kk = g >> 1, key half kt = g & 1)
    return *(const bf16x8*)(kb + (32 * (g & 1) + lr) * 128 + (((2 * (g >> 1)) << 4) ^ ksw));
  };
  // PV slot g = (d half dt = g & 1, key step s4 = g >> 1): consecutive MFMAs alternate between the two O accumulators (a dependent
  // 32x32x16 chain issues every ~64 cycles, independent ones every 32); each accumulator still sums its key steps in order
  auto rdV = [&](const char* vb, const int g) {
    const char* va = vb + va_off + 64 * ((g & 1) ^ vsw);
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(va + (2 * (g >> 1)) * 8 * 128));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(va + (2 * (g >> 1) + 1) * 8 * 128));
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
  };

  // prologue: S(0)
  f32x16 sA[2], sB[2];                                 // S of the current / the next tile: the two sets swap roles every iteration (no copies)
  {
    const char* kb = smem;
    bf16x8 kf[8];
#pragma unroll
    for (int g = 0; g < 8; ++g) kf[g] = rdK(kb, g);
#pragma unroll
    for (int g = 0; g < 8; ++g) sA[g & 1] = DMX_MFMA_32x32x16(kf[g], qf[g >> 1], (g >> 1) == 0 ? zero : sA[g & 1]);
  }
  bf16x8 pA[4], pB[4];                                 // P of the previous / the current tile, likewise; zeros in front of the first tile
#pragma unroll
  for (int i = 0; i < 4; ++i) pA[i] = __builtin_bit_cast(bf16x8, (u32x4){0u, 0u, 0u, 0u});
  int npf = 0;

  // one iteration: sc = S(it) (in), sn = S(it+1) (out), pfp = P(it-1) (in), pfc = P(it) (out)
#ifdef DMX_ATTN_TIMING
  long long tm_slots = 0, tm_wait = 0, tm_bar = 0, tm_top = 0, tm_t0 = __builtin_amdgcn_s_memtime();
#define TM_STAMP(acc_) { const long long n_ = __builtin_amdgcn_s_memtime(); acc_ += n_ - tm_t0; tm_t0 = n_; }
#else
#define TM_STAMP(acc_)
#endif
  auto iteration = [&](const int it, f32x16 (&sc)[2], f32x16 (&sn)[2], const bf16x8 (&pfp)[4], bf16x8 (&pfc)[4]) {
    const int kv0 = it * 64;
    // K(it+2) goes over K(it), V(it+1) over V(it-2): both consumed by every wave before the barrier that ended iteration it-1.  The four
    // 1-KB pieces of this wave are issued INSIDE the slot stream (an LDS-DMA instruction costs its wave ~60 cycles among MFMAs and ~190 when
    // four of them queue up back to back); sources are clamped instead of branched on - behind the end of the key stream the last tile is
    // fetched again into a buffer nobody reads
    const int tk = min(it + 2, ntiles - 1) * 64, tv = min(it + 1, ntiles - 1) * 64;
    const int lastk = p.Skv - 1 - tk, lastv = p.Skv - 1 - tv;
    const char* dsrc[4]; char* ddst[4];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r0 = 8 * (wave + NW * i);
      dsrc[i] = kdma + (size_t)tk * ldkb + ((unsigned)min(r0 + drow, lastk) * ldkb + kcb);
      ddst[i] = smem + (it & 1) * KTB + r0 * 128;
      dsrc[2 + i] = vdma + (size_t)tv * ldvb + ((unsigned)min(r0 + drow, lastv) * ldvb + vcb);
      ddst[2 + i] = smem + 2 * KTB + ((it + 1) % 3) * VTB + r0 * 128;
    }
    if (!DMX_ATTN_DMA_SLOTS) {
#pragma unroll
      for (int i = 0; i < 4; ++i) dmx_dma16(dsrc[i], DMX_LDS_ADDR(ddst[i]));
    }
    const char* kn = smem + ((it + 1) & 1) * KTB;                            // K(it+1) (behind the last tile: a stale tile, the result is not used)
    const char* vp = smem + 2 * KTB + (it == 0 ? 0 : (it - 1) % 3) * VTB;     // V(it-1) (first tile: V(0) against P = 0)

    if (kv0 + 64 > p.Skv) {
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kv0 + 32 * kt + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (key >= p.Skv) sc[kt][r] = -INFINITY;
        }
    }

    TM_STAMP(tm_top)
    // ---- the slots.  Slot g issues: the operand reads of slot g + PD, its MFMA, and three INDEPENDENT pieces of the softmax chain -
    // the scale fma of pair g + 1, the two exp2 of pair g, the row-sum add and the bf16 pack of pair g - 1 - so that no instruction of a
    // slot waits for its predecessor (a lone wave per SIMD has nobody else to fill those bubbles)
    float ps0 = 0.f, ps1 = 0.f;                        // (even, odd) scores of the pairs - SCALAR f32 instructions: beside MFMAs a v_pk_fma_f32 / v_pk_add_f32
    const float mc = -m_run * sl2;                     // costs ~25 cycles more than the two plain instructions it replaces (MI355X_MICROARCH.md, cycle constants)
    unsigned int w[4];
    constexpr int PD = DMX_ATTN_PD, NR = PD + 1;       // operand prefetch distance in slots, ring size
    bf16x8 opnd[NR];
#pragma unroll
    for (int g = 0; g < PD; ++g) opnd[g] = rdK(kn, g);
    float av0[17], av1[17], pp0[17], pp1[17];
#define DMX_AV(g) { constexpr int kt_ = (g) >> 3, u_ = ((g) >> 2) & 1, e_ = (g) & 3;                                                 \
                    av0[g] = __builtin_fmaf(sc[kt_][8 * u_ + 2 * e_], sl2, mc); av1[g] = __builtin_fmaf(sc[kt_][8 * u_ + 2 * e_ + 1], sl2, mc); }
#define DMX_FIN(g) { constexpr int kt_ = (g) >> 3, u_ = ((g) >> 2) & 1, e_ = (g) & 3;                                                \
                     ps0 += pp0[g]; ps1 += pp1[g]; w[e_] = pack_bf2(pp0[g], pp1[g]);                                                 \
                     if (e_ == 3) { const u32x4 wv_ = {w[0], w[1], w[2], w[3]}; pfc[2 * kt_ + u_] = __builtin_bit_cast(bf16x8, wv_); } }
    DMX_AV(0)
    // (a macro, not a loop: the group sizes of sched_group_barrier must be literals)
#define DMX_SLOT(g, NRD)                                                                                                            \
    {                                                                                                                                \
      if (!(DMX_ATTN_ABL & 4)) {                                                                                                     \
        if (g + PD < 8) opnd[(g + PD) % NR] = rdK(kn, g + PD);                                                                       \
        else if (g + PD < 16) opnd[(g + PD) % NR] = rdV(vp, g + PD - 8);                                                             \
      }                                                                                                                              \
      if (!(DMX_ATTN_ABL & 2)) {                                                                                                     \
        if (g < 8) sn[g & 1] = DMX_MFMA_32x32x16(opnd[g % NR], qf[g >> 1], (g >> 1) == 0 ? zero : sn[g & 1]);                        \
        else o[(g - 8) & 1] = DMX_MFMA_32x32x16(opnd[g % NR], pfp[(g - 8) >> 1], o[(g - 8) & 1]);                                    \
      } else if (g < 8 && (g >> 1) == 0) { sn[g & 1] = zero; sn[g & 1][0] = __builtin_bit_cast(float, (int)opnd[g % NR][0]) * 1e-30f; } \
      if (g + 1 < 16) DMX_AV(g + 1)                                                                                                  \
      pp0[g] = (DMX_ATTN_ABL & 1) ? av0[g] : __builtin_amdgcn_exp2f(av0[g]); pp1[g] = (DMX_ATTN_ABL & 1) ? av1[g] : __builtin_amdgcn_exp2f(av1[g]); \
      if (g > 0) DMX_FIN(g - 1)                                                                                                      \
      if (DMX_ATTN_DMA_SLOTS && (g & 3) == 2) dmx_dma16(dsrc[g >> 2], DMX_LDS_ADDR(ddst[g >> 2])); \
      __builtin_amdgcn_sched_group_barrier(0x100, NRD, 0);  /* the operand reads of slot g + PD */                                    \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);    /* this slot's MFMA */                                                    \
      __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);    /* the softmax pieces underneath */                                       \
      __builtin_amdgcn_sched_barrier(0);                                                                                             \
    }
#define DMX_NRD(g) ((g) + PD < 8 ? 1 : ((g) + PD < 16 ? 2 : 0))
    DMX_SLOT(0, DMX_NRD(0)) DMX_SLOT(1, DMX_NRD(1)) DMX_SLOT(2, DMX_NRD(2)) DMX_SLOT(3, DMX_NRD(3)) DMX_SLOT(4, DMX_NRD(4)) DMX_SLOT(5, DMX_NRD(5))
    DMX_SLOT(6, DMX_NRD(6)) DMX_SLOT(7, DMX_NRD(7)) DMX_SLOT(8, DMX_NRD(8)) DMX_SLOT(9, DMX_NRD(9)) DMX_SLOT(10, DMX_NRD(10)) DMX_SLOT(11, DMX_NRD(11))
    DMX_SLOT(12, DMX_NRD(12)) DMX_SLOT(13, DMX_NRD(13)) DMX_SLOT(14, DMX_NRD(14)) DMX_SLOT(15, DMX_NRD(15))
    DMX_FIN(15)
#undef DMX_SLOT
#undef DMX_NRD
#undef DMX_AV
#undef DMX_FIN
    float psum = ps0 + ps1;
    if (__any(!(psum <= 8192.0f))) {                   // some score is far above the reference max (or there is none yet): rare, wave-uniform
      float mx = -INFINITY;
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sc[kt][r]);
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      const float m_new = fmaxf(m_run, mx);
      const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * sl2);
      m_run = m_new;
      l_run *= alpha;
#pragma unroll
      for (int i = 0; i < 16; ++i) { o[0][i] *= alpha; o[1][i] *= alpha; }      // (O holds P(it-1) V(it-1) already: still relative to the old max)
      const float mc2 = -m_run * sl2;
      f32x2 q2 = {0.f, 0.f};
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          unsigned int w2[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const f32x2 sv = {sc[kt][8 * u + 2 * e], sc[kt][8 * u + 2 * e + 1]};
            const f32x2 av = __builtin_elementwise_fma(sv, (f32x2){sl2, sl2}, (f32x2){mc2, mc2});
            const float p0 = __builtin_amdgcn_exp2f(av.x), p1 = __builtin_amdgcn_exp2f(av.y);
            q2 += (f32x2){p0, p1};
            w2[e] = pack_bf2(p0, p1);
          }
          const u32x4 wv = {w2[0], w2[1], w2[2], w2[3]};
          pfc[2 * kt + u] = __builtin_bit_cast(bf16x8, wv);
        }
      psum = q2.x + q2.y;
    }
    l_run += psum;
    if (it == 0) {
      // weight prefetch for the launches that follow (AttnArgs.pf), as in dmx_attn_d64_kernel: requested BEHIND this iteration's tile pieces and
      // left in flight by its counted wait
      const int nblk = gridDim.x * gridDim.y * gridDim.z, blk = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int nb = p.pf_bytes[r];
        for (int u = blk * NW + wave; u * 1024 < nb && npf < 3; u += nblk * NW, ++npf) {
          int off = u * 1024 + lane * 16; if (off > nb - 16) off = nb - 16;
          dmx_dma16(((const char*)p.pf[r] + off), DMX_LDS_ADDR(pf_dump));
        }
      }
    }
    TM_STAMP(tm_slots)

    if (it > 0 || npf == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (npf == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else if (npf == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    TM_STAMP(tm_wait)
    __syncthreads();
    TM_STAMP(tm_bar)
  };
  for (int it = 0; it < ntiles; it += 2) {
    iteration(it, sA, sB, pA, pB);
    if (it + 1 < ntiles) iteration(it + 1, sB, sA, pB, pA);
  }
  // ---- the last tile's P V (its P sits in pB after an even tile index, in pA after an odd one)
  {
    const char* vp = smem + 2 * KTB + ((ntiles - 1) % 3) * VTB;
    if ((ntiles - 1) & 1) {
#pragma unroll
      for (int g = 0; g < 8; ++g) o[g & 1] = DMX_MFMA_32x32x16(rdV(vp, g), pA[g >> 1], o[g & 1]);
    } else {
#pragma unroll
      for (int g = 0; g < 8; ++g) o[g & 1] = DMX_MFMA_32x32x16(rdV(vp, g), pB[g >> 1], o[g & 1]);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // nothing may outlive the block (prefetch units of a single-tile stream)
  {
    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv = 1.0f / l_tot;
    const int qrow = q0 + lr;
#ifdef DMX_ATTN_TIMING
    if (p.lse && lane == 0) {
      float* d = p.lse + ((size_t)((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 4 + wave) * 4;
      d[0] = (float)tm_top; d[1] = (float)tm_slots; d[2] = (float)tm_wait; d[3] = (float)tm_bar;
    }
#else
    if (p.lse && qrow < p.Sq && lh == 0) p.lse[((size_t)b * p.H + h) * p.Sq + qrow] = m_run * sl2 + log2f(l_tot);
#endif
    if (qrow < p.Sq) {
      bf16* op = p.o + ((size_t)b * p.Sq + qrow) * p.ldo + h * 64 + 4 * lh;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          u32x2 pk = {pack_bf2(o[dt][4 * g] * inv, o[dt][4 * g + 1] * inv),
                      pack_bf2(o[dt][4 * g + 2] * inv, o[dt][4 * g + 3] * inv)};
          *(u32x2*)(op + 32 * dt + 8 * g) = pk;
        }
    }
  }
}

#endif  // DMX_ATTN_PIPE

// block shape: 128 query rows (4 waves x 32).  256-row blocks (4 waves x 64 rows, 8 waves x 32 rows) lose on every shape of the
// pass (EXPERIMENTS.md); the template keeps the R / NW parameters, only <., 1, 4> is instantiated.
int dmx_attention_launch(const AttnArgs& a, hipStream_t stream) {
  DMX_REQUIRE(a.B > 0 && a.H > 0 && a.Sq > 0 && a.Skv > 0, "attention: empty problem");
  DMX_REQUIRE(a.kv_rows >= a.Skv, "attention: kv_rows=%d < Skv=%d", a.kv_rows, a.Skv);
  DMX_REQUIRE(a.ldq % 8 == 0 && a.ldk % 8 == 0 && a.ldo % 4 == 0, "attention: strides must be multiples of 8 (ldq=%d ldk=%d)", a.ldq, a.ldk);
  dim3 grid(cdiv(a.Sq, 128), a.H, a.B);
  if (a.v) {
    DMX_REQUIRE(a.ldv % 8 == 0, "attention: ldv=%d must be a multiple of 8", a.ldv);
#ifdef DMX_ATTN_PIPE
#ifndef DMX_ATTN_PIPE_MIN
#define DMX_ATTN_PIPE_MIN 1
#endif
    if (a.Skv >= DMX_ATTN_PIPE_MIN) {                  // the software-pipelined loop: long key streams (measured per shape: EXPERIMENTS.md round 5)
      hipLaunchKernelGGL(dmx_attn_d64_pipe_kernel, grid, dim3(256), 0, stream, a);
      dmx_profile_note_symbol("dmx_attn_d64_pipe_kernel(AttnArgs)");
      return dmx_check_launch("dmx_attn_d64_pipe_kernel");
    }
#endif
#ifdef DMX_PROBES
#define PB(N) else if (pb == N) hipLaunchKernelGGL((dmx_attn_d64_kernel<true, 1, 4, N>), grid, dim3(256), 0, stream, a);
    if (const int pb = getenv("DMX_ATTN_PROBE_BITS") ? atoi(getenv("DMX_ATTN_PROBE_BITS")) : 0; pb == 0) hipLaunchKernelGGL((dmx_attn_d64_kernel<true, 1, 4>), grid, dim3(256), 0, stream, a);
    PB(1) PB(2) PB(4) PB(8) PB(16) PB(32) PB(96) PB(3) PB(7) PB(24) PB(31) PB(127) PB(120)
#undef PB
#else
    hipLaunchKernelGGL((dmx_attn_d64_kernel<true, 1, 4>), grid, dim3(256), 0, stream, a);
#endif
    dmx_profile_note_symbol("void dmx_attn_d64_kernel<true, 1, 4, 0>(AttnArgs)");
  } else {
    grid = dim3(cdiv(a.Sq, 128), a.H, a.B);
    DMX_REQUIRE(a.vt && a.ldvt % 8 == 0 && a.skv_stride % 8 == 0, "attention: V^T strides must be multiples of 8 (ldvt=%d skv_stride=%d)", a.ldvt, a.skv_stride);
    DMX_REQUIRE(a.skv_stride >= (a.Skv + 7) / 8 * 8, "attention: skv_stride=%d < Skv=%d rounded to 8", a.skv_stride, a.Skv);
    hipLaunchKernelGGL((dmx_attn_d64_kernel<false, 1, 4>), grid, dim3(256), 0, stream, a);
  }
  return dmx_check_launch("dmx_attn_d64_kernel");
}
