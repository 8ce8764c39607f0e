// Row-local chains of the BasicTransformerBlock at the 64x64 level (C = 320) as ONE kernel each (SURVEY.md 8a K7 / K8;
// /root/reference/app.ipynb:814 -> diffusers BasicTransformerBlock.forward):
//
//   MODE 0   h1 = a1 Wo1^T + bo1 + h0                      (attn1.to_out + residual)
//            q2 = LN2(h1) Wq2^T                            (attn2.to_q, LayerNorm folded: raw rows x W*gamma, normalised in the epilogue)
//   MODE 2   h0 = n Wpi^T + bpi                           (proj_in on the GroupNorm output)
//            qkv = LN1(h0) [Wq | Wk | Wv]^T                (attn1's stacked projections, LayerNorm folded; three 320-wide GEMMs on the same fragments)
//   MODE 1   h2 = a2 Wo2^T + bo2 + h1                      (attn2.to_out + residual)
//            h3 = h2 + FF2(GEGLU(LN3(h2) W1^T))            (ff.net, hidden 4C never leaves the CU)
//            y  = h3 Wpo^T + bpo + x                       (proj_out + the Transformer2DModel residual)
//
// Outside the two attention cores every op of the block is per query row.  As separate GEMMs they are 9 launches of 12-60 us whose
// K = 320 loops are 70 % prologue + epilogue (scripts/attic/linear_timeline.py), and the 4C-wide hidden tensor makes an 84 MB round
// trip.  Here a block owns 64 rows: the activation operand of the running GEMM lives in REGISTERS (each wave keeps its 32
// rows x 320 k as twenty 16x32 fragments), the weights stream through a three-slot LDS ring of [320 n][64 k] tiles by
// LDS-DMA (global_load_lds, 16 B per lane, XOR swizzle applied on the source side), two tiles (80 KB) always in flight
// across phase boundaries - all 256 blocks walk the same weight stream, so it is L2-resident and the DMA runs at
// ~120 GB/s per CU (scripts/attic/probes/dma_depth_probe.hip), not at the ~25 GB/s per CU of an HBM stream.  A phase's output goes
// through the ring slot that was computed last (bf16, [64][320] swizzled) to become the next phase's register fragments.
// Eight waves = 2 (rows) x 4 (80 output columns each), v_mfma_f32_16x16x32 with the weight tile as the A operand, so a lane
// ends up with 4 consecutive output channels of one row (8-byte LDS writes, and a value / gate pair of GEGLU in one lane:
// FF1's rows are fetched in the order [v0 v1 g0 g1] per lane quad straight from the packed 64-row groups of the arena).
// Residual tensors arrive as DMA tiles in the exchange layout (the epilogue reads the residual at the LDS address it then writes
// its output to); the column vectors of the epilogues are staged in LDS once per block.  The LDS-DMA instructions ride between
// the MFMAs, staggered by wave (mode 1) - the address path takes one 1-KB instruction per ~25 cycles per CU, and eight waves
// issuing at one program point queue behind each other with the matrix pipes idle (EXPERIMENTS.md, round 3 item 0).
// Deterministic: no atomics, fixed summation order.
#include "common.h"
#include "kernels.h"

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

namespace {
constexpr int XC = 320;                        // channel width this kernel is built for
constexpr int XBM = 64;                        // rows per block
constexpr int XSLOT = XC * 64 * 2;             // one [320 n][64 k] weight tile: 40 KB
constexpr int XFF_LD = 384;                    // bytes per row of the GEGLU chunk buffer (160 hidden columns used of 192)
constexpr int XFF_OFF = 3 * XSLOT;
constexpr int XST_OFF = XFF_OFF + XBM * XFF_LD;          // row statistics partials [64][4 column waves][2] floats
constexpr int XVEC_OFF = XST_OFF + XBM * 4 * 2 * 4;      // three fp32 [320] column vectors of the 320-wide epilogues (loaded once per block)
constexpr int XFFV_OFF = XVEC_OFF + 3 * XC * 4;          // c1 | c2 of the running feed-forward chunk (320 floats each, 3 KB with padding) + 1 KB the idle waves' lanes write
constexpr int XDUMP = XFFV_OFF + 4096;                   // 1 KB the weight prefetch for the next launches lands in (XfChainArgs.pf)
constexpr int XLDS = XDUMP + 1024;                       // 158 464 bytes
constexpr int XCHUNK = 160;                    // hidden columns per feed-forward chunk (= 320 packed FF1 rows)
}  // namespace

// 16-byte chunk c of row m in a [rows][8 k chunks] swizzled image: the XOR touches the low three chunk bits only
__device__ __forceinline__ int xsw(int c, int m) { return (c & ~7) | ((c & 7) ^ ((m >> 1) & 7)); }

// ABL (probe builds only, results invalid): bit 0 no MFMA phase, bit 1 no DMA refills
template <int MODE, int ABL = 0>
__global__ __launch_bounds__(512, 2) void dmx_xf_chain_kernel(const XfChainArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = w >> 2, wn = w & 3, lr = lane & 15, lh = lane >> 4;
  const int m0 = blockIdx.x * XBM;
  constexpr int NCH = 4 * XC / XCHUNK;                               // 8 feed-forward chunks
  // tile stream: five [320][64] weight tiles per 320-wide GEMM, then one RESIDUAL tile (this block's 64 rows of the tensor the
  // epilogue adds, fetched in the exchange-image layout: the epilogue reads it from LDS at the address it then writes its output to)
  constexpr int FF0 = 6;                                             // first feed-forward tile (mode 1)
  constexpr int NTILES = MODE == 0 ? 11 : MODE == 2 ? 20 : FF0 + NCH * 8 + 1 + 6;

  // ---- thread t fills LDS chunk positions t + 512 i; weight tiles: row t/8 + 64 i, physical chunk t & 7
  const int d_row = t >> 3;
  const int d_csrc = (t & 7) ^ ((t >> 4) & 7);                       // the source chunk the swizzle assigns to this position
  const int d_prow = 32 * ((t >> 4) & 1) + 2 * (t >> 5) + ((t >> 3) & 1);   // FF1: LDS row 4q + r <-> hidden 2q + (r & 1), value (r < 2) / gate of a packed 64-row group
  int gi = 0, si = 0;                                                // next tile to request and its ring slot
  // prepare_next(): decode the next tile (uniform branches; rare extra piece) into a base pointer + five per-thread byte
  // offsets; dma(i): the i-th of its five 16-byte-per-lane loads - branch-free, so that the MFMA phase can carry them
  // BETWEEN its MFMAs (an LDS-DMA instruction stalls its wave for 60-180 cycles when the address path is busy: issued as a
  // burst behind the barrier, both waves of a SIMD pay that at the same time and the matrix pipe idles)
  const char* dbase = nullptr; char* ddst = nullptr; unsigned doff = 0, dstep = 0; int dres = 0;      // tile = base + doff + i * dstep (weights) | residual rows
  auto prepare_next = [&]() {
    const bf16* base; unsigned ldb = XC * 2; int perm = 0, half = 0, resid = 0, vec = -1;
    if constexpr (MODE == 0) {
      if (gi < 5) base = p.w0 + 64 * gi;
      else if (gi == 5) { base = p.res; ldb = (unsigned)p.ldres * 2; resid = 1; }
      else if (gi < NTILES) base = p.w1 + 64 * (gi - 6);
      else base = p.w0;                                              // past the end: dummy tile (keeps the counted waits uniform)
    } else if constexpr (MODE == 2) {
      if (gi < 5) base = p.w0 + 64 * gi;
      else if (gi < NTILES) { const int gg = gi - 5, sl = gg / 5, j = gg - sl * 5; base = p.w1 + (size_t)sl * XC * XC + 64 * j; }   // rows 320 sl .. of [3C][C]
      else base = p.w0;
    } else {
      if (gi < 5) base = p.w0 + 64 * gi;
      else if (gi == 5) { base = p.res; ldb = (unsigned)p.ldres * 2; resid = 1; }
      else if (gi < FF0 + NCH * 8) {
        const int gg = gi - FF0, c = gg >> 3, j = gg & 7;
        if (j < 5) { base = p.wf1 + (size_t)c * (2 * XCHUNK) * XC + 64 * j; perm = 1; if (j == 0) vec = c; }
        else { base = p.wf2 + c * XCHUNK + 64 * (j - 5); ldb = 4 * XC * 2; half = (j == 7); }
      } else if (gi == FF0 + NCH * 8) { base = p.h_out; ldb = (unsigned)p.ldh * 2; resid = 1; }
      else if (gi < NTILES - 1) base = p.wpo + 64 * (gi - (FF0 + NCH * 8 + 1));
      else if (gi == NTILES - 1) { base = p.xres; ldb = (unsigned)p.ldxres * 2; resid = 1; }
      else base = p.w0;
    }
    ddst = smem + si * XSLOT + w * 1024;
    if (vec >= 0) {
      // the chunk's folded-LayerNorm vectors ride with its first weight tile: positions 0..79 = c1 slice, 80..159 = c2 slice
      const int pos = w * 64 + lane;
      const float* vs = pos < 80 ? p.c1 + vec * (2 * XCHUNK) + 4 * pos : pos < 160 ? p.c2 + vec * (2 * XCHUNK) + 4 * (pos - 80) : p.c1;
      __builtin_amdgcn_global_load_lds((gptr_t)vs, (lptr_t)(smem + XFFV_OFF + (w < 3 ? w : 3) * 1024), 16, 0, 0);
    }
    dres = resid;
    {   // the tile base is wave-uniform: pinned into SGPRs so that the loads take the (SGPR base + 32-bit VGPR offset) form - no
        // 64-bit address arithmetic per load
      const unsigned long long bq = (unsigned long long)(resid ? (const char*)base + (size_t)m0 * ldb : (const char*)base);
      const unsigned blo = __builtin_amdgcn_readfirstlane((unsigned)bq), bhi = __builtin_amdgcn_readfirstlane((unsigned)(bq >> 32));
      dbase = (const char*)(((unsigned long long)bhi << 32) | blo);
    }
    if (resid) {
      dstep = __builtin_amdgcn_readfirstlane(ldb);
    } else {
      // (arithmetic selects: a ?: between the captured per-thread constants becomes a select of their stack ADDRESSES - two
      //  dependent scratch / flat loads with vmcnt(0) in the hot loop, which also drained the DMA queue every step)
      const unsigned row = (unsigned)d_row + (unsigned)perm * (unsigned)(d_prow - d_row);
      const unsigned cs = (unsigned)d_csrc & (half ? 3u : 7u);       // half tile (32 valid k): both halves fetch the valid one
      doff = row * ldb + cs * 16; dstep = __builtin_amdgcn_readfirstlane(64 * ldb);
    }
    ++gi; si = (si == 2) ? 0 : si + 1;
  };
  auto dma = [&](const int i) {
    if constexpr (ABL & 2) { if (gi > 2) return; }
    __builtin_amdgcn_global_load_lds((gptr_t)((dbase + (size_t)i * dstep) + doff), (lptr_t)(ddst + i * 8192), 16, 0, 0);   // (SGPR base + i * SGPR step) + 32-bit VGPR offset
  };
  auto dma_all = [&]() {                                             // (the only way residual tiles are issued: their steps have no MFMA phase)
    if constexpr (ABL & 2) { if (gi > 2) return; }
    if (dres) {
#pragma unroll
      for (int i = 0; i < 5; ++i) {
        const int pch = t + 512 * i, row = pch / 40, ch = pch - row * 40;
        __builtin_amdgcn_global_load_lds((gptr_t)(dbase + ((unsigned)row * dstep + (unsigned)(xsw(ch, row) << 4))), (lptr_t)(ddst + i * 8192), 16, 0, 0);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 5; ++i) dma(i);
    }
  };

  // ---- per-lane geometry
  int ml[2];                                                         // block-local rows of this lane (fragment / accumulator row)
#pragma unroll
  for (int b = 0; b < 2; ++b) ml[b] = 32 * wm + 16 * b + lr;
  int wad[2];                                                        // weight fragment offsets inside a slot: rows 80 wn + 16 a + lr -> + a * 2048 (the swizzle key (n >> 1) & 7 does not depend on a)
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) wad[kk] = (80 * wn + lr) * 128 + (((4 * kk + lh) ^ ((lr >> 1) & 7)) << 4);
  int xoff[2];                                                       // exchange image: row base + swizzle key
#pragma unroll
  for (int b = 0; b < 2; ++b) xoff[b] = ml[b] * (XC * 2);

  bf16x8 xf[2][10];                                                  // activation operand of the running GEMM: rows ml[b], k = 32 ks + 8 lh .. +8
  f32x4 acc[5][2];
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  auto clear = [&](f32x4 (&c)[5][2]) {
#pragma unroll
    for (int a = 0; a < 5; ++a) { c[a][0] = zero4; c[a][1] = zero4; }
  };
  // one [320][64] weight tile against k-steps 2J, 2J+1 of the register operand, with the five loads of the prepared tile between the MFMAs
  // rn: the prepared tile is a residual tile (its loads are not affine in i: issued as a burst up front, not between the MFMAs)
  auto mma_tile = [&](const char* st, const int J, f32x4 (&c)[5][2], const bool rn = false) {
    if constexpr (ABL & 1) { dma_all(); return; }
    if (rn) dma_all();
    // the five loads are STAGGERED by wave: wave w issues load i behind MFMA 4 i + (w & 3) of the tile's twenty, so that at any
    // moment only the two waves of one SIMD (which alternate on its matrix pipe anyway) reach for the address path - issued at
    // the same program point by all eight waves every LDS-DMA instruction queued behind seven others (~200 cycles each).
    // (mode 1; in mode 0's short loops the pinned static order measured better: 2.6 vs 3.1 us per five tiles)
    const int wq = w & 3;
    if constexpr (MODE != 1) {
      bf16x8 wf[2][5];
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int a = 0; a < 5; ++a) wf[kk][a] = *(const bf16x8*)(st + wad[kk] + a * 2048);
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int a = 0; a < 5; ++a) {
          c[a][0] = DMX_MFMA_16x16x32(wf[kk][a], xf[0][2 * J + kk], c[a][0]);
          c[a][1] = DMX_MFMA_16x16x32(wf[kk][a], xf[1][2 * J + kk], c[a][1]);
          if (!rn && ((kk * 5 + a) & 1)) dma((kk * 5 + a) >> 1);
        }
      // pin the order: the ten fragment reads first (the second k-step's land under the first one's MFMAs), then 4 MFMAs + 1 DMA, five times
      __builtin_amdgcn_sched_group_barrier(0x100, 10, 0);
      if (rn) __builtin_amdgcn_sched_group_barrier(0x008, 20, 0);
      else {
#pragma unroll
        for (int q = 0; q < 5; ++q) {
          __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
          __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
        }
      }
    } else {                                                         // (the feed-forward loop holds two accumulator sets: one k-step of fragments at a time)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        bf16x8 wf[5];
#pragma unroll
        for (int a = 0; a < 5; ++a) wf[a] = *(const bf16x8*)(st + wad[kk] + a * 2048);
#pragma unroll
        for (int a = 0; a < 5; ++a) {
#pragma unroll
          for (int b = 0; b < 2; ++b) {
            c[a][b] = DMX_MFMA_16x16x32(wf[a], xf[b][2 * J + kk], c[a][b]);
            const int m = kk * 10 + 2 * a + b;
            if (!rn && (m & 3) == wq) dma(m >> 2);
          }
        }
      }
    }
  };
  int sc = 0;                                                        // ring slot of the tile being computed
#define XSTEP(...)                                                                                               \
  {                                                                                                              \
    asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)" ::: "memory");   /* this wave's pieces of the tile have landed (the next tile's five may be in flight); its LDS reads of the previous tile are complete */ \
    __builtin_amdgcn_s_barrier();                      /* everybody's pieces; and everybody is done with the previous tile's slot */ \
    prepare_next();                                                                                              \
    const char* st = smem + sc * XSLOT;                                                                          \
    __VA_ARGS__;                                                                                                 \
    sc = (sc == 2) ? 0 : sc + 1;                                                                                 \
  }

  // ---- epilogue pieces
  float mean[2] = {0.f, 0.f}, rstd[2] = {1.f, 1.f};
  u32x2 hq[5][2];                                                    // the phase's output, rounded: 4 consecutive channels per (a, b)
  // out = [rstd * (acc - mean * c1) +] bias [+ res]; optional row statistics of the rounded values
  auto finish = [&](const f32x4 (&c)[5][2], const int vbias, const int vc1, const bool res, const bool stats) {
    float ps[2] = {0.f, 0.f}, pq[2] = {0.f, 0.f};
    const char* X = smem + (sc == 0 ? 2 : sc - 1) * XSLOT;           // the residual tile (the slot computed last), in the exchange layout
    const float* vec = (const float*)(smem + (MODE == 2 ? XFF_OFF : XVEC_OFF));
#pragma unroll
    for (int a = 0; a < 5; ++a) {
      const int n = 80 * wn + 16 * a + 4 * lh;
      const f32x4 bv = *(const f32x4*)(vec + vbias * XC + n);
      f32x4 cv = zero4;
      if (vc1 >= 0) cv = *(const f32x4*)(vec + vc1 * XC + n);
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = vc1 >= 0 ? rstd[b] * (c[a][b][r] - mean[b] * cv[r]) + bv[r] : c[a][b][r] + bv[r];
        if (res) {
          const u32x2 rq = *(const u32x2*)(X + xoff[b] + (xsw(n >> 3, ml[b]) << 4) + (n & 7) * 2);
          v[0] += h2f_lo(rq[0]); v[1] += h2f_hi(rq[0]); v[2] += h2f_lo(rq[1]); v[3] += h2f_hi(rq[1]);
        }
        hq[a][b] = (u32x2){pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
        if (stats) {
          const float f0 = h2f_lo(hq[a][b][0]), f1 = h2f_hi(hq[a][b][0]), f2 = h2f_lo(hq[a][b][1]), f3 = h2f_hi(hq[a][b][1]);
          ps[b] += (f0 + f1) + (f2 + f3);
          pq[b] += (f0 * f0 + f1 * f1) + (f2 * f2 + f3 * f3);
        }
      }
    }
    if (stats) {
      float* sp = (float*)(smem + XST_OFF);
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        ps[b] += __shfl_xor(ps[b], 16); pq[b] += __shfl_xor(pq[b], 16);
        ps[b] += __shfl_xor(ps[b], 32); pq[b] += __shfl_xor(pq[b], 32);
        if (lh == 0) { sp[(ml[b] * 4 + wn) * 2] = ps[b]; sp[(ml[b] * 4 + wn) * 2 + 1] = pq[b]; }
      }
    }
  };
  // hq -> the exchange image in the slot computed last -> (optionally) a coalesced global copy -> the next phase's fragments
  auto exchange = [&](bf16* gout, int ldg, bool stats, bool reload) {
    char* X = smem + (sc == 0 ? 2 : sc - 1) * XSLOT;
    __builtin_amdgcn_s_barrier();                      // every wave has finished reading that slot's weight tile
#pragma unroll
    for (int a = 0; a < 5; ++a) {
      const int n = 80 * wn + 16 * a + 4 * lh;
#pragma unroll
      for (int b = 0; b < 2; ++b) *(u32x2*)(X + xoff[b] + (xsw(n >> 3, ml[b]) << 4) + (n & 7) * 2) = hq[a][b];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (reload) {
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int ks = 0; ks < 10; ++ks) xf[b][ks] = *(const bf16x8*)(X + xoff[b] + (xsw(4 * ks + lh, ml[b]) << 4));
    }
    if (stats) {
      const float* sp = (const float*)(smem + XST_OFF);
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const float* q = sp + ml[b] * 8;
        const float s = (q[0] + q[2]) + (q[4] + q[6]), sq = (q[1] + q[3]) + (q[5] + q[7]);
        const float mu = s * (1.0f / XC);
        float var = sq * (1.0f / XC) - mu * mu; var = var < 0.f ? 0.f : var;
        mean[b] = mu; rstd[b] = rsqrtf(var + p.eps);
      }
    }
    if (gout) {
#pragma unroll
      for (int i = 0; i < 5; ++i) {
        const int pch = t + 512 * i, row = pch / 40, ch = pch - row * 40;
        const u32x4 v = *(const u32x4*)(X + row * (XC * 2) + (xsw(ch, row) << 4));
        *(u32x4*)(gout + (size_t)(m0 + row) * ldg + ch * 8) = v;
      }
      if (MODE == 1 && p.colstats && gout == p.y) {
        // GroupNorm statistics of the block output for its next reader (the fused GroupNorm -> conv launch of conv_halo.hip): per-channel
        // (sum, sum of squares) of the ROUNDED values in the exchange image, 8 row groups x 40 octets, folded in a fixed order through
        // the (idle) feed-forward chunk buffer, one DmxStat add per channel and block
        float* red = (float*)(smem + XFF_OFF);         // [8][320][2]
        if (t < 320) {
          const int o = t % 40, rg = t / 40;
          float sa[8], sq[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) { sa[e] = 0.f; sq[e] = 0.f; }
#pragma unroll
          for (int r8 = 0; r8 < 8; ++r8) {
            const int row = rg * 8 + r8;
            float f[8]; unpack_bf8(*(const u32x4*)(X + row * (XC * 2) + (xsw(o, row) << 4)), f);
#pragma unroll
            for (int e = 0; e < 8; ++e) { sa[e] += f[e]; sq[e] += f[e] * f[e]; }
          }
#pragma unroll
          for (int e = 0; e < 8; ++e) { red[((rg * XC) + o * 8 + e) * 2] = sa[e]; red[((rg * XC) + o * 8 + e) * 2 + 1] = sq[e]; }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (t < XC) {
          float sa = 0.f, sq = 0.f;
#pragma unroll
          for (int k = 0; k < 8; ++k) { sa += red[(k * XC + t) * 2]; sq += red[(k * XC + t) * 2 + 1]; }
          dmx_stat_add(p.colstats + ((size_t)(m0 / p.cs_rows) * XC + t) * DMX_STAT_WORDS, sa, sq);
        }
      }
    }
    // (the next XSTEP's barrier comes before the DMA that reuses this slot)
  };

  long long* tmo = p.timing ? p.timing + (size_t)blockIdx.x * 8 : nullptr;      // measurement aid: s_memrealtime (10 ns) at the phase boundaries
  auto stamp = [&](int i) { if (tmo && t == 0) tmo[i] = (long long)__builtin_amdgcn_s_memrealtime(); };
  stamp(0);
  // ---- weight prefetch for the launches that follow (XfChainArgs.pf): 1-KB units over (block, wave), the oldest requests of each wave
  int pf_left = 4;                                     // (at most four units per wave)
#pragma unroll
  for (int r_ = 0; r_ < 4; ++r_) {
    const int nb_ = p.pf_bytes[r_];
    for (int u_ = blockIdx.x * 8 + w; u_ * 1024 < nb_ && pf_left > 0; u_ += gridDim.x * 8, --pf_left) {
      int off_ = u_ * 1024 + lane * 16; if (off_ > nb_ - 16) off_ = nb_ - 16;
      __builtin_amdgcn_global_load_lds((gptr_t)((const char*)p.pf[r_] + off_), (lptr_t)(smem + XDUMP), 16, 0, 0);
    }
  }
  // ---- start: two tiles in flight, the first operand straight from global into fragments
  prepare_next(); dma_all(); prepare_next(); dma_all();
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const bf16* xp = p.x + (size_t)(m0 + ml[b]) * p.ldx + 8 * lh;
#pragma unroll
    for (int ks = 0; ks < 10; ++ks) xf[b][ks] = *(const bf16x8*)(xp + 32 * ks);
  }
  if constexpr (MODE == 2) {                                         // b0 | c1[3C] | c2[3C] -> the (otherwise unused) GEGLU chunk buffer
    for (int e = t; e < 7 * 80; e += 512) {
      const int v = e / 80, q4 = e - v * 80;
      const float* vsrc = v == 0 ? p.b0 : v < 4 ? p.c1 + (v - 1) * XC : p.c2 + (v - 4) * XC;
      *(f32x4*)(smem + XFF_OFF + (v * XC + 4 * q4) * 4) = *(const f32x4*)(vsrc + 4 * q4);
    }
  } else if (t < 240) {                                              // the three column vectors of the 320-wide epilogues -> LDS
    const int v = t / 80, q4 = t - v * 80;
    const float* vsrc = MODE == 0 ? (v == 0 ? p.b0 : v == 1 ? p.c1 : p.c2) : (v == 0 ? p.b0 : v == 1 ? p.bf2 : p.bpo);
    *(f32x4*)(smem + XVEC_OFF + (v * XC + 4 * q4) * 4) = *(const f32x4*)(vsrc + 4 * q4);
  }
  if constexpr (MODE == 2) {
    if (p.gn_st) {
      // GroupNorm of the transformer entry folded into the operand load: per-channel affine a = rstd * gamma, s = beta - mean * a of this
      // block's sample from the statistics records of x (group reduction and variance in double, the order of dmx_gn_apply_kernel), then
      // y = x * a + s on the fragments - one launch and one round trip of the 64x64 tensor less per transformer block
      double* cs = (double*)(smem + XFF_OFF + 12288);                // [2][320] (behind the seven column vectors)
      float* coef = (float*)(smem + XVEC_OFF);                       // a[320] | s[320]
      const int smp = m0 / p.gn_rows;
      if (t < XC) {
        const long long* q = p.gn_st + ((size_t)smp * XC + t) * DMX_STAT_WORDS;
        cs[t] = dmx_stat_sum(q[0]); cs[XC + t] = dmx_stat_sumsq(q[1], q[2]);
      }
      __syncthreads();
      if (t < XC) {
        const int cpg = XC / p.gn_groups, g = t / cpg;
        double a = 0.0, q = 0.0;
        for (int k = 0; k < cpg; ++k) { a += cs[g * cpg + k]; q += cs[XC + g * cpg + k]; }
        const double inv_n = 1.0 / ((double)p.gn_rows * (double)cpg);
        const double mean = a * inv_n;
        double var = q * inv_n - mean * mean;
        var = var < 0.0 ? 0.0 : var;
        const float mf = (float)mean, rf = (float)(1.0 / __builtin_sqrt(var + (double)p.gn_eps));
        const float A = rf * p.gn_gamma[t];
        coef[t] = A; coef[XC + t] = p.gn_beta[t] - mf * A;
      }
      __syncthreads();
#pragma unroll
      for (int ks = 0; ks < 10; ++ks) {
        const int c0 = 32 * ks + 8 * lh;
        const f32x4 a0 = *(const f32x4*)(coef + c0), a1 = *(const f32x4*)(coef + c0 + 4);
        const f32x4 s0 = *(const f32x4*)(coef + XC + c0), s1 = *(const f32x4*)(coef + XC + c0 + 4);
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          float f[8]; unpack_bf8(__builtin_bit_cast(u32x4, xf[b][ks]), f);
#pragma unroll
          for (int i = 0; i < 8; ++i) f[i] = f[i] * (i < 4 ? a0[i] : a1[i - 4]) + (i < 4 ? s0[i] : s1[i - 4]);
          xf[b][ks] = __builtin_bit_cast(bf16x8, pack_bf8(f));
        }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int ks = 0; ks < 10; ++ks) asm volatile("" : "+v"(xf[b][ks]));      // fresh values: no pending-load bookkeeping follows them into the loop

  stamp(1);
  // ---- phase 1: h = x W0^T + b0 + res
  clear(acc);
  if constexpr (MODE == 2) {
    XSTEP(mma_tile(st, 0, acc)) XSTEP(mma_tile(st, 1, acc)) XSTEP(mma_tile(st, 2, acc)) XSTEP(mma_tile(st, 3, acc)) XSTEP(mma_tile(st, 4, acc))
    stamp(2);
    finish(acc, 0, -1, false, true);
  } else {
    XSTEP(mma_tile(st, 0, acc)) XSTEP(mma_tile(st, 1, acc)) XSTEP(mma_tile(st, 2, acc)) XSTEP(mma_tile(st, 3, acc, true)) XSTEP(mma_tile(st, 4, acc))
    XSTEP(dma_all())                                       // the residual tile
    stamp(2);
    finish(acc, 0, -1, true, true);
  }
  exchange(p.h_out, p.ldh, true, true);
  stamp(3);

  if constexpr (MODE == 0) {
    // ---- phase 2: y = LN(h) W1'^T  (folded: rstd * (h W1'^T - mean * c1) + c2)
    clear(acc);
    XSTEP(mma_tile(st, 0, acc)) XSTEP(mma_tile(st, 1, acc)) XSTEP(mma_tile(st, 2, acc)) XSTEP(mma_tile(st, 3, acc)) XSTEP(mma_tile(st, 4, acc))
    stamp(4);
    finish(acc, 2, 1, false, false);
    exchange(p.y, p.ldy, false, false);
  } else if constexpr (MODE == 2) {
    // ---- q | k | v = LN(h) W1'^T, 320 output columns at a time on the same register fragments
#pragma unroll
    for (int sl = 0; sl < 3; ++sl) {
      clear(acc);
      XSTEP(mma_tile(st, 0, acc)) XSTEP(mma_tile(st, 1, acc)) XSTEP(mma_tile(st, 2, acc)) XSTEP(mma_tile(st, 3, acc)) XSTEP(mma_tile(st, 4, acc))
      if (sl == 0) stamp(4);
      finish(acc, 4 + sl, 1 + sl, false, false);
      exchange(p.y + sl * XC, p.ldy, false, false);
    }
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the copy of h (the residual of the feed-forward, read back below) has left this wave
    // ---- feed-forward over 8 chunks of 160 hidden columns: FF1 chunk -> GEGLU -> LDS -> partial FF2 into accf
    f32x4 accf[5][2];
    clear(accf);
    char* FB = smem + XFF_OFF;
    // one [320][64] tile of FF2 against k-steps 2JJ (.. 2JJ + NKK - 1) of the GEGLU chunk in LDS, carrying the prepared tile's five loads
    auto ff2_tile = [&](const char* st, const int JJ, const int NKK, const bool rn = false) {
      if constexpr (ABL & 1) { dma_all(); return; }
      if (rn) dma_all();
      const int wq = w & 3;
#pragma unroll
      for (int kk = 0; kk < NKK; ++kk) {
        bf16x8 wf[5], tf[2];
#pragma unroll
        for (int a = 0; a < 5; ++a) wf[a] = *(const bf16x8*)(st + wad[kk] + a * 2048);
#pragma unroll
        for (int b = 0; b < 2; ++b) tf[b] = *(const bf16x8*)(FB + ml[b] * XFF_LD + (xsw(4 * (2 * JJ + kk) + lh, ml[b]) << 4));
#pragma unroll
        for (int a = 0; a < 5; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b) {
            accf[a][b] = DMX_MFMA_16x16x32(wf[a], tf[b], accf[a][b]);
            const int m = kk * 10 + 2 * a + b;
            if (rn) continue;
            if (NKK == 2) { if ((m & 3) == wq) dma(m >> 2); }       // staggered by wave like mma_tile
            else if ((m & 1) == (wq & 1)) dma(m >> 1);
          }
      }
    };
#pragma unroll 1
    for (int c = 0; c < NCH; ++c) {
      clear(acc);
      XSTEP(mma_tile(st, 0, acc)) XSTEP(mma_tile(st, 1, acc)) XSTEP(mma_tile(st, 2, acc)) XSTEP(mma_tile(st, 3, acc)) XSTEP(mma_tile(st, 4, acc))
      // GEGLU: accumulator quad = [value(2q), value(2q+1), gate(2q), gate(2q+1)], q = 20 wn + 4 a + lh
#pragma unroll
      for (int a = 0; a < 5; ++a) {
        const int hc = 2 * (20 * wn + 4 * a + lh);                   // hidden column inside the chunk (even)
        const int rv = 64 * (hc >> 5) + (hc & 31);                   // packed FF1 row of value(hc) inside the chunk; gate(hc) is 32 rows further
        const char* fv = smem + XFFV_OFF + rv * 4;
        const f32x2 c1v = *(const f32x2*)fv, c1g = *(const f32x2*)(fv + 32 * 4);
        const f32x2 c2v = *(const f32x2*)(fv + XC * 4), c2g = *(const f32x2*)(fv + (XC + 32) * 4);
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          const f32x2 av = {acc[a][b][0], acc[a][b][1]}, ag = {acc[a][b][2], acc[a][b][3]};
          const f32x2 v = __builtin_elementwise_fma(av - c1v * mean[b], (f32x2)(rstd[b]), c2v);
          const f32x2 g = __builtin_elementwise_fma(ag - c1g * mean[b], (f32x2)(rstd[b]), c2g);
          const f32x2 tt = v * gelu_erf_f2(g);
          *(unsigned int*)(FB + ml[b] * XFF_LD + (xsw(hc >> 3, ml[b]) << 4) + (hc & 7) * 2) = pack_bf2(tt.x, tt.y);
        }
      }
      // partial FF2: K = 160 = tiles of 64, 64, 32 (the next XSTEP's barrier orders the chunk-buffer writes before these reads)
      XSTEP(ff2_tile(st, 0, 2)) XSTEP(if (dres) ff2_tile(st, 1, 2, true); else ff2_tile(st, 1, 2)) XSTEP(ff2_tile(st, 2, 1))
    }
    XSTEP(dma_all())                                       // the residual tile: the copy of h this block wrote after phase 1
    stamp(4);
    // ---- h3 = ff + b2 + h (read back from the copy this block wrote after phase 1)
    finish(accf, 1, -1, true, false);
    exchange(nullptr, 0, false, true);
    stamp(5);
    // ---- y = h3 Wpo^T + bpo + x_res
    clear(acc);
    XSTEP(mma_tile(st, 0, acc)) XSTEP(mma_tile(st, 1, acc)) XSTEP(mma_tile(st, 2, acc)) XSTEP(mma_tile(st, 3, acc, true)) XSTEP(mma_tile(st, 4, acc))
    XSTEP(dma_all())                                       // the residual tile
    stamp(6);
    finish(acc, 2, -1, true, false);
    exchange(p.y, p.ldy, false, false);
  }
#undef XSTEP
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the two dummy tiles behind the last real one land before the LDS is released
  stamp(7);
}

size_t dmx_xf_chain_lds_bytes() { return XLDS; }

bool dmx_xf_chain_supported(int M, int C) { return C == XC && M > 0 && M % XBM == 0; }
// A block owns 64 rows, one block per CU, so the time of a launch goes with the number of ROUNDS of M / 64 blocks over the CUs,
// while the separate full-chip GEMMs go with M.  Measured (scripts/ab_pass.py xf_chain --batch b [--latent 96]; chains forced on
// vs off, ms per 50-step pass): 64 blocks 236 vs 224, 128 blocks 293 vs 293, 192 blocks 340 vs 346, 256 blocks 349 vs 359,
// 288 blocks (768 px, batch 2) 457 vs 442 - (round 3, before the prefetch plan) - see below.
bool dmx_xf_chain_pays(int M, int C) {
  if (!dmx_xf_chain_supported(M, C)) return false;
  static int n_cu = 0;
  if (!n_cu) { int dev = 0; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev); if (n_cu <= 0) n_cu = 256; }
  // (round 4, with the weight prefetch plan - the chains no longer wait for cold weights: 64 blocks (B = 1) 239.1 vs 231.4, 128 blocks (B = 2)
  // 286.8 vs 296.6, 192 blocks (B = 3) 328.6 vs 346.0, 288 blocks (768 px, batch 2) 484.2 vs 479.0 -> the LAST round at least half full)
  const int blocks = M / XBM, rounds = (blocks + n_cu - 1) / n_cu, last = blocks - (rounds - 1) * n_cu;
  return 2 * last >= n_cu;
}

int dmx_xf_chain_launch(const XfChainArgs& a, int mode, hipStream_t stream) {
  DMX_REQUIRE(mode >= 0 && mode <= 2, "xf_chain: mode %d", mode);
  DMX_REQUIRE(dmx_xf_chain_supported(a.M, a.C), "xf_chain: M=%d C=%d unsupported (C = 320, M %% 64 == 0)", a.M, a.C);
  DMX_REQUIRE(a.x && (a.res || mode == 2) && a.w0 && a.b0 && a.h_out && a.y && a.c1 && a.c2, "xf_chain: null operand");
  if (a.colstats) DMX_REQUIRE(mode == 1 && a.cs_rows > 0 && a.cs_rows % 64 == 0 && a.M % a.cs_rows == 0, "xf_chain: output statistics need mode 1 and samples of a multiple of 64 rows");
  DMX_REQUIRE(a.ldx % 8 == 0 && a.ldres % 4 == 0 && a.ldh % 8 == 0 && a.ldy % 8 == 0, "xf_chain: row strides must be multiples of 8 elements");
  if (a.gn_st) DMX_REQUIRE(mode == 2 && a.gn_gamma && a.gn_beta && a.gn_groups > 0 && a.C % a.gn_groups == 0 && a.gn_rows > 0 && a.gn_rows % 64 == 0 && a.M % a.gn_rows == 0,
                           "xf_chain: the folded GroupNorm needs mode 2, gamma / beta, groups dividing C and samples of a multiple of 64 rows");
  if (mode != 1) DMX_REQUIRE(a.w1 != nullptr, "xf_chain: null operand");
  else DMX_REQUIRE(a.wf1 && a.wf2 && a.bf2 && a.wpo && a.bpo && a.xres && a.ldxres % 4 == 0, "xf_chain: null operand");
  const double M = a.M, C = a.C;
  const double flops = mode == 0 ? 2.0 * M * C * C * 2 : mode == 2 ? 2.0 * M * C * C * 4 : 2.0 * M * C * C * 2 + 2.0 * M * C * (8 * C) + 2.0 * M * (4 * C) * C;
  const double bytes = mode == 0 ? 2.0 * (4 * M * C + 2 * C * C) : mode == 2 ? 2.0 * (5 * M * C + 4 * C * C) : 2.0 * (4 * M * C + 2 * C * C + 12 * C * C);
  char tag[64]; snprintf(tag, sizeof(tag), "M=%d C=%d mode=%d", a.M, a.C, mode);
  ProfScope ps(PROF_XFCHAIN, stream, flops, bytes, tag);
  const dim3 grid(a.M / XBM), block(512);
#define XLAUNCH(MODE_, ABL_)                                                                       \
  {                                                                                                \
    DMX_LDS_OPT_IN((dmx_xf_chain_kernel<MODE_, ABL_>), XLDS);                                      \
    hipLaunchKernelGGL((dmx_xf_chain_kernel<MODE_, ABL_>), grid, block, XLDS, stream, a);          \
    dmx_profile_note_symbol("void dmx_xf_chain_kernel<" #MODE_ ", " #ABL_ ">(XfChainArgs)");       \
  }
#ifdef DMX_PROBES
  if (a.dbg & 3) {
    const int k = a.dbg & 3;
    if (mode == 0) { if (k == 1) XLAUNCH(0, 1) else if (k == 2) XLAUNCH(0, 2) else XLAUNCH(0, 3) }
    else if (mode == 2) { if (k == 1) XLAUNCH(2, 1) else if (k == 2) XLAUNCH(2, 2) else XLAUNCH(2, 3) }
    else { if (k == 1) XLAUNCH(1, 1) else if (k == 2) XLAUNCH(1, 2) else XLAUNCH(1, 3) }
    return dmx_check_launch("dmx_xf_chain_kernel");
  }
#else
  DMX_REQUIRE(a.dbg == 0, "xf_chain: the ablation switches exist in -DDMX_PROBES builds only");
#endif
  if (mode == 0) XLAUNCH(0, 0) else if (mode == 2) XLAUNCH(2, 0) else XLAUNCH(1, 0)
#undef XLAUNCH
  return dmx_check_launch("dmx_xf_chain_kernel");
}
