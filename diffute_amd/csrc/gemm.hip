// Fused implicit-GEMM conv / linear kernel for gfx950 (bf16 MFMA, fp32 accumulate).
//
// One kernel covers SURVEY.md 8a rows K1 (conv3x3 s1), K1s (stride 2, incl. the VAE's
// asymmetric pad), K2 (conv1x1 / resnet shortcut), K7 (linear), K8 (GEGLU feed-forward)
// and K10 (nearest x2 upsample and skip-concat folded into the operand gather).
//
//   out[m][n] = epilogue( sum_k X[m][k] * W[n][k] )
//
// X is never materialised: row m is an output pixel of an NHWC tensor and k walks
// (tap, channel) of the 3x3 window, read straight from the activation tensor(s) with
// `global_load_lds` (16 B per lane, DMA into LDS).  Padding / tails read a zero page.
// A second source splits the channel range (UNet skip concat) and an optional third
// K-segment appends the 1x1 shortcut of a ResnetBlock2D so conv2 + shortcut + residual
// is one launch.
//
// Tiling (template dmx_gemm_kernel<WM, TN, BKT, NSTAGE, TM, NP, NWN>): NWN*WM MFMA waves (WM along m x NWN = 2 along n), each a
// (32*TM) x (32*TN) sub-tile of v_mfma_f32_32x32x16_bf16; NP extra waves only issue DMA (warp specialisation).  The
// weight tile is the MFMA A operand (rows = n) and the activation tile the B operand (cols = m) so each lane ends up
// with 4 consecutive output channels of one pixel.  LDS tiles are [rows][BKT] bf16 (64- or 128-byte rows) with the
// 16-byte k-chunk XOR-swizzled on the SOURCE side (glds writes lane-linear), which makes the ds_read_b128 fragment
// reads bank-conflict free.  An NSTAGE-deep LDS ring keeps NSTAGE-1 K-tiles of DMA in flight behind a counted
// `s_waitcnt vmcnt(N)` + raw s_barrier; the queue is never drained inside the loop.  The instances and what each is for
// are listed at kCfg below; DESIGN.md section 4 has the measurements behind them.
#include "common.h"
#include "kernels.h"
#include <stdio.h>
#include <vector>

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

struct RowSrc {          // per-thread state for one staged activation row
  int bbase;             // b * IH * IW   (pixel index base) or plain row index
  int iy0, ix0;          // oy*stride - pad_t, ox*stride - pad_l
  int valid;             // m < M
};

// Tile configurations (plan id -> instance), see kCfg:
//   0 <2,2,32,4>      128x128x32, 4 waves, 64 KB ring -> 2 blocks/CU   (mid-size GEMMs, split-K convolutions)
//   1 <2,1,32,4>      128x 64x32, 4 waves, 48 KB                        (N <= 64)
//   2 <4,2,64,3>      256x128x64, 8 waves, 144 KB -> 1 block/CU         (large GEMMs: 128-byte DMA rows, 85 FLOP/B)
//   6 <2,2,64,3,4,4>  warp-specialised 256x128x64: waves 0-3 MFMA (128x64 sub-tiles, software-pipelined fragment reads),
//                     waves 4-7 issue the LDS-DMA refills                (large K)
//   7..9 <4,1,64,3,1> / <4,2,32,4,1> / <4,2,64,3,1>  128x64 / 128x128 tiles with EIGHT waves (32x32 / 32x64 sub-tiles): small
//                     grids are bound by the per-CU LDS-DMA fill rate, which doubles with 8 issuing waves
//   10 <4,5,64,2,1,0,1> 128x160x64: FOUR waves in one column (NWN = 1), 32x160 sub-tiles, 2-buffer ring (72 KB -> 2 blocks/CU).
//                     160 divides every channel count of the UNet (320 k): no wasted tile columns where N = 320 / 640 / 960;
//                     its own coalesced epilogue (20 octets per row do not divide the block); bias / row bias / residual only
//   11 <4,5,64,2,1,0,2> 128x320x64: eight waves (4 x 2) of 32x160 sub-tiles, 2-buffer ring (112 KB, 1 block/CU); five whole packed
//                     GEGLU groups per tile, same epilogue as 10 plus folded LayerNorm + GEGLU: the FF1 GEMM of the 16x16 level
//                     (1024 x 10240 x 1280) becomes 256 tiles = one full round instead of 640 tiles on 512 slots
//   3, 4, 5           retired (two-stage 128x64, deep-ring 128x128x64, 256x256x32: measured, not adopted - EXPERIMENTS.md)
// measurement aid: 100 MHz wall ticks, or (dbg bit 2) shader-clock cycles - their ratio is the effective clock
__device__ __forceinline__ long long dmx_now(int dbg) {
  return (dbg & 4) ? (long long)__builtin_amdgcn_s_memtime() : (long long)__builtin_amdgcn_s_memrealtime();
}
// CS: this instance also emits the GroupNorm column statistics of its output (GemmArgs.colstats) - a separate instantiation so
// that the 16 registers the partials cost do not touch the occupancy of the plain instances
template <int WM, int TN, int BKT, int NSTAGE, int TM = 2, int NP = 0, int NWN = 2, bool PS = false, int MF = 32, bool CS = false>
__global__ __launch_bounds__(64 * (NWN * WM + NP), (TN == 5 && NWN == 2) ? 1 : ((WM == 4 && TM == 1 && TN <= 2 && NP == 0 && NWN == 2) ? 4 : 2)) void dmx_gemm_kernel(const GemmArgs p) {
  static_assert(NWN == 1 || NWN == 2, "one or two waves along n");
  // MF = MFMA fragment size: 32 -> v_mfma_f32_32x32x16 (wave tile 32 TM x 32 TN), 16 -> v_mfma_f32_16x16x32 (16 TM x 16 TN: the
  // 64 x 80 wave tiles of the 256x160 persistent instance - 25 % fewer LDS fragment bytes per FLOP than 32 x 160 wave tiles)
  static_assert(MF == 32 || (MF == 16 && TN == 5 && PS && NP == 0 && BKT == 64), "16x16 fragments: 160-column persistent instance only");
  constexpr int NGRP = MF == 32 ? 4 : 1;               // float4 groups per accumulator fragment and lane
  typedef float accv __attribute__((ext_vector_type(4 * NGRP)));
  constexpr int BM = MF * TM * WM, BN = MF * TN * NWN;
  constexpr int NC = NWN * WM;                         // consumer (MFMA) waves: WM along m x NWN along n
  constexpr int NT = 64 * (NC + NP);                   // block threads
  constexpr int NL = NP ? 64 * NP : NT;                // threads that stage tiles (all of them unless warp-specialised)
  constexpr int CPR = BKT / 8;                         // 16-byte chunks per LDS row
  constexpr int ROWB = BKT * 2;                        // LDS row bytes
  constexpr int RSTEP = NL / CPR;                      // row distance between a thread's consecutive chunks
  constexpr int XL = BM * CPR / NL;                    // DMA loads per thread per K-tile (activations)
  constexpr int WL = (BN * CPR + NL - 1) / NL;         // ... and weights; when BN*CPR is not a multiple of the staging threads (160 rows
  constexpr int WREM = BN * CPR - NL * (WL - 1);       // over 512 threads) the last round covers WREM chunks: the upper waves repeat the lower
                                                       // waves' loads (same bytes to the same LDS address) so every thread issues NLOADS
  static_assert(BM * CPR % NL == 0 && (WREM & (WREM - 1)) == 0 && WREM % 64 == 0, "staging rounds");
  constexpr int NLOADS = XL + WL;
  constexpr int X_BYTES = BM * ROWB, W_BYTES = BN * ROWB, STAGE = X_BYTES + W_BYTES;
  constexpr int KSTEPS = BKT / (MF == 32 ? 16 : 32);   // MFMA k-steps per K-tile
  constexpr int CPS = 64 / MF;                         // 16-byte k-chunks per k-step and fragment row (2 lane halves / 4 lane quarters)
  static_assert(XL >= 1 && WL >= 1, "tile too small for the block");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  auto swz = [](int r) { return BKT == 32 ? ((r >> 2) & 3) : ((r >> 1) & 7); };

  long long tm0 = 0, tm1 = 0, tm2 = 0;
  if (p.timing) tm0 = dmx_now(p.dbg);
  // weight prefetch for the next launches (GemmArgs.pf): 1-KB units dealt over (block, wave); they are the oldest requests of the
  // wave, so every counted vmcnt wait of the K loop covers them and the vmcnt(0) behind the loop drains them.  Only waves that stage
  // tiles issue them (the consumer waves of the warp-specialised instance never wait on vmcnt: a request of theirs could still be
  // writing the dump slot after the block has gone) - in the warp-specialised instance the call sits INSIDE the loader branch, so that
  // "every path from an LDS-DMA request to s_endpgm passes a vmcnt(0)" holds path by path (scripts/isa_audit.py rule L checks the binary)
  auto issue_prefetch = [&]() {
    if (p.pf_dump_off <= 0) return;
    const int nblk_ = gridDim.x * gridDim.y, blk_ = blockIdx.y * gridDim.x + blockIdx.x, wv_ = (int)(threadIdx.x >> 6) - (NP > 0 ? NC : 0);
    constexpr int NWV = NP > 0 ? NP : NT / 64;           // issuing waves per block
    int left_ = 4;                                     // at most four units per wave: a launch of few blocks (batch 1) must not turn into a weight stream
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int nb = p.pf_bytes[r];
      for (int u = blk_ * NWV + wv_; u * 1024 < nb && left_ > 0; u += nblk_ * NWV, --left_) {
        int off = u * 1024 + (threadIdx.x & 63) * 16; if (off > nb - 16) off = nb - 16;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)((const char*)p.pf[r] + off),
                                         (__attribute__((address_space(3))) void*)(smem + p.pf_dump_off), 16, 0, 0);
      }
    }
  };
  if constexpr (NP == 0) issue_prefetch();

  // ---- work items.  Classic launch: one block = one (tile, K-slice) - tile from blockIdx.x, split-K slice from blockIdx.y.
  // Persistent stream-K launch (p.persist; grid = one block per CU): the flattened (tile, K-tile) iteration space is cut
  // into gridDim.x equal contiguous ranges; a block walks its range tile by tile.  A range that starts inside a tile makes
  // the block a HELPER for that tile (its first work item): it parks its fp32 accumulators in slab[position] and raises
  // flag[position].  The block whose range holds the tile's K-tile 0 is the tile's OWNER: that item is the LAST of its
  // range, so by then the helpers (who met the tile FIRST in theirs) are long done; it adds their slabs in K order (fixed
  // order -> deterministic) and runs the epilogue.  No reduce pass, no tile quantisation.
  // Blocks are dealt round-robin to the 8 XCDs, so first give every XCD a contiguous range (of tile ids / of the iteration
  // space), then rasterise tile ids in GROUP_M x tiles_n super-tiles: the blocks resident on one XCD at a time cover a
  // compact (m, n) patch and share both their activation rows and their weight rows through that XCD's L2 (otherwise
  // every n-tile of a conv re-reads the whole activation tensor from HBM / Infinity Cache).
  const int nblk = gridDim.x;
  // ups2 (phase-decomposed nearest-x2 upsample + 3x3 conv): the m-tiles are 4 phase blocks of tiles over the SOURCE grid
  const int Mlim = p.ups2 ? p.M4 : p.M;               // rows of the gathered operand (per phase)
  const int tiles_mp = (Mlim + BM - 1) / BM;
  const int tiles_m = p.ups2 ? 4 * tiles_mp : tiles_mp;
  const int tiles_n = (p.N + BN - 1) / BN;
  const int gm = p.group_m > 0 ? p.group_m : 8;
  const int width = gm * tiles_n;
  const int nkt_total = p.K / BKT;
  const int pos = ((nblk & 7) == 0) ? (blockIdx.x & 7) * (nblk >> 3) + (blockIdx.x >> 3) : blockIdx.x;
  const long long it_total = (long long)tiles_m * tiles_n * nkt_total;
  long long it, it_last;                               // [it, it_last): this block's part of the iteration space
  // range boundary of position q: it_total * q / nblk, and a boundary that falls INSIDE a tile is moved PS_BIAS K-tiles later:
  // the owner of a shared tile (the range holding its K-tile 0) gets a little more of it than the helpers, so the helpers have
  // published their slabs by the time the owner's K loop ends (with equal shares - e.g. 2 blocks per tile - the owner sat
  // waiting for the helper's store + release, measured ~4 us of a 46 us kernel)
  constexpr int PS_BIAS = 2;
  auto bound = [&](int q) -> long long {
    long long b = it_total * q / nblk;
    if (q > 0 && q < nblk) {
      const long long r = b % nkt_total;
      if (r != 0 && nkt_total >= 8 * PS_BIAS) b += (r + PS_BIAS < nkt_total) ? PS_BIAS : (nkt_total - 1 - r);
    }
    return b;
  };
  if constexpr (PS) {
    it = bound(pos); it_last = bound(pos + 1);
  } else {
    int kb = 0, ke = nkt_total;
    if (p.splitk > 1) { kb = blockIdx.y * p.kt_per_split; ke = min(kb + p.kt_per_split, nkt_total); }
    it = (long long)pos * nkt_total + kb; it_last = it + (ke - kb);
  }
  bool later_item = false;
  do {
  // (persistent instances: the thread index is made opaque per work item, so nothing derived from it is hoisted out of the
  // item loop and carried through the epilogue of every item - that costs 50+ VGPRs and spills)
  int t = threadIdx.x;
  if constexpr (PS) asm volatile("" : "+v"(t));
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const bool is_loader = (NP == 0) || wave >= NC, is_consumer = (NP == 0) || wave < NC;
  const int lt = NP ? t - 64 * NC : t;                 // index among the staging threads (loaders only)
  const int lwave = NP ? wave - NC : wave;
  const int wm = wave % WM, wn = NWN == 2 ? (wave / WM) & 1 : 0;
  const int tile_id = (int)(it / nkt_total);
  const int kt_begin = (int)(it - (long long)tile_id * nkt_total);
  const int kt_end = (int)min((long long)nkt_total, kt_begin + (it_last - it));
  it += kt_end - kt_begin;
  if (PS && later_item) {                              // the previous item's epilogue / slab pass is done with LDS
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  later_item = true;
  const int gid = tile_id / width;
  const int first_m = gid * gm;
  const int gsz = min(tiles_m - first_m, gm);
  const int rem_id = tile_id - gid * width;
  const int tile_mv = first_m + rem_id % gsz;
  const int tile_n = rem_id / gsz;
  const int phase = p.ups2 ? tile_mv / tiles_mp : 0;   // (output row parity, output column parity)
  const int pa = phase >> 1, pb = phase & 1;
  const int tile_m = tile_mv - phase * tiles_mp;
  const int m0 = tile_m * BM;
  const int n0 = tile_n * BN;
  // output row of gathered row m: identity, or pixel (2i+pa, 2j+pb) of the upsampled grid for source pixel (i, j)
  auto out_row = [&](int m) -> size_t {
    if (!p.ups2) return (size_t)m;
    const int q = m / p.IW;                            // n*IH + i
    return (size_t)2 * m + (size_t)2 * p.IW * q + (size_t)(pa * 2 * p.IW + pb);
  };

  // ---- per-thread staging rows (loader threads): 16-byte chunk q = lt + NL*i -> row q/CPR, slot q%CPR
  const int slot = lt % CPR;
  RowSrc xr[XL];
  int kcx[XL];
#pragma unroll
  for (int i = 0; i < XL; ++i) {
    const int r = lt / CPR + RSTEP * i;
    const int m = m0 + r;
    kcx[i] = (slot ^ swz(r)) * 8;
    xr[i].valid = m < Mlim;
    if (p.direct) {
      xr[i].bbase = m; xr[i].iy0 = 0; xr[i].ix0 = 0;
    } else {
      const int ohw = p.OH * p.OW;
      const int b = m / ohw;
      const int rem = m - b * ohw;
      const int oy = rem / p.OW;
      const int ox = rem - oy * p.OW;
      xr[i].bbase = b * p.IH * p.IW;
      xr[i].iy0 = p.ups2 ? oy - (1 - pa) : oy * p.stride - p.pad;     // ups2: the 2x2 window starts at (i - 1 + pa, j - 1 + pb)
      xr[i].ix0 = p.ups2 ? ox - (1 - pb) : ox * p.stride - p.pad;
    }
  }
  const int eh = p.ups ? 2 * p.IH : p.IH;   // extent of the (virtually upsampled) input grid
  const int ew = p.ups ? 2 * p.IW : p.IW;

  // ---- producer state.  K is walked in SEGMENTS inside which the source tensor and the filter tap are fixed
  // (tap x {x0,x1}, then the fused shortcut {s0,s1}); inside a segment staging a K-tile is NLOADS DMA loads from
  // per-row pointers, then pointer += one LDS row.  All address decoding lives in segment_setup (rare).
  const char* xp[XL]; int xinc[XL];
  const char* wp[WL]; int winc[WL];
#pragma unroll
  for (int i = 0; i < WL; ++i) {
    const int ltw = (i == WL - 1) ? (lt & (WREM - 1)) : lt;
    const int r = ltw / CPR + RSTEP * i;
    const int n = n0 + r;
    const int kc = (slot ^ swz(r)) * 8;
    if (n < p.N) { wp[i] = (const char*)(p.w + (size_t)phase * p.w_phase_stride + (size_t)n * p.ldw + (size_t)kt_begin * BKT + kc); winc[i] = ROWB; }
    else { wp[i] = (const char*)p.zeros; winc[i] = 0; }
  }
  int p_kt = kt_begin, p_left = 0;
  auto segment_setup = [&](int k0) {
    const bf16* src; int ld, ci, cend, dy = 0, dx = 0; bool sc = false;
    if (k0 < p.Ktaps) {
      int tap = 0; ci = k0;
      if (p.ksize == 3) { tap = k0 / p.Cin; ci = k0 - tap * p.Cin; dy = tap / 3; dx = tap - dy * 3; }
      else if (p.ksize == 2) { tap = k0 / p.Cin; ci = k0 - tap * p.Cin; dy = tap >> 1; dx = tap & 1; }
      if (ci < p.cx0) { src = p.x0 + ci; ld = p.ldx0; cend = p.cx0; } else { src = p.x1 + (ci - p.cx0); ld = p.ldx1; cend = p.Cin; }
    } else {
      sc = true; ci = k0 - p.Ktaps; const int ctot = p.K - p.Ktaps;
      if (ci < p.cs0) { src = p.s0 + ci; ld = p.lds0; cend = p.cs0; } else { src = p.s1 + (ci - p.cs0); ld = p.lds1; cend = ctot; }
    }
    p_left = (cend - ci) / BKT;
#pragma unroll
    for (int i = 0; i < XL; ++i) {
      const bf16* g = nullptr;
      if (xr[i].valid) {
        if (p.direct) {
          g = src + (size_t)xr[i].bbase * ld;
        } else if (sc) {       // shortcut: 1x1 at the output pixel (shortcut source has the output grid)
          g = src + (size_t)(m0 + lt / CPR + RSTEP * i) * ld;
        } else {
          const int iy = xr[i].iy0 + dy, ix = xr[i].ix0 + dx;
          if (iy >= 0 && iy < eh && ix >= 0 && ix < ew) {
            const int sy = p.ups ? (iy >> 1) : iy, sx = p.ups ? (ix >> 1) : ix;
            g = src + (size_t)(xr[i].bbase + sy * p.IW + sx) * ld;
          }
        }
      }
      if (g) { xp[i] = (const char*)(g + kcx[i]); xinc[i] = ROWB; }
      else { xp[i] = (const char*)p.zeros; xinc[i] = 0; }       // padding / M tail: re-read the zero page
    }
  };
  // Issues exactly NLOADS global_load_lds per thread; past the end: dummy loads of the zero page into a buffer
  // nobody reads, so the counted vmcnt below stays uniform through the pipeline tail.  The (rare, bulky) segment
  // switch is kept out of this hot helper: callers run `advance_segment()` once per K-tile before it.
  auto advance_segment = [&]() {
    if (p_left == 0 && p_kt < kt_end) segment_setup(p_kt * BKT);
  };
  auto produce = [&](int buf) {
    char* xs = smem + buf * STAGE;
    char* ws = xs + X_BYTES;
    if (p_kt < kt_end) {
#pragma unroll
      for (int i = 0; i < XL; ++i) {
        __builtin_amdgcn_global_load_lds((gptr_t)xp[i], (lptr_t)(xs + (lwave * 64 + NL * i) * 16), 16, 0, 0);
        xp[i] += xinc[i];
      }
#pragma unroll
      for (int i = 0; i < WL; ++i) {
        __builtin_amdgcn_global_load_lds((gptr_t)wp[i], (lptr_t)(ws + ((i == WL - 1 ? (lwave * 64) & (WREM - 1) : lwave * 64) + NL * i) * 16), 16, 0, 0);
        wp[i] += winc[i];
      }
      ++p_kt; --p_left;
    } else {
#pragma unroll
      for (int i = 0; i < NLOADS; ++i)
        __builtin_amdgcn_global_load_lds((gptr_t)p.zeros, (lptr_t)(xs + (lwave * 64 + NL * (i % XL)) * 16), 16, 0, 0);
    }
  };

  accv acc[TN][TM];
#pragma unroll
  for (int a = 0; a < TN; ++a)
#pragma unroll
    for (int b = 0; b < TM; ++b)
#pragma unroll
      for (int r = 0; r < 4 * NGRP; ++r) acc[a][b][r] = 0.f;

  auto mfma_ = [](const bf16x8& w_, const bf16x8& x_, const accv& c_) -> accv {
    if constexpr (MF == 32) return DMX_MFMA_32x32x16(w_, x_, c_);
    else return DMX_MFMA_16x16x32(w_, x_, c_);
  };
  // fragment read addresses for this lane (stage offset is an immediate: the loop is unrolled over the ring)
  const int lr = lane & (MF - 1), lh = lane / MF;
  int xad[TM][KSTEPS], wad[TN][KSTEPS];
#pragma unroll
  for (int b = 0; b < TM; ++b) {
    const int r = wm * (MF * TM) + b * MF + lr;
#pragma unroll
    for (int kk = 0; kk < KSTEPS; ++kk) xad[b][kk] = r * ROWB + (((CPS * kk + lh) ^ swz(r)) << 4);
  }
#pragma unroll
  for (int a = 0; a < TN; ++a) {
    const int r = wn * MF * TN + a * MF + lr;
#pragma unroll
    for (int kk = 0; kk < KSTEPS; ++kk) wad[a][kk] = X_BYTES + r * ROWB + (((CPS * kk + lh) ^ swz(r)) << 4);
  }
  // MFMA phase of one K-tile.  Fragments are double-buffered in registers: the ds_reads of k-step kk+1 are issued
  // before the MFMAs of k-step kk, so LDS latency hides under the matrix pipe instead of serialising with it.
  auto compute = [&](const int J) {
    const char* st = smem + J * STAGE;
    if constexpr (TN <= 2) {
      bf16x8 xf[2][TM], wf[2][TN];
#pragma unroll
      for (int b = 0; b < TM; ++b) xf[0][b] = *(const bf16x8*)(st + xad[b][0]);
#pragma unroll
      for (int a = 0; a < TN; ++a) wf[0][a] = *(const bf16x8*)(st + wad[a][0]);
#pragma unroll
      for (int kk = 0; kk < KSTEPS; ++kk) {
        const int cur = kk & 1, nxt = cur ^ 1;
        if (kk + 1 < KSTEPS) {
#pragma unroll
          for (int b = 0; b < TM; ++b) xf[nxt][b] = *(const bf16x8*)(st + xad[b][kk + 1]);
#pragma unroll
          for (int a = 0; a < TN; ++a) wf[nxt][a] = *(const bf16x8*)(st + wad[a][kk + 1]);
        }
#pragma unroll
        for (int a = 0; a < TN; ++a)
#pragma unroll
          for (int b = 0; b < TM; ++b)
            acc[a][b] = mfma_(wf[cur][a], xf[cur][b], acc[a][b]);
        // pin the issue order the scheduler would otherwise undo: [ds_reads of k-step kk+1] then [MFMAs of kk]
        if (kk + 1 < KSTEPS) __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, TM * TN, 0);
      }
    } else {                                           // wide wave tile (64 x 128): 128 accumulator registers, single-buffered fragments
#pragma unroll
      for (int kk = 0; kk < KSTEPS; ++kk) {
        bf16x8 xf[TM], wf[TN];
#pragma unroll
        for (int b = 0; b < TM; ++b) xf[b] = *(const bf16x8*)(st + xad[b][kk]);
#pragma unroll
        for (int a = 0; a < TN; ++a) wf[a] = *(const bf16x8*)(st + wad[a][kk]);
#pragma unroll
        for (int a = 0; a < TN; ++a)
#pragma unroll
          for (int b = 0; b < TM; ++b)
            acc[a][b] = mfma_(wf[a], xf[b], acc[a][b]);
      }
    }
  };

  // MFMA phase of one K-tile for the ping-pong loop below: register double-buffered fragments for ANY wave tile (the ds_reads of
  // k-step kk+1 ride in the gaps of the MFMAs of k-step kk, two per gap), s_setprio 1 around it so that this wave's MFMAs win
  // the issue arbitration against its SIMD partner, which is in its DMA phase.
  auto compute_pp = [&](const int J) {
    const char* st = smem + J * STAGE;
    bf16x8 xf[2][TM], wf[2][TN];
    auto rd = [&](const int buf, const int kk) {
#pragma unroll
      for (int b = 0; b < TM; ++b) xf[buf][b] = *(const bf16x8*)(st + xad[b][kk]);
#pragma unroll
      for (int a = 0; a < TN; ++a) wf[buf][a] = *(const bf16x8*)(st + wad[a][kk]);
    };
    rd(0, 0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kk = 0; kk < KSTEPS; ++kk) {
      const int cur = kk & 1;
      __builtin_amdgcn_sched_barrier(0);
      if (kk + 1 < KSTEPS) rd(cur ^ 1, kk + 1);
#pragma unroll
      for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TM; ++b)
          acc[a][b] = mfma_(wf[cur][a], xf[cur][b], acc[a][b]);
      if (kk + 1 < KSTEPS) {
#pragma unroll
        for (int q_ = 0; q_ < (TM + TN + 1) / 2; ++q_) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, TM * TN - (TM + TN + 1) / 2, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_s_setprio(0);
  };

  // ---- folded LayerNorm (consumer side): the GEMM runs on the RAW rows x with W' = W*diag(gamma); the epilogue
  // applies y = rstd*(acc - mean*c1[n]) + c2[n].  Row mean / rstd come from the producer's per-n-tile partial sums.
  constexpr int EPI_BYTES = BM * (BN + 4) * 4;                                  // whole-tile fp32 staging (epilogue)
  constexpr int EPI_PASS = TN == 5 ? EPI_BYTES / 2 : (EPI_BYTES <= 152 * 1024) ? EPI_BYTES : EPI_BYTES / WM;  // or one slab of rows per pass
  constexpr int LN_OFF = (NSTAGE * STAGE > EPI_PASS) ? NSTAGE * STAGE : EPI_PASS;
  float* lnst = (float*)(smem + LN_OFF);               // [BM][2] = (mean, rstd); only allocated when ln_stats != null
  if (p.ln_stats) {
    for (int r = t; r < BM; r += NT) {
      int m = m0 + r; if (m >= p.M) m = p.M - 1;
      float sa = 0.f, sq = 0.f;
      for (int j = 0; j < p.ln_tiles; ++j) {
        const float* q = p.ln_stats + ((size_t)j * p.M + m) * 2;
        sa += q[0]; sq += q[1];
      }
      const float mean = sa / (float)p.ln_C;
      float var = sq / (float)p.ln_C - mean * mean; var = var < 0.f ? 0.f : var;
      lnst[2 * r] = mean; lnst[2 * r + 1] = rsqrtf(var + p.ln_eps);
    }
  }

  // ---- NSTAGE-deep LDS ring, NSTAGE-1 K-tiles of DMA in flight, counted vmcnt (never drained to 0 in the loop)
  if constexpr (NP > 0) {
    // warp-specialised: loader waves and MFMA waves run separate loops that meet at one s_barrier per K-tile
    // (separate loops keep the loaders' pointer state and the consumers' accumulators out of each other's live ranges)
    if (is_loader) {
      issue_prefetch();
#pragma unroll
      for (int s = 0; s < NSTAGE - 1; ++s) { advance_segment(); produce(s); }
      int kt = kt_begin;
#define DMX_LSUB(J)                                                                          \
  {                                                                                          \
    advance_segment();                                                                       \
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSTAGE - 2) * NLOADS) : "memory");             \
    __builtin_amdgcn_s_barrier();                                                            \
    if (!(p.dbg & 2)) produce((J + NSTAGE - 1) % NSTAGE);   /* dbg bit1: ablation, no DMA */   \
    if (++kt >= kt_end) break;                                                               \
  }
      for (;;) {
        DMX_LSUB(0) DMX_LSUB(1) DMX_LSUB(2)
        if constexpr (NSTAGE >= 4) DMX_LSUB(3)
      }
#undef DMX_LSUB
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      // MFMA waves: software-pipelined over k-steps AND K-tiles.  Two register fragment sets; the ds_reads of the
      // next k-step (or of k-step 0 of the NEXT tile, right after its barrier) are always issued before the 8 MFMAs
      // of the current one, so the matrix pipe never waits for LDS latency.  The barrier for tile kt+1 is taken once
      // tile kt is completely in registers (lgkmcnt(0)), which is also what lets the loaders refill its buffer.
      static_assert(NP == 0 || KSTEPS == 4, "warp-specialised MFMA loop is written for 4 k-steps per tile");
      if (p.timing) tm1 = dmx_now(p.dbg);
      bf16x8 xf[2][TM], wf[2][TN];
      auto rd = [&](const int buf, const int J, const int kk) {
        const char* st = smem + J * STAGE;
        wf[buf][0] = *(const bf16x8*)(st + wad[0][kk]);          // in the order the MFMAs consume them
#pragma unroll
        for (int b = 0; b < TM; ++b) xf[buf][b] = *(const bf16x8*)(st + xad[b][kk]);
#pragma unroll
        for (int a = 1; a < TN; ++a) wf[buf][a] = *(const bf16x8*)(st + wad[a][kk]);
      };
      auto mm = [&](const int buf) {
#pragma unroll
        for (int a = 0; a < TN; ++a)
#pragma unroll
          for (int b = 0; b < TM; ++b)
            acc[a][b] = mfma_(wf[buf][a], xf[buf][b], acc[a][b]);
      };
#define DMX_SB __builtin_amdgcn_sched_barrier(0)
      // one scheduling region = [TM+TN ds_reads of the next fragment set] interleaved, two per gap, behind the first
      // MFMAs of the current one (a ds_read issued in an MFMA's shadow is free; clustered between MFMA groups
      // they cost ~100 cycles per group)
#define DMX_PHASE(RB, RJ, RK, MB)                                                            \
  {                                                                                          \
    DMX_SB; rd(RB, RJ, RK); mm(MB);                                                          \
    _Pragma("unroll") for (int q_ = 0; q_ < (TM + TN) / 2; ++q_) {                           \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                     \
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                                     \
    }                                                                                        \
    __builtin_amdgcn_sched_group_barrier(0x008, TM * TN - (TM + TN) / 2, 0);                 \
    DMX_SB;                                                                                  \
  }
      int kt = kt_begin;
      __builtin_amdgcn_s_barrier();
      rd(0, 0, 0);
#define DMX_CSUB(J)                                                                          \
  {                                                                                          \
    DMX_PHASE(1, J, 1, 0)                                                                    \
    DMX_PHASE(0, J, 2, 1)                                                                    \
    DMX_PHASE(1, J, 3, 0)                                                                    \
    ++kt;                                                                                    \
    if (kt < kt_end) {                                                                       \
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                     \
      __builtin_amdgcn_s_barrier();                                                          \
    }                                                                                        \
    DMX_PHASE(0, (J + 1) % NSTAGE, 0, 1)  /* after the last tile: a harmless read of a dead buffer */ \
    if (kt >= kt_end) break;                                                                 \
  }
      for (;;) {
        DMX_CSUB(0) DMX_CSUB(1) DMX_CSUB(2)
        if constexpr (NSTAGE >= 4) DMX_CSUB(3)
      }
#undef DMX_PHASE
#undef DMX_CSUB
#undef DMX_SB
    }
  } else if constexpr (PS && BM >= 256) {
    // ---- two-group ping-pong (persistent big-tile instances).  Waves w and w + NC/2 share a SIMD; group A = waves [0, NC/2), group B the
    // rest.  Per K-tile every wave runs a DMA phase (its share of tile j+1 -> stage (j+1) % NSTAGE), a barrier, an MFMA phase
    // (tile j), a barrier.  Both groups run the SAME instruction stream, but B takes one extra barrier up front (and A one at the
    // end), so B is always one phase behind: on every SIMD one wave is in its MFMA phase while its partner issues LDS-DMA.  (An
    // LDS-DMA instruction costs its wave ~60-180 issue cycles; in a lock-step loop both waves of a SIMD pay that at the same
    // time and the matrix pipe idles: measured 1.29 us per 256x160x64 K-tile against 0.88 MFMA-only / 0.64 DMA-only.)
    //   barrier k:   A: dma(j+1) |2j+1| mfma(j) |2j+2| ...        B: |1| ... dma(j+1) |2j+2| mfma(j) |2j+3| ...
    // A wave confirms its pieces of tile j+1 (vmcnt(0)) at the END of mfma(j) - they were issued a whole phase earlier - so
    // every piece of tile j+1 is confirmed by barrier 2j+3 at the latest (B's), the barrier before A's mfma(j+1).  Stage
    // (j+1) % NSTAGE held tile j-2 (NSTAGE = 3), last read by B before barrier 2j-1; A refills it after barrier 2j.
    static_assert(NSTAGE >= 3 && NC % 2 == 0, "ping-pong loop: three stages, an even number of waves");
    const bool grpB = wave >= NC / 2;
    advance_segment(); produce(0);
    if (p.timing) tm1 = dmx_now(p.dbg);
    if (grpB) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
    int kt = kt_begin;
#define DMX_PP(J)                                                                            \
  {                                                                                          \
    advance_segment();                                                                       \
    if (!(p.dbg & 2)) produce((J + 1) % NSTAGE);                                             \
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLOADS) : "memory");                            \
    __builtin_amdgcn_s_barrier();                                                            \
    if (!(p.dbg & 1)) compute_pp(J);                                                         \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                         \
    __builtin_amdgcn_s_barrier();                                                            \
    if (++kt >= kt_end) break;                                                               \
  }
    for (;;) {
      DMX_PP(0) DMX_PP(1) DMX_PP(2)
      if constexpr (NSTAGE >= 4) DMX_PP(3)
    }
#undef DMX_PP
    if (!grpB) __builtin_amdgcn_s_barrier();           // (group B's extra barrier of the prologue)
  } else {
#pragma unroll
    for (int s = 0; s < NSTAGE - 1; ++s) { advance_segment(); produce(s); }
    if (p.timing) tm1 = dmx_now(p.dbg);
    int kt = kt_begin;
#define DMX_SUBITER(J)                                                                      \
  {                                                                                         \
    advance_segment();                                                                      \
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSTAGE - 2) * NLOADS) : "memory");            \
    __builtin_amdgcn_s_barrier(); /* tile kt is in LDS for every wave; tile kt-1's buffer is free */ \
    if (!(p.dbg & 2)) produce((J + NSTAGE - 1) % NSTAGE);                                   \
    if (!(p.dbg & 1)) compute(J);                                                           \
    if (++kt >= kt_end) break;                                                              \
  }
    for (;;) {
      DMX_SUBITER(0) DMX_SUBITER(1)
      if constexpr (NSTAGE >= 3) DMX_SUBITER(2)
      if constexpr (NSTAGE >= 4) DMX_SUBITER(3)
      if constexpr (NSTAGE >= 5) DMX_SUBITER(4)
      if constexpr (NSTAGE >= 6) DMX_SUBITER(5)
    }
#undef DMX_SUBITER
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // drain the dummy tail loads before LDS is released
  }
  if (p.timing) tm2 = dmx_now(p.dbg);

  // ---------------------------------------------------------------- stream-K fix-up (persistent launches)
  // slab[position]: the block's accumulators in register order, [16*TN*TM/4 float4 groups][consumer thread] - every store /
  // load is a fully coalesced 16 B per lane; no row-major staging.
  constexpr int NCT = 64 * NC;                         // consumer threads
  constexpr int SLAB_F4 = TN * TM * NGRP * NCT;           // float4 groups per slab (= BM*BN/4)
  bool helper = false;
  if constexpr (PS) {
  __builtin_amdgcn_sched_barrier(0);                   // keep the epilogue's loads below the fix-up (register pressure)
  if (kt_begin > 0) {
    helper = true;
    if (is_consumer) {
      // WRITE-THROUGH (sc1) 16-byte stores: the slab leaves this XCD's L2 as it is written, so publishing it needs no agent-scope
      // release (buffer_wbl2 walks the whole L2: ~2 us clean, 6+ us with the slab freshly dirty - cdna guide, "publish-large")
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)((f32x4*)p.partial + (size_t)pos * SLAB_F4), 0, SLAB_F4 * 16, 0x00020000);
      int off = t * 16;
#pragma unroll
      for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TM; ++b)
#pragma unroll
          for (int g = 0; g < NGRP; ++g) {
            const f32x4 v = {acc[a][b][4 * g], acc[a][b][4 * g + 1], acc[a][b][4 * g + 2], acc[a][b][4 * g + 3]};
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, off, 0, 16);
            off += NCT * 16;
            __builtin_amdgcn_sched_barrier(0);
          }
    }
    // publish: every wave's write-through stores drained -> block barrier -> the flag (relaxed, agent scope)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t == 0) __hip_atomic_store(p.flags + pos, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else if (kt_end < nkt_total) {
    // owner of a tile whose K tail other blocks computed: positions pos+1 ... while their range starts inside this tile
    const long long tile_end_it = (long long)(tile_id + 1) * nkt_total;
    int q_end = pos + 1;
    while (q_end < nblk && bound(q_end) < tile_end_it) ++q_end;
    if (t == 0) {
      // bounded: a helper that is not resident (the grid is one block per CU, but another stream may hold CUs) must neither hang the GPU
      // nor pass silently - after ~40 ms the owner raises the device error (common.h) and goes on
      const long long t0 = __builtin_amdgcn_s_memrealtime();
      for (int q = pos + 1; q < q_end; ++q)
        while (__hip_atomic_load(p.flags + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
          __builtin_amdgcn_s_sleep(4);
          if (__builtin_amdgcn_s_memrealtime() - t0 > 4000000) { dmx_dev_raise(p.err, DMX_DEVK_STREAMK_HELPER, pos, tile_id, q, nblk); break; }
        }
    }
    __syncthreads();
    if (is_consumer) {
      for (int q = pos + 1; q < q_end; ++q) {
        // the slabs were stored write-through (sc1), so sc1 loads (L1 bypassed, served by L2 / the fabric) see them without an
        // agent-scope acquire (buffer_inv: ~1.7 us per block) - cdna guide, Guideline 16 R1
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)((f32x4*)p.partial + (size_t)q * SLAB_F4), 0, SLAB_F4 * 16, 0x00020000);
        // the block is alone on its CU and the slab comes from L2 / the fabric: what the pass costs is round trips, so all
        // loads of a batch are issued before the first add (SB groups = 4*SB registers in flight)
        constexpr int NG = TN * TM * NGRP, SB = NG % 5 == 0 ? 5 : (NG % 8 == 0 ? 8 : 4);
        static_assert(NG % SB == 0, "slab batches");
#pragma unroll
        for (int g0 = 0; g0 < NG; g0 += SB) {
          f32x4 v[SB];
#pragma unroll
          for (int u = 0; u < SB; ++u) v[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (t + (g0 + u) * NCT) * 16, 0, 16));
#pragma unroll
          for (int u = 0; u < SB; ++u) {
            const int gi = g0 + u, ab = gi / NGRP, g = gi % NGRP, a = ab / TM, b = ab % TM;
            acc[a][b][4 * g] += v[u][0]; acc[a][b][4 * g + 1] += v[u][1]; acc[a][b][4 * g + 2] += v[u][2]; acc[a][b][4 * g + 3] += v[u][3];
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  }   // PS
  // ---- GroupNorm statistics of the tile's output (GemmArgs.colstats): per-channel (sum, sum of squares) of the ROUNDED outputs.
  // Inside the tile everything is float arithmetic in a fixed order (per-thread partials over the thread's rows, then a fixed-
  // order fold of the row lanes through LDS); across tiles the per-tile sums are added as 64-bit FIXED-POINT integers with
  // global atomics (integer addition is associative: the totals are bit-reproducible whatever order the tiles finish in).
  // colstats[(sample*N + n)*2 + {0,1}] += {sum * 2^20, sumsq * 2^32}; the buffer is zero before the launch.
  auto publish_colstats = [&](const float* cs_s, const float* cs_q, int o, int rl, int nrl, bool active, float* red) {
    __builtin_amdgcn_s_barrier();                      // every thread is done with the staged tile
    if (active) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { red[(rl * BN + o * 8 + e) * 2] = cs_s[e]; red[(rl * BN + o * 8 + e) * 2 + 1] = cs_q[e]; }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int c = t; c < BN; c += NT) {
      if (n0 + c >= p.N) continue;
      float sa = 0.f, sq = 0.f;
      for (int k = 0; k < nrl; ++k) { sa += red[(k * BN + c) * 2]; sq += red[(k * BN + c) * 2 + 1]; }
      const int smp = m0 / p.cs_rows;
      dmx_stat_add(p.colstats + ((size_t)smp * p.N + n0 + c) * DMX_STAT_WORDS, sa, sq);
    }
  };
  // ---------------------------------------------------------------- epilogue
  // acc[a][b][4g+e] = out[m = m0 + wm*64 + b*32 + lr][n = n0 + wn*32*TN + a*32 + 8g + 4lh + e]
  if (helper) {
    // nothing more: the tile's owner finishes it
  } else if (p.splitk > 1) {
    float* part = p.partial + (size_t)blockIdx.y * p.M * p.N;
    if (is_consumer)
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
      for (int b = 0; b < TM; ++b) {
        const int m = m0 + wm * (MF * TM) + b * MF + lr;
        if (m >= Mlim) continue;
        const size_t orow = out_row(m);                  // partials are kept in output-row order: the reduce pass needs no map
#pragma unroll
        for (int g = 0; g < NGRP; ++g) {
          const int n = n0 + wn * MF * TN + a * MF + 8 * g + 4 * lh;
          float* o = part + orow * p.N + n;
          if (n + 3 < p.N && (p.N & 3) == 0) {
            f32x4 v = {acc[a][b][4 * g], acc[a][b][4 * g + 1], acc[a][b][4 * g + 2], acc[a][b][4 * g + 3]};
            *(f32x4*)o = v;
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) if (n + e < p.N) o[e] = acc[a][b][4 * g + e];
          }
        }
      }
  } else if (TN == 5 && !p.out_f32 && (p.N & 7) == 0 && (p.ldo & 7) == 0 && (p.res == nullptr || (p.ldres & 7) == 0)) {
    // ---- coalesced epilogue of the 160 / 320-column tiles (BN / 8 = 20 or 40 octets per row do not divide the block, so
    // the (row, octet) of an item varies per thread and the column vectors are fetched per item): two passes of BM/2 rows
    // staged in LDS as fp32; bias or the folded-LayerNorm affine, time-embedding row bias, residual, GEGLU (320-column
    // tile: 5 whole packed 64-column groups), bf16 out.  Row statistics / GELU stay on the power-of-two tiles (plan).
    if constexpr (TN == 5) {
      constexpr int LDT = BN + 4, EP = 2, RPP = BM / EP;
      static_assert(WM % EP == 0, "160-column epilogue: whole waves per pass");
      float* tile = (float*)smem;
      const bool geglu = NWN == 2 && p.geglu != 0;
      const float* bsrc = p.ln_stats ? p.ln_c2 : p.bias;
      // the tile's column vectors (bias | c2, c1, time-embedding row bias) go to LDS once: fetched per item from global memory
      // (three dependent L2 round trips inside the item loop) they made this epilogue 4-5 us per tile
      float* colv = lnst + 2 * BM;                       // [3][BN]
      const bool rb_one = p.rowbias && (m0 / p.rows_per_group == (min(m0 + BM, Mlim) - 1) / p.rows_per_group);   // whole tile inside one row-bias group
      for (int c = t; c < BN; c += NT) {
        const int n = min(n0 + c, p.N - 1);
        colv[c] = bsrc ? bsrc[n] : 0.f;
        colv[BN + c] = p.ln_stats ? p.ln_c1[n] : 0.f;
        colv[2 * BN + c] = rb_one ? p.rowbias[(size_t)(m0 / p.rows_per_group) * p.ldrb + n] : 0.f;
      }
      float cs_s[8], cs_q[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) { cs_s[e] = 0.f; cs_q[e] = 0.f; }
#pragma unroll
      for (int ep = 0; ep < EP; ++ep) {
        __builtin_amdgcn_s_barrier();
        if (is_consumer && wm / (WM / EP) == ep) {
#pragma unroll
          for (int a = 0; a < TN; ++a)
#pragma unroll
            for (int b = 0; b < TM; ++b) {
              const int r = (wm % (WM / EP)) * (MF * TM) + b * MF + lr;
#pragma unroll
              for (int g = 0; g < NGRP; ++g) {
                const f32x4 v = {acc[a][b][4 * g], acc[a][b][4 * g + 1], acc[a][b][4 * g + 2], acc[a][b][4 * g + 3]};
                *(f32x4*)(tile + r * LDT + wn * MF * TN + a * MF + 8 * g + 4 * lh) = v;
              }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        // Items (row, output octet) are processed U at a time: all LDS reads (and residual loads) of a group are issued before
        // the first one is consumed - one item at a time the loop paid the LDS round trip (and a non-constant division) per item.
        auto affine = [&](const f32x4& v0, const f32x4& v1, int tc, float mean, float rstd, float* v) {
          const f32x4 b0 = *(const f32x4*)(colv + tc), b1 = *(const f32x4*)(colv + tc + 4);
          if (p.ln_stats) {
            const f32x4 c0 = *(const f32x4*)(colv + BN + tc), c1 = *(const f32x4*)(colv + BN + tc + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] = rstd * (v0[e] - mean * c0[e]) + b0[e]; v[4 + e] = rstd * (v1[e] - mean * c1[e]) + b1[e]; }
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] = v0[e] + b0[e]; v[4 + e] = v1[e] + b1[e]; }
          }
        };
        if (geglu) {
          constexpr int OCG = BN / 16;                   // output octets per tile row (GEGLU halves the columns)
          constexpr int NIT = (RPP * OCG + NT - 1) / NT;
#pragma unroll
          for (int k0 = 0; k0 < NIT; k0 += 3) {
            f32x4 a0[3], a1[3], g0[3], g1[3];
#pragma unroll
            for (int u = 0; u < 3; ++u) {
              const int q = min(t + NT * (k0 + u), RPP * OCG - 1);
              const int r = q / OCG, o = q - r * OCG, tc = 64 * (o >> 2) + (o & 3) * 8;
              a0[u] = *(const f32x4*)(tile + r * LDT + tc); a1[u] = *(const f32x4*)(tile + r * LDT + tc + 4);
              g0[u] = *(const f32x4*)(tile + r * LDT + tc + 32); g1[u] = *(const f32x4*)(tile + r * LDT + tc + 36);
            }
#pragma unroll
            for (int u = 0; u < 3; ++u) {
              const int q = t + NT * (k0 + u);
              if (k0 + u >= NIT || q >= RPP * OCG) continue;
              const int r = q / OCG, o = q - r * OCG, Gg = o >> 2, jj = (o & 3) * 8;
              const int m = m0 + ep * RPP + r, na = n0 + 64 * Gg + jj;
              if (m >= Mlim || na >= p.N) continue;
              float mean = 0.f, rstd = 1.f;
              if (p.ln_stats) { mean = lnst[2 * (ep * RPP + r)]; rstd = lnst[2 * (ep * RPP + r) + 1]; }
              float v[8], gt[8];
              affine(a0[u], a1[u], 64 * Gg + jj, mean, rstd, v);
              affine(g0[u], g1[u], 64 * Gg + 32 + jj, mean, rstd, gt);
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] *= gelu_erf_f(gt[e]);
              *(u32x4*)((bf16*)p.out + (size_t)m * p.ldo + (n0 >> 1) + 32 * Gg + jj) = pack_bf8(v);
            }
          }
        } else if (CS && p.colstats) {
          // statistics wanted: (row lane, octet) item mapping with the octet FIXED per thread (NT / OCP row lanes, a few threads
          // idle, one partial sweep more) so the column partials accumulate in registers across sweeps and passes
          constexpr int OCP = BN / 8, RL = NT / OCP, NITB = (RPP + RL - 1) / RL;
          const bool act_t = t < RL * OCP;
          const int o = t % OCP, rl = t / OCP;
#pragma unroll
          for (int u0 = 0; u0 < NITB; u0 += 3) {
            f32x4 x0[3], x1[3]; u32x4 rr[3];
#pragma unroll
            for (int u = 0; u < 3; ++u) {
              const int r = min(rl + RL * (u0 + u), RPP - 1);
              x0[u] = *(const f32x4*)(tile + r * LDT + o * 8); x1[u] = *(const f32x4*)(tile + r * LDT + o * 8 + 4);
              if (p.res) {
                int m = m0 + ep * RPP + r; if (m >= Mlim) m = Mlim - 1;
                const int n = min(n0 + o * 8, p.N - 8);
                rr[u] = *(const u32x4*)(p.res + (size_t)m * p.ldres + n);
              }
            }
#pragma unroll
            for (int u = 0; u < 3; ++u) {
              const int r = rl + RL * (u0 + u);
              const int m = m0 + ep * RPP + r, n = n0 + o * 8;
              if (u0 + u >= NITB || !act_t || r >= RPP || m >= Mlim || n >= p.N) continue;
              float mean = 0.f, rstd = 1.f;
              if (p.ln_stats) { mean = lnst[2 * (ep * RPP + r)]; rstd = lnst[2 * (ep * RPP + r) + 1]; }
              float v[8];
              affine(x0[u], x1[u], o * 8, mean, rstd, v);
              if (p.rowbias) {
                const float* rb = rb_one ? colv + 2 * BN + o * 8 : p.rowbias + (size_t)(m / p.rows_per_group) * p.ldrb + n;
                const f32x4 b0 = *(const f32x4*)rb, b1 = *(const f32x4*)(rb + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] += b0[e]; v[4 + e] += b1[e]; }
              }
              if (p.res) {
                float rf[8]; unpack_bf8(rr[u], rf);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += rf[e];
              }
              const u32x4 pk = pack_bf8(v);
              *(u32x4*)((bf16*)p.out + out_row(m) * p.ldo + n) = pk;
              float f[8]; unpack_bf8(pk, f);
#pragma unroll
              for (int e = 0; e < 8; ++e) { cs_s[e] += f[e]; cs_q[e] += f[e] * f[e]; }
            }
          }
          if (ep == EP - 1) publish_colstats(cs_s, cs_q, o, rl, RL, act_t, tile);
        } else {
          constexpr int OCP = BN / 8;
          constexpr int NIT = RPP * OCP / NT;            // 5 for both instances (64 rows x 20 | 40 octets over 256 | 512 threads)
          static_assert(RPP * OCP % NT == 0, "160-column epilogue: items divide over the block");
          f32x4 x0[NIT], x1[NIT]; u32x4 rr[NIT];
#pragma unroll
          for (int u = 0; u < NIT; ++u) {
            const int q = t + NT * u, r = q / OCP, o = q - r * OCP;
            x0[u] = *(const f32x4*)(tile + r * LDT + o * 8); x1[u] = *(const f32x4*)(tile + r * LDT + o * 8 + 4);
            if (p.res) {
              int m = m0 + ep * RPP + r; if (m >= Mlim) m = Mlim - 1;
              const int n = min(n0 + o * 8, p.N - 8);
              rr[u] = *(const u32x4*)(p.res + (size_t)m * p.ldres + n);
            }
          }
#pragma unroll
          for (int u = 0; u < NIT; ++u) {
            const int q = t + NT * u, r = q / OCP, o = q - r * OCP;
            const int m = m0 + ep * RPP + r, n = n0 + o * 8;
            if (m >= Mlim || n >= p.N) continue;
            float mean = 0.f, rstd = 1.f;
            if (p.ln_stats) { mean = lnst[2 * (ep * RPP + r)]; rstd = lnst[2 * (ep * RPP + r) + 1]; }
            float v[8];
            affine(x0[u], x1[u], o * 8, mean, rstd, v);
            if (p.rowbias) {
              const float* rb = rb_one ? colv + 2 * BN + o * 8 : p.rowbias + (size_t)(m / p.rows_per_group) * p.ldrb + n;
              const f32x4 b0 = *(const f32x4*)rb, b1 = *(const f32x4*)(rb + 4);
#pragma unroll
              for (int e = 0; e < 4; ++e) { v[e] += b0[e]; v[4 + e] += b1[e]; }
            }
            if (p.res) {
              float rf[8]; unpack_bf8(rr[u], rf);
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] += rf[e];
            }
            *(u32x4*)((bf16*)p.out + out_row(m) * p.ldo + n) = pack_bf8(v);
          }
        }
      }
    }
  } else if (TN != 5 && !p.out_f32 && (p.N & 7) == 0 && (p.ldo & 7) == 0 && (p.res == nullptr || (p.ldres & 7) == 0)) {
    // ---- coalesced epilogue: accumulators (+ time-embedding row bias) -> LDS as fp32 (the ring is free now) ->
    // each thread finishes 8 consecutive channels of one pixel: + bias + residual (or GEGLU a*gelu(b)), one
    // rounding to bf16, one 16-byte store; a wave writes whole 128/256-byte row segments.  Raw s_barrier +
    // lgkmcnt only: __syncthreads() would also wait for the global stores (vmcnt counts stores on gfx950), and
    // every global load is issued before the first store for the same reason.
    if constexpr (TN != 5) {
    constexpr int LDT = BN + 4;                        // padded row stride (floats): conflict-free b128 writes
    constexpr int OC = BN / 8;                         // output octets per tile row (NT % OC == 0: o fixed per thread)
    constexpr int EP = (EPI_BYTES <= 152 * 1024) ? 1 : WM;   // passes: the 256x256 tile stages 64 rows (one wm) at a time
    constexpr int RPP = BM / EP;                       // tile rows per pass
    constexpr int OCT = RPP * OC / NT;                 // (row, octet) items per thread per pass
    float* tile = (float*)smem;
    const bool geglu = p.geglu != 0;
    if (p.rowbias && is_consumer) {                    // uniform branch; columns past N are clamped (never stored)
#pragma unroll
      for (int b = 0; b < TM; ++b) {
        int m = m0 + wm * (MF * TM) + b * MF + lr; if (m >= p.M) m = p.M - 1;
        const float* rb = p.rowbias + (size_t)(m / p.rows_per_group) * p.ldrb;
#pragma unroll
        for (int a = 0; a < TN; ++a) {
          if (TN * TM > 4) asm volatile("" ::: "memory");   // wide tiles: do not hoist all 8 x 4 row-bias loads at once (VGPRs)
#pragma unroll
          for (int g = 0; g < NGRP; ++g) {
            int nn = n0 + wn * MF * TN + a * MF + 8 * g + 4 * lh; if (nn > p.N - 4) nn = p.N - 4;
            const f32x4 bv = *(const f32x4*)(rb + nn);
            acc[a][b][4 * g] += bv[0]; acc[a][b][4 * g + 1] += bv[1]; acc[a][b][4 * g + 2] += bv[2]; acc[a][b][4 * g + 3] += bv[3];
          }
        }
      }
    }
    // per-thread column data (the octet is the same for every item of a thread)
    const int og = t % (BN / 16), Gg = og >> 2, jjg = (og & 3) * 8;      // GEGLU: output octet -> packed 64-col group
    const int o = t % OC;
    const bool nvalid = geglu ? (n0 + 64 * Gg + jjg < p.N) : (n0 + o * 8 < p.N);
    const int n = nvalid ? n0 + o * 8 : 0;             // clamped: loads stay in range, nothing is stored
    const int na = nvalid ? n0 + 64 * Gg + jjg : 0;
    float bs[8], c1[8], bg[8], cg[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { bs[e] = 0.f; c1[e] = 0.f; bg[e] = 0.f; cg[e] = 0.f; }
    auto load_cols = [&]() {
      const float* bsrc = p.ln_stats ? p.ln_c2 : p.bias;     // folded LayerNorm: c2 already contains the bias
      const int nb = geglu ? na : n;
      if (bsrc) {
        const f32x4 b0 = *(const f32x4*)(bsrc + nb), b1 = *(const f32x4*)(bsrc + nb + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { bs[e] = b0[e]; bs[4 + e] = b1[e]; }
        if (geglu) {
          const f32x4 g0 = *(const f32x4*)(bsrc + nb + 32), g1 = *(const f32x4*)(bsrc + nb + 36);
#pragma unroll
          for (int e = 0; e < 4; ++e) { bg[e] = g0[e]; bg[4 + e] = g1[e]; }
        }
      }
      if (p.ln_stats) {
        const f32x4 y0 = *(const f32x4*)(p.ln_c1 + nb), y1 = *(const f32x4*)(p.ln_c1 + nb + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { c1[e] = y0[e]; c1[4 + e] = y1[e]; }
        if (geglu) {
          const f32x4 g0 = *(const f32x4*)(p.ln_c1 + nb + 32), g1 = *(const f32x4*)(p.ln_c1 + nb + 36);
#pragma unroll
          for (int e = 0; e < 4; ++e) { cg[e] = g0[e]; cg[4 + e] = g1[e]; }
        }
      }
    };
    // 128-accumulator wave tiles: fetch the column vectors only after the accumulators are staged (VGPR budget - a
    // spill reload after the first global store would wait for that store: vmcnt counts stores)
    constexpr bool LATE_COLS = (TN * TM > 4);
    if constexpr (!LATE_COLS) load_cols();
    float cs_s[8], cs_q[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { cs_s[e] = 0.f; cs_q[e] = 0.f; }
#pragma unroll
    for (int ep = 0; ep < EP; ++ep) {
      __builtin_amdgcn_s_barrier();                    // K loop / previous pass is done with this LDS
      if (is_consumer && (EP == 1 || wm == ep)) {
#pragma unroll
        for (int a = 0; a < TN; ++a)
#pragma unroll
          for (int b = 0; b < TM; ++b) {
            const int r = (EP == 1 ? wm * (MF * TM) : 0) + b * MF + lr;
#pragma unroll
            for (int g = 0; g < NGRP; ++g) {
              const f32x4 v = {acc[a][b][4 * g], acc[a][b][4 * g + 1], acc[a][b][4 * g + 2], acc[a][b][4 * g + 3]};
              *(f32x4*)(tile + r * LDT + wn * MF * TN + a * MF + 8 * g + 4 * lh) = v;
            }
          }
      }
      if constexpr (LATE_COLS) { if (ep == 0) load_cols(); }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      const int rbase = ep * RPP;                      // first tile row of this pass
      if (geglu) {
        if constexpr (BN >= 64) {                      // (a 64-column tile holds exactly one packed group: 32 'a' + 32 gate columns)
          // BN/2 output columns per tile row: octet og -> packed group Gg ('a' rows 64Gg.., gate rows 64Gg+32..)
          constexpr int GI = RPP * (BN / 16) / NT;
#pragma unroll
          for (int k = 0; k < GI; ++k) {
            asm volatile("" ::: "memory");
            const int r = t / (BN / 16) + (NT / (BN / 16)) * k;
            const int m = m0 + rbase + r;
            const float* ta = tile + r * LDT + 64 * Gg + jjg;
            float mean = 0.f, rstd = 1.f;
            if (p.ln_stats) { mean = lnst[2 * (rbase + r)]; rstd = lnst[2 * (rbase + r) + 1]; }
            float v[8];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
              const f32x4 av = *(const f32x4*)(ta + 4 * q), gv = *(const f32x4*)(ta + 32 + 4 * q);
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const float a_ = rstd * (av[e] - mean * c1[4 * q + e]) + bs[4 * q + e];
                const float g_ = rstd * (gv[e] - mean * cg[4 * q + e]) + bg[4 * q + e];
                v[4 * q + e] = a_ * gelu_erf_f(g_);
              }
            }
            if (nvalid && m < p.M) *(u32x4*)((bf16*)p.out + (size_t)m * p.ldo + (n0 >> 1) + 32 * Gg + jjg) = pack_bf8(v);
          }
        }
      } else {
        u32x4 rv[OCT];
        if (p.res) {
#pragma unroll
          for (int k = 0; k < OCT; ++k) {
            int m = m0 + rbase + t / OC + (NT / OC) * k; if (m >= p.M) m = p.M - 1;
            rv[k] = *(const u32x4*)(p.res + (size_t)m * p.ldres + n);
          }
        }
#pragma unroll
        for (int k = 0; k < OCT; ++k) {
          asm volatile("" ::: "memory");               // keep each item's LDS reads in its own iteration (VGPR pressure)
          const int r = t / OC + (NT / OC) * k;
          const int m = m0 + rbase + r;
          const bool live = nvalid && m < Mlim;
          const f32x4 v0 = *(const f32x4*)(tile + r * LDT + o * 8), v1 = *(const f32x4*)(tile + r * LDT + o * 8 + 4);
          float v[8];
          if (p.ln_stats) {
            const float mean = lnst[2 * (rbase + r)], rstd = lnst[2 * (rbase + r) + 1];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              v[e] = rstd * (v0[e] - mean * c1[e]) + bs[e];
              v[4 + e] = rstd * (v1[e] - mean * c1[4 + e]) + bs[4 + e];
            }
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] = v0[e] + bs[e]; v[4 + e] = v1[e] + bs[4 + e]; }
          }
          if (p.act == 1) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = gelu_erf_f(v[e]);
          }
          if (p.res) {
            float rf[8]; unpack_bf8(rv[k], rf);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += rf[e];
          }
          const u32x4 pk = pack_bf8(v);
          if (live) *(u32x4*)((bf16*)p.out + out_row(m) * p.ldo + n) = pk;
          if (CS && p.colstats && live) {
            float f[8]; unpack_bf8(pk, f);
#pragma unroll
            for (int e = 0; e < 8; ++e) { cs_s[e] += f[e]; cs_q[e] += f[e] * f[e]; }
          }
          if (p.rowstats_out) {
            // per-row (sum, sumsq) of the rounded outputs over this n-tile: the OC lanes of a row are adjacent lanes
            float f[8]; unpack_bf8(pk, f);
            float sa = 0.f, sq = 0.f;
            if (live) {
#pragma unroll
              for (int e = 0; e < 8; ++e) { sa += f[e]; sq += f[e] * f[e]; }
            }
            // the OC lanes of a row are adjacent; OC | 16: an inclusive scan with DPP row shifts (no LDS round trips: the
            // bpermute butterfly was three dependent LDS exchanges per item); the row's last lane holds the total
            constexpr bool DPP = OC <= 16 && (OC & (OC - 1)) == 0;
            if constexpr (DPP) {
              if (OC > 1) { sa += dpp_row_shr<1>(sa); sq += dpp_row_shr<1>(sq); }
              if (OC > 2) { sa += dpp_row_shr<2>(sa); sq += dpp_row_shr<2>(sq); }
              if (OC > 4) { sa += dpp_row_shr<4>(sa); sq += dpp_row_shr<4>(sq); }
              if (OC > 8) { sa += dpp_row_shr<8>(sa); sq += dpp_row_shr<8>(sq); }
            } else {
#pragma unroll
              for (int d = 1; d < OC; d <<= 1) { sa += __shfl_xor(sa, d); sq += __shfl_xor(sq, d); }
            }
            if (o == (DPP ? OC - 1 : 0) && m < p.M) {
              float* q = p.rowstats_out + ((size_t)tile_n * p.M + m) * 2;
              q[0] = sa; q[1] = sq;
            }
          }
        }
      }
    }
    if (CS && p.colstats && !geglu) publish_colstats(cs_s, cs_q, o, t / OC, NT / OC, true, tile);
    }   // if constexpr (TN != 5)
  } else {
    // ---- generic epilogue (fp32 output, channel counts that are not multiples of 8): per-lane 4-channel groups
    const bool vec_ok = ((p.N & 3) == 0) && ((p.ldo & 3) == 0);
    if (is_consumer)
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
      for (int b = 0; b < TM; ++b) {
        const int m = m0 + wm * (MF * TM) + b * MF + lr;
        if (m >= Mlim) continue;
        const size_t orow = out_row(m);
        const float* rb = p.rowbias ? (p.rowbias + (size_t)(m / p.rows_per_group) * p.ldrb) : nullptr;
#pragma unroll
        for (int g = 0; g < NGRP; ++g) {
          const int n = n0 + wn * MF * TN + a * MF + 8 * g + 4 * lh;
          if (n >= p.N) continue;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if (n + e >= p.N) continue;
            float x = acc[a][b][4 * g + e];
            if (p.bias) x += p.bias[n + e];
            if (rb) x += rb[n + e];
            if (p.res) x += bf_bits2f(*(const unsigned short*)(p.res + (size_t)m * p.ldres + n + e));
            if (p.out_f32) ((float*)p.out)[orow * p.ldo + n + e] = x;
            else ((unsigned short*)p.out)[orow * p.ldo + n + e] = f2bf_bits(x);
          }
        }
      }
    (void)vec_ok;
  }
  } while (PS && it < it_last);   // work items
  if (p.timing && threadIdx.x == 0) {        // measurement aid: per-block timeline in 10 ns ticks (s_memrealtime)
    long long* o = p.timing + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 4;
    o[0] = tm0; o[1] = tm1; o[2] = tm2; o[3] = dmx_now(p.dbg);
  }
}

// split-K second pass: sum partials in fixed order, then the same epilogue.  One float4 per thread; the partial loads of a
// thread are issued four at a time (independent, all in flight) and summed in split order - the pass is latency-bound,
// not bandwidth-bound (a few MB), so what matters is loads in flight per thread and enough blocks to cover the chip.
__global__ __launch_bounds__(256) void dmx_splitk_reduce_kernel(const GemmArgs p) {
  const size_t total4 = (size_t)p.M * p.N / 4;
  const size_t MN = (size_t)p.M * p.N;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total4) return;
  const size_t e0 = i * 4;
  const int m = (int)(e0 / p.N);
  const int n = (int)(e0 - (size_t)m * p.N);
  // epilogue operands first: their latency overlaps the partial loads
  f32x4 bv = {0.f, 0.f, 0.f, 0.f}, rb = {0.f, 0.f, 0.f, 0.f};
  u32x2 rv = {0u, 0u};
  if (p.bias) bv = *(const f32x4*)(p.bias + n);
  if (p.rowbias) rb = *(const f32x4*)(p.rowbias + (size_t)(m / p.rows_per_group) * p.ldrb + n);
  if (p.res) rv = *(const u32x2*)(p.res + (size_t)m * p.ldres + n);
  const float* q = p.partial + e0;
  f32x4 s = *(const f32x4*)q;
  int k = 1;
  for (; k + 3 < p.splitk; k += 4) {
    const f32x4 a0 = *(const f32x4*)(q + (size_t)k * MN), a1 = *(const f32x4*)(q + (size_t)(k + 1) * MN);
    const f32x4 a2 = *(const f32x4*)(q + (size_t)(k + 2) * MN), a3 = *(const f32x4*)(q + (size_t)(k + 3) * MN);
    s += a0; s += a1; s += a2; s += a3;
  }
  for (; k < p.splitk; ++k) s += *(const f32x4*)(q + (size_t)k * MN);
  float v[4] = {s[0] + bv[0] + rb[0], s[1] + bv[1] + rb[1], s[2] + bv[2] + rb[2], s[3] + bv[3] + rb[3]};
  if (p.res) {
    v[0] += h2f_lo(rv[0]); v[1] += h2f_hi(rv[0]);
    v[2] += h2f_lo(rv[1]); v[3] += h2f_hi(rv[1]);
  }
  if (p.out_f32) {
    f32x4 o = {v[0], v[1], v[2], v[3]};
    *(f32x4*)((float*)p.out + (size_t)m * p.ldo + n) = o;
  } else {
    u32x2 pk = {pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
    *(u32x2*)((bf16*)p.out + (size_t)m * p.ldo + n) = pk;
  }
}

// ------------------------------------------------------------------------- host side
static const bf16* g_zero_page = nullptr;

int dmx_zero_page(const bf16** out) {
  if (!g_zero_page) {
    void* z = nullptr;
    DMX_HIP(hipMalloc(&z, 4096));
    DMX_HIP(hipMemset(z, 0, 4096));   // staging pointers into this page never advance (increment 0)
    g_zero_page = (const bf16*)z;
  }
  *out = g_zero_page;
  return DMX_OK;
}

// Tile / split-K plan.  Config ids: 0 = 128x128x32 (2 blocks/CU), 1 = 128x64x32, 2 = 256x128x64 (1 block/CU).
// The UNet's GEMMs are small next to 256 CUs (M = 256..16384 rows), so the plan trades tile efficiency against
// filling the CUs, with split-K (fp32 partials + a reduce pass) when tiles alone leave CUs idle.  Costs are in
// units of one 128x128x32 K-tile step of one block; constants fitted on scripts/tune_gemm.py measurements.
struct TileCfg { int bm, bn, bk, slots; double per_ktile, fixed; };
static const TileCfg kCfg[16] = {            // measured (scripts/attic/gemm_timeline.py): 0.45 / 0.35 / 1.05 us per K-tile,
    {128, 128, 32, 512, 1.00, 9.0},         // ~4 us of prologue + epilogue per block
    {128, 64, 32, 512, 0.78, 6.0},
    {256, 128, 64, 256, 2.35, 10.0},        // 4x the FLOPs of config 0 per K-tile at ~1.7x its rate
    {128, 64, 64, 768, 0.55, 5.0},          // 128x64x64, eight waves, TWO-stage ring (49 KB): three blocks per CU so one block's prologue / epilogue
                                            // overlaps the others' K loops (the K = C linears are 70 % prologue + epilogue); force_tn = 4 / tuned table
    {128, 128, 64, 256, 1.00, 7.0},         // deep ring (4 x 32 KB, 1 block/CU)
    {256, 256, 32, 256, 2.40, 16.0},        // 128 FLOP/B (experimental, force_tn = 6 only)
    {256, 128, 64, 256, 1.90, 11.0},        // warp-specialised 256x128x64: 4 MFMA waves + 4 DMA waves (~0.85 us per K-tile)
    {128, 64, 64, 512, 0.55, 5.0},          // 128x64x64 with EIGHT waves (32x32 each): small grids are bound by the per-CU LDS-DMA fill
                                            // rate, which doubles with 8 issuing waves (~95 vs ~50 GB/s); force_tn = 8 / tuned table
    {128, 128, 32, 512, 1.00, 8.0},         // 128x128x32 with eight waves (32x64 each), 4-stage ring; force_tn = 9 / tuned table
    {128, 128, 64, 256, 1.00, 8.0},         // 128x128x64 with eight waves, 3-stage ring (96 KB); force_tn = 10 / tuned table
    {128, 160, 64, 512, 1.10, 8.0},         // 128x160x64, four waves with 32x160 sub-tiles (one wave column): 160 divides every channel
                                            // count of this UNet (320 k), so no column of a tile is wasted; force_tn = 11 / tuned table
    {128, 320, 64, 256, 2.00, 10.0},        // 128x320x64, eight waves (4 x 2), 32x160 sub-tiles, 2-buffer ring (112 KB, 1 block/CU): five whole
                                            // GEGLU groups per tile - the feed-forward GEMMs (N = 8C) tile without a partial round; force_tn = 12
    // 12..14: PERSISTENT stream-K instances (one block per CU walks (tile, K-range) items, fix-up in the kernel, no reduce pass;
    //         two-group ping-pong K loop over a 3+ stage ring):
    {256, 160, 64, 256, 1.6, 8.0},          // 12: 256x160x64, eight waves in one column (32x160 sub-tiles), 3-stage ring (156 KB); force_tn = 13
    {256, 256, 32, 256, 1.2, 9.0},          // 13: 256x256x32, eight waves (4 x 2) of 64x128 sub-tiles, 4-stage ring (128 KB); force_tn = 14
    {256, 128, 64, 256, 1.3, 7.0},          // 14: 256x128x64, eight waves in one column (32x128 sub-tiles), 3-stage ring (144 KB); force_tn = 15
    {256, 160, 64, 256, 1.4, 8.0},          // 15: 256x160x64 with 16x16x32 MFMAs: eight waves (4 x 2) of 64x80 sub-tiles, 3-stage ring; force_tn = 16
};

static bool cfg_persistent(int c) { return c >= 12; }

// number of blocks of a persistent launch: one per CU, fewer when the iteration space is small (>= 4 K-tiles per block),
// a multiple of 8 so the XCD-contiguous remap applies
static int persist_grid(const GemmArgs& a, int c) {
  const int n_cu = n_cus();                            // of the current device (conv_halo.hip)
  const TileCfg& T = kCfg[c];
  const long tiles = (long)(a.ups2 ? 4 * cdiv(a.M4, T.bm) : cdiv(a.M, T.bm)) * cdiv(a.N, T.bn);
  const long total = tiles * (a.K / T.bk);
  long g = total / 4; if (g > n_cu) g = n_cu; if (g < 1) g = 1;
  if (g >= 8) g &= ~7L;
  return (int)g;
}

// what an instance can run: K-tile alignment of every K segment, and what its epilogue covers
static bool cfg_applicable(const GemmArgs& a, int c) {
  const TileCfg& T = kCfg[c];
  if (c == 3 || c == 4 || c == 5 || c == 13) return false;   // retired / unused instances (13: the 256x256 tile spills - 128 accumulators + the ping-pong loop) (EXPERIMENTS.md); ids kept so the tuned table's numbering is stable
  if (a.K % T.bk != 0 || a.Cin % T.bk != 0 || a.cx0 % T.bk != 0 || a.Ktaps % T.bk != 0 || (a.Ktaps < a.K && a.cs0 % T.bk != 0)) return false;
  const bool col160 = (c == 10 || c == 11 || c == 12 || c == 15);   // the 160 / 320-column epilogue: bias | folded LayerNorm, row bias, residual (+ GEGLU on 320)
  if (col160 && (a.rowstats_out || a.act || a.out_f32 || (a.N & 7) || (c != 11 && a.geglu))) return false;
  if (cfg_persistent(c) && ((a.N & 7) || a.out_f32)) return false;   // stream-K owners finish through the coalesced bf16 epilogue
  return true;
}

static double plan_cost(const GemmArgs& a, int c, int sk) {
  const TileCfg& T = kCfg[c];
  const int nkt = a.K / T.bk;
  const long tiles = (long)cdiv(a.M, T.bm) * cdiv(a.N, T.bn);
  if (cfg_persistent(c)) {                                     // every block gets total / grid K-tiles; ~2 tile boundaries per block
    const int g = persist_grid(a, c);
    return (double)cdiv((int)(tiles * nkt), g) * T.per_ktile + T.fixed * (1.0 + (double)tiles / g);
  }
  const long nb = tiles * sk;
  const double t_block = cdiv(nkt, sk) * T.per_ktile + T.fixed;
  const int per_round = 256;                                   // CUs
  double rounds = (double)((nb + per_round - 1) / per_round);
  if (T.slots == 512) {                                        // two co-resident blocks overlap each other's stalls
    if (nb <= 256) rounds = 1.2; 
  }
  double t = rounds * t_block;
  if (sk > 1) t += 12.0 + ((double)a.M * a.N * 4.0 * (sk + 1) / 5.0e12) / 0.45e-6;   // reduce pass: ~5 us + traffic
  return t;
}

#include "gemm_tuned.h"

// Tuning aid (scripts/tune_in_situ.py): plan overrides set at run time take precedence over the compiled table, so a
// candidate plan can be timed inside the real UNet pass - with the shape's real epilogue, neighbours and cache state.
static std::vector<TunedPlan> g_plan_overrides;
void dmx_gemm_plan_override_set(int M, int N, int K, int st, int ups, int cfg, int sk) {
  dmx_plan_epoch_bump();
  if (cfg < 0) { g_plan_overrides.clear(); return; }
  for (TunedPlan& tp : g_plan_overrides)
    if (tp.M == M && tp.N == N && tp.K == K && tp.st == st && tp.ups == ups) { tp.cfg = cfg; tp.sk = sk; return; }
  g_plan_overrides.push_back(TunedPlan{M, N, K, 0, st, ups, cfg, sk});
}

// force_tn -> plan id: 1 -> 1 (128x64), 2 -> 0 (128x128), n >= 3 -> n - 1
static int force_to_cfg(int force_tn) { return force_tn == 1 ? 1 : force_tn == 2 ? 0 : force_tn - 1; }

void dmx_gemm_plan(const GemmArgs& a, int* cfg_out, int* splitk_out, int* ktps_out) {
  const bool no_split = a.rowstats_out || a.ln_stats || a.geglu || a.act || (a.N % 4) != 0;   // those epilogues live in the GEMM kernel
  auto take = [&](int c, int sk) -> bool {
    if (c < 0 || c >= 16 || !cfg_applicable(a, c)) return false;
    const int nkt = a.K / kCfg[c].bk;
    if (cfg_persistent(c)) { *cfg_out = c; *splitk_out = 1; *ktps_out = nkt; return true; }
    if (sk < 1 || (sk > 1 && (no_split || nkt / sk < 4))) return false;
    const int ktps = cdiv(nkt, sk);
    *cfg_out = c; *splitk_out = cdiv(nkt, ktps); *ktps_out = ktps;
    return true;
  };
  if (!a.force_tn && !a.force_splitk && (a.N % 4) == 0) {
    for (const TunedPlan& tp : g_plan_overrides)
      if (tp.M == a.M && tp.N == a.N && tp.K == a.K && tp.st == a.stride && tp.ups == (a.ups2 ? 2 : a.ups)) {
        if (take(tp.cfg, tp.sk)) return;
        break;                                                 // not applicable: normal plan
      }
    for (const TunedPlan& tp : kTuned)      // keyed on the GEMM view (M, N, K) + gather flavour; tap structure does not matter
      if (tp.M == a.M && tp.N == a.N && tp.K == a.K && tp.st == a.stride && tp.ups == (a.ups2 ? 2 : a.ups) && take(tp.cfg, tp.sk)) return;
  }
  int best_c = 0, best_sk = 1; double best = 1e300;
  for (int c = 0; c < 16; ++c) {
    const TileCfg& T = kCfg[c];
    if (!cfg_applicable(a, c)) continue;
    if (a.force_tn) { if (c != force_to_cfg(a.force_tn)) continue; }
    else {
      if (c >= 7) continue;                               // eight-wave / 160-column / persistent instances: tuned table or force_tn only
      if (c != 1 && a.N <= 64) continue;
      if ((c == 2 || c == 6) && ((long)a.M * a.N < 256L * 128 * 96)) continue;     // big tiles only for big outputs
      // 3x3 convolutions over <= 128 input channels (the 512^2 / 256^2 levels of the autoencoder, K = 1152): the weight operand is
      // as large as the activation operand per tile, so the 256-row tile's reuse buys nothing and its longer prologue / epilogue
      // shows - measured at batch 32 (scripts/attic/vae_conv_probe.py): 128->128 3.54 vs 3.79 ms, 128->256 1.61 vs 1.76 ms
      if (c == 2 && a.ksize == 3 && !a.direct && a.Cin <= 128 && a.Ktaps == a.K) continue;
    }
    const int nkt = a.K / T.bk;
    const int max_sk = (no_split || cfg_persistent(c)) ? 1 : 16;
    for (int sk = 1; sk <= max_sk; ++sk) {
      if (a.force_splitk && sk != a.force_splitk && !cfg_persistent(c)) continue;
      if (sk > 1 && nkt / sk < (T.bk == 64 ? 4 : 8)) break;
      if (c == 6 && !a.force_tn && nkt / sk < 40) break;      // the warp-specialised loop pays off from ~2.5k of K per block
      const double cst = plan_cost(a, c, sk);
      if (cst < best) { best = cst; best_c = c; best_sk = sk; }
    }
  }
  const int nkt = a.K / kCfg[best_c].bk;
  int ktps = cdiv(nkt, best_sk);
  best_sk = cdiv(nkt, ktps);
  *cfg_out = best_c; *splitk_out = best_sk; *ktps_out = ktps;
}

bool dmx_gemm_colstats_ok(const GemmArgs& a) {
  if (a.out_f32 || a.geglu || (a.N & 7) || (a.ldo & 7) || (a.res && (a.ldres & 7)) || a.cs_rows <= 0) return false;
  int c, sk, ktps;
  dmx_gemm_plan(a, &c, &sk, &ktps);
  if (sk > 1 || c == 6) return false;                        // the reduce pass finishes those tiles; the warp-specialised instance has no twin
  if (c == 12 || c == 15) return false;                      // persistent 160-column instances: the statistics cost the producer +5 us per launch
                                                             // (measured in situ: 53.8 -> 59.1 us), as much as the consumer saves
  const int rows = a.ups2 ? a.M4 : a.M;
  return a.cs_rows % kCfg[c].bm == 0 && rows % a.cs_rows == 0;
}

int dmx_gemm_persist_blocks(const GemmArgs& a) {
  int c, sk, ktps;
  dmx_gemm_plan(a, &c, &sk, &ktps);
  return cfg_persistent(c) ? persist_grid(a, c) : 0;
}

int dmx_gemm_tiles_n(const GemmArgs& a) {
  int c, sk, ktps;
  dmx_gemm_plan(a, &c, &sk, &ktps);
  return cdiv(a.N, kCfg[c].bn);
}

// split-K: sk fp32 planes of the output.  Persistent stream-K: 256 bytes of flags per 64 blocks, then one accumulator slab per block.
static size_t persist_flag_bytes(int grid) { return align_up((size_t)grid * sizeof(int), 256); }
size_t dmx_gemm_workspace_bytes(const GemmArgs& a) {
  int c, sk, ktps;
  dmx_gemm_plan(a, &c, &sk, &ktps);
  if (cfg_persistent(c)) {
    const int g = persist_grid(a, c);
    return persist_flag_bytes(g) + (size_t)g * kCfg[c].bm * kCfg[c].bn * sizeof(float);
  }
  return sk > 1 ? (size_t)sk * a.M * a.N * sizeof(float) : 0;
}

template <int WM, int TN, int BKT, int NST, int TM = 2, int NP = 0, int NWN = 2, bool PS = false, int MF = 32, bool CS = false>
static void launch_cfg_(const GemmArgs& a, dim3 grid, hipStream_t stream) {
  constexpr int BM = MF * TM * WM, BN = MF * TN * NWN;
  size_t lds = (size_t)NST * (BM + BN) * BKT * 2;
  size_t lds_epi = (size_t)BM * (BN + 4) * sizeof(float);             // fp32 staging tile of the coalesced epilogue
  if (TN == 5) lds_epi /= 2;                                           // the 160 / 320-column tiles stage half their rows per pass
  else if (lds_epi > 152 * 1024) lds_epi /= WM;                        // the 256x256 tiles stage one 64-row slab (one wave row) per pass
  if (lds_epi > lds) lds = lds_epi;
  if (a.ln_stats || TN == 5) lds += (size_t)BM * 2 * sizeof(float);    // (mean, rstd) per row of the folded LayerNorm
  if (TN == 5) lds += (size_t)3 * BN * sizeof(float);                  // the tile's column vectors (160 / 320-column epilogue)
  {
    char sym[112];
    snprintf(sym, sizeof(sym), "void dmx_gemm_kernel<%d, %d, %d, %d, %d, %d, %d, %s, %d, %s>(GemmArgs)", WM, TN, BKT, NST, TM, NP, NWN, PS ? "true" : "false", MF, CS ? "true" : "false");
    dmx_profile_note_symbol(sym);
  }
  GemmArgs ap = a;                                                     // (the prefetch dump slot: 1 KB behind everything else, where it fits)
  ap.pf_dump_off = 0;
  if (a.pf_bytes[0] > 0 && lds + 1024 <= 163840 && (PS ? false : true)) { ap.pf_dump_off = (int)align_up(lds, 16); lds = (size_t)ap.pf_dump_off + 1024; }
  static bool attr[64] = {};                          // the dynamic-LDS opt-in is per device
  int dev = 0; (void)hipGetDevice(&dev);
  if (dev < 64 && !attr[dev]) { (void)hipFuncSetAttribute((const void*)dmx_gemm_kernel<WM, TN, BKT, NST, TM, NP, NWN, PS, MF, CS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lds + 1040 + BM * 2 * sizeof(float) + 3 * BN * sizeof(float) > 163840 ? 163840 : lds + 1040 + BM * 2 * sizeof(float) + 3 * BN * sizeof(float))); attr[dev] = true; }
  hipLaunchKernelGGL((dmx_gemm_kernel<WM, TN, BKT, NST, TM, NP, NWN, PS, MF, CS>), grid, dim3(64 * (NWN * WM + NP)), lds, stream, ap);
}

// the statistics-emitting twin of an instance is used exactly when GemmArgs.colstats is set (the warp-specialised instance has none)
template <int WM, int TN, int BKT, int NST, int TM = 2, int NP = 0, int NWN = 2, bool PS = false, int MF = 32>
static void launch_cfg(const GemmArgs& a, dim3 grid, hipStream_t stream) {
  if constexpr (NP == 0 && !(PS && TN == 5)) { if (a.colstats) { launch_cfg_<WM, TN, BKT, NST, TM, NP, NWN, PS, MF, true>(a, grid, stream); return; } }
  launch_cfg_<WM, TN, BKT, NST, TM, NP, NWN, PS, MF, false>(a, grid, stream);
}

int dmx_gemm_launch(GemmArgs a, void* workspace, size_t workspace_bytes, hipStream_t stream) {
  DMX_REQUIRE(a.M > 0 && a.N > 0 && a.K > 0, "gemm: empty problem M=%d N=%d K=%d", a.M, a.N, a.K);
  DMX_REQUIRE(a.K % 32 == 0, "gemm: K=%d must be a multiple of 32", a.K);
  DMX_REQUIRE(a.ldw % 8 == 0 && a.ldx0 % 8 == 0, "gemm: leading dimensions must be multiples of 8 (ldw=%d ldx0=%d)", a.ldw, a.ldx0);
  DMX_REQUIRE(a.cx0 % 32 == 0 && a.Cin % 32 == 0, "gemm: channel splits must be multiples of 32 (Cin=%d cx0=%d)", a.Cin, a.cx0);
  DMX_REQUIRE(a.Ktaps % 32 == 0 && a.Ktaps <= a.K, "gemm: bad Ktaps=%d K=%d", a.Ktaps, a.K);
  if (a.Ktaps < a.K) DMX_REQUIRE(a.s0 != nullptr && a.cs0 % 32 == 0, "gemm: shortcut segment needs s0 and aligned cs0");
  if (a.ln_stats) DMX_REQUIRE(a.ln_c1 && a.ln_c2 && a.ln_tiles > 0 && a.ln_C > 0 && !a.out_f32 && a.N % 8 == 0 && a.ldo % 8 == 0, "gemm: folded LayerNorm needs c1/c2, bf16 output and N %% 8 == 0");
  if (a.rowstats_out) DMX_REQUIRE(!a.out_f32 && a.N % 8 == 0 && a.ldo % 8 == 0 && !a.geglu, "gemm: row statistics need the bf16 coalesced epilogue");
  if (a.act) DMX_REQUIRE(a.act == 1 && !a.out_f32 && !a.geglu && a.N % 8 == 0 && a.ldo % 8 == 0 && (a.res == nullptr || a.ldres % 8 == 0), "gemm: the GELU epilogue needs the bf16 coalesced path (N %% 8 == 0)");
  if (a.geglu) DMX_REQUIRE((a.bias || a.ln_stats) && a.N % 128 == 0 && !a.out_f32 && !a.res && !a.rowbias && a.ldo % 8 == 0, "gemm: GEGLU needs bias, N%%128==0, bf16 out");
  if (a.ups2) {
    DMX_REQUIRE(a.ksize == 2 && !a.direct && !a.ups && a.stride == 1 && a.OH == a.IH && a.OW == a.IW && a.M4 > 0 && a.M == 4 * a.M4 &&
                a.Ktaps == a.K && a.K == 4 * a.Cin && a.w_phase_stride > 0, "gemm: inconsistent phase-decomposed upsample conv arguments");
    DMX_REQUIRE(!a.res && !a.rowbias && !a.geglu && !a.act && !a.ln_stats && !a.rowstats_out, "gemm: the phase-decomposed upsample conv takes a bias only");
  }
  DMX_REQUIRE(a.force_tn < 4 || a.force_tn > 6, "gemm: tile instance %d was retired", a.force_tn);
  if (a.colstats) DMX_REQUIRE(dmx_gemm_colstats_ok(a), "gemm: this plan cannot emit column statistics (split-K, fp32 / GEGLU output or tiles straddling samples)");
  if (a.force_tn) DMX_REQUIRE(a.force_tn <= 16 && cfg_applicable(a, force_to_cfg(a.force_tn)), "gemm: tile instance %d cannot run this problem (K-tile alignment / epilogue)", a.force_tn);
  int rc = dmx_zero_page(&a.zeros);
  if (rc) return rc;
  int c, sk, ktps;
  dmx_gemm_plan(a, &c, &sk, &ktps);
  a.splitk = sk; a.kt_per_split = ktps; a.persist = 0;
  const TileCfg& T = kCfg[c];
  dim3 grid((a.ups2 ? 4 * cdiv(a.M4, T.bm) : cdiv(a.M, T.bm)) * cdiv(a.N, T.bn), sk, 1);
  if (cfg_persistent(c)) {
    const int g = persist_grid(a, c);
    const size_t fb = persist_flag_bytes(g), need = fb + (size_t)g * T.bm * T.bn * sizeof(float);
    if (workspace == nullptr || workspace_bytes < need) {
      dmx_set_error("gemm: the persistent stream-K plan needs %zu bytes of workspace, got %zu", need, workspace_bytes);
      return DMX_ERR_WORKSPACE;
    }
    a.persist = 1; a.partial = (float*)((char*)workspace + fb); a.err = dmx_dev_err_words();
    if (!a.flags) {                                            // standalone call: the executors hand out slices of a pool they zero once per forward
      a.flags = (int*)workspace;
      if (const int zr = dmx_zero16_launch(a.flags, fb, stream)) return zr;      // a kernel node, not a memset node: exec.hip Exec::zero_pool
    }
    grid = dim3(g, 1, 1);
  } else if (sk > 1) {
    const size_t need = (size_t)sk * a.M * a.N * sizeof(float);
    if (workspace == nullptr || workspace_bytes < need) {
      dmx_set_error("gemm: split-K needs %zu bytes of workspace, got %zu", need, workspace_bytes);
      return DMX_ERR_WORKSPACE;
    }
    a.partial = (float*)workspace;
  }
  // algorithmic work of this launch: 2*M*N*K flops; bytes = activations read once + weights + output
  const double flops = 2.0 * a.M * (double)a.N * a.K;
  const double bytes = 2.0 * ((double)(a.ups2 ? a.M4 : a.M) * (a.K / (a.direct ? 1 : (a.ksize * a.ksize))) + (double)a.N * a.K * (a.ups2 ? 4 : 1) + (double)a.M * (a.geglu ? a.N / 2 : a.N));
  char tag[96];
  snprintf(tag, sizeof(tag), "M=%d N=%d K=%d ks=%d st=%d ups=%d tn=%d sk=%d", a.M, a.N, a.K, a.direct ? 1 : a.ksize, a.stride, a.ups2 ? 2 : a.ups, c == 0 ? 2 : (c == 1 ? 1 : c + 1), sk);
  {
    ProfScope ps((ProfClass)(PROF_GEMM_CFG0 + c), stream, flops, bytes, tag);
    if (c == 0) launch_cfg<2, 2, 32, 4>(a, grid, stream);
    else if (c == 1) launch_cfg<2, 1, 32, 4>(a, grid, stream);
    else if (c == 2) launch_cfg<4, 2, 64, 3>(a, grid, stream);
    else if (c == 6) launch_cfg<2, 2, 64, 3, 4, 4>(a, grid, stream);
    else if (c == 7) launch_cfg<4, 1, 64, 3, 1>(a, grid, stream);
    else if (c == 8) launch_cfg<4, 2, 32, 4, 1>(a, grid, stream);
    else if (c == 9) launch_cfg<4, 2, 64, 3, 1>(a, grid, stream);
    else if (c == 10) launch_cfg<4, 5, 64, 2, 1, 0, 1>(a, grid, stream);
    else if (c == 11) launch_cfg<4, 5, 64, 2, 1, 0, 2>(a, grid, stream);
    else if (c == 12) launch_cfg<8, 5, 64, 3, 1, 0, 1, true>(a, grid, stream);
    else if (c == 14) launch_cfg<8, 4, 64, 3, 1, 0, 1, true>(a, grid, stream);
    else if (c == 15) launch_cfg<4, 5, 64, 3, 4, 0, 2, true, 16>(a, grid, stream);
    else { dmx_set_error("gemm: plan id %d has no instance", c); return DMX_ERR_ARG; }
  }
  rc = dmx_check_launch("dmx_gemm_kernel");
  if (rc) return rc;
  if (sk > 1 && !a.defer_reduce) rc = dmx_splitk_reduce_launch(a, stream);
  return rc;
}

// the reduce pass of a split-K GEMM on its own (a: the arguments dmx_gemm_launch ran with, partial / splitk filled in)
int dmx_splitk_reduce_launch(const GemmArgs& a, hipStream_t stream) {
  const size_t total4 = (size_t)a.M * a.N / 4;
  const int blocks = (int)((total4 + 255) / 256);
  char tag[96];
  snprintf(tag, sizeof(tag), "M=%d N=%d K=%d ks=%d st=%d ups=%d tn=0 sk=%d", a.M, a.N, a.K, a.direct ? 1 : a.ksize, a.stride, a.ups2 ? 2 : a.ups, a.splitk);
  ProfScope ps(PROF_SPLITK, stream, 0.0, 4.0 * a.splitk * (double)a.M * a.N, tag);
  hipLaunchKernelGGL(dmx_splitk_reduce_kernel, dim3(blocks), dim3(256), 0, stream, a);
  return dmx_check_launch("dmx_splitk_reduce_kernel");
}

// ------------------------------------------------------------------------- folded-LayerNorm weight preparation
// One wave per output row n: W'[n][k] = bf16(W[n][k] * gamma[k]); c1[n] = sum_k W'[n][k] (the values the MFMA will
// actually multiply); c2[n] = sum_k beta[k] * W[n][k] (+ bias[n]).
__global__ __launch_bounds__(256) void dmx_ln_fold_kernel(const bf16* w_raw, bf16* w_out, const float* gamma, const float* beta,
                                                          const float* bias, float* c1, float* c2, int N, int K) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  float s1 = 0.f, s2 = 0.f;
  for (int k = lane * 8; k < K; k += 512) {
    float w[8]; unpack_bf8(*(const u32x4*)(w_raw + (size_t)n * K + k), w);
    float o[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { o[i] = w[i] * gamma[k + i]; s2 += beta[k + i] * w[i]; }
    const u32x4 pk = pack_bf8(o);
    float r[8]; unpack_bf8(pk, r);
#pragma unroll
    for (int i = 0; i < 8; ++i) s1 += r[i];
    *(u32x4*)(w_out + (size_t)n * K + k) = pk;
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) { s1 += __shfl_xor(s1, d); s2 += __shfl_xor(s2, d); }
  if (lane == 0) { c1[n] = s1; c2[n] = s2 + (bias ? bias[n] : 0.f); }
}
int dmx_ln_fold_launch(const bf16* w_raw, bf16* w_out, const float* gamma, const float* beta, const float* bias,
                       float* c1, float* c2, int N, int K, hipStream_t stream) {
  DMX_REQUIRE(K % 8 == 0, "ln_fold: K=%d must be a multiple of 8", K);
  hipLaunchKernelGGL(dmx_ln_fold_kernel, dim3(cdiv(N, 4)), dim3(256), 0, stream, w_raw, w_out, gamma, beta, bias, c1, c2, N, K);
  return dmx_check_launch("dmx_ln_fold_kernel");
}

// ------------------------------------------------------------------------- phase weights of the upsample conv
// conv3x3(nearest_x2(x)) at output pixel (2i+pa, 2j+pb) only sees the 2x2 source window starting at (i-1+pa, j-1+pb):
// the taps that land on the same source pixel are summed once here (fp32 sum of the bf16 taps, one rounding),
//   rows: pa=0: {ky=0} | {ky=1,2}     pa=1: {ky=0,1} | {ky=2}      (columns likewise with pb, kx)
// wp[phase = 2*pa+pb][n][(2*ty+tx)*Cin + ci].  2.25x fewer multiply-adds than the gather over the virtual upsampled grid.
__global__ __launch_bounds__(256) void dmx_ups_phase_weights_kernel(const bf16* w3, int ldw3, bf16* wp, int N, int Cin) {
  const size_t total = (size_t)4 * N * 4 * (Cin / 8);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int oc = (int)(i % (Cin / 8)); size_t r = i / (Cin / 8);
    const int tap = (int)(r % 4); r /= 4;
    const int n = (int)(r % N); const int phase = (int)(r / N);
    const int pa = phase >> 1, pb = phase & 1, ty = tap >> 1, tx = tap & 1;
    const int ky0 = pa == 0 ? (ty == 0 ? 0 : 1) : (ty == 0 ? 0 : 2), ky1 = pa == 0 ? (ty == 0 ? 0 : 2) : (ty == 0 ? 1 : 2);
    const int kx0 = pb == 0 ? (tx == 0 ? 0 : 1) : (tx == 0 ? 0 : 2), kx1 = pb == 0 ? (tx == 0 ? 0 : 2) : (tx == 0 ? 1 : 2);
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    for (int ky = ky0; ky <= ky1; ++ky)
      for (int kx = kx0; kx <= kx1; ++kx) {
        float f[8]; unpack_bf8(*(const u32x4*)(w3 + (size_t)n * ldw3 + (size_t)(ky * 3 + kx) * Cin + oc * 8), f);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += f[e];
      }
    *(u32x4*)(wp + ((size_t)phase * N + n) * 4 * Cin + (size_t)tap * Cin + oc * 8) = pack_bf8(acc);
  }
}
int dmx_ups_phase_weights_launch(const bf16* w3, int ldw3, bf16* wp, int N, int Cin, hipStream_t stream) {
  DMX_REQUIRE(Cin % 8 == 0 && ldw3 % 8 == 0 && ldw3 >= 9 * Cin, "ups_phase_weights: Cin=%d / ldw=%d unsupported", Cin, ldw3);
  const size_t total = (size_t)4 * N * 4 * (Cin / 8);
  int blocks = (int)((total + 255) / 256); if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(dmx_ups_phase_weights_kernel, dim3(blocks), dim3(256), 0, stream, w3, ldw3, wp, N, Cin);
  return dmx_check_launch("dmx_ups_phase_weights_kernel");
}
