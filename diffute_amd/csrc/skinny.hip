// Weight-streaming convolution / linear for the SKINNY levels of the UNet (M = B*H*W <= 256 output rows: the 8x8 level at batch 4, the
// 16x16 and 8x8 levels at batch 1 - ResnetBlock2D conv1 / conv2 behind unet(...), app.ipynb:814).
//
// STATUS (round 6): correct (tests/test_ops_gpu.py::test_skinny_conv), exported as dmx_skinny_conv, and NOT taken by the model executors: in a captured
// graph it measures 32 us (42.9 with the GroupNorm fused) for 256 x 1280 x 11520 against 25 + ~10 us for the tiled split-K GEMM + GroupNorm launch it
// would replace.  EXPERIMENTS.md round 6 item 2 has the per-block timelines and ablations (profiles/r06_skinny_probe.jsonl): cross-block exchange 10.7 us,
// K loop 12.4 us against a 7 us load floor of its double-buffered chunks, prologue 3 us (9.5 us with the statistics round trip of the fused GroupNorm).
//
// The design.  There the problem is a weight stream: 256 x 1280 x 11520 moves 29.5 MB of weights for 0.65 MB of activations, the HBM floor (4.7 us) is
// above the matrix-pipe floor (3 us), and every weight element is used by ONE block.  So:
//   * the layer's weights exist a second time in MFMA-FRAGMENT order (dmx_skinny_pack_launch, at pack time): fragment f of n-block nb is the 1 KB a wave
//     needs as the A operand of one v_mfma_f32_32x32x16 (lane l: row n = l & 31, k = 8 (l >> 5) .. + 8), fragments ordered (segment, 16-channel k-step, tap):
//     the fragments of a chunk are ONE contiguous run, staged with coalesced 1-KB LDS-DMA instructions (68 KB per CU in flight);
//   * a block = (32 output channels) x (one K slice); K is sliced BY CHANNEL RANGE, so a slice's activations are [M][64 channel] chunks staged ONCE in LDS
//     (128-byte rows, 16-byte pieces XOR-swizzled with (row >> 1) & 7 on the source side: conflict-free ds_read_b128) and the nine taps are nine shifted
//     fragment reads of a chunk (rows outside the image read the buffer's zero row);
//   * the GroupNorm + SiLU in front of the conv (ResnetBlock2D norm1 / norm2) is applied to the staged chunk in place, from the DmxStat records of the
//     input(s) - no separate GroupNorm launch;
//   * eight symmetric waves (two per SIMD): each stages its eighth of chunk c + 1, computes its eighth of chunk c's (k-step, tap) items, normalises what it
//     staged; one barrier per chunk.  (First built with four compute + four loader waves and a hand-scheduled asm stream: one compute wave per SIMD issues
//     in order, and hipcc copies / spills registers that asm ds_reads are still filling - EXPERIMENTS.md.)
//   * tiles x slices = N / 32 x S ~ one block per CU: all 256 CUs stream disjoint weight slices.  The partial sums meet in the kernel: every block folds its
//     waves' accumulators through LDS in wave order and publishes the [M][32] fp32 tile write-through (sc1) + flag; then block s of a tile finishes rows
//     [M s / S, M (s + 1) / S): adds the S tiles in slice order (fixed -> bit-reproducible), bias + time-embedding row + residual, one rounding, the
//     DmxStat records of the output.  The blocks of a tile have adjacent block ids, so they are dispatched together; the wait is bounded (~40 ms) and a
//     block that gives up RAISES the device error (common.h) - never a silent wrong tile.
#include "common.h"
#include "kernels.h"
#include <stdio.h>

namespace {

constexpr int SK_NT = 512;                 // 4 compute + 4 loader waves
constexpr int SK_CKS = 4;                  // k-steps (16 channels) per staged chunk
constexpr int SK_ROWB = SK_CKS * 32;       // LDS bytes per staged pixel row (64 channels)
constexpr int SK_WBUF = SK_CKS * 9 * 1024; // a chunk's weight fragments: <= 36 x 1 KB
constexpr int SK_LDT = 36;                 // floats per row of a wave's accumulator tile in LDS (32 + 4: conflict-free b128 rows)

// LDS map.  K loop: two activation chunk buffers [M + 1][128 B] (row M stays zero: what a masked tap reads), two weight chunk buffers (the chunk's
// fragments in item order), GroupNorm tables.  Epilogue: four accumulator tiles reuse everything.
template <int MB> struct SkL {
  static constexpr int M = 32 * MB;
  static constexpr int BUF = (M + 1) * SK_ROWB;
  static constexpr int BUF0 = 0, BUF1 = BUF;
  static constexpr int WB0 = 2 * BUF, WB1 = WB0 + SK_WBUF;
  static constexpr int GST = WB1 + SK_WBUF;                                   // (mean, rstd) per (sample, group slot): 4 samples x 16 slots x 8 B
  static constexpr int COEF = GST + 512;                                      // (a, s) per (sample, GroupNorm'ed channel of the slice): 4 x SK_COEF_CH x 8 B
  static constexpr int KLOOP = COEF + 4 * SK_COEF_CH * 8;
  static constexpr int RED = 4 * M * SK_LDT * 4;                              // epilogue: four [M][36] fp32 tiles (tile 0 ends up holding their sum; tile 1 the rounded outputs)
  static constexpr int TOTAL = KLOOP > RED ? KLOOP : RED;
};

// (DBG: p.dbg's measurement switches are live - uniform branches; the one instantiation serves both.)
template <int MB, bool DBG>
__global__ __launch_bounds__(SK_NT, 2) void dmx_skinny_kernel(const SkinnyArgs p) {
  typedef SkL<MB> L;
  constexpr int M = L::M;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int S = p.S;
  const int tile = blockIdx.x / S, sl = blockIdx.x - tile * S;
  const int n0 = tile * 32;
  const SkinnyPlanSlice& PS = p.plan[sl];
  const int nchunk = PS.nchunk;
  const int HW = p.H * p.W;
  const int wsh = p.wsh, hwsh = p.hwsh;

  long long tm[6] = {0, 0, 0, 0, 0, 0};
  if (p.timing) tm[0] = __builtin_amdgcn_s_memrealtime();
  // the zero rows (row M of both activation buffers: what a masked tap reads) - visible after the first barrier
  if (t < 16) *(u32x4*)(smem + (t < 8 ? L::BUF0 : L::BUF1) + M * SK_ROWB + (t & 7) * 16) = u32x4{0u, 0u, 0u, 0u};

  f32x16 acc[MB];
#pragma unroll
  for (int i = 0; i < MB; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  // ================================================================= K loop: eight symmetric waves (two per SIMD)
  // Every wave stages its eighth of chunk c + 1 (LDS-DMA: activations and weight fragments), computes its eighth of chunk c's items, then normalises the
  // pieces it staged itself; one barrier per chunk.  Two waves per SIMD hide each other's LDS latency and MFMA issue: the stream needs no hand
  // scheduling (a single compute wave per SIMD with dedicated loader waves was built first and measured: EXPERIMENTS.md round 6).
  const int lr = lane & 31, lh = lane >> 5;
  const bool gn_any = p.gn_groups > 0;
  auto stage = [&](int c) {
    const SkinnyChunk ch = PS.ch[c];
    const SkinnySeg& sg = p.seg[ch.seg];
    const int buf = (c & 1) ? L::BUF1 : L::BUF0;
    const int r = lane >> 3, slot = lane & 7;           // one DMA instruction = 8 pixel rows x 8 pieces of 16 B
    for (int j = wave; j < M / 8; j += 8) {
      const int px = 8 * j + r;
      const int g = slot ^ ((px >> 1) & 7);              // (128-byte rows: two rows per bank sweep, so the swizzle key is (row >> 1) & 7)
      const bf16* src = (g < 2 * ch.nks) ? sg.x + (size_t)px * sg.ld + (ch.k0 * 16 + g * 8) : p.zeros;
      dmx_dma16(src, (unsigned)(buf + j * 1024));
    }
    // ... and the chunk's weight fragments: nks x taps consecutive 1-KB fragments of this n-block, in item order
    const int nfr = (int)ch.nks * sg.taps;
    const bf16* wsrc = p.wp + ((size_t)tile * p.frags_per_nb + ch.frag) * 512 + lane * 8;
    const unsigned wb = (c & 1) ? L::WB1 : L::WB0;
    for (int j = wave; j < nfr; j += 8) dmx_dma16(wsrc + (size_t)j * 512, wb + (unsigned)(j * 1024));
  };
  // in place: y = silu(x a + s) for the pieces this lane requested itself (its own vmcnt wait is all the ordering it needs)
  auto normalise = [&](int c) {
    const SkinnyChunk ch = PS.ch[c];
    const SkinnySeg& sg = p.seg[ch.seg];
    if (!sg.st || (DBG && (p.dbg & 8))) return;
    const int buf = (c & 1) ? L::BUF1 : L::BUF0;
    const int r = lane >> 3, slot = lane & 7;
    const float* cf = (const float*)(smem + L::COEF) + (size_t)ch.coef0 * 2;
    for (int j = wave; j < M / 8; j += 8) {
      const int px = 8 * j + r;
      const int g = slot ^ ((px >> 1) & 7);
      if (g >= 2 * ch.nks) continue;
      const int b = px >> hwsh;
      char* q = smem + buf + j * 1024 + lane * 16;
      const u32x4 xv = *(const u32x4*)q;
      float x[8]; unpack_bf8(xv, x);
      const float* cc = cf + ((size_t)b * SK_COEF_CH + g * 8) * 2;
      float y[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float v = __builtin_fmaf(x[e], cc[2 * e], cc[2 * e + 1]);
        if (p.silu) v *= __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896f * v));
        y[e] = v;
      }
      *(u32x4*)q = pack_bf8(y);
    }
  };
  // (a, s) of every GroupNorm'ed channel of the slice for every sample, from the group statistics in GST (slot = group - g_first of the segment's piece)
  auto coef_tables = [&]() {
    float* cf = (float*)(smem + L::COEF);
    const float* gs = (const float*)(smem + L::GST);
    for (int c = 0; c < nchunk; ++c) {
      const SkinnyChunk ch = PS.ch[c];
      const SkinnySeg& sg = p.seg[ch.seg];
      if (!sg.st) continue;
      for (int i = t; i < p.B * 16 * ch.nks; i += SK_NT) {
        const int cl = i % (16 * ch.nks), b = i / (16 * ch.nks);
        const int cseg = ch.k0 * 16 + cl;            // channel inside the segment
        const int gslot = PS.slot0[ch.seg] + (sg.gn_c0 + cseg) / p.gn_cpg - PS.g_first[ch.seg];
        const float mean = gs[(b * 16 + gslot) * 2], rstd = gs[(b * 16 + gslot) * 2 + 1];
        const float a = rstd * sg.gamma[cseg];
        cf[((size_t)b * SK_COEF_CH + ch.coef0 + cl) * 2] = a;
        cf[((size_t)b * SK_COEF_CH + ch.coef0 + cl) * 2 + 1] = sg.beta[cseg] - mean * a;
      }
    }
  };
  if (nchunk > 0) stage(0);
  if (gn_any && PS.g_count > 0) {
    // 16 threads per (sample, group slot): sum the group's channel records (integers: exact, any order), mean / variance in double.  Every record load of a
    // thread is issued before the first is used (cpg / 16 <= 5 records: the loads are L2 / memory round trips, ~2 us each when they are serial)
    const int sub = t & 15, pair = t >> 4;             // 32 pairs per round
    for (int pr = pair; pr < p.B * PS.g_count; pr += 32) {
      const int b = pr / PS.g_count, gslot = pr - b * PS.g_count, g = PS.slot_group[gslot];
      long long r0[5], r1[5], r2[5];
#pragma unroll
      for (int k5 = 0; k5 < 5; ++k5) {
        const int cg = g * p.gn_cpg + sub + 16 * k5;
        const long long* rec = nullptr;                // which GroupNorm'ed segment holds channel cg of the concatenated tensor
        if (cg < (g + 1) * p.gn_cpg) {
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const SkinnySeg& sg = p.seg[k];
            if (k < p.nseg && sg.st && cg >= sg.gn_c0 && cg < sg.gn_c0 + sg.C) rec = sg.st + ((size_t)b * sg.C + (cg - sg.gn_c0)) * DMX_STAT_WORDS;
          }
        }
        r0[k5] = rec ? rec[0] : 0; r1[k5] = rec ? rec[1] : 0; r2[k5] = rec ? rec[2] : 0;
      }
      long long s0 = 0, qh = 0, ql = 0;
#pragma unroll
      for (int k5 = 0; k5 < 5; ++k5) { s0 += r0[k5]; qh += r1[k5]; ql += r2[k5]; }
#pragma unroll
      for (int d = 1; d < 16; d <<= 1) { s0 += __shfl_xor(s0, d); qh += __shfl_xor(qh, d); ql += __shfl_xor(ql, d); }
      if (sub == 0) {
        const double n = (double)p.gn_cpg * (double)HW;
        const double mean = dmx_stat_sum(s0) / n;
        double var = dmx_stat_sumsq(qh, ql) / n - mean * mean;
        var = var < 0.0 ? 0.0 : var;
        float* gs = (float*)(smem + L::GST);
        gs[(b * 16 + gslot) * 2] = (float)mean; gs[(b * 16 + gslot) * 2 + 1] = (float)(1.0 / __builtin_sqrt(var + (double)p.gn_eps));      // (as dmx_gn_apply_kernel: same bits on every path)
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                        // #0: zero rows + group statistics published; chunk 0 landed (raw)
  if (gn_any) {
    coef_tables();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                      // #1: the slice's coefficient table published (every lane reads entries other lanes wrote)
    if (nchunk > 0) normalise(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                      // #2: chunk 0 normalised
  }
  if (p.timing) tm[1] = __builtin_amdgcn_s_memrealtime();
  {
    const int Hm = p.H - 1, Wm = p.W - 1;
    const unsigned wlane = (unsigned)(lane * 16);
    for (int c = 0; c < nchunk; ++c) {
      if (c + 1 < nchunk && !(DBG && (p.dbg & 2))) stage(c + 1);       // (buffers (c + 1) & 1 were last read in chunk c - 1: every wave passed barrier c - 1)
      const SkinnyChunk ch = PS.ch[c];
      const int taps = p.seg[ch.seg].taps, nks = ch.nks, n = nks * taps;
      const int lo = (n * wave) >> 3, hi = (n * (wave + 1)) >> 3;      // this wave's eighth of the chunk's items, flattened tap-major (i' = tap nks + kk)
      const unsigned abuf = (c & 1) ? L::BUF1 : L::BUF0, wbuf = (c & 1) ? L::WB1 : L::WB0;
      int tp = 0, kk = lo;
      while (kk >= nks) { kk -= nks; ++tp; }
      int cur_tap = -1; unsigned sw = 0; unsigned rowsel[MB];
      for (int ip = lo; ip < hi; ++ip) {
        const int tap = taps == 9 ? tp : 4;
        if (tap != cur_tap) {                            // per-tap set-up: the tap-shifted source row of every m-block (the buffer's zero row outside the image)
          const int dy = tap / 3 - 1, dx = tap % 3 - 1;
          const int dlt = (dy << wsh) + dx;
          sw = (unsigned)(((lr + dlt) >> 1) & 7);
#pragma unroll
          for (int mb = 0; mb < MB; ++mb) {
            const int px = mb * 32 + lr, y = ((px >> wsh) & Hm) + dy, x = (px & Wm) + dx;
            const bool ok = (unsigned)y <= (unsigned)Hm && (unsigned)x <= (unsigned)Wm;
            rowsel[mb] = abuf + (unsigned)((ok ? px + dlt : M) * SK_ROWB);
          }
          cur_tap = tap;
        }
        const unsigned pos = (unsigned)(kk * taps + (taps == 9 ? tp : 0));
        const unsigned pc = ((unsigned)(2 * kk + lh) ^ sw) << 4;
        if (!(DBG && (p.dbg & 4))) {
          const bf16x8 afrag = *(const bf16x8*)(smem + wbuf + pos * 1024u + wlane);
          bf16x8 bfr[MB];
#pragma unroll
          for (int mb = 0; mb < MB; ++mb) bfr[mb] = *(const bf16x8*)(smem + rowsel[mb] + pc);
          if (!(DBG && (p.dbg & 1))) {
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) acc[mb] = DMX_MFMA_32x32x16(afrag, bfr[mb], acc[mb]);
            // pin the issue order the scheduler would otherwise undo (one read, wait, one MFMA - eight exposed LDS latencies per item): all reads, then the MFMAs
            __builtin_amdgcn_sched_group_barrier(0x100, MB + 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, MB, 0);
          }
        }
        if (++kk == nks) { kk = 0; ++tp; }
      }
      if (c + 1 < nchunk) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (gn_any) normalise(c + 1);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                    // chunk c + 1 ready; every wave is done with chunk c's buffers
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (every path from an LDS-DMA request to s_endpgm passes a drain: scripts/isa_audit.py rule L checks the binary)
  }
  if (p.timing) tm[2] = __builtin_amdgcn_s_memrealtime();
  // ================================================================= epilogue
  // fold the four compute waves' accumulators through LDS (wave order), publish the [M][32] fp32 tile, finish this block's rows from the S tiles
  float* red = (float*)smem;
  {
    // the eight waves' accumulators -> four [M][36] fp32 tiles: waves 0-3 write, then waves 4-7 add theirs to the tile of wave w - 4 (same lanes, same
    // addresses: tile_w = acc_w + acc_(w+4)); fixed order -> deterministic
    float* mine = red + (size_t)(wave & 3) * M * SK_LDT;
    if (wave < 4) {
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 v = {acc[mb][4 * g], acc[mb][4 * g + 1], acc[mb][4 * g + 2], acc[mb][4 * g + 3]};
          *(f32x4*)(mine + (mb * 32 + lr) * SK_LDT + 8 * g + 4 * lh) = v;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wave >= 4) {
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float* q = mine + (mb * 32 + lr) * SK_LDT + 8 * g + 4 * lh;
          const f32x4 o = *(const f32x4*)q;
          const f32x4 v = {o[0] + acc[mb][4 * g], o[1] + acc[mb][4 * g + 1], o[2] + acc[mb][4 * g + 2], o[3] + acc[mb][4 * g + 3]};
          *(f32x4*)q = v;
        }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  const size_t slab_elems = (size_t)M * 32;
  const size_t tile_slot = (size_t)tile * S;
  const int r_lo = (M * sl) / S, r_hi = (M * (sl + 1)) / S;    // this block's rows of the tile
  auto wave_sum = [&](int row, int c8, float* v) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const f32x4 a = *(const f32x4*)(red + ((size_t)k * M + row) * SK_LDT + c8 * 8), b = *(const f32x4*)(red + ((size_t)k * M + row) * SK_LDT + c8 * 8 + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[e] += a[e]; v[4 + e] += b[e]; }
    }
  };
  {
    // tile 0 <- tile 0 + 1 + 2 + 3 (wave order) for every (row, octet): each item is private to one thread; rows of OTHER slices also go to this block's slab
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.slabs + (tile_slot + sl) * slab_elems), 0, (int)(slab_elems * 4), 0x00020000);
    for (int it = t; it < M * 4; it += SK_NT) {
      const int row = it >> 2, c8 = it & 3;
      float v[8]; wave_sum(row, c8, v);
      const f32x4 v0 = {v[0], v[1], v[2], v[3]}, v1 = {v[4], v[5], v[6], v[7]};
      if (row >= r_lo && row < r_hi) {
        *(f32x4*)(red + (size_t)row * SK_LDT + c8 * 8) = v0; *(f32x4*)(red + (size_t)row * SK_LDT + c8 * 8 + 4) = v1;
      } else if (S > 1) {
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v0), rs, (row * 32 + c8 * 8) * 4, 0, 16);          // write-through (sc1)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v1), rs, (row * 32 + c8 * 8 + 4) * 4, 0, 16);
      }
    }
  }
  if (S > 1) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t == 0) {
      __hip_atomic_store(p.flags + tile_slot + sl, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (p.timing) tm[3] = __builtin_amdgcn_s_memrealtime();
      // peers' tiles: bounded spin (the blocks of a tile have adjacent ids: dispatched together).  A lost peer must never hang the GPU and never pass
      // silently: after ~40 ms the block RAISES the device error (common.h) and goes on
      const long long t0 = __builtin_amdgcn_s_memrealtime();
      for (int s = 0; s < S; ++s) {
        if (s == sl) continue;
        while (__hip_atomic_load(p.flags + tile_slot + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
          __builtin_amdgcn_s_sleep(2);
          if (__builtin_amdgcn_s_memrealtime() - t0 > 4000000) { dmx_dev_raise(p.err, DMX_DEVK_SKINNY_PEER, (int)blockIdx.x, tile, s, S); break; }
        }
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __syncthreads();
  if (p.timing) tm[4] = __builtin_amdgcn_s_memrealtime();
  // ---- items (row of this block, octet): sum the S tiles in slice order, + bias + time-embedding row + residual, one rounding, store
  float* stq = red + (size_t)M * SK_LDT;                // (tile 1's region: free once tile 0 holds the sum)
  const int nrows = r_hi - r_lo;
  for (int it = t; it < nrows * 4; it += SK_NT) {
    const int row = r_lo + (it >> 2), c8 = it & 3;
    const int n = n0 + c8 * 8;
    const int b = row >> hwsh;
    f32x4 pv[8][2];
    if (S > 1) {
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        if (s >= S || s == sl) continue;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.slabs + (tile_slot + s) * slab_elems), 0, (int)(slab_elems * 4), 0x00020000);
        pv[s][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (row * 32 + c8 * 8) * 4, 0, 16));       // sc1: served by L2 / the fabric
        pv[s][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (row * 32 + c8 * 8 + 4) * 4, 0, 16));
      }
    }
    u32x4 rr = {0u, 0u, 0u, 0u};
    if (p.res) rr = *(const u32x4*)(p.res + (size_t)row * p.ldres + n);
    float own[8];
    { const f32x4 a = *(const f32x4*)(red + (size_t)row * SK_LDT + c8 * 8), b_ = *(const f32x4*)(red + (size_t)row * SK_LDT + c8 * 8 + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { own[e] = a[e]; own[4 + e] = b_[e]; } }
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = 0.f;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      if (s >= S) continue;
      if (s == sl) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += own[e];
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] += pv[s][0][e]; v[4 + e] += pv[s][1][e]; }
      }
    }
    if (p.bias) {
      const f32x4 b0 = *(const f32x4*)(p.bias + n), b1 = *(const f32x4*)(p.bias + n + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[e] += b0[e]; v[4 + e] += b1[e]; }
    }
    if (p.rowbias) {
      const float* rb = p.rowbias + (size_t)b * p.ldrb + n;
      const f32x4 b0 = *(const f32x4*)rb, b1 = *(const f32x4*)(rb + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[e] += b0[e]; v[4 + e] += b1[e]; }
    }
    if (p.res) {
      float rf[8]; unpack_bf8(rr, rf);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += rf[e];
    }
    const u32x4 pk = pack_bf8(v);
    *(u32x4*)(p.out + (size_t)row * p.ldo + n) = pk;
    if (p.colstats) {
      float f[8]; unpack_bf8(pk, f);
      float* q = stq + (size_t)(it >> 2) * 32 + c8 * 8;
      *(f32x4*)q = f32x4{f[0], f[1], f[2], f[3]}; *(f32x4*)(q + 4) = f32x4{f[4], f[5], f[6], f[7]};
    }
  }
  if (p.colstats) {
    // per-(sample, channel) sums of the ROUNDED outputs of this block's rows: float in row order inside the block, 64-bit fixed point across blocks (DmxStat)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
    const int b_lo = r_lo >> hwsh, b_hi = (r_hi - 1) >> hwsh;
    const int c = t & 31, bi = t >> 5;                   // up to 16 samples per block's row range
    if (nrows > 0 && b_lo + bi <= b_hi) {
      const int b = b_lo + bi;
      const int ra = max(r_lo, b << hwsh), rb = min(r_hi, (b + 1) << hwsh);
      float sa = 0.f, sq = 0.f;
      for (int row = ra; row < rb; ++row) { const float f = stq[(size_t)(row - r_lo) * 32 + c]; sa += f; sq += f * f; }
      dmx_stat_add(p.colstats + ((size_t)b * p.N + n0 + c) * DMX_STAT_WORDS, sa, sq);
    }
  }
  if (p.timing && t == 0) {                            // measurement aid: per-block phase stamps in 10 ns ticks
    tm[5] = __builtin_amdgcn_s_memrealtime();
    long long* o_ = p.timing + (size_t)blockIdx.x * 6;
#pragma unroll
    for (int i = 0; i < 6; ++i) o_[i] = tm[i];
  }
}

// ---- weights [N][ldw] (k = tap * tap_stride + koff + c) -> fragment order: [N / 32][frag f = (segment, k-step, tap)][lane][8]
__global__ __launch_bounds__(256) void dmx_skinny_pack_kernel(const bf16* w, int ldw, bf16* wp, int N, SkinnyPackDesc d) {
  const size_t total = (size_t)(N / 32) * d.frags_per_nb * 64;          // one thread = one lane's 16 bytes
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int lane = (int)(i & 63); size_t r = i >> 6;
    const int f = (int)(r % d.frags_per_nb); const int nb = (int)(r / d.frags_per_nb);
    int sg = 0;
    while (sg + 1 < d.nseg && f >= d.frag0[sg + 1]) ++sg;
    const int fl = f - d.frag0[sg], ks = fl / d.taps[sg], tp = fl - ks * d.taps[sg];
    const int n = nb * 32 + (lane & 31), c = ks * 16 + 8 * (lane >> 5);
    const bf16* src = w + (size_t)n * ldw + (size_t)tp * d.tap_stride[sg] + d.koff[sg] + c;
    *(u32x4*)(wp + i * 8) = *(const u32x4*)src;
  }
}

}  // namespace

static int g_skinny = 1;
extern "C" int dmx_set_skinny(int on) { const int old = g_skinny; g_skinny = on; dmx_plan_switch(DMX_SW_SKINNY, on); return old; }
bool dmx_skinny_enabled() { return g_skinny != 0; }

static bool pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }
static int ilog2(int v) { int s = 0; while ((1 << s) < v) ++s; return s; }

bool dmx_skinny_supported(const SkinnyArgs& a) {
  const int M = a.B * a.H * a.W;
  if (!(M == 64 || M == 128 || M == 256) || !pow2(a.H) || !pow2(a.W) || a.B < 1 || a.B > 4) return false;
  if (a.N <= 0 || a.N % 32 || a.nseg < 1 || a.nseg > 4 || (a.ldo & 7) || (a.res && (a.ldres & 7))) return false;
  for (int k = 0; k < a.nseg; ++k) {
    const SkinnySeg& s = a.seg[k];
    if (!s.x || s.C <= 0 || s.C % 16 || (s.ld & 7) || !(s.taps == 9 || s.taps == 1)) return false;
    if (s.st && (!s.gamma || !s.beta)) return false;
  }
  if (a.gn_groups > 0) { if (a.gn_Ctot % a.gn_groups) return false; }
  return true;
}

int dmx_skinny_frags_per_nb(const SkinnyArgs& a) {
  int f = 0;
  for (int k = 0; k < a.nseg; ++k) f += (a.seg[k].C / 16) * a.seg[k].taps;
  return f;
}

// slices: S = CUs / tiles (<= 8); every slice takes the same share of every segment's k-steps, in chunks of <= 8 k-steps
static int skinny_plan(SkinnyArgs& a) {
  const int tiles = a.N / 32;
  int S = n_cus() / tiles; if (S > 8) S = 8; if (S < 1) S = 1;
  if (a.force_S > 0) S = a.force_S > 8 ? 8 : a.force_S;
  // a slice should hold >= ~4 k-steps of work in total
  int ks_total = 0; for (int k = 0; k < a.nseg; ++k) ks_total += a.seg[k].C / 16;
  while (S > 1 && ks_total / S < 2) --S;
  a.S = S;
  a.frags_per_nb = dmx_skinny_frags_per_nb(a);
  a.gn_cpg = a.gn_groups > 0 ? a.gn_Ctot / a.gn_groups : 1;
  int frag0[4]; { int f = 0; for (int k = 0; k < a.nseg; ++k) { frag0[k] = f; f += (a.seg[k].C / 16) * a.seg[k].taps; } }
  for (int s = 0; s < S; ++s) {
    SkinnyPlanSlice& P = a.plan[s];
    P.nchunk = 0; P.g_count = 0;
    int coef = 0;
    for (int k = 0; k < a.nseg; ++k) {
      const int nks = a.seg[k].C / 16, lo = nks * s / S, hi = nks * (s + 1) / S;
      P.g_first[k] = 0; P.slot0[k] = 0;
      for (int k0 = lo; k0 < hi; k0 += SK_CKS) {
        if (P.nchunk >= SK_MAX_CHUNKS) return DMX_ERR_UNSUPPORTED;
        SkinnyChunk& c = P.ch[P.nchunk++];
        c.seg = (unsigned char)k; c.k0 = (unsigned short)k0; c.nks = (unsigned char)((hi - k0) < SK_CKS ? (hi - k0) : SK_CKS);
        c.frag = frag0[k] + k0 * a.seg[k].taps;
        c.coef0 = (unsigned short)coef; c.pad_ = 0;
        if (a.seg[k].st) coef += 16 * c.nks;
      }
      if (coef > SK_COEF_CH) return DMX_ERR_UNSUPPORTED;
      if (a.seg[k].st && hi > lo) {                    // the groups this segment's piece touches get consecutive slots (a group that straddles two sources gets one per source)
        const int g_lo = (a.seg[k].gn_c0 + lo * 16) / a.gn_cpg, g_hi = (a.seg[k].gn_c0 + hi * 16 - 1) / a.gn_cpg;
        if (g_hi > 255 || P.g_count + (g_hi - g_lo + 1) > 16) return DMX_ERR_UNSUPPORTED;
        P.g_first[k] = (unsigned char)g_lo; P.slot0[k] = (unsigned char)P.g_count;
        for (int g = g_lo; g <= g_hi; ++g) P.slot_group[P.g_count++] = (unsigned char)g;
      }
    }
  }
  return DMX_OK;
}

int dmx_skinny_flag_count(const SkinnyArgs& a_) { SkinnyArgs a = a_; if (skinny_plan(a)) return 0; return a.S > 1 ? (a.N / 32) * a.S : 0; }
static size_t sk_flag_bytes(int n) { return align_up((size_t)n * sizeof(int), 256); }
size_t dmx_skinny_workspace_bytes(const SkinnyArgs& a_) {
  SkinnyArgs a = a_; if (skinny_plan(a)) return 0;
  if (a.S <= 1) return 0;
  const int M = a.B * a.H * a.W;
  return sk_flag_bytes((a.N / 32) * a.S) + (size_t)(a.N / 32) * a.S * M * 32 * sizeof(float);
}

template <int MB, bool DBG> static int skinny_launch_t2(const SkinnyArgs& a, hipStream_t stream) {
  typedef SkL<MB> L;
  DMX_LDS_OPT_IN((dmx_skinny_kernel<MB, DBG>), L::TOTAL);
  char sym[64]; snprintf(sym, sizeof(sym), "void dmx_skinny_kernel<%d, %s>(SkinnyArgs)", MB, DBG ? "true" : "false");
  dmx_profile_note_symbol(sym);
  hipLaunchKernelGGL((dmx_skinny_kernel<MB, DBG>), dim3((a.N / 32) * a.S), dim3(SK_NT), L::TOTAL, stream, a);
  return dmx_check_launch("dmx_skinny_kernel");
}
template <int MB> static int skinny_launch_t(const SkinnyArgs& a, hipStream_t stream) { return skinny_launch_t2<MB, true>(a, stream); }

int dmx_skinny_launch(SkinnyArgs a, void* workspace, size_t workspace_bytes, hipStream_t stream) {
  DMX_REQUIRE(dmx_skinny_supported(a), "skinny conv: unsupported problem (B=%d H=%d W=%d N=%d segments=%d)", a.B, a.H, a.W, a.N, a.nseg);
  DMX_REQUIRE(a.wp && a.out, "skinny conv: null argument");
  int rc = skinny_plan(a);
  if (rc) { dmx_set_error("skinny conv: the slice plan does not fit (too many chunks / groups per slice)"); return rc; }
  rc = dmx_zero_page(&a.zeros);
  if (rc) return rc;
  const int M = a.B * a.H * a.W;
  a.wsh = ilog2(a.W); a.hwsh = ilog2(a.H * a.W);
  a.err = dmx_dev_err_words();
  if (a.S > 1) {
    const int nfl = (a.N / 32) * a.S;
    const size_t fb = sk_flag_bytes(nfl), need = fb + (size_t)nfl * M * 32 * sizeof(float);
    if (!workspace || workspace_bytes < need) { dmx_set_error("skinny conv: needs %zu bytes of workspace, got %zu", need, workspace_bytes); return DMX_ERR_WORKSPACE; }
    a.slabs = (float*)((char*)workspace + fb);
    if (!a.flags) {                                            // standalone call: the executors hand out slices of a pool they zero once per forward
      a.flags = (int*)workspace;
      if (const int zr = dmx_zero16_launch(a.flags, fb, stream)) return zr;
    }
  }
  double K = 0; for (int k = 0; k < a.nseg; ++k) K += (double)a.seg[k].C * a.seg[k].taps;
  char tag[96]; snprintf(tag, sizeof(tag), "M=%d N=%d K=%d S=%d gn=%d", M, a.N, (int)K, a.S, a.gn_groups > 0 ? 1 : 0);
  ProfScope ps(PROF_SKINNY, stream, 2.0 * M * (double)a.N * K, 2.0 * ((double)a.N * K + (double)M * (K / 9.0) + (double)M * a.N), tag);
  if (M == 256) return skinny_launch_t<8>(a, stream);
  if (M == 128) return skinny_launch_t<4>(a, stream);
  return skinny_launch_t<2>(a, stream);
}

int dmx_skinny_pack_launch(const bf16* w, int ldw, bf16* wp, int N, const SkinnyPackDesc& d, hipStream_t stream) {
  DMX_REQUIRE(w && wp && N % 32 == 0 && (ldw & 7) == 0 && d.nseg >= 1 && d.nseg <= 4, "skinny pack: bad arguments");
  const size_t total = (size_t)(N / 32) * d.frags_per_nb * 64;
  int blocks = (int)((total + 255) / 256); if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(dmx_skinny_pack_kernel, dim3(blocks), dim3(256), 0, stream, w, ldw, wp, N, d);
  return dmx_check_launch("dmx_skinny_pack_kernel");
}
