// Halo-tiled 3x3 convolution with the preceding GroupNorm(+SiLU) applied in LDS (SURVEY.md 8a rows K1 + K3 as ONE launch; north_star:
// "NHWC conv2d with LDS-staged input tiles ... GroupNorm/SiLU fused per-channel in LDS").  Every ResnetBlock2D behind
// app.ipynb:814 / train_diffute_v1.py:913 (UNet) and app.ipynb:793,819 (AutoencoderKL) is [GroupNorm -> SiLU -> conv3x3] x 2.
//
//   out[b, y, x, n] = bias[n] + rowbias[b][n] + res[b, y, x, n]
//                   + sum_{dy, dx, c} W[n][(3 dy + dx) Cin + c] * G(in)[b, y + dy - 1, x + dx - 1, c]      (zero outside the image)
//                   + sum_{c'} W[n][9 Cin + c'] * sc[b, y, x, c']                                           (fused 1x1 shortcut, raw input)
//   G(in) = SiLU(GroupNorm(in)) with the statistics taken from the DmxStat records of in's producer(s) (common.h), or in itself.
//
// Structure (gemm.hip re-fetches every input pixel once per tap through the LDS-DMA path, which is the per-CU resource that bounds
// it - EXPERIMENTS.md; here the input is fetched once per 64-channel chunk):
//   * block = 256 output pixels (TH x TW = 8 x 32 or 16 x 16 of one image) x BN = 160 / 128 output channels, 8 waves = 4 (pixels) x 2
//     (channels) of 64 x 80 / 64 x 64 wave tiles of v_mfma_f32_16x16x32, weights as the A operand (a lane ends up with 4 consecutive
//     channels of one pixel);
//   * per 64-channel CHUNK the (TH + 2) x (TW + 2) x 64 input patch is DMA'd ONCE into one of two LDS patch buffers (128-byte rows,
//     16-byte pieces XOR-swizzled with row & 7 on the SOURCE address: conflict-free ds_read_b128 at every tap shift -
//     scripts/probes/halo_bank_check.py); the nine taps are nine shifted fragment reads of that patch, and only the [BN][64] weight
//     tiles (20 KB) stream through a three-stage ring: ~25 KB of LDS-DMA per tap instead of 52 KB;
//   * the patch of chunk c + 1 lands during the first taps of chunk c and is NORMALISED IN PLACE during the others - y = x a + s,
//     SiLU, one rounding, zero for the padding pixels - one 16-byte piece per thread and tap, riding in the MFMA shadows;
//     a = rstd gamma, s = beta - mean a per (sample, channel) come from a 64-entry table built per chunk from the group statistics
//     (integer sums of the producers' records -> double mean / variance, once per block) and gamma / beta (one small DMA per chunk);
//   * one barrier per tap; the schedule of a chunk is static (fully unrolled, per-step vmcnt immediates, the queue is never drained);
//   * K split over `splits` blocks per tile (whole tap rows): every block keeps 256 / splits pixel rows of the tile, publishes the
//     other rows of its fp32 accumulators as write-through (sc1) slabs + flag, and finishes its own rows from LDS + the peers' slabs in
//     K order (reduce-scatter: no reduce pass, no idle helper, deterministic); grid <= one block per CU so the peers are co-resident;
//   * epilogue: + bias + time-embedding row + residual, one rounding, 16-byte stores, and the DmxStat records of the OUTPUT for the
//     next GroupNorm.
#include "common.h"
#include "kernels.h"
#include <stdio.h>
#include <type_traits>

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

namespace {

constexpr int HB_NT = 512;                 // threads
constexpr int HB_PATCH = 44032;            // patch buffer: 2752 pieces of 16 B (340 rows of 128 B, rounded up to whole 1-KB DMA instructions)

template <int NF> struct HaloLds {
  static constexpr int BN = 32 * NF;
  static constexpr int WSTAGE = BN * 128;
  static constexpr int PATCH0 = 3 * WSTAGE, PATCH1 = PATCH0 + HB_PATCH;
  static constexpr int GB = PATCH1 + HB_PATCH;        // gamma | beta of a chunk, two 1-KB slots (lanes 32..63 of the DMA land in the second half)
  static constexpr int COEF = GB + 2048;              // (a, s) of the 64 channels of a chunk, 512 B
  static constexpr int GST = COEF + 512;              // (mean, rstd) of the 32 groups, 256 B (+ 256 spare)
  static constexpr int JUNK = GST + 512;              // destination of the dummy DMA instructions that keep the vmcnt arithmetic static, 1 KB
  static constexpr int TOTAL = JUNK + 1024;
  // epilogue: fp32 staging of 128 rows + the statistics fold
  static constexpr int LDT = BN + 4;
  static constexpr int EPI_FOLD = 128 * LDT * 4;
  static constexpr int OCP = BN / 8, RL = HB_NT / OCP;
  static constexpr int EPI_TOTAL = EPI_FOLD + RL * BN * 8;
  static_assert(EPI_TOTAL <= TOTAL && TOTAL <= 163840, "LDS budget");
};

template <int NF>
__global__ __launch_bounds__(HB_NT, 2) void dmx_conv_halo_kernel(const HaloConvArgs p) {
  typedef HaloLds<NF> L;
  constexpr int BN = L::BN, WSTAGE = L::WSTAGE;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave & 3, wn = wave >> 2;
  const int lr = lane & 15, lq = lane >> 4;
  // measurement aids, probe builds only (-DDMX_PROBES; they cost registers in the K loop): HaloConvArgs.dbg ablation switches (results
  // invalid: 1 no MFMA phase, 2 no weight DMA, 4 no normalisation, 8 no patch DMA, 16 no barriers) and .timing phase timestamps
#ifdef DMX_PROBES
  const int DBG = p.dbg; long long* const TIMING = p.timing;
#else
  constexpr int DBG = 0; constexpr long long* TIMING = nullptr;
#endif
  long long tm[6] = {0, 0, 0, 0, 0, 0};                // 100 MHz ticks at the phase boundaries
  if (TIMING) tm[0] = __builtin_amdgcn_s_memrealtime();

  // ---- work item: (n-tile, K slice) combos are dealt XCD-contiguously, pixel tiles inside a combo: the blocks resident on one XCD
  // stream the SAME weight slice (the large operand of the deep levels) through that XCD's L2
  const int TW = p.TW, TH = p.TH, PW = TW + 2;
  const int tiles_x = p.W / TW, tiles_img = tiles_x * (p.H / TH);
  const int tiles_m = p.B * tiles_img, S = p.splits;
  const int nb = gridDim.x;
  const int Lb = ((nb & 7) == 0) ? (blockIdx.x & 7) * (nb >> 3) + (blockIdx.x >> 3) : blockIdx.x;
  const int combo = Lb / tiles_m, tile_m = Lb - combo * tiles_m;
  const int tile_n = combo / S, r = combo - tile_n * S;
  const int b = tile_m / tiles_img, ti = tile_m - b * tiles_img;
  const int ty0 = (ti / tiles_x) * TH, tx0 = (ti % tiles_x) * TW;
  const int n0 = tile_n * BN;
  const int twsh = (TW == 32) ? 5 : 4;                 // TW is 16 or 32
  const int nprow = (TH + 2) * PW, npiece = nprow * 8;

  // ---- K steps: main chunk c = steps [9c, 9c + 9) (tap = step % 9), then one step per 64 shortcut channels.  Slice r = steps [sb, se),
  // boundaries inside the main part on whole tap rows.
  const int nc = p.Cin >> 6, nsc = p.Csc >> 6, T = 9 * nc + nsc;
  auto bnd = [&](int q) { int v = (int)((long long)T * q / S); if (v < 9 * nc) v = (v + 1) / 3 * 3; return v; };
  const int sb = bnd(r), se = bnd(r + 1);
  const int ch_first = sb < 9 * nc ? sb / 9 : nc + (sb - 9 * nc);         // chunk ids: 0 .. nc-1 main, nc + j shortcut
  const int ch_last = (se - 1) < 9 * nc ? (se - 1) / 9 : nc + (se - 1 - 9 * nc);

  // ---- per-thread DMA geometry.  Patch piece i of thread t: q = t + 512 i -> patch row q >> 3 (pixel (py, px) of the halo tile),
  // 16-byte slot q & 7 holding source chunk slot ^ (row & 7).  ppix[i] = pixel index in the image tensor, -1 = padding / beyond the patch.
  int ppix[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int q = t + HB_NT * i, prow = q >> 3;
    const int py = prow / PW, px = prow - py * PW;
    const int iy = ty0 - 1 + py, ix = tx0 - 1 + px;
    ppix[i] = (q < npiece && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) ? (b * p.H + iy) * p.W + ix : -1;
  }
  const int pslot = ((t & 7) ^ ((t >> 3) & 7)) * 8;    // source channel octet of every piece of this thread ((q >> 3) & 7 = (t >> 3) & 7)
  // weight pieces: instruction j = wave + 8 i covers tile rows 8 j .. 8 j + 7
  constexpr int WI = (BN * 8 + HB_NT - 1) / HB_NT;     // DMA instructions per thread and weight tile (3 for 160 columns, the last round half dummy)
  // (byte offset of this thread's piece of instruction 0 inside the [BN][ldw] weight slab; instruction i is 64 rows further)
  const char* const wslab = (const char*)(p.w + (size_t)n0 * p.ldw);
  const unsigned woff0 = (unsigned)(((t >> 3) * p.ldw + (((t & 7) ^ ((t >> 3) & 7)) * 8)) * 2);
  const unsigned wstep = (unsigned)(64 * p.ldw * 2);
  const char* zp = (const char*)p.zeros;

  auto dma = [&](const char* src, int lds_off) {
    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(smem + lds_off), 16, 0, 0);
  };
  // weight tile at K offset `koff` (elements; < 0: a dummy that keeps the vmcnt arithmetic static) -> ring stage `st`
  auto issue_w = [&](long koff, int st) {
#pragma unroll
    for (int i = 0; i < WI; ++i) {
      const bool ok = koff >= 0 && (wave + 8 * i) * 64 < BN * 8;       // wave-uniform
      unsigned off = woff0 + wstep * i;
      asm volatile("" : "+v"(off));
      dma(ok ? wslab + koff * 2 + off : zp, ok ? st * WSTAGE + (wave * 64 + HB_NT * i) * 16 : L::JUNK);
    }
  };
  // K offset of the weight tile of pipeline step g (-1 outside this block's slice)
  auto koff_of = [&](int g) -> long {
    if (g < sb || g >= se) return -1;
    if (g < 9 * nc) { const int c = g / 9, tap = g - 9 * c; return (long)tap * p.Cin + c * 64; }
    return (long)9 * p.Cin + (g - 9 * nc) * 64;
  };
  // what a chunk's P instructions fetch: mode 1 = main chunk (gamma | beta + six pieces of the halo patch), 2 = shortcut chunk (four
  // pieces: the 256 centre pixels, row = tile pixel index), 0 = nothing (dummies)
  struct PDesc { const bf16* base; int ld; int mode; int ch; };
  auto pdesc = [&](int ch) -> PDesc {
    if (ch < 0) return PDesc{p.x0, 0, 0, 0};
    if (ch < nc) { const int c0 = ch * 64; return c0 < p.cx0 ? PDesc{p.x0 + c0, p.ldx0, 1, ch} : PDesc{p.x1 + (c0 - p.cx0), p.ldx1, 1, ch}; }
    const int c0 = (ch - nc) * 64; return c0 < p.cs0 ? PDesc{p.s0 + c0, p.lds0, 2, ch} : PDesc{p.s1 + (c0 - p.cs0), p.lds1, 2, ch};
  };
  auto spix = [&](int i) {                             // shortcut pieces: image pixel of tile pixel (t + 512 i) >> 3
    int tt = t; asm volatile("" : "+v"(tt));
    const int pp = (tt + HB_NT * i) >> 3; return (b * p.H + ty0 + (pp >> twsh)) * p.W + tx0 + (pp & (TW - 1));
  };
  auto issue_coef = [&](const PDesc& d) {
    const bool on = d.mode == 1 && p.gn;
    const float* g = ((lane & 16) ? p.beta : p.gamma) + d.ch * 64 + (lane & 15) * 4;
    dma(on ? (const char*)g : zp, on ? L::GB + (d.ch & 1) * 1024 : L::JUNK);
  };
  auto issue_piece = [&](const PDesc& d, int pb, const int i) {
    int pix = d.mode == 1 ? ppix[i] : ((d.mode == 2 && i < 4) ? spix(i) : -1);
    asm volatile("" : "+v"(pix));                      // keep the address arithmetic here: hoisted out of the chunk loop it is 40 registers of pointers
    const char* src = pix >= 0 ? (const char*)(d.base + (size_t)pix * d.ld + pslot) : zp;
    const bool live = d.mode == 1 ? (wave * 64 + HB_NT * i) < npiece : (d.mode == 2 && i < 4);      // wave-uniform
    dma(src, live ? pb + (wave * 64 + HB_NT * i) * 16 : L::JUNK);
  };

  // ---- fragment addresses: m-fragment i of this wave = tile pixels wm*64 + 16 i + lr, one tile row (TW = 16) or half a row (TW = 32);
  // n-fragment j = weight-tile rows wn*16NF + 16 j + lr.  Rows 16 apart share row & 7, so the swizzle term is the same for every fragment:
  // one base register each, the fragment index is an immediate / uniform offset.
  const int pp0 = wm * 64 + lr;
  const int xrow0 = (pp0 >> twsh) * PW + (pp0 & (TW - 1));                 // main taps: patch row of fragment 0 at tap (0, 0)
  const int xd1 = TW == 32 ? 16 : PW, xd2 = TW == 32 ? PW : 2 * PW, xd3 = TW == 32 ? PW + 16 : 3 * PW;   // ... of fragments 1..3 relative to it (uniform)
  const int xsc0 = pp0 * 128 + ((lq ^ (pp0 & 7)) << 4);                    // shortcut step: row = tile pixel index; fragment i is 2048 bytes further
  const int wn0 = wn * (16 * NF) + lr;
  const int wad0 = wn0 * 128 + ((lq ^ (wn0 & 7)) << 4);                    // fragment j is 2048 bytes further
  f32x4 acc[NF][4];
#pragma unroll
  for (int j = 0; j < NF; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};

  // one tap: 2 k-steps of 32 channels.  The four pixel fragments of a k-step stay in registers, the weight fragments pass through two
  // register sets one at a time (j-major MFMA order, 40 fragment registers); the reads for the next group of four MFMAs are pinned in
  // front of the current group (sched_group_barrier), so LDS latency hides under the matrix pipe
  auto mma = [&](const char* xb, const int* xa, const char* wbase) {
    const char* wb = wbase + wad0;
    const char* wb1 = wbase + (wad0 ^ 64);             // k-step 1: chunk index ^ 4 (an XOR on the swizzled offset, not + 64)
    bf16x8 xf[2][4], wf[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) xf[0][i] = *(const bf16x8*)(xb + xa[i]);
    wf[0] = *(const bf16x8*)(wb);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
      for (int j = 0; j < NF; ++j) {
        const int cur = (kk * NF + j) & 1;
        int nrd = 0;
        if (j + 1 < NF) { wf[cur ^ 1] = *(const bf16x8*)((kk ? wb1 : wb) + (j + 1) * 2048); ++nrd; }
        else if (kk == 0) { wf[cur ^ 1] = *(const bf16x8*)(wb1); ++nrd; }
        if (kk == 0 && j < 4) { xf[1][j] = *(const bf16x8*)(xb + (xa[j] ^ 64)); ++nrd; }
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[j][i] = DMX_MFMA_16x16x32(wf[cur], xf[kk][i], acc[j][i]);
        if (nrd == 2) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        else if (nrd == 1) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  auto mma_main = [&](int pb, int tapoff, int st) {
    int pr = xrow0;
    asm volatile("" : "+v"(pr));                       // (the tap addresses are loop-invariant: hoisted they would live across the whole K loop)
    pr += tapoff;
    const int rr4[4] = {pr, pr + xd1, pr + xd2, pr + xd3};
    int xa[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) xa[i] = rr4[i] * 128 + ((lq ^ (rr4[i] & 7)) << 4);
    mma(smem + pb, xa, smem + st * WSTAGE);
  };
  auto mma_sc = [&](int pb, int st) {
    int x0 = xsc0;
    asm volatile("" : "+v"(x0));
    const int xa[4] = {x0, x0 + 2048, x0 + 4096, x0 + 6144};
    mma(smem + pb, xa, smem + st * WSTAGE);
  };

  // ---- GroupNorm pieces
  // (a, s) of the 64 channels of main chunk `ch` from the group statistics and the chunk's gamma | beta slot; every wave writes the
  // same 64 entries (no divergence, no extra barrier)
  const int cpg = p.gn ? p.Cin / p.groups : 1;
  auto coef_table = [&](int ch) {
    if (!p.gn) return;
    const int c = ch * 64 + lane;
    const int g = c / cpg;
    const float* gb = (const float*)(smem + L::GB + (ch & 1) * 1024);
    const float* gs = (const float*)(smem + L::GST);
    const float a = gs[2 * g + 1] * gb[lane];
    float* cf = (float*)(smem + L::COEF);
    cf[2 * lane] = a; cf[2 * lane + 1] = gb[64 + lane] - gs[2 * g] * a;
  };
  // normalise piece i of the patch in buffer `pb` in place
  auto norm_piece = [&](int pb, const int i) {
    if (!p.gn) return;
    const int q = t + HB_NT * i;
    if (q >= npiece) return;
    u32x4* ptr = (u32x4*)(smem + pb + q * 16);
    const float* cf = (const float*)(smem + L::COEF) + pslot * 2;
    const f32x4 c0 = *(const f32x4*)cf, c1 = *(const f32x4*)(cf + 4), c2 = *(const f32x4*)(cf + 8), c3 = *(const f32x4*)(cf + 12);
    float f[8]; unpack_bf8(*ptr, f);
    const float av[8] = {c0[0], c0[2], c1[0], c1[2], c2[0], c2[2], c3[0], c3[2]};
    const float sv[8] = {c0[1], c0[3], c1[1], c1[3], c2[1], c2[3], c3[1], c3[3]};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float y = __builtin_fmaf(f[e], av[e], sv[e]);
      if (p.silu) y *= __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896f * y));
      f[e] = y;
    }
    u32x4 o = pack_bf8(f);
    if (ppix[i] < 0) o = u32x4{0u, 0u, 0u, 0u};        // the conv pads the NORMALISED tensor with zeros
    *ptr = o;
  };

  // ---- prologue: group statistics of this block's sample (main chunks in range and gn), first patch, first two weight tiles
  if (p.gn && ch_first < nc) {
    // 16 threads per group sum the group's channels' records (integers: exact, any order), then mean / variance in double
    const int g = t >> 4, sub = t & 15;
    long long s0 = 0, qh = 0, ql = 0;
    if (g < p.groups) {
      for (int c = g * cpg + sub; c < (g + 1) * cpg; c += 16) {
        const long long* rec = c < p.cx0 ? p.st0 + ((size_t)b * p.cx0 + c) * DMX_STAT_WORDS
                                         : p.st1 + ((size_t)b * (p.Cin - p.cx0) + (c - p.cx0)) * DMX_STAT_WORDS;
        s0 += rec[0]; qh += rec[1]; ql += rec[2];
      }
    }
#pragma unroll
    for (int d = 1; d < 16; d <<= 1) {
      s0 += __shfl_xor(s0, d); qh += __shfl_xor(qh, d); ql += __shfl_xor(ql, d);
    }
    if (sub == 0 && g < p.groups) {
      const double n = (double)cpg * (double)p.H * (double)p.W;
      const double mean = dmx_stat_sum(s0) / n;
      double var = dmx_stat_sumsq(qh, ql) / n - mean * mean;
      var = var < 0.0 ? 0.0 : var;
      float* gs = (float*)(smem + L::GST);
      gs[2 * g] = (float)mean; gs[2 * g + 1] = (float)(1.0 / __builtin_sqrt(var + (double)p.eps));
    }
  }
  int cur = ch_first, seq = 0;
  int gstep = cur < nc ? 9 * cur : 9 * nc + (cur - nc);   // pipeline step of the chunk's first step (inactive steps of a partial chunk included)
  {
    const PDesc d = pdesc(cur);
    issue_coef(d);
#pragma unroll
    for (int i = 0; i < 6; ++i) issue_piece(d, L::PATCH0, i);
  }
  issue_w(koff_of(gstep), gstep % 3);
  issue_w(koff_of(gstep + 1), (gstep + 1) % 3);
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * WI) : "memory");     // the patch (and gamma | beta) landed; the weight tiles stay in flight
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                        // (also publishes the group statistics)
  if (cur < nc && p.gn) {
    coef_table(cur);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int i = 0; i < 6; ++i) norm_piece(L::PATCH0, i);
  }

  if (TIMING) tm[1] = __builtin_amdgcn_s_memrealtime();
  // ---- K loop: the main chunks of the slice, then its shortcut steps (two loops: one loop with both bodies makes the compiler
  // shuffle the 80 accumulators at the merge)
  while (cur < nc) {
    const int next = cur < ch_last ? cur + 1 : -1;
    const int pb = (seq & 1) ? L::PATCH1 : L::PATCH0, pbn = (seq & 1) ? L::PATCH0 : L::PATCH1;
    const PDesc dn = pdesc(next);
    // main chunk: nine statically scheduled steps.  Step s: [wait: weights of s landed] [barrier] [DMA: weights of s + 2; steps 0 / 1
    // also the next chunk's patch] [MFMA tap s] [steps >= 3: normalise one piece of the next patch]
    auto step = [&](auto S_) {
      constexpr int s = decltype(S_)::value;
      constexpr int nwait = s == 1 ? (4 + WI) : s == 2 ? (3 + WI) : WI;     // DMA instructions issued after the group this step needs
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(nwait) : "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (!(DBG & 16)) __builtin_amdgcn_s_barrier();
      if (!(DBG & 8)) {
        if constexpr (s == 0) { issue_coef(dn); issue_piece(dn, pbn, 0); issue_piece(dn, pbn, 1); issue_piece(dn, pbn, 2); }
        if constexpr (s == 1) { issue_piece(dn, pbn, 3); issue_piece(dn, pbn, 4); issue_piece(dn, pbn, 5); }
      }
      if (!(DBG & 2)) issue_w(koff_of(gstep + s + 2), (s + 2) % 3);
      if constexpr (s == 2) { if (dn.mode == 1) coef_table(next); }
      const int g = gstep + s;
      if (g >= sb && g < se && !(DBG & 1)) mma_main(pb, (s / 3) * PW + (s % 3), s % 3);
      if constexpr (s >= 3) { if (dn.mode == 1 && !(DBG & 4)) norm_piece(pbn, s - 3); }
    };
    step(std::integral_constant<int, 0>{}); step(std::integral_constant<int, 1>{}); step(std::integral_constant<int, 2>{});
    step(std::integral_constant<int, 3>{}); step(std::integral_constant<int, 4>{}); step(std::integral_constant<int, 5>{});
    step(std::integral_constant<int, 6>{}); step(std::integral_constant<int, 7>{}); step(std::integral_constant<int, 8>{});
    gstep += 9; ++seq;
    if (next < 0) { cur = -1; break; }
    cur = next;
  }
  while (cur >= nc) {
    // shortcut step: the next shortcut patch is requested one step ahead (P before W, so vmcnt(WI) covers it)
    const int next = cur < ch_last ? cur + 1 : -1;
    const int pb = (seq & 1) ? L::PATCH1 : L::PATCH0, pbn = (seq & 1) ? L::PATCH0 : L::PATCH1;
    const PDesc dn = pdesc(next);
    const int st = gstep % 3;
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WI) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    issue_piece(dn, pbn, 0); issue_piece(dn, pbn, 1); issue_piece(dn, pbn, 2); issue_piece(dn, pbn, 3);
    issue_w(koff_of(gstep + 2), (gstep + 2) % 3);
    mma_sc(pb, st);
    gstep += 1; ++seq;
    cur = next;                                        // (-1 ends the loop)
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the dummy tail loads must land before LDS is reused
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  if (TIMING) tm[2] = __builtin_amdgcn_s_memrealtime();
  // ---------------------------------------------------------------- epilogue
  // acc[j][i][e] = out[pixel wm*64 + i*16 + lr][channel n0 + wn*16NF + j*16 + 4 lq + e].  Rows (tile pixels) [own0, own0 + RO) are
  // this block's; the other rows of its accumulators go to its slab for their owners.
  constexpr int LDT = L::LDT, OCP = L::OCP, RL = L::RL;
  const int RO = 256 / S, own0 = r * RO;
  const size_t slab_elems = (size_t)256 * BN;
  const size_t tile_slot = ((size_t)tile_n * tiles_m + tile_m) * S;
  if (S > 1) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.slabs + (tile_slot + r) * slab_elems), 0, (int)(slab_elems * 4), 0x00020000);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row0 = wm * 64 + i * 16;
      if (row0 / RO == r) continue;                    // wave-uniform
#pragma unroll
      for (int j = 0; j < NF; ++j)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[j][i]), rs, ((row0 + lr) * BN + wn * 16 * NF + j * 16 + 4 * lq) * 4, 0, 16);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // write-through (sc1) stores drained -> block barrier -> flag (relaxed, agent scope)
    __syncthreads();
    if (t == 0) __hip_atomic_store(p.flags + tile_slot + r, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (TIMING) tm[3] = __builtin_amdgcn_s_memrealtime();
  float* tile = (float*)smem;
  const int o = t % OCP, rl = t / OCP;
  const bool act = t < RL * OCP;
  float bs[8];
  {
    const int n = n0 + o * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) bs[e] = 0.f;
    if (p.bias) { const f32x4 v0 = *(const f32x4*)(p.bias + n), v1 = *(const f32x4*)(p.bias + n + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { bs[e] = v0[e]; bs[4 + e] = v1[e]; } }
    if (p.rowbias) { const float* rb = p.rowbias + (size_t)b * p.ldrb + n; const f32x4 v0 = *(const f32x4*)rb, v1 = *(const f32x4*)(rb + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { bs[e] += v0[e]; bs[4 + e] += v1[e]; } }
  }
  float cs_s[8], cs_q[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { cs_s[e] = 0.f; cs_q[e] = 0.f; }
  const int RPP = RO < 128 ? RO : 128, passes = RO / RPP;
  for (int ep = 0; ep < passes; ++ep) {
    const int prow0 = own0 + ep * RPP;                 // first tile pixel of this pass
    if (ep) __builtin_amdgcn_s_barrier();              // the previous pass is done with the staging tile
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row0 = wm * 64 + i * 16;
      if (row0 < prow0 || row0 >= prow0 + RPP) continue;
#pragma unroll
      for (int j = 0; j < NF; ++j) *(f32x4*)(tile + (row0 - prow0 + lr) * LDT + wn * 16 * NF + j * 16 + 4 * lq) = acc[j][i];
    }
    if (S > 1 && ep == 0 && t == 0) {
      // peers' slabs: bounded spin (a lost peer must never hang the GPU: after ~40 ms the block goes on with what is there)
      const long long t0 = __builtin_amdgcn_s_memrealtime();
      for (int s = 0; s < S; ++s) {
        if (s == r) continue;
        while (__hip_atomic_load(p.flags + tile_slot + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
          __builtin_amdgcn_s_sleep(2);
          if (__builtin_amdgcn_s_memrealtime() - t0 > 4000000) break;
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
    if (TIMING && ep == 0) tm[4] = __builtin_amdgcn_s_memrealtime();
    // items: (row lane rl, octet o), the octet fixed per thread so the column partial sums stay in registers
    auto items = [&](auto S_, auto U_) {
      constexpr int SS = decltype(S_)::value, U = decltype(U_)::value;
      for (int k0 = 0; k0 * RL < RPP; k0 += U) {
        u32x4 rr[U]; f32x4 pv[U][SS][2];               // (indexed by slice: entry r stays unused - static register indices)
#pragma unroll
        for (int u = 0; u < U; ++u) {
          int row = rl + RL * (k0 + u); if (row >= RPP) row = RPP - 1;
          const int pp = prow0 + row;
          const size_t pix = (size_t)(b * p.H + ty0 + (pp >> twsh)) * p.W + tx0 + (pp & (TW - 1));
          if (p.res) rr[u] = *(const u32x4*)(p.res + pix * p.ldres + n0 + o * 8);
          if constexpr (SS > 1) {
#pragma unroll
            for (int s = 0; s < SS; ++s) {
              if (s == r) continue;
              // sc1 loads of the write-through slabs: served by L2 / the fabric, no agent-scope acquire needed (gemm.hip stream-K)
              const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.slabs + (tile_slot + s) * slab_elems), 0, (int)(slab_elems * 4), 0x00020000);
              const int off = (pp * BN + o * 8) * 4;
              pv[u][s][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 16));
              pv[u][s][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off + 16, 0, 16));
            }
          }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int row = rl + RL * (k0 + u);
          if (!act || row >= RPP) continue;
          const int pp = prow0 + row;
          const size_t pix = (size_t)(b * p.H + ty0 + (pp >> twsh)) * p.W + tx0 + (pp & (TW - 1));
          const f32x4 m0 = *(const f32x4*)(tile + row * LDT + o * 8), m1 = *(const f32x4*)(tile + row * LDT + o * 8 + 4);
          float v[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = 0.f;
          // K order: slice 0, 1, ... (this block's own part sits at position r)
#pragma unroll
          for (int s = 0; s < SS; ++s) {
            if (s == r) {
#pragma unroll
              for (int e = 0; e < 4; ++e) { v[e] += m0[e]; v[4 + e] += m1[e]; }
            } else if constexpr (SS > 1) {
#pragma unroll
              for (int e = 0; e < 4; ++e) { v[e] += pv[u][s][0][e]; v[4 + e] += pv[u][s][1][e]; }
            }
          }
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += bs[e];
          if (p.res) {
            float rf[8]; unpack_bf8(rr[u], rf);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += rf[e];
          }
          const u32x4 pk = pack_bf8(v);
          *(u32x4*)(p.out + pix * p.ldo + n0 + o * 8) = pk;
          if (p.colstats) {
            float f[8]; unpack_bf8(pk, f);
#pragma unroll
            for (int e = 0; e < 8; ++e) { cs_s[e] += f[e]; cs_q[e] += f[e] * f[e]; }
          }
        }
      }
    };
    if (S == 1) items(std::integral_constant<int, 1>{}, std::integral_constant<int, 3>{});
    else if (S == 2) items(std::integral_constant<int, 2>{}, std::integral_constant<int, 3>{});
    else if (S == 4) items(std::integral_constant<int, 4>{}, std::integral_constant<int, 3>{});
    else items(std::integral_constant<int, 8>{}, std::integral_constant<int, 1>{});
  }
  if (p.colstats) {
    // per-channel (sum, sum of squares) of the ROUNDED outputs of this block's rows: float inside the block in a fixed order, 64-bit
    // fixed point across blocks (DmxStat)
    float* red = (float*)(smem + L::EPI_FOLD);
    if (act) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { red[(rl * BN + o * 8 + e) * 2] = cs_s[e]; red[(rl * BN + o * 8 + e) * 2 + 1] = cs_q[e]; }
    }
    __syncthreads();
    for (int c = t; c < BN; c += HB_NT) {
      float sa = 0.f, sq = 0.f;
      for (int k = 0; k < RL; ++k) { sa += red[(k * BN + c) * 2]; sq += red[(k * BN + c) * 2 + 1]; }
      dmx_stat_add(p.colstats + ((size_t)b * p.N + n0 + c) * DMX_STAT_WORDS, sa, sq);
    }
  }
  if (TIMING && t == 0) {
    long long* o_ = TIMING + (size_t)blockIdx.x * 8;
    tm[5] = __builtin_amdgcn_s_memrealtime();
#pragma unroll
    for (int i = 0; i < 6; ++i) o_[i] = tm[i];
    o_[6] = r; o_[7] = se - sb;
  }
}

// statistics of a tensor whose producer emitted none: block = 64 rows x all channels of one sample
__global__ __launch_bounds__(256) void dmx_colstats_kernel(const bf16* x, int ldx, int HW, int C, long long* st, int rows_per_block) {
  const int b = blockIdx.y, row0 = blockIdx.x * rows_per_block;
  const int oc = C >> 3;
  for (int o = threadIdx.x; o < oc; o += 256) {
    float s[8], q[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { s[e] = 0.f; q[e] = 0.f; }
    const int row1 = min(row0 + rows_per_block, HW);
    for (int row = row0; row < row1; row += 4) {
      u32x4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = *(const u32x4*)(x + ((size_t)b * HW + min(row + u, row1 - 1)) * ldx + o * 8);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (row + u >= row1) continue;
        float f[8]; unpack_bf8(v[u], f);
#pragma unroll
        for (int e = 0; e < 8; ++e) { s[e] += f[e]; q[e] += f[e] * f[e]; }
      }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) dmx_stat_add(st + ((size_t)b * C + o * 8 + e) * DMX_STAT_WORDS, s[e], q[e]);
  }
}

int halo_nf(const HaloConvArgs& a) { return a.N % 160 == 0 ? 5 : (a.N % 128 == 0 ? 4 : 0); }

void halo_geometry(HaloConvArgs& a) {
  if (a.W % 32 == 0 && a.H % 8 == 0) { a.TW = 32; a.TH = 8; }
  else if (a.W % 16 == 0 && a.H % 16 == 0) { a.TW = 16; a.TH = 16; }
  else { a.TW = 0; a.TH = 0; }
  const int nf = halo_nf(a);
  if (!a.TW || !nf) { a.splits = 0; return; }
  static int n_cu = 0;
  if (!n_cu) { int dev = 0; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev); if (n_cu <= 0) n_cu = 256; }
  const long tiles = (long)a.B * (a.H / a.TH) * (a.W / a.TW) * (a.N / (32 * nf));
  const int T = 9 * (a.Cin / 64) + a.Csc / 64;
  int s = 1;
  if (a.force_split) s = a.force_split;
  else while (s < 8 && tiles * (s * 2) <= n_cu && T / (s * 2) >= 9) s *= 2;
  a.splits = s;
}

}  // namespace

bool dmx_conv_halo_supported(const HaloConvArgs& a0) {
  HaloConvArgs a = a0;
  if (a.Cin <= 0 || a.Cin % 64 || a.cx0 % 64 || a.cx0 > a.Cin || a.Csc % 64 || (a.Csc && a.cs0 % 64) || a.N % 8 || a.ldo % 8) return false;
  if (a.ldx0 % 8 || (a.x1 && a.ldx1 % 8) || a.ldw % 8 || (a.res && a.ldres % 8)) return false;
  if (a.gn && (a.groups != 32 || a.Cin % a.groups)) return false;
  halo_geometry(a);
  if (!a.splits) return false;
  const long blocks = (long)a.B * (a.H / a.TH) * (a.W / a.TW) * (a.N / (32 * halo_nf(a))) * a.splits;
  if (a.splits > 1) {
    static int n_cu = 0;
    if (!n_cu) { int dev = 0; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev); if (n_cu <= 0) n_cu = 256; }
    if (blocks > n_cu) return false;                   // the blocks of a tile wait for each other: all of them must be resident
    if ((a.splits & (a.splits - 1)) || a.splits > 8) return false;
    const int T = 9 * (a.Cin / 64) + a.Csc / 64;
    if (T / a.splits < 3) return false;
  }
  return true;
}

int dmx_conv_halo_flag_count(const HaloConvArgs& a0) {
  HaloConvArgs a = a0; halo_geometry(a);
  if (a.splits <= 1) return 0;
  return (int)((long)a.B * (a.H / a.TH) * (a.W / a.TW) * (a.N / (32 * halo_nf(a))) * a.splits);
}

static size_t halo_flag_bytes(int n) { return align_up((size_t)n * sizeof(int), 256); }

size_t dmx_conv_halo_workspace_bytes(const HaloConvArgs& a0) {
  const int n = dmx_conv_halo_flag_count(a0);
  if (!n) return 0;
  return halo_flag_bytes(n) + (size_t)n * 256 * (32 * halo_nf(a0)) * sizeof(float);
}

template <int NF> static int halo_launch_(const HaloConvArgs& a, int blocks, hipStream_t stream) {
  DMX_LDS_OPT_IN((dmx_conv_halo_kernel<NF>), HaloLds<NF>::TOTAL);
  hipLaunchKernelGGL((dmx_conv_halo_kernel<NF>), dim3(blocks), dim3(HB_NT), HaloLds<NF>::TOTAL, stream, a);
  return dmx_check_launch("dmx_conv_halo_kernel");
}

int dmx_conv_halo_launch(HaloConvArgs a, void* workspace, size_t workspace_bytes, hipStream_t stream) {
  DMX_REQUIRE(a.x0 && a.w && a.out, "conv_halo: null argument");
  DMX_REQUIRE(dmx_conv_halo_supported(a), "conv_halo: unsupported problem (H=%d W=%d Cin=%d cx0=%d Csc=%d N=%d split=%d)", a.H, a.W, a.Cin, a.cx0, a.Csc, a.N, a.force_split);
  if (a.gn) DMX_REQUIRE(a.st0 && a.gamma && a.beta && (a.cx0 == a.Cin || a.st1), "conv_halo: the fused GroupNorm needs statistics records, gamma and beta");
  if (a.Csc) DMX_REQUIRE(a.s0 && (a.cs0 == a.Csc || a.s1), "conv_halo: the shortcut segment needs its source tensor(s)");
  DMX_REQUIRE(a.ldw >= 9 * a.Cin + a.Csc, "conv_halo: weight rows shorter than K");
  if (!a.x1) { a.x1 = a.x0; a.ldx1 = a.ldx0; }
  if (!a.s1) { a.s1 = a.s0; a.lds1 = a.lds0; }
  int rc = dmx_zero_page(&a.zeros);
  if (rc) return rc;
  halo_geometry(a);
  const int nf = halo_nf(a);
  const int tiles = a.B * (a.H / a.TH) * (a.W / a.TW) * (a.N / (32 * nf));
  const int blocks = tiles * a.splits;
  if (a.splits > 1) {
    const size_t fb = halo_flag_bytes(blocks), need = fb + (size_t)blocks * 256 * (32 * nf) * sizeof(float);
    if (!workspace || workspace_bytes < need) { dmx_set_error("conv_halo: the K split needs %zu bytes of workspace, got %zu", need, workspace_bytes); return DMX_ERR_WORKSPACE; }
    a.slabs = (float*)((char*)workspace + fb);
    if (!a.flags) { a.flags = (int*)workspace; DMX_HIP(hipMemsetAsync(a.flags, 0, fb, stream)); }
  }
  const double flops = 2.0 * a.B * a.H * a.W * (double)a.N * (9.0 * a.Cin + a.Csc);
  const double bytes = 2.0 * ((double)a.B * a.H * a.W * (a.Cin + a.Csc + a.N) + (double)a.N * (9.0 * a.Cin + a.Csc));
  char tag[96];
  snprintf(tag, sizeof(tag), "M=%d N=%d K=%d halo gn=%d sk=%d", a.B * a.H * a.W, a.N, 9 * a.Cin + a.Csc, a.gn, a.splits);
  ProfScope ps(PROF_HALO, stream, flops, bytes, tag);
  return nf == 5 ? halo_launch_<5>(a, blocks, stream) : halo_launch_<4>(a, blocks, stream);
}

int dmx_colstats_launch(const bf16* x, int ldx, int B, int HW, int C, long long* st, hipStream_t stream) {
  DMX_REQUIRE(x && st && C % 8 == 0 && ldx % 8 == 0, "colstats: C and ld must be multiples of 8");
  int rpb = 64;
  while ((long)B * cdiv(HW, rpb) > 4096) rpb *= 2;
  char tag[96]; snprintf(tag, sizeof(tag), "rows=%d C=%d colstats", B * HW, C);
  ProfScope ps(PROF_GNORM, stream, 0.0, 2.0 * (double)B * HW * C, tag);
  hipLaunchKernelGGL(dmx_colstats_kernel, dim3(cdiv(HW, rpb), B), dim3(256), 0, stream, x, ldx, HW, C, st, rpb);
  return dmx_check_launch("dmx_colstats_kernel");
}
