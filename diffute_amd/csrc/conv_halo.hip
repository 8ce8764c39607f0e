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
//     scripts/attic/probes/halo_bank_check.py); the nine taps are nine shifted fragment reads of that patch, and only the [BN][64] weight
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

int n_cus() {                                          // of the CURRENT device (the K-split plans need a tile's blocks co-resident on it)
  static int n_cu[64] = {};
  int dev = 0; (void)hipGetDevice(&dev);
  int& n = n_cu[dev & 63];
  if (!n) { (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); if (n <= 0) n = 256; }
  return n;
}

// dmx_set_exclusive_device(0): other streams / other kernels may hold CUs while a launch of the library runs (micro-batches on several streams,
// a collective on a side stream).  The in-kernel K-split exchange needs the S blocks of a tile co-resident - symmetric peers wait for each
// other - so without the device to itself the planner takes no split: shapes that need one fall back to GroupNorm + the implicit GEMM.
// (The stream-K GEMMs are not affected: an owner only ever waits for blocks dispatched AFTER it, which start as soon as any block retires.)
int g_exclusive_device = 1;
extern "C" int dmx_set_exclusive_device(int on) { const int old = g_exclusive_device; g_exclusive_device = on ? 1 : 0; dmx_plan_switch(DMX_SW_EXCLUSIVE, g_exclusive_device); return old; }
int dmx_exclusive_device() { return g_exclusive_device; }
extern "C" int dmx_get_exclusive_device(void) { return g_exclusive_device; }

namespace {

// x / d for x < 2^20, 2 <= d < 2^12 with magic = 2^32 / d + 1; d = 1: magic 0 (host: halo_magic)
__device__ __forceinline__ int hb_div(int x, unsigned magic) { return magic ? (int)(((unsigned long long)(unsigned)x * magic) >> 32) : x; }
static unsigned halo_magic(int d) { return d <= 1 ? 0u : (unsigned)((1ull << 32) / (unsigned)d + 1); }

constexpr int HB_PATCH = 44032;            // patch buffer: 2752 pieces of 16 B (340 rows of 128 B, rounded up to whole 1-KB DMA instructions)

// WMW = waves along the pixel axis: 4 -> 4 x 2 waves of 64-pixel x 16 NF-channel tiles (BN = 32 NF = 160 / 128 columns),
//                                   8 -> 8 x 1 waves of 32-pixel x 16 NF-channel tiles (BN = 16 NF = 80 / 64 columns: twice the tiles - levels
//                                        where the wider tile would need a K split, i.e. an exchange of fp32 slabs through memory)
//   WNW = waves along the channel axis (1 or 2).  WMW x WNW = 8 waves: two-group ping-pong K loop; 16 waves (8 x 2, 32 x 16 NF wave tiles,
//   <= 128 registers per lane, four waves per SIMD): lock step, one barrier per tap - what bounds the 8-wave loop is every wave's SERIAL
//   non-MFMA work per tap (DMA issue, normalisation, addressing), and twice the waves halve it per wave at 1.55 x the LDS fragment bytes
//   WS (warp-specialised; 4 x 1 compute waves of 64-pixel x 160 / 128-column tiles + four LOADER waves): the compute waves only read
//   fragments and issue MFMAs, the loaders issue every LDS-DMA instruction and normalise the patches - one barrier per tap.  What bounds a
//   ping-pong step is the DMA phase of a group (each LDS-DMA instruction holds its wave ~100 cycles; 6-7 per wave and tap plus the
//   normalisation is longer than the partner's 640 cycles of MFMAs), and the 64 x 80 wave tiles read 144 KB of fragments per tap (88 % of
//   the LDS bandwidth under the MFMAs); here the loaders have the whole tap for the same DMA work and the fragment reads are 112 KB
template <int NF, int WMW, int WNW, bool WS_ = false> struct HaloLds {
  static constexpr bool WS = WS_;
  static constexpr int NCW = WMW * WNW;               // compute waves
  static constexpr int NT = 64 * NCW + (WS ? 256 : 0);   // threads (WS: + four loader waves)
  static constexpr int DNT = WS ? 256 : NT;           // threads that issue DMA / normalise (WS: the last 256 threads of the block)
  static constexpr bool PP = !WS && (NCW == 8);       // two-group ping-pong (8 waves)
  static constexpr int BN = WNW * 16 * NF;
  static constexpr int MFR = 16 / WMW;                // 16-pixel fragments per wave
  // TPS = taps per pipeline step: the narrow tiles have the LDS for TWO weight tiles per ring stage (2 x 10 KB = one stage of the
  // 160-column tile): per step the same DMA / MFMA work as the wide tile with half the barriers, DMA-phase overheads and
  // normalisation slots per FLOP - and twice the tiles, i.e. no K split (no fp32 slab exchange) at the 64 x 64 level
  static constexpr int TPS = (WMW == 8 && WNW == 1) ? 2 : 1;
  static constexpr int WTILE = BN * 128;              // one tap's [BN][64] weight tile
  static constexpr int WSTAGE = TPS * WTILE;
  // weight ring: 3 stages = prefetch distance 1 under the two-group ping-pong (stage (s - 1) % 3 is the partner's, s % 3 this wave's next).
  // A 4-stage ring / distance 2 with a counted vmcnt was built for the tiles that have the LDS for it and measured SLOWER (64x64x320,
  // 80 columns: 72 vs 62 us): what bounds a step is each wave's serial non-MFMA work, not the landing time of the tile (EXPERIMENTS.md)
  static constexpr int NSTG = 3;
  static constexpr int PATCH0 = NSTG * WSTAGE, PATCH1 = PATCH0 + HB_PATCH;
  static constexpr int GB = PATCH1 + HB_PATCH;        // gamma | beta of a chunk, two 1-KB slots (lanes 32..63 of the DMA land in the second half)
  static constexpr int COEF = GB + 2048;              // (a, s) of the 64 channels of a chunk, 512 B
  static constexpr int GST = COEF + 512;              // (mean, rstd) of the 32 groups, 256 B (+ 256 spare)
  static constexpr int DUMP = GST + 512;              // 1 KB the weight prefetch for the NEXT launches lands in (HaloConvArgs.pf)
  static constexpr int TOTAL = DUMP + 1024;
  // epilogue: fp32 staging of 128 rows + the statistics fold
  static constexpr int LDT = BN + 4;
  static constexpr int EPI_FOLD = 128 * LDT * 4;
  static constexpr int OCP = BN / 8, RL = NT / OCP;
  static constexpr int EPI_TOTAL = EPI_FOLD + RL * BN * 8;
  static_assert(EPI_TOTAL <= TOTAL && TOTAL <= 163840, "LDS budget");
};

// lgkmcnt of the wait in front of MFMA group g of the hand-scheduled stream of TAPS taps (see mma_stream): reads issued so far minus the
// position of the last read the group needs.  Group g = (tap t, k-step kk, weight fragment j), gl = g % (2 NF).  Issue order:
// X0(tap 0)[0 .. MFR) before the barrier, W[0], W[1], then in front of group g: X1(t)[gl] (gl < MFR), X0(t + 1)[gl - NF]
// (NF <= gl < NF + MFR, not the last tap), W[g + 2].
// PF: the last tap also prefetches the X0 fragments of the NEXT call's first tap (they stay in flight across the call boundary).
constexpr int hb_wait(int NF, int MFR, int TAPS, int g, bool PF = false) {
  const int G = TAPS * 2 * NF;
  int pos = MFR + 2, posW[48] = {}, posX0[4][4] = {}, posX1[4][4] = {};
  for (int i = 0; i < MFR; ++i) posX0[0][i] = i;
  posW[0] = MFR; posW[1] = MFR + 1;
  int issued_at_wait = 0;
  for (int q = 0; q < G; ++q) {
    const int t = q / (2 * NF), gl = q - t * 2 * NF;
    if (q == g) issued_at_wait = pos;
    if (gl < MFR) posX1[t][gl] = pos++;
    if ((t + 1 < TAPS || PF) && gl >= NF && gl < NF + MFR) posX0[t + 1][gl - NF] = pos++;
    if (q + 2 < G) posW[q + 2] = pos++;
  }
  const int t = g / (2 * NF), gl = g - t * 2 * NF;
  int need = posW[g];
  if (gl == 0 && posX0[t][MFR - 1] > need) need = posX0[t][MFR - 1];
  if (gl == NF && posX1[t][MFR - 1] > need) need = posX1[t][MFR - 1];
  return issued_at_wait - 1 - need;
}

template <int NF, int WMW, int WNW, bool WS = false>
__global__ __launch_bounds__((HaloLds<NF, WMW, WNW, WS>::NT), ((HaloLds<NF, WMW, WNW, WS>::NT + 255) / 256)) void dmx_conv_halo_kernel(const HaloConvArgs p) {
  typedef HaloLds<NF, WMW, WNW, WS> L;
  constexpr int BN = L::BN, WSTAGE = L::WSTAGE, WTILE = L::WTILE, TPS = L::TPS, MFR = L::MFR, NSTG = L::NSTG, HB_NT = L::NT, DNT = L::DNT, NCW = L::NCW;
  constexpr bool PP = L::PP;
  constexpr int PD = PP ? 1 : 2;                       // prefetch distance of the weight tiles (taps): the ping-pong keeps one stage for the partner group
  constexpr int NPP = (HB_PATCH / 16 + DNT - 1) / DNT; // patch pieces per DMA thread (6; WS: 11)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave % WMW, wn = wave / WMW;
  // DMA / normalisation roles: every thread (dt = t), or the loader waves of the warp-specialised instances (the last DNT threads)
  const int dt = t - (HB_NT - DNT), dwave = wave - (HB_NT - DNT) / 64;
  const bool loader = !WS || wave >= NCW;
  const int lr = lane & 15, lq = lane >> 4;
  // measurement aids, probe builds only (-DDMX_PROBES; they cost registers in the K loop): HaloConvArgs.dbg ablation switches (results
  // invalid: 1 no MFMA phase, 2 no weight DMA, 4 no normalisation, 8 no patch DMA, 16 no barriers) and .timing phase timestamps
#ifndef DMX_HALO_DBG
#define DMX_HALO_DBG 0                                 // (compile-time ablation builds: scripts/attic/halo_ablate_build.sh)
#endif
  constexpr int DBG = DMX_HALO_DBG;
#ifdef DMX_PROBES
  long long* const TIMING = p.timing;
#else
  constexpr long long* TIMING = nullptr;
#endif
  long long tm[8] = {0, 0, 0, 0, 0, 0, 0, 0};                // 100 MHz ticks at the phase boundaries
  long long tp[4] = {0, 0, 0, 0};                      // ... inside the prologue: requests issued, statistics done, patch landed + barrier
  if (TIMING) tm[0] = __builtin_amdgcn_s_memrealtime();

  // ---- work item: (n-tile, K slice) combos are dealt XCD-contiguously, pixel tiles inside a combo: the blocks resident on one XCD
  // stream the SAME weight slice (the large operand of the deep levels) through that XCD's L2
  const int TW = p.TW, TH = p.TH, PW = TW + 2;
  const int tiles_x = p.tiles_x, tiles_img = p.tiles_img;
  const int tiles_m = p.tiles_m, S = p.splits;
  const int sshift = S == 1 ? 0 : S == 2 ? 1 : S == 4 ? 2 : 3;
  const int nb = gridDim.x;
  const int Lb = ((nb & 7) == 0) ? (blockIdx.x & 7) * (nb >> 3) + (blockIdx.x >> 3) : blockIdx.x;
  const int ncombo = p.ncombo;
  // (xcd_tile_major: pixel tiles XCD-contiguous, their column tiles / K slices inside - the blocks of an XCD share the ACTIVATION tiles)
  // (no integer divisions in the block decode: hb_div with the host's magic numbers - a scalar division is ~25 instructions through
  // the vector unit, and a dozen of them sat in front of the first DMA request)
  // (peers_local: the K slice r is the FASTEST index of the decode in both orders, so the S blocks of a tile are neighbours in Lb and sit on one XCD -
  // their fp32 slabs then change hands inside that XCD's L2; the weights-major order deals (n-tile, pixel tile) pairs instead of (n-tile, slice) combos)
  int tile_n, tile_m, r;
  if (p.peers_local && !p.xcd_tile_major) {
    const int L2i = Lb >> sshift;
    tile_n = hb_div(L2i, p.mg_tiles_m); tile_m = L2i - tile_n * tiles_m; r = Lb & (S - 1);
  } else {
    const int q_nc = hb_div(Lb, p.mg_ncombo), q_tm = hb_div(Lb, p.mg_tiles_m);
    const int combo = p.xcd_tile_major ? Lb - q_nc * ncombo : q_tm;
    tile_m = p.xcd_tile_major ? q_nc : Lb - q_tm * tiles_m;
    tile_n = combo >> sshift; r = combo & (S - 1);
  }
  // this block's XCC id + 1, published at once beside the flags; the peers' ids are read in the prologue (xcc_seen) and compared before the slabs are stored
  const size_t tile_slot = ((size_t)tile_n * tiles_m + tile_m) * S;
  const int xcc_mine = 1 + (int)(__builtin_amdgcn_s_getreg(6164) & 15);    // hwreg(HW_REG_XCC_ID = 20, offset 0, 4 bits)
  if (S > 1 && p.peers_local && t == 0) __hip_atomic_store(p.flags + nb + tile_slot + r, xcc_mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const int b = hb_div(tile_m, p.mg_tiles_img), ti = tile_m - b * tiles_img;
  const int tyq = hb_div(ti, p.mg_tiles_x);
  const int ty0 = tyq * TH, tx0 = (ti - tyq * tiles_x) * TW;
  const int n0 = tile_n * BN;
  const int twsh = (TW == 32) ? 5 : 4;                 // TW is 16 or 32
  const int nprow = (TH + 2) * PW, npiece = nprow * 8;

  // ---- K steps: main chunk c = taps [9c, 9c + 9), then one tap per 64 shortcut channels.  Slice r = taps [sb, se): a chunk cut by a
  // boundary runs all its pipeline steps in both blocks, each with the MFMA phases of its own taps only.
  const int nc = p.Cin >> 6, nsc = p.Csc >> 6, T = 9 * nc + nsc;
  auto bnd = [&](int q) {
    int v = (T * q) >> sshift;
    if (v < 9 * nc) { if (TPS == 1) v = (v + 1) / 3 * 3; else { const int c = v / 9, tp = v - 9 * c; v = 9 * c + (tp & ~1); } }   // whole tap rows / whole steps
    return v;
  };
  const int sb = bnd(r), se = bnd(r + 1);
  const int ch_first = sb < 9 * nc ? sb / 9 : nc + (sb - 9 * nc);         // chunk ids: 0 .. nc-1 main, nc + j shortcut
  const int ch_last = (se - 1) < 9 * nc ? (se - 1) / 9 : nc + (se - 1 - 9 * nc);

  // ---- per-thread DMA geometry.  Patch piece i of thread t: q = t + 512 i -> patch row q >> 3 (pixel (py, px) of the halo tile),
  // 16-byte slot q & 7 holding source chunk slot ^ (row & 7).  ppix[i] = pixel index in the image tensor, -1 = padding / beyond the patch.
  int ppix[NPP];
#pragma unroll
  for (int i = 0; i < NPP; ++i) {
    const int q = (dt < 0 ? 0 : dt) + DNT * i, prow = q >> 3;
    const int py = hb_div(prow, p.mg_pw), px = prow - py * PW;
    const int iy = ty0 - 1 + py, ix = tx0 - 1 + px;
    ppix[i] = (q < npiece && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) ? (b * p.H + iy) * p.W + ix : -1;
  }
  const int pslot = ((dt & 7) ^ ((dt >> 3) & 7)) * 8;  // source channel octet of every piece of this thread ((q >> 3) & 7 = (dt >> 3) & 7)
  // weight pieces: instruction j = wave + 8 i covers tile rows 8 j .. 8 j + 7
  constexpr int WI = (BN * 8 + DNT - 1) / DNT;         // rounds of DMA instructions per weight tile (BN / 8 instructions over the DMA waves)
  constexpr int NWV = DNT / 64;
  // (byte offset of this thread's piece of instruction 0 inside the [BN][ldw] weight slab; instruction i is 64 rows further)
  const char* const wslab = (const char*)(p.w + (size_t)n0 * p.ldw);
  const unsigned woff0 = (unsigned)(((dt >> 3) * p.ldw + (((dt & 7) ^ ((dt >> 3) & 7)) * 8)) * 2);
  const unsigned wstep = (unsigned)((DNT / 8) * p.ldw * 2);
  const char* zp = (const char*)p.zeros;

  auto dma = [&](const char* src, int lds_off) {
    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(smem + lds_off), 16, 0, 0);
  };
  // weight tile at K offset `koff` (elements) -> ring stage `st`
  auto issue_w = [&](long koff, int lds_tile) -> int {  // (lds_tile: byte offset of the destination tile in the ring)
    int n = 0;
#pragma unroll
    for (int i = 0; i < WI; ++i) {
      if ((dwave + NWV * i) * 64 >= BN * 8) continue;  // wave-uniform (BN / 8 instructions over the waves: the last round is the low waves only)
      unsigned off = woff0 + wstep * i;
      asm volatile("" : "+v"(off));
      dma(wslab + koff * 2 + off, lds_tile + (dwave * 64 + DNT * i) * 16);
      ++n;
    }
    return n;
  };
  // wait until at most n of this wave's DMA instructions are outstanding (n is wave-uniform, 0 .. 3)
  auto wait_vm = [&](int n) {
    if (n <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (n == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else if (n == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if (n == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if (n == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if (n == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else if (n == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if (n == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    else if (n == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (n == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
  };
  // K offset of the weight tile of pipeline step g (-1 outside this block's slice)
  auto koff_of = [&](int g) -> long {
    if (g < sb || g >= se) return -1;
    if (g < 9 * nc) { const int c = g / 9, tap = g - 9 * c; return (long)tap * p.Cin + c * 64; }
    return (long)9 * p.Cin + (g - 9 * nc) * 64;
  };
  // what a chunk's patch instructions fetch: mode 1 = main chunk (gamma | beta + six pieces of the halo patch), 2 = shortcut chunk (four
  // pieces: the 256 centre pixels, row = tile pixel index), 0 = nothing
  struct PDesc { const bf16* base; int ld; int mode; int ch; };
  auto pdesc = [&](int ch) -> PDesc {
    if (ch < 0) return PDesc{p.x0, 0, 0, 0};
    if (ch < nc) { const int c0 = ch * 64; return c0 < p.cx0 ? PDesc{p.x0 + c0, p.ldx0, 1, ch} : PDesc{p.x1 + (c0 - p.cx0), p.ldx1, 1, ch}; }
    const int c0 = (ch - nc) * 64; return c0 < p.cs0 ? PDesc{p.s0 + c0, p.lds0, 2, ch} : PDesc{p.s1 + (c0 - p.cs0), p.lds1, 2, ch};
  };
  auto spix = [&](int i) {                             // shortcut pieces: image pixel of tile pixel (t + 512 i) >> 3
    int tt = dt; asm volatile("" : "+v"(tt));
    const int pp = (tt + DNT * i) >> 3; return (b * p.H + ty0 + (pp >> twsh)) * p.W + tx0 + (pp & (TW - 1));
  };
  auto issue_coef = [&](const PDesc& d) -> int {        // (all return the number of DMA instructions this wave issued)
    if (d.mode != 1 || !p.gn) return 0;
    const float* g = ((lane & 16) ? p.beta : p.gamma) + d.ch * 64 + (lane & 15) * 4;
    dma((const char*)g, L::GB + (d.ch & 1) * 1024);
    return 1;
  };
  auto issue_piece = [&](const PDesc& d, int pb, const int i) -> int {
    if (d.mode == 0 || (d.mode == 2 && (dwave * 64 + DNT * i) >= 2048)) return 0;                 // (shortcut chunk: 2048 pieces = the 256 centre pixels)
    if (d.mode == 1 && (dwave * 64 + DNT * i) >= npiece) return 0;         // wave-uniform: this instruction lies beyond the patch
    int pix = d.mode == 1 ? ppix[i] : spix(i);
    asm volatile("" : "+v"(pix));                      // keep the address arithmetic here: hoisted out of the chunk loop it is 40 registers of pointers
    const char* src = pix >= 0 ? (const char*)(d.base + (size_t)pix * d.ld + pslot) : zp;
    dma(src, pb + (dwave * 64 + DNT * i) * 16);
    return 1;
  };

  // ---- fragment addresses: m-fragment i of this wave = tile pixels wm*64 + 16 i + lr, one tile row (TW = 16) or half a row (TW = 32);
  // n-fragment j = weight-tile rows wn*16NF + 16 j + lr.  Rows 16 apart share row & 7, so the swizzle term is the same for every fragment:
  // one base register each, the fragment index is an immediate / uniform offset.
  const unsigned lds0 = (unsigned)(size_t)(lptr_t)smem;                    // LDS byte address of smem[0]
  const int pp0 = wm * (16 * MFR) + lr;
  const int xrow0 = (pp0 >> twsh) * PW + (pp0 & (TW - 1));                 // main taps: patch row of fragment 0 at tap (0, 0)
  auto xdelta = [&](int i) { return ((16 * i) >> twsh) * PW + ((16 * i) & (TW - 1)); };                  // ... of fragment i relative to it (uniform)
  const int xsc0 = pp0 * 128 + ((lq ^ (pp0 & 7)) << 4);                    // shortcut step: row = tile pixel index; fragment i is 2048 bytes further
  const int wn0 = wn * (16 * NF) + lr;
  const unsigned wad0 = lds0 + wn0 * 128 + ((lq ^ (wn0 & 7)) << 4);        // fragment j is 2048 bytes further; k-step 1 = this ^ 64
  f32x4 acc[NF][MFR];
#pragma unroll
  for (int j = 0; j < NF; ++j)
#pragma unroll
    for (int i = 0; i < MFR; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- MFMA phase of one tap: 2 k-steps of 32 channels = 2 NF groups of four MFMAs (j-major: one weight fragment x four pixel fragments).
  // Hand-scheduled: the fragment reads are inline-asm ds_read_b128 with COUNTED lgkmcnt waits (hipcc waits lgkmcnt(0) in front of
  // every other group of this loop, i.e. for the prefetches it has just issued).  Read order: X0[0..3] (issued before the barrier: the
  // patch is stable), W[0], W[1], then in front of group g: X1[g] (g < 4) and W[g + 2]; weight fragments rotate through three
  // register sets.  Group g needs W[g] (and X0 / X1 at g = 0 / NF): reads issued after it = [g - 1 < 4] + [g + 1 < 2 NF].
#define HB_DSR(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off) : "memory")
#ifdef DMX_F16
#define HB_MFMA_OP "v_mfma_f32_16x16x32_f16"
#else
#define HB_MFMA_OP "v_mfma_f32_16x16x32_bf16"
#endif
  u32x4 fx0[MFR], fx1[MFR], fw[3];
  auto mma_pre = [&](const unsigned* xa) {
#pragma unroll
    for (int i = 0; i < MFR; ++i) HB_DSR(fx0[i], xa[i], 0);
  };
  // xa: [TAPS][MFR] fragment addresses of the taps, wbase: [TAPS] LDS addresses of their weight tiles
  auto mma_stream = [&](auto TAPS_, const unsigned* xa, const unsigned* wbase, auto PF_) {
    constexpr int TAPS = decltype(TAPS_)::value, G = TAPS * 2 * NF;
    constexpr bool PF = decltype(PF_)::value;          // xa[TAPS * MFR ..): the X0 fragments of the next call's first tap, requested under the last tap's MFMAs
    unsigned wb[TAPS][2];
#pragma unroll
    for (int tt = 0; tt < TAPS; ++tt) { wb[tt][0] = wbase[tt] + wad0; wb[tt][1] = wbase[tt] + (wad0 ^ 64); }
    auto rd_w = [&](const int g2) {
      const int t2 = g2 / (2 * NF), gl2 = g2 - t2 * 2 * NF, k2 = gl2 / NF, j2 = gl2 - k2 * NF;
      const unsigned wsel = wb[t2][k2];
      switch (j2) {
        case 0: HB_DSR(fw[g2 % 3], wsel, 0); break;
        case 1: HB_DSR(fw[g2 % 3], wsel, 2048); break;
        case 2: HB_DSR(fw[g2 % 3], wsel, 2 * 2048); break;
        case 3: HB_DSR(fw[g2 % 3], wsel, 3 * 2048); break;
        case 4: HB_DSR(fw[g2 % 3], wsel, 4 * 2048); break;
        case 5: HB_DSR(fw[g2 % 3], wsel, 5 * 2048); break;
        case 6: HB_DSR(fw[g2 % 3], wsel, 6 * 2048); break;
        case 7: HB_DSR(fw[g2 % 3], wsel, 7 * 2048); break;
        case 8: HB_DSR(fw[g2 % 3], wsel, 8 * 2048); break;
        default: HB_DSR(fw[g2 % 3], wsel, 9 * 2048); break;
      }
    };
    rd_w(0); rd_w(1);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const int tt = g / (2 * NF), gl = g - tt * 2 * NF, kk = gl / NF, j = gl - kk * NF;
      // the wait is tied to the registers the group reads ("+v"): the MFMAs cannot be scheduled in front of it
#define HB_WAIT(N_)                                                                                                                   \
      if (gl == 0) {                                                                                                                  \
        if constexpr (MFR == 4) asm volatile("s_waitcnt lgkmcnt(" #N_ ")" : "+v"(fx0[0]), "+v"(fx0[1]), "+v"(fx0[2]), "+v"(fx0[3]), "+v"(fw[g % 3]) :: "memory"); \
        else asm volatile("s_waitcnt lgkmcnt(" #N_ ")" : "+v"(fx0[0]), "+v"(fx0[1]), "+v"(fw[g % 3]) :: "memory");                      \
      } else if (gl == NF) {                                                                                                          \
        if constexpr (MFR == 4) asm volatile("s_waitcnt lgkmcnt(" #N_ ")" : "+v"(fx1[0]), "+v"(fx1[1]), "+v"(fx1[2]), "+v"(fx1[3]), "+v"(fw[g % 3]) :: "memory"); \
        else asm volatile("s_waitcnt lgkmcnt(" #N_ ")" : "+v"(fx1[0]), "+v"(fx1[1]), "+v"(fw[g % 3]) :: "memory");                      \
      } else asm volatile("s_waitcnt lgkmcnt(" #N_ ")" : "+v"(fw[g % 3]) :: "memory");
      switch (hb_wait(NF, MFR, TAPS, g, PF)) {           // (g is a constant after unrolling; the immediate has to be a literal)
        case 0: HB_WAIT(0) break;
        case 1: HB_WAIT(1) break;
        case 2: HB_WAIT(2) break;
        case 3: HB_WAIT(3) break;
        default: HB_WAIT(4) break;
      }
#undef HB_WAIT
      __builtin_amdgcn_sched_barrier(0);
      if (gl < MFR) { const unsigned x1 = xa[tt * MFR + gl] ^ 64; HB_DSR(fx1[gl], x1, 0); }
      if ((tt + 1 < TAPS || PF) && gl >= NF && gl < NF + MFR) HB_DSR(fx0[gl - NF], xa[(tt + 1) * MFR + gl - NF], 0);
      if (g + 2 < G) rd_w(g + 2);
      // (the MFMAs as asm with the accumulator tied in place: left to the compiler every MFMA of this stream writes a NEW register quad -
      // the accumulator set migrates through the file and the stream needs ~100 registers on top of it; nothing reads an accumulator
      // sooner than a whole k-step later, so no software wait states are due)
#pragma unroll
      for (int i = 0; i < MFR; ++i) {
        if (kk) asm volatile(HB_MFMA_OP " %0, %1, %2, %0" : "+v"(acc[j][i]) : "v"(fw[g % 3]), "v"(fx1[i]));
        else asm volatile(HB_MFMA_OP " %0, %1, %2, %0" : "+v"(acc[j][i]) : "v"(fw[g % 3]), "v"(fx0[i]));
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_s_setprio(0);
  };
  auto mma_post = [&](const unsigned* xa, unsigned wbase, auto&&) { mma_stream(std::integral_constant<int, 1>{}, xa, &wbase, std::false_type{}); };
  auto xa_main = [&](int pb, int tapoff, unsigned* xa) {
    int pr = xrow0;
    asm volatile("" : "+v"(pr));                       // (the tap addresses are loop-invariant: hoisted they would live across the whole K loop)
    pr += tapoff;
#pragma unroll
    for (int i = 0; i < MFR; ++i) { const int rr = pr + xdelta(i); xa[i] = lds0 + pb + rr * 128 + ((lq ^ (rr & 7)) << 4); }
  };
  auto xa_sc = [&](int pb, unsigned* xa) {
    int x0 = xsc0;
    asm volatile("" : "+v"(x0));
#pragma unroll
    for (int i = 0; i < MFR; ++i) xa[i] = lds0 + pb + x0 + 2048 * i;
  };

  // ---- GroupNorm pieces
  // (a, s) of the 64 channels of main chunk `ch` from the group statistics and the chunk's gamma | beta slot; every wave writes the
  // same 64 entries (no divergence, no extra barrier)
  const int cpg = p.cpg;
  auto coef_table = [&](int ch) {
    if (!p.gn) return;
    const int c = ch * 64 + lane;
    const int g = hb_div(c, p.mg_cpg);
    const float* gb = (const float*)(smem + L::GB + (ch & 1) * 1024);
    const float* gs = (const float*)(smem + L::GST);
    const float a = gs[2 * g + 1] * gb[lane];
    float* cf = (float*)(smem + L::COEF);
    cf[2 * lane] = a; cf[2 * lane + 1] = gb[64 + lane] - gs[2 * g] * a;
  };
  // normalise piece i of the patch in buffer `pb` in place (a thread normalises the pieces it requested itself: its own vmcnt wait
  // is all the ordering this needs).  Three parts so that the K loop can put the arithmetic in the shadow of its MFMAs: nrm_load (LDS
  // reads of the piece and of its 8 channels' (a, s), in the DMA phase), nrm_slice(e) (element e: fma, SiLU - after MFMA group e), nrm_store.
  u32x4 nrm_x; f32x4 nrm_c[4]; float nrm_y[8]; int nrm_q = -1;
  auto nrm_load = [&](int pb, const int i) {           // (call only with p.gn; every lane loads - lanes beyond the patch re-read piece 0 and store nothing)
    const int q = dt + DNT * i;
    nrm_q = q >= npiece ? -1 : (ppix[i] < 0 ? -2 - q : q);                 // (padding pieces are written as zeros: the conv pads the NORMALISED tensor)
    nrm_x = *(const u32x4*)(smem + pb + (q >= npiece ? 0 : q) * 16);
    const float* cf = (const float*)(smem + L::COEF) + pslot * 2;
#pragma unroll
    for (int k = 0; k < 4; ++k) nrm_c[k] = *(const f32x4*)(cf + 4 * k);
  };
  // (a use of the loaded registers right behind the loads, in the same branch: the compiler's wait for them lands HERE and not in front
  // of every slice, where it would also wait for the fragment prefetches in flight)
  auto nrm_ready = [&]() { asm volatile("" : "+v"(nrm_x), "+v"(nrm_c[0]), "+v"(nrm_c[1]), "+v"(nrm_c[2]), "+v"(nrm_c[3])); };
  auto nrm_slice = [&](const int e) {
    const float x = (e & 1) ? h2f_hi(nrm_x[e >> 1]) : h2f_lo(nrm_x[e >> 1]);
    float y = __builtin_fmaf(x, nrm_c[e >> 1][(e & 1) * 2], nrm_c[e >> 1][(e & 1) * 2 + 1]);
    if (p.silu) y *= __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896f * y));
    nrm_y[e] = y;
  };
  auto nrm_store = [&](int pb) {
    if (nrm_q == -1) return;
    u32x4 o = pack_bf8(nrm_y);
    int q = nrm_q;
    if (q < 0) { o = u32x4{0u, 0u, 0u, 0u}; q = -2 - q; }
    *(u32x4*)(smem + pb + q * 16) = o;
  };
  auto norm_piece = [&](int pb, const int i) {
    if (!p.gn) return;
    nrm_load(pb, i);
#pragma unroll
    for (int e = 0; e < 8; ++e) nrm_slice(e);
    nrm_store(pb);
  };

  // ---- weight prefetch for the launches that follow (HaloConvArgs.pf): 1-KB units dealt over (block, wave), the oldest requests of
  // every wave (the prologue's vmcnt(0) covers them)
  int pf_left = 4;                                     // (at most four units per wave)
#pragma unroll
  for (int r_ = 0; r_ < 2; ++r_) {
    const int nb_ = p.pf_bytes[r_];
    for (int u_ = blockIdx.x * (HB_NT / 64) + wave; u_ * 1024 < nb_ && pf_left > 0; u_ += gridDim.x * (HB_NT / 64), --pf_left) {
      int off_ = u_ * 1024 + lane * 16; if (off_ > nb_ - 16) off_ = nb_ - 16;
      dma((const char*)p.pf[r_] + off_, L::DUMP);
    }
  }
  // ---- prologue: the first patch and weight tile are requested FIRST (their latency covers the statistics), then the group statistics
  // of this block's sample
  int cur = ch_first, seq = 0;
  int gstep = cur < nc ? 9 * cur : 9 * nc + (cur - nc);   // pipeline step of the chunk's first step (inactive steps of a partial chunk included)
  if (loader) {
    const PDesc d = pdesc(cur);
    issue_coef(d);
#pragma unroll
    for (int i = 0; i < NPP; ++i) issue_piece(d, L::PATCH0, i);
    if constexpr (TPS == 1) {
#pragma unroll
      for (int d = 0; d < PD; ++d) { const long k0 = koff_of(gstep + d); if (k0 >= 0) issue_w(k0, ((gstep + d) % NSTG) * WSTAGE); }
    } else {
#pragma unroll
      for (int u = 0; u < TPS; ++u) { const long k0 = koff_of(gstep + u); if (k0 >= 0 && (cur < nc || u == 0)) issue_w(k0, u * WTILE); }     // ring step 0: the first step's taps
    }
  }
  if (TIMING) tp[0] = __builtin_amdgcn_s_memrealtime();
  // lane s of every wave asks for peer s's XCC word (one request per wave; it lands under the statistics below and the wait for the patch)
  int xcc_seen = xcc_mine;
  if (S > 1 && p.peers_local && lane < S) xcc_seen = __hip_atomic_load(p.flags + nb + tile_slot + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (p.gn && ch_first < nc) {
    // 16 threads per group sum the group's channels' records (integers: exact, any order), then mean / variance in double
    const int g = t >> 4, sub = t & 15;                // (threads 0 .. 511)
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
      gs[2 * g] = (float)mean; gs[2 * g + 1] = (float)(1.0 / __builtin_sqrt(var + (double)p.eps));      // (as dmx_gn_apply_kernel and the chain's folded GroupNorm: same bits on every path)
    }
  }
  if (TIMING) tp[1] = __builtin_amdgcn_s_memrealtime();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                        // the group statistics are published (the patch pieces a thread normalises are its own)
  if (TIMING) tp[2] = __builtin_amdgcn_s_memrealtime();
  // every peer is confirmed on THIS XCD (a peer that had not started yet, or sits elsewhere: the write-through exchange of round 3) - wave-uniform
  const bool slabs_local = S > 1 && p.peers_local && __builtin_amdgcn_ballot_w64(xcc_seen != xcc_mine) == 0;
  if (cur < nc && p.gn) {
    coef_table(cur);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    {
      // (every thread of the block takes pieces here - the requests are confirmed and behind a barrier; in the K loop a DMA thread
      // normalises the pieces it requested itself)
      constexpr int NPA = (HB_PATCH / 16 + HB_NT - 1) / HB_NT;
      const int pslot_a = ((t & 7) ^ ((t >> 3) & 7)) * 8;
      u32x4 px[NPA];
#pragma unroll
      for (int i = 0; i < NPA; ++i) { const int q = t + HB_NT * i; px[i] = *(const u32x4*)(smem + L::PATCH0 + (q >= npiece ? 0 : q) * 16); }
      const float* cf = (const float*)(smem + L::COEF) + pslot_a * 2;
      f32x4 cc[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) cc[k] = *(const f32x4*)(cf + 4 * k);
#pragma unroll
      for (int i = 0; i < NPA; ++i) {
        const int q = t + HB_NT * i;
        bool pad;
        if constexpr (WS) {
          const int prow = q >> 3, py = hb_div(prow, p.mg_pw), px_ = prow - py * PW;
          const int iy = ty0 - 1 + py, ix = tx0 - 1 + px_;
          pad = !(iy >= 0 && iy < p.H && ix >= 0 && ix < p.W);
        } else pad = ppix[i] < 0;
        float f[8]; unpack_bf8(px[i], f);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float y = __builtin_fmaf(f[e], cc[e >> 1][(e & 1) * 2], cc[e >> 1][(e & 1) * 2 + 1]);
          if (p.silu) y *= __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896f * y));
          f[e] = y;
        }
        u32x4 o = pack_bf8(f);
        if (pad) o = u32x4{0u, 0u, 0u, 0u};
        if (q < npiece) *(u32x4*)(smem + L::PATCH0 + q * 16) = o;
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  if (TIMING) tm[1] = __builtin_amdgcn_s_memrealtime();
  // ---- K loop, main chunks: TWO-GROUP PING-PONG.  Waves w and w + 4 share a SIMD; group A = waves 0..3, B = waves 4..7.  Per tap every
  // wave runs a DMA phase (weights of the next tap; in steps 1 / 2 the next chunk's patch; one piece of normalisation; the first fragment
  // reads), a barrier, an MFMA phase (40 MFMAs back to back), a barrier.  Both groups run the same instruction stream, B one barrier
  // behind: on every SIMD one wave is in its MFMA phase while its partner issues LDS-DMA and does the normalisation VALU work (an
  // LDS-DMA instruction stalls its wave 60-180 cycles; in lock step both waves of a SIMD pay that together and the matrix pipe idles).
  //   A: dma(s) |X| mfma(s) |Y| dma(s+1) ...          B: |.| dma(s) |X'| mfma(s) |Y'| ...      X' = A's Y
  // Weights of step s + 1 go to stage (s + 1) % 3 while the partner reads stage (s - 1) % 3 and this wave reads s % 3 next; a wave
  // confirms its own requests (vmcnt(0)) at the END of its MFMA phase - a DMA + an MFMA phase after issuing them - so both groups'
  // pieces of step s + 1 are confirmed by the barrier in front of mfma(s + 1).  The next chunk's patch is requested in steps 1 / 2 (in
  // step 0 the partner still reads that buffer for step 8 of the previous chunk) and normalised in steps 3 .. 8 by the threads that
  // requested it.
#ifdef DMX_PROBES
  long long pa_dma = 0, pa_wx = 0, pa_mma = 0, pa_wy = 0, pa_t = __builtin_amdgcn_s_memtime();   // shader cycles in: DMA phase, barrier X, MFMA phase, barrier Y
#define HB_STAMP(acc_) { const long long n_ = __builtin_amdgcn_s_memtime(); acc_ += n_ - pa_t; pa_t = n_; }
#else
#define HB_STAMP(acc_)
#endif
  if constexpr (PP && TPS == 2) {
  // ---- K loop of the narrow tiles: two-group ping-pong (below) with TWO taps per step - steps (0,1) (2,3) (4,5) (6,7) (8) of a chunk.
  // The weight tiles of the next step go to ring stage (rs + 1) % 3 (two 10-KB slots); the two taps of a step run as ONE MFMA stream
  // (the second tap's fragments are prefetched under the first's MFMAs).  The next chunk's patch is requested in steps 1 / 2 and
  // normalised one piece per PHASE in steps 2 .. 4: in the DMA phase and again behind the MFMAs of the MFMA phase.
  const bool grpB = wave >= 4;
  const bool pingpong = cur >= 0 && cur < nc;
  int rs = 0;                                          // ring step
  if (pingpong && grpB) __builtin_amdgcn_s_barrier();
  while (cur >= 0 && cur < nc) {
    const int next = cur < ch_last ? cur + 1 : -1;
    const int pb = (seq & 1) ? L::PATCH1 : L::PATCH0, pbn = (seq & 1) ? L::PATCH0 : L::PATCH1;
    const PDesc dn = pdesc(next);
    const bool nrm = dn.mode == 1 && p.gn && !(DBG & 4);
    auto step = [&](auto K_) {
      constexpr int k = decltype(K_)::value, t0 = 2 * k, nt = k < 4 ? 2 : 1;
      const int g0 = gstep + t0;
      const bool active = g0 >= sb && g0 < se;           // (slices are cut on step starts: a step is active or not as a whole)
      // ---- DMA phase
      if (!(DBG & 8)) {
        if constexpr (k == 1) { issue_coef(dn); issue_piece(dn, pbn, 0); issue_piece(dn, pbn, 1); issue_piece(dn, pbn, 2); }
        if constexpr (k == 2) { issue_piece(dn, pbn, 3); issue_piece(dn, pbn, 4); issue_piece(dn, pbn, 5); }
      }
      if (!(DBG & 2)) {
        // the next step's taps: 2 (k + 1) .. of this chunk, or the first tap(s) of the next chunk (one tap if that is a shortcut chunk)
        const int n0g = k < 4 ? g0 + 2 : gstep + 9;
        const int nn = (k + 1 < 4) ? 2 : ((k + 1 == 4) ? 1 : ((next >= 0 && next < nc) ? 2 : 1));
        const int dst = ((rs + 1) % 3) * WSTAGE;
        { const long kn = koff_of(n0g); if (kn >= 0) issue_w(kn, dst); }
        if (nn == 2) { const long kn = koff_of(n0g + 1); if (kn >= 0) issue_w(kn, dst + WTILE); }
      }
      if constexpr (k == 2) { if (dn.mode == 1) coef_table(next); }
      if constexpr (k >= 2) { if (nrm) norm_piece(pbn, 2 * (k - 2)); }
      unsigned xa[2 * MFR];
      if (active) { xa_main(pb, (t0 / 3) * PW + (t0 % 3), xa); if (nt == 2) xa_main(pb, ((t0 + 1) / 3) * PW + ((t0 + 1) % 3), xa + MFR); }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (active && !(DBG & 1)) mma_pre(xa);
      HB_STAMP(pa_dma)
      if (!(DBG & 16)) __builtin_amdgcn_s_barrier();
      HB_STAMP(pa_wx)
      // ---- MFMA phase
      if (active && !(DBG & 1)) {
        const unsigned wbs[2] = {lds0 + (rs % 3) * WSTAGE, lds0 + (rs % 3) * WSTAGE + WTILE};
        mma_stream(std::integral_constant<int, nt>{}, xa, wbs, std::false_type{});
      }
      if constexpr (k >= 2) { if (nrm) norm_piece(pbn, 2 * (k - 2) + 1); }
      wait_vm(0);                                        // this wave's requests of this step's DMA phase have landed
      HB_STAMP(pa_mma)
      if (!(DBG & 16)) __builtin_amdgcn_s_barrier();
      HB_STAMP(pa_wy)
      ++rs;
    };
    step(std::integral_constant<int, 0>{}); step(std::integral_constant<int, 1>{}); step(std::integral_constant<int, 2>{});
    step(std::integral_constant<int, 3>{}); step(std::integral_constant<int, 4>{});
    gstep += 9; ++seq;
    cur = next;
  }
  if (pingpong && !grpB) __builtin_amdgcn_s_barrier(); // (group B's extra barrier of the entry: the groups are level again)
  // ---- shortcut steps (one tap each): lock step, the next shortcut patch requested one step ahead into the other patch buffer
  while (cur >= nc) {
    const int next = cur < ch_last ? cur + 1 : -1;
    const int pb = (seq & 1) ? L::PATCH1 : L::PATCH0, pbn = (seq & 1) ? L::PATCH0 : L::PATCH1;
    const PDesc dn = pdesc(next);
    issue_piece(dn, pbn, 0); issue_piece(dn, pbn, 1); issue_piece(dn, pbn, 2); issue_piece(dn, pbn, 3);
    { const long kn = koff_of(gstep + 1); if (kn >= 0) issue_w(kn, ((rs + 1) % 3) * WSTAGE); }
    unsigned xa[MFR];
    xa_sc(pb, xa);
    mma_pre(xa);
    __builtin_amdgcn_s_barrier();
    const unsigned wb1 = lds0 + (rs % 3) * WSTAGE;
    mma_stream(std::integral_constant<int, 1>{}, xa, &wb1, std::false_type{});
    wait_vm(0);                                        // (the next shortcut patch and weight tile have landed)
    __builtin_amdgcn_s_barrier();
    gstep += 1; ++seq; ++rs;
    cur = next;                                        // (-1 ends the loop)
  }
  } else
  if constexpr (PP) {
  const bool grpB = wave >= 4;
  const bool pingpong = cur >= 0 && cur < nc;
  if (pingpong && grpB) __builtin_amdgcn_s_barrier();
  while (cur >= 0 && cur < nc) {
    const int next = cur < ch_last ? cur + 1 : -1;
    const int pb = (seq & 1) ? L::PATCH1 : L::PATCH0, pbn = (seq & 1) ? L::PATCH0 : L::PATCH1;
    const PDesc dn = pdesc(next);
    auto step = [&](auto S_) {
      constexpr int s = decltype(S_)::value;
      const int g = gstep + s;
      const bool active = g >= sb && g < se;
      // ---- DMA phase
      if (!(DBG & 8)) {
        if constexpr (s == 1) { issue_coef(dn); issue_piece(dn, pbn, 0); issue_piece(dn, pbn, 1); issue_piece(dn, pbn, 2); }
        if constexpr (s == 2) { issue_piece(dn, pbn, 3); issue_piece(dn, pbn, 4); issue_piece(dn, pbn, 5); }
        static_assert(!PP || NPP == 6, "ping-pong schedule: six patch pieces per thread");
      }
      int nw = 0;                                        // (patch pieces first, the weight tile last: the counted wait below leaves only IT in flight)
      if (!(DBG & 2)) { const long kn = koff_of(g + PD); if (kn >= 0) nw = issue_w(kn, ((g + PD) % NSTG) * WSTAGE); }
      if constexpr (s == 2) { if (dn.mode == 1) coef_table(next); }
      if constexpr (s >= 3) { if (dn.mode == 1 && !(DBG & 4)) norm_piece(pbn, s - 3); }
      unsigned xa[MFR];
      if (active) xa_main(pb, (s / 3) * PW + (s % 3), xa);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (active && !(DBG & 1)) mma_pre(xa);
      HB_STAMP(pa_dma)
      if (!(DBG & 16)) __builtin_amdgcn_s_barrier();
      HB_STAMP(pa_wx)
      // ---- MFMA phase.  (The normalisation arithmetic as per-group slices in here - loads in the DMA phase, one element after each
      // MFMA group - was built and measured slower, 65 vs 60 us: one wave per SIMD is in this phase, so VALU between its MFMAs delays them)
      if (active && !(DBG & 1)) mma_post(xa, lds0 + (g % NSTG) * WSTAGE, [&](auto, const int) {});
      // this wave's pieces of the NEXT tap's weight tile (requested a DMA + an MFMA phase ago) and everything older have landed
      wait_vm(0); (void)nw;
      HB_STAMP(pa_mma)
      if (!(DBG & 16)) __builtin_amdgcn_s_barrier();
      HB_STAMP(pa_wy)
    };
    step(std::integral_constant<int, 0>{}); step(std::integral_constant<int, 1>{}); step(std::integral_constant<int, 2>{});
    step(std::integral_constant<int, 3>{}); step(std::integral_constant<int, 4>{}); step(std::integral_constant<int, 5>{});
    step(std::integral_constant<int, 6>{}); step(std::integral_constant<int, 7>{}); step(std::integral_constant<int, 8>{});
    gstep += 9; ++seq;
    cur = next;
  }
  if (pingpong && !grpB) __builtin_amdgcn_s_barrier(); // (group B's extra barrier of the entry: the groups are level again)
  // ---- shortcut steps: lock step, the next shortcut patch requested one step ahead into the other patch buffer
  while (cur >= nc) {
    const int next = cur < ch_last ? cur + 1 : -1;
    const int pb = (seq & 1) ? L::PATCH1 : L::PATCH0, pbn = (seq & 1) ? L::PATCH0 : L::PATCH1;
    const PDesc dn = pdesc(next);
    issue_piece(dn, pbn, 0); issue_piece(dn, pbn, 1); issue_piece(dn, pbn, 2); issue_piece(dn, pbn, 3);
    int nw = 0;
    { const long kn = koff_of(gstep + PD); if (kn >= 0) nw = issue_w(kn, ((gstep + PD) % NSTG) * WSTAGE); }
    unsigned xa[MFR];
    xa_sc(pb, xa);
    mma_pre(xa);
    __builtin_amdgcn_s_barrier();
    mma_post(xa, lds0 + (gstep % NSTG) * WSTAGE, [&](auto, const int) {});
    wait_vm(0); (void)nw;                              // (the next shortcut patch and weight tile have landed)
    __builtin_amdgcn_s_barrier();
    gstep += 1; ++seq;
    cur = next;                                        // (-1 ends the loop)
  }
  } else {
  // ---- K loop, warp-specialised instances: waves 0 .. 3 COMPUTE (fragment reads + MFMAs of tap s), waves 4 .. 7 LOAD (the weight tile of
  // tap s + 2 into the ring stage tap s - 1 has just freed; in steps 0 .. 2 the next chunk's patch; step 3: its (a, s) table; steps 3 .. 8:
  // two pieces of its normalisation per step).  ONE barrier per tap: behind barrier s the loaders' requests of step s - 1 and older have
  // landed (counted vmcnt: only the requests of the step itself stay in flight) and their LDS writes are complete, and the compute waves
  // have finished the fragment reads of tap s - 1.  A compute wave requests the first fragments of tap s + 1 under the last MFMAs of tap s
  // (same patch: stable), so only the first tap of a chunk starts with an exposed LDS round trip.
  static_assert(!WS || (NSTG == 3 && PD == 2), "warp-specialised schedule: three ring stages, weights two taps ahead");
  static_assert(!WS || NPP <= 12, "warp-specialised schedule: at most twelve patch pieces per loader thread");
  // (two separate loops, not one loop with a role branch inside: the compiler cannot know that the role never changes, and would keep
  // the loaders' piece table and the compute waves' fragments alive through each other's code on top of the 160 accumulator registers)
  if (loader) {
    while (cur >= 0 && cur < nc) {
      const int next = cur < ch_last ? cur + 1 : -1;
      const int pbn = (seq & 1) ? L::PATCH0 : L::PATCH1;
      const PDesc dn = pdesc(next);
      auto step = [&](auto S_) {
        constexpr int s = decltype(S_)::value;
        const int g = gstep + s;
        if (!(DBG & 16)) __builtin_amdgcn_s_barrier();
        int n = 0;
        if (!(DBG & 8)) {
          if constexpr (s == 0) n += issue_coef(dn);
          if constexpr (s < 3) {
#pragma unroll
            for (int i = 4 * s; i < 4 * s + 4 && i < NPP; ++i) n += issue_piece(dn, pbn, i);
          }
        }
        if (!(DBG & 2)) { const long kn = koff_of(g + PD); if (kn >= 0) n += issue_w(kn, ((g + PD) % NSTG) * WSTAGE); }
        if constexpr (s == 3) { if (dn.mode == 1) coef_table(next); }
        if constexpr (s >= 3) {
          if (dn.mode == 1 && !(DBG & 4)) {
            norm_piece(pbn, 2 * (s - 3));
            if constexpr (2 * (s - 3) + 1 < NPP) norm_piece(pbn, 2 * (s - 3) + 1);
          }
        }
        wait_vm(n);                                      // everything requested before this step has landed
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      };
      step(std::integral_constant<int, 0>{}); step(std::integral_constant<int, 1>{}); step(std::integral_constant<int, 2>{});
      step(std::integral_constant<int, 3>{}); step(std::integral_constant<int, 4>{}); step(std::integral_constant<int, 5>{});
      step(std::integral_constant<int, 6>{}); step(std::integral_constant<int, 7>{}); step(std::integral_constant<int, 8>{});
      gstep += 9; ++seq;
      cur = next;
    }
    // shortcut steps: the next shortcut patch requested one step ahead into the other patch buffer (confirmed in the same step: its request
    // goes out BEFORE the weight tile's, the counted wait leaves only that in flight), the weight tile two steps ahead
    while (cur >= nc) {
      const int next = cur < ch_last ? cur + 1 : -1;
      const int pbn = (seq & 1) ? L::PATCH0 : L::PATCH1;
      const PDesc dn = pdesc(next);
      __builtin_amdgcn_s_barrier();
#pragma unroll
      for (int i = 0; i < NPP; ++i) issue_piece(dn, pbn, i);
      int nwt = 0;
      { const long kn = koff_of(gstep + PD); if (kn >= 0) nwt = issue_w(kn, ((gstep + PD) % NSTG) * WSTAGE); }
      wait_vm(nwt);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      gstep += 1; ++seq;
      cur = next;                                        // (-1 ends the loop)
    }
  } else {
    while (cur >= 0 && cur < nc) {
      const int next = cur < ch_last ? cur + 1 : -1;
      const int pb = (seq & 1) ? L::PATCH1 : L::PATCH0;
      auto step = [&](auto S_) {
        constexpr int s = decltype(S_)::value;
        const int g = gstep + s;
        const bool active = g >= sb && g < se;
        if (!(DBG & 16)) __builtin_amdgcn_s_barrier();
        // (ONE branch around the whole tap and no variants inside it: with run-time variants of the stream the accumulators come out of
        // the merges in different registers - copies and spills by the hundred)
        if (active && !(DBG & 1)) {
          unsigned xa[2 * MFR];
          xa_main(pb, (s / 3) * PW + (s % 3), xa);
          if constexpr (s < 8) xa_main(pb, ((s + 1) / 3) * PW + ((s + 1) % 3), xa + MFR);
          // taps 1 .. 8 find their first fragments requested by the previous tap (same patch: stable) unless the slice starts here
          if (s == 0 || g == sb) mma_pre(xa);
          const unsigned wb1 = lds0 + (g % NSTG) * WSTAGE;
          mma_stream(std::integral_constant<int, 1>{}, xa, &wb1, std::integral_constant<bool, (s < 8)>{});
        }
      };
      step(std::integral_constant<int, 0>{}); step(std::integral_constant<int, 1>{}); step(std::integral_constant<int, 2>{});
      step(std::integral_constant<int, 3>{}); step(std::integral_constant<int, 4>{}); step(std::integral_constant<int, 5>{});
      step(std::integral_constant<int, 6>{}); step(std::integral_constant<int, 7>{}); step(std::integral_constant<int, 8>{});
      gstep += 9; ++seq;
      cur = next;
    }
    while (cur >= nc) {
      const int next = cur < ch_last ? cur + 1 : -1;
      const int pb = (seq & 1) ? L::PATCH1 : L::PATCH0;
      __builtin_amdgcn_s_barrier();
      unsigned xa[MFR];
      xa_sc(pb, xa);
      mma_pre(xa);
      const unsigned wb1 = lds0 + (gstep % NSTG) * WSTAGE;
      mma_stream(std::integral_constant<int, 1>{}, xa, &wb1, std::false_type{});
      gstep += 1; ++seq;
      cur = next;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                        // every wave is done with the ring and the patches: LDS is the epilogue's
  }
#undef HB_DSR
#undef HB_MFMA_OP

  if (TIMING) tm[2] = __builtin_amdgcn_s_memrealtime();
  // ---------------------------------------------------------------- epilogue
  // acc[j][i][e] = out[pixel wm*16MFR + i*16 + lr][channel n0 + wn*16NF + j*16 + 4 lq + e].  Rows (tile pixels) [own0, own0 + RO) are
  // this block's; the other rows of its accumulators go to its slab for their owners.
  constexpr int LDT = L::LDT, OCP = L::OCP, RL = L::RL;
  const int RO = 256 / S, own0 = r * RO;
  const size_t slab_elems = (size_t)256 * BN;
  if (S > 1) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.slabs + (tile_slot + r) * slab_elems), 0, (int)(slab_elems * 4), 0x00020000);
    // peers on this XCD: PLAIN stores - the lines stay in the XCD's L2, where the peers' sc1 loads (L1 bypassed, L2 served) find them; otherwise
    // write-through (sc1) stores, which reach memory and drop the line (MI355X_MICROARCH.md, "stores of each flavour")
    if (slabs_local) {
#pragma unroll
      for (int i = 0; i < MFR; ++i) {
        const int row0 = wm * (16 * MFR) + i * 16;
        if (row0 / RO == r || (WS && wave >= NCW)) continue;   // wave-uniform (the loader waves hold no accumulators)
#pragma unroll
        for (int j = 0; j < NF; ++j)
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[j][i]), rs, ((row0 + lr) * BN + wn * 16 * NF + j * 16 + 4 * lq) * 4, 0, 0);
      }
    } else {
#pragma unroll
      for (int i = 0; i < MFR; ++i) {
        const int row0 = wm * (16 * MFR) + i * 16;
        if (row0 / RO == r || (WS && wave >= NCW)) continue;
#pragma unroll
        for (int j = 0; j < NF; ++j)
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[j][i]), rs, ((row0 + lr) * BN + wn * 16 * NF + j * 16 + 4 * lq) * 4, 0, 16);
      }
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
  const int RPP = RO < 128 ? RO : 128, passes = RO / RPP;    // (S = 1: two passes of 128 rows; S > 1: one pass of 256 / S rows)
  // items of one pass: (row lane rl, octet o), the octet fixed per thread so the column partial sums stay in registers; U rows per
  // thread in flight (loads of the residual and of the peers' slabs first, then the arithmetic and the stores)
  auto items = [&](auto S_, auto U_, const int prow0) {
    constexpr int SS = decltype(S_)::value, U = decltype(U_)::value;
    for (int k0 = 0; k0 * RL < RPP; k0 += U) {
      u32x4 rr[U]; f32x4 pv[U][SS][2];                 // (indexed by slice: entry r stays unused - static register indices)
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
#ifdef HB_LOCAL_AUX
            if (slabs_local) {
              pv[u][s][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, HB_LOCAL_AUX));
              pv[u][s][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off + 16, 0, HB_LOCAL_AUX));
              continue;
            }
#endif
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
  auto stage = [&](const int prow0) {
#pragma unroll
    for (int i = 0; i < MFR; ++i) {
      const int row0 = wm * (16 * MFR) + i * 16;
      if (row0 < prow0 || row0 >= prow0 + RPP || (WS && wave >= NCW)) continue;
#pragma unroll
      for (int j = 0; j < NF; ++j) *(f32x4*)(tile + (row0 - prow0 + lr) * LDT + wn * 16 * NF + j * 16 + 4 * lq) = acc[j][i];
    }
  };
  constexpr int KPP = (128 + RL - 1) / RL;             // rows per thread of a 128-row pass
  if (S == 1) {
    for (int ep = 0; ep < passes; ++ep) {
      if (ep) __builtin_amdgcn_s_barrier();            // the previous pass is done with the staging tile
      stage(own0 + ep * RPP);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __syncthreads();
      if (TIMING && ep == 0) tm[4] = __builtin_amdgcn_s_memrealtime();
      items(std::integral_constant<int, 1>{}, std::integral_constant<int, PP ? 3 : 2>{}, own0 + ep * RPP);
    }
  } else {
    // one pass: after the staging the accumulators are dead, so every row of a thread is in flight at once (8-wave instances)
    stage(own0);
    if (t == 0) {
      // peers' slabs: bounded spin.  A lost peer must never hang the GPU, and never pass silently either: after ~40 ms the block RAISES
      // the device error (common.h dmx_dev_raise -> DMX_ERR_DEVICE at the next launch check / dmx_device_error()) and goes on
      const long long t0 = __builtin_amdgcn_s_memrealtime();
      for (int s = 0; s < S; ++s) {
        if (s == r) continue;
        while (__hip_atomic_load(p.flags + tile_slot + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
          __builtin_amdgcn_s_sleep(2);
          if (__builtin_amdgcn_s_memrealtime() - t0 > 4000000) { dmx_dev_raise(p.err, DMX_DEVK_HALO_PEER, (int)blockIdx.x, tile_slot, s, S); break; }
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
    if (TIMING) tm[4] = __builtin_amdgcn_s_memrealtime();
    // (the slab loads are sc1 in both exchange modes: L1 bypassed, served by the XCD's L2 where the peers left the lines there, by the fabric otherwise.
    // nt / sc0 / plain loads of L2-resident slabs measured no faster - EXPERIMENTS.md round 6)
    if (S == 2) items(std::integral_constant<int, 2>{}, std::integral_constant<int, PP ? KPP : (WS ? (KPP + 1) / 2 : 1)>{}, own0);
    else if (S == 4) items(std::integral_constant<int, 4>{}, std::integral_constant<int, PP ? (KPP + 1) / 2 : (WS ? 2 : 1)>{}, own0);
    else if constexpr (PP || WS) items(std::integral_constant<int, 8>{}, std::integral_constant<int, 1>{}, own0);      // (8-way splits: 8-wave instances only - the plan sees to it)
  }
  if (TIMING) tm[6] = __builtin_amdgcn_s_memrealtime();
  if (p.colstats) {
    // per-channel (sum, sum of squares) of the ROUNDED outputs of this block's rows: float inside the block in a fixed order, 64-bit
    // fixed point across blocks (DmxStat)
    float* red = (float*)(smem + L::EPI_FOLD);
    if (act) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { red[(rl * BN + o * 8 + e) * 2] = cs_s[e]; red[(rl * BN + o * 8 + e) * 2 + 1] = cs_q[e]; }
    }
    __syncthreads();
    {
      // channel c is folded by FT adjacent lanes (row lanes k = part, part + FT, ...; fixed order inside a lane, then a fixed shuffle tree)
      constexpr int FT = (HB_NT / BN) >= 2 ? 2 : 1;
      const int c = t / FT, part = t % FT;
      float sa = 0.f, sq = 0.f;
      if (c < BN)
        for (int k = part; k < RL; k += FT) { sa += red[(k * BN + c) * 2]; sq += red[(k * BN + c) * 2 + 1]; }
      if (FT == 2) { sa += __shfl_xor(sa, 1); sq += __shfl_xor(sq, 1); }
      if (c < BN && part == 0) dmx_stat_add(p.colstats + ((size_t)b * p.N + n0 + c) * DMX_STAT_WORDS, sa, sq);
    }
  }
#ifdef DMX_PROBES
  if (TIMING && (t == 0 || t == 256)) {                // phase sums of wave 0 (group A) and wave 4 (group B), behind the 4096 block records
    long long* o2 = TIMING + (size_t)(4096 + blockIdx.x * 2 + (t >> 8)) * 8;
    o2[0] = pa_dma; o2[1] = pa_wx; o2[2] = pa_mma; o2[3] = pa_wy; o2[4] = gstep; o2[5] = tp[0]; o2[6] = tp[1]; o2[7] = tp[2];
  }
#endif
  if (TIMING && t == 0) {
    long long* o_ = TIMING + (size_t)blockIdx.x * 8;
    tm[5] = __builtin_amdgcn_s_memrealtime();
#pragma unroll
    for (int i = 0; i < 7; ++i) o_[i] = tm[i];
    o_[7] = (long long)(se - sb) | ((long long)(slabs_local ? 1 : 0) << 32) | ((long long)xcc_mine << 40) | ((long long)r << 48);      // (taps of the slice | exchange through L2 | XCC id + 1 | slice)
  }
}

// statistics of a tensor whose producer emitted none: block = `rows_per_block` rows x one slab of CW channels of one sample; thread =
// (row lane, channel octet), 4 rows in flight per thread, row lanes folded through LDS in a fixed order (two stages), one DmxStat add per
// channel and block - the integer atomics are what a launch of many small blocks pays for (4096 x 1280: 50 us with 344 blocks x 1280
// channels, 1.3 M atomics on 15 k addresses), so a block takes a narrow slab over many rows
template <int CW>
__global__ __launch_bounds__(512) void dmx_colstats_kernel(const bf16* x, int ldx, int HW, int C, long long* st, int rows_per_block) {
  constexpr int OC = CW / 8, RL = 512 / OC, P2 = RL >= 8 ? 8 : RL;       // row lanes, second-stage parts
  __shared__ float red[RL][CW][2];
  __shared__ float red2[P2][CW][2];
  const int b = blockIdx.z, c0 = blockIdx.y * CW, row0 = blockIdx.x * rows_per_block, row1 = min(row0 + rows_per_block, HW);
  const int t = threadIdx.x, o = t % OC, rl = t / OC;
  float s[8], q[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { s[e] = 0.f; q[e] = 0.f; }
  for (int row = row0 + rl; row < row1; row += 4 * RL) {
    u32x4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = *(const u32x4*)(x + ((size_t)b * HW + min(row + u * RL, row1 - 1)) * ldx + c0 + o * 8);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (row + u * RL >= row1) continue;
      float f[8]; unpack_bf8(v[u], f);
#pragma unroll
      for (int e = 0; e < 8; ++e) { s[e] += f[e]; q[e] += f[e] * f[e]; }
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) { red[rl][o * 8 + e][0] = s[e]; red[rl][o * 8 + e][1] = q[e]; }
  __syncthreads();
  if (t < P2 * CW) {
    const int c = t % CW, part = t / CW;
    float sa = 0.f, sq = 0.f;
    for (int k = part * (RL / P2); k < (part + 1) * (RL / P2); ++k) { sa += red[k][c][0]; sq += red[k][c][1]; }
    red2[part][c][0] = sa; red2[part][c][1] = sq;
  }
  __syncthreads();
  if (t < CW) {
    float sa = 0.f, sq = 0.f;
#pragma unroll
    for (int k = 0; k < P2; ++k) { sa += red2[k][t][0]; sq += red2[k][t][1]; }
    dmx_stat_add(st + ((size_t)b * C + c0 + t) * DMX_STAT_WORDS, sa, sq);
  }
}


// Plan: tile geometry, column width BN (160 / 128: 4 x 2 waves; 80 / 64: 8 x 1 waves) and K split.  A K split costs an exchange of fp32
// slabs through memory ((S - 1) / S x 256 x BN x 4 bytes per block, written and read back) and the co-residency of a tile's blocks, so the
// plan takes the narrow tiles where they make the split unnecessary or smaller; with tiles to spare the wide tile wins (twice the work per
// weight byte and per barrier).  Costs in us, fitted on scripts/attic/halo_probe.py.
int g_halo_peers = 1;
int g_halo_ws = 1;                                     // dmx_set_halo_ws: 0 = the planner never takes the warp-specialised instances (A/B aid)
struct HaloPlan { int TH, TW, nf, wmw, bn, splits, waves; };
HaloPlan halo_plan(const HaloConvArgs& a) {
  HaloPlan P{0, 0, 0, 0, 0, 0, 8};
  if (a.W % 32 == 0 && a.H % 8 == 0) { P.TW = 32; P.TH = 8; }
  else if (a.W % 16 == 0 && a.H % 16 == 0) { P.TW = 16; P.TH = 16; }
  else return P;
  const long tiles_m = (long)a.B * (a.H / P.TH) * (a.W / P.TW);
  const int T = 9 * (a.Cin / 64) + a.Csc / 64;
  double best = 1e300;
  const int cand[4][3] = {{5, 4, 160}, {4, 4, 128}, {5, 8, 80}, {4, 8, 64}};
  for (int c = 0; c < 4; ++c) {
    const int bn = cand[c][2];
    if (a.N % bn) continue;
    if (a.force_bn && a.force_bn != bn) continue;
    const long tiles = tiles_m * (a.N / bn);
    for (int s = 1; s <= 8; s *= 2) {
      if (a.force_split && s != a.force_split) continue;
      if (s > 1 && (!g_exclusive_device || tiles * s > n_cus() || T / s < 3)) continue;
      const double rounds = (double)((tiles * s + n_cus() - 1) / n_cus());
      const double step = bn >= 128 ? 1.45 : 0.96;                         // us per tap and block (measured, B = 4 64x64x320: K = 2880 / 5760 / 8640)
      double cost = rounds * ((double)((T + s - 1) / s) * step + 13.0);    // + prologue / epilogue of a block
      if (s > 1) cost += 3.5 + 2.0 * (double)(s - 1) * bn / 160.0;         // publish + wait + the peers' slabs
      if (cost < best) { best = cost; P.nf = cand[c][0]; P.wmw = cand[c][1]; P.bn = bn; P.splits = s; }
    }
  }
  // waves: 8 = two-group ping-pong, 4 = warp-specialised (4 compute + 4 loader waves; the wide tiles only).  (16-wave lock-step instances
  // of the wide tiles were built in round 4 and measured slower than the ping-pong - 68 / 95 / 130 vs 60 / 87 / 118 us - and removed.)
  // Measured (B = 4, stand-alone incl. GroupNorm, us; ping-pong / 4 + 4 / 8 + 4): 64x64x320 K = 2880 / 5760 / 8640 (two-way splits): 62.2 /
  // 85.3 / 113.1 - 61.0 / 81.7 / 107.0 - 62.6 / 84.2 / 111.1; 32x32x640 K = 5760 (four-way): 59.5 - 64.3 - 66.9; 16x16x1280 K = 11520
  // (eight-way): 58.8 - 70.0 - 77.6: the warp-specialised K loop is ~10 % faster per tap (1.02 vs 1.13 us), its epilogue with four / eight
  // slabs per row in flight has fewer registers to hide them in -> 4 + 4 up to two-way splits, the ping-pong beyond
  P.waves = 8;
#ifndef DMX_F16
  // (bf16 build only: in the fp16 build the compiler spills an accumulator of the 160-column instance INSIDE the tap loop, which the
  // asm MFMAs make a correctness bug - non-deterministic results at 768 px, EXPERIMENTS.md round 4 item 1b; tests/test_build_cpu.py
  // checks the ISA of both builds)
  if (P.bn >= 128 && P.splits <= 2 && g_halo_ws) P.waves = 4;
  if (P.bn >= 128 && (a.force_waves == 4 || a.force_waves == 12 || a.force_waves == 8)) P.waves = a.force_waves;
#else
  if (a.force_waves == 8) P.waves = 8;
#endif
  if (a.force_waves && a.force_waves != P.waves) P.splits = 0;
  return P;
}

}  // namespace

// what the model executors ask: the fused launch is at least as fast IN SITU as GroupNorm + the implicit-GEMM conv (+ its split-K reduce).
// Measured per shape inside the 50-step pass (B = 4; halo incl. GroupNorm vs conv + reduce + GroupNorm): 64x64 level 61 vs 69, 87 vs 105,
// 119 vs 129 us; 32x32 level 63 vs 66, 89 vs 98, 124 vs 125; 16x16 level 64 vs 61, 85 vs 77, 48 vs 44 - the blocks of the deep levels
// are 8-way K splits whose fp32 slab exchange costs what the fusion saves.
// dmx_set_halo_peers: 0 = round-5 dealing and write-through slabs everywhere (A/B and the tests' second opinion); 1 = a tile's K-split peers on one XCD
extern "C" int dmx_set_halo_peers(int on) { const int old = g_halo_peers; g_halo_peers = on ? 1 : 0; dmx_plan_switch(DMX_SW_HALO_PEERS, g_halo_peers); return old; }
extern "C" int dmx_set_halo_ws(int on) { const int old = g_halo_ws; g_halo_ws = on; dmx_plan_switch(DMX_SW_HALO_WS, on); return old; }
// ONE place for "could a fused GroupNorm -> conv launch consume statistics records of an H x W tensor": the tile geometries of halo_plan and the
// level rule of dmx_conv_halo_pays.  The executors ask this before they spend a statistics pass / a statistics epilogue on a tensor
// (Exec::ensure_stats, Exec::chain_stats); `everywhere` = dmx_set_halo_conv(2): wherever the kernel takes the problem.
bool dmx_conv_halo_wants_stats(int H, int W, bool everywhere) {
  const bool geometry = (W % 32 == 0 && H % 8 == 0) || (W % 16 == 0 && H % 16 == 0);
  return geometry && (everywhere || (long)H * W >= 1024);
}
bool dmx_conv_halo_pays(const HaloConvArgs& a) {
  if (!dmx_conv_halo_supported(a)) return false;
  return dmx_conv_halo_wants_stats(a.H, a.W, false);
}

static long halo_blocks(const HaloConvArgs& a, const HaloPlan& P);
// the block decode divides by multiplication (hb_div: magic = 2^32 / d + 1), exact only while x * d < 2^32 for every dividend x
// (block index, channel, patch row) and divisor d of the decode
static bool halo_decode_ok(const HaloConvArgs& a, const HaloPlan& P) {
  const long tiles_x = a.W / P.TW, tiles_img = tiles_x * (a.H / P.TH), tiles_m = (long)a.B * tiles_img, ncombo = (long)(a.N / P.bn) * P.splits;
  long dmax = tiles_m; for (long d : {ncombo, tiles_img, tiles_x, (long)P.TW + 2, (long)(a.gn ? a.Cin / a.groups : 1)}) if (d > dmax) dmax = d;
  long xmax = halo_blocks(a, P); for (long x : {(long)a.Cin, (long)(P.TH + 2) * (P.TW + 2) * 8}) if (x > xmax) xmax = x;
  return xmax < (1l << 31) && dmax < (1l << 31) && (unsigned long long)xmax * (unsigned long long)dmax < (1ull << 32);
}

bool dmx_conv_halo_supported(const HaloConvArgs& a) {
  if (a.Cin <= 0 || a.Cin % 64 || a.cx0 % 64 || a.cx0 > a.Cin || a.Csc % 64 || (a.Csc && a.cs0 % 64) || a.N % 8 || a.ldo % 8) return false;
  if (a.ldx0 % 8 || (a.x1 && a.ldx1 % 8) || a.ldw % 8 || (a.res && a.ldres % 8)) return false;
  if (a.gn && (a.groups != 32 || a.Cin % a.groups)) return false;
  if (a.force_split && (a.force_split & (a.force_split - 1) || a.force_split > 8)) return false;
  const HaloPlan P = halo_plan(a);
  return P.splits > 0 && halo_decode_ok(a, P);         // (larger problems: GroupNorm + the implicit GEMM, whose decode is 64-bit)
}

static long halo_blocks(const HaloConvArgs& a, const HaloPlan& P) {
  return (long)a.B * (a.H / P.TH) * (a.W / P.TW) * (a.N / P.bn) * P.splits;
}

// one completion flag per block + one XCC-id word per block (peers_local)
int dmx_conv_halo_flag_count(const HaloConvArgs& a) {
  const HaloPlan P = halo_plan(a);
  return P.splits > 1 ? 2 * (int)halo_blocks(a, P) : 0;
}

static size_t halo_flag_bytes(int n) { return align_up((size_t)2 * n * sizeof(int), 256); }

size_t dmx_conv_halo_workspace_bytes(const HaloConvArgs& a) {
  const HaloPlan P = halo_plan(a);
  if (P.splits <= 1) return 0;
  const long n = halo_blocks(a, P);
  return halo_flag_bytes((int)n) + (size_t)n * 256 * P.bn * sizeof(float);
}

template <int NF, int WMW, int WNW, bool WS = false> static int halo_launch_(const HaloConvArgs& a, int blocks, hipStream_t stream) {
  DMX_LDS_OPT_IN((dmx_conv_halo_kernel<NF, WMW, WNW, WS>), (HaloLds<NF, WMW, WNW, WS>::TOTAL));
  {
    char sym[112];
    snprintf(sym, sizeof(sym), "void (anonymous namespace)::dmx_conv_halo_kernel<%d, %d, %d, %s>(HaloConvArgs)", NF, WMW, WNW, WS ? "true" : "false");
    dmx_profile_note_symbol(sym);
  }
  hipLaunchKernelGGL((dmx_conv_halo_kernel<NF, WMW, WNW, WS>), dim3(blocks), dim3(HaloLds<NF, WMW, WNW, WS>::NT), (HaloLds<NF, WMW, WNW, WS>::TOTAL), stream, a);
  return dmx_check_launch("dmx_conv_halo_kernel");
}

int dmx_conv_halo_launch(HaloConvArgs a, void* workspace, size_t workspace_bytes, hipStream_t stream) {
  DMX_REQUIRE(a.x0 && a.w && a.out, "conv_halo: null argument");
  DMX_REQUIRE(dmx_conv_halo_supported(a), "conv_halo: unsupported problem (H=%d W=%d Cin=%d cx0=%d Csc=%d N=%d split=%d bn=%d)", a.H, a.W, a.Cin, a.cx0, a.Csc, a.N, a.force_split, a.force_bn);
  if (a.gn) DMX_REQUIRE(a.st0 && a.gamma && a.beta && (a.cx0 == a.Cin || a.st1), "conv_halo: the fused GroupNorm needs statistics records, gamma and beta");
  if (a.Csc) DMX_REQUIRE(a.s0 && (a.cs0 == a.Csc || a.s1), "conv_halo: the shortcut segment needs its source tensor(s)");
  DMX_REQUIRE(a.ldw >= 9 * a.Cin + a.Csc, "conv_halo: weight rows shorter than K");
  if (!a.x1) { a.x1 = a.x0; a.ldx1 = a.ldx0; }
  if (!a.s1) { a.s1 = a.s0; a.lds1 = a.lds0; }
  int rc = dmx_zero_page(&a.zeros);
  if (rc) return rc;
  a.err = dmx_dev_err_words();
  const HaloPlan P = halo_plan(a);
  a.TH = P.TH; a.TW = P.TW; a.splits = P.splits;
  const int blocks = (int)halo_blocks(a, P);
  a.tiles_x = a.W / P.TW; a.tiles_img = a.tiles_x * (a.H / P.TH); a.tiles_m = a.B * a.tiles_img; a.ncombo = (a.N / P.bn) * P.splits;
  a.cpg = a.gn ? a.Cin / a.groups : 1;
  DMX_REQUIRE(halo_decode_ok(a, P), "conv_halo: problem too large for the block decode (%d blocks, %d pixel tiles)", blocks, a.tiles_m);
  a.mg_tiles_x = halo_magic(a.tiles_x); a.mg_tiles_img = halo_magic(a.tiles_img); a.mg_tiles_m = halo_magic(a.tiles_m);
  a.mg_ncombo = halo_magic(a.ncombo); a.mg_pw = halo_magic(P.TW + 2); a.mg_cpg = halo_magic(a.cpg);
  // block -> XCD dealing: weights-major when the weight slab is the larger stream of an XCD, pixel-tile-major otherwise
  {
    const double wbytes = 2.0 * a.N * (9.0 * a.Cin + a.Csc), abytes = 2.0 * a.B * a.H * a.W * (double)(a.Cin + a.Csc) * 1.33;
    const int combos = (a.N / P.bn) * P.splits;
    // weights-major: every XCD reads its combos' weights once, the activations combos / 8 .. combos times; tile-major: the reverse
    a.xcd_tile_major = (abytes * (combos >= 8 ? combos / 8.0 : 1.0) + wbytes > abytes + wbytes * 8.0) ? 1 : 0;
  }
  // the S blocks of a tile on one XCD: r is the fastest index of the decode and an XCD's share of the grid (blocks / 8 consecutive Lb) holds whole tiles
  a.peers_local = (g_halo_peers && a.splits > 1 && blocks % 8 == 0 && (blocks / 8) % a.splits == 0) ? 1 : 0;
  if (a.splits > 1) {
    const size_t fb = halo_flag_bytes(blocks), need = fb + (size_t)blocks * 256 * P.bn * sizeof(float);
    if (!workspace || workspace_bytes < need) { dmx_set_error("conv_halo: the K split needs %zu bytes of workspace, got %zu", need, workspace_bytes); return DMX_ERR_WORKSPACE; }
    a.slabs = (float*)((char*)workspace + fb);
    if (!a.flags) { a.flags = (int*)workspace; if (const int zr = dmx_zero16_launch(a.flags, fb, stream)) return zr; }      // a kernel node, not a memset node: exec.hip Exec::zero_pool
  }
  const double flops = 2.0 * a.B * a.H * a.W * (double)a.N * (9.0 * a.Cin + a.Csc);
  const double bytes = 2.0 * ((double)a.B * a.H * a.W * (a.Cin + a.Csc + a.N) + (double)a.N * (9.0 * a.Cin + a.Csc));
  char tag[96];
  snprintf(tag, sizeof(tag), "M=%d N=%d K=%d halo gn=%d bn=%d sk=%d", a.B * a.H * a.W, a.N, 9 * a.Cin + a.Csc, a.gn, P.bn, a.splits);
  ProfScope ps(PROF_HALO, stream, flops, bytes, tag);
#ifndef DMX_F16
  if (P.waves == 4) return P.nf == 5 ? halo_launch_<10, 4, 1, true>(a, blocks, stream) : halo_launch_<8, 4, 1, true>(a, blocks, stream);
  if (P.waves == 12) return P.nf == 5 ? halo_launch_<5, 4, 2, true>(a, blocks, stream) : halo_launch_<4, 4, 2, true>(a, blocks, stream);
#endif
  if (P.nf == 5 && P.wmw == 4) return halo_launch_<5, 4, 2>(a, blocks, stream);
  if (P.nf == 4 && P.wmw == 4) return halo_launch_<4, 4, 2>(a, blocks, stream);
  if (P.nf == 5 && P.wmw == 8) return halo_launch_<5, 8, 1>(a, blocks, stream);
  return halo_launch_<4, 8, 1>(a, blocks, stream);
}

int dmx_colstats_launch(const bf16* x, int ldx, int B, int HW, int C, long long* st, hipStream_t stream) {
  DMX_REQUIRE(x && st && C % 8 == 0 && ldx % 8 == 0 && C >= 8, "colstats: C and ld must be multiples of 8");
  const int cw = C % 64 == 0 ? 64 : (C % 32 == 0 ? 32 : 8), RL = 512 / (cw / 8);
  int rpb = 4 * RL;                                    // >= 4 rows per thread; fewer, larger blocks once the chip is covered twice
  while ((long)B * (C / cw) * cdiv(HW, rpb) > 2 * n_cus() && rpb < HW) rpb *= 2;
  char tag[96]; snprintf(tag, sizeof(tag), "rows=%d C=%d colstats", B * HW, C);
  ProfScope ps(PROF_GNORM, stream, 0.0, 2.0 * (double)B * HW * C, tag);
  const dim3 grid(cdiv(HW, rpb), C / cw, B);
  if (cw == 64) hipLaunchKernelGGL(dmx_colstats_kernel<64>, grid, dim3(512), 0, stream, x, ldx, HW, C, st, rpb);
  else if (cw == 32) hipLaunchKernelGGL(dmx_colstats_kernel<32>, grid, dim3(512), 0, stream, x, ldx, HW, C, st, rpb);
  else hipLaunchKernelGGL(dmx_colstats_kernel<8>, grid, dim3(512), 0, stream, x, ldx, HW, C, st, rpb);
  return dmx_check_launch("dmx_colstats_kernel");
}
