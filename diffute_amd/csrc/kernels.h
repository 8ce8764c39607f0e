// Internal (C++) launch interfaces shared by the C-ABI wrappers and the model executors.
#pragma once
#include <vector>
#include "common.h"

// ------------------------------------------------------------------ profiler (exec.hip)
// Optional per-kernel-class timing with hipEvents on the launch stream (bench.py roofline leg).
enum ProfClass { PROF_GEMM128 = 0, PROF_GEMM64 = 1, PROF_SPLITK = 2, PROF_ATTN = 3, PROF_GNORM = 4, PROF_LNORM = 5,
                 PROF_OTHER = 6, PROF_GEMM256 = 7, PROF_WGRAD = 8, PROF_GEMM256WS = 9,
                 PROF_GEMM_CFG0 = 10 /* one class per GEMM tile config (template instance): 10 + plan id */, PROF_XFCHAIN = 26 /* xf_chain.hip */, PROF_HALO = 27 /* conv_halo.hip */, PROF_SKINNY = 28 /* skinny.hip */, PROF_NCLASS = 29 };
int n_cus();                                           // compute units of the CURRENT device (conv_halo.hip)
// exec.hip: the plan signature (key of captured graphs / cached workspace sizes): every plan-changing switch records its value
enum DmxPlanSwitch { DMX_SW_EXCLUSIVE = 0, DMX_SW_GN_STATS, DMX_SW_DEFER, DMX_SW_HALO, DMX_SW_PREFETCH, DMX_SW_XF_CHAIN, DMX_SW_HALO_WS, DMX_SW_OVERRIDES, DMX_SW_SKINNY, DMX_SW_HALO_PEERS, DMX_SW_ATTN_BALANCED, DMX_SW_COUNT };
void dmx_plan_switch(int slot, int value);
void dmx_plan_epoch_bump();                            // (dmx_gemm_plan_override: a counter)
extern "C" int dmx_plan_epoch(void);
int dmx_exclusive_device();                            // dmx_set_exclusive_device (conv_halo.hip): 0 = plans that need co-resident blocks are off
void dmx_profile_note_symbol(const char* sym);         // kernel symbol of the launch inside the innermost open ProfScope (exec.hip)
struct ProfScope {
  ProfScope(ProfClass c, hipStream_t s, double flops, double bytes, const char* tag = nullptr);
  ~ProfScope();
  int slot = -1;
  hipStream_t stream_ = nullptr;
};

// ------------------------------------------------------------------ gemm.hip
struct GemmArgs {
  // activation operand (rows m): up to two sources split on the channel axis
  const bf16* x0; const bf16* x1;
  int ldx0, ldx1;          // row strides, elements
  int cx0;                 // channels [0,cx0) of a tap come from x0, [cx0,Cin) from x1
  int direct;              // 1: row m of X is row m of x0/x1 (linear / 1x1 s1); 0: conv gather
  int IH, IW, OH, OW;      // source grid (before the optional x2 nearest upsample), output grid
  int stride, pad, ups, ksize;
  // ups2 = 1: nearest-x2 upsample + 3x3 conv computed as four 2x2 convolutions on the SOURCE grid, one per output pixel
  // parity (pa, pb), with pre-summed weights (dmx_ups_phase_weights_launch): 4*Cin instead of 9*Cin multiply-adds per
  // output.  ksize = 2, OH/OW = IH/IW, M4 = B*IH*IW gathered rows per phase, M = 4*M4 output rows, K = Ktaps = 4*Cin,
  // w = [4 phases][N][ldw], w_phase_stride = elements between phases; bias only.
  int ups2, M4; long long w_phase_stride;
  int Cin;                 // channels per tap
  int Ktaps;               // ksize*ksize*Cin ; K - Ktaps = shortcut channels
  const bf16* s0; const bf16* s1; int lds0, lds1, cs0;   // fused 1x1 shortcut segment
  const bf16* w; int ldw;  // weights [N][K]
  int M, N, K;
  const float* bias;       // [N] fp32 or null
  const float* rowbias;    // [M/rows_per_group][ldrb] fp32 or null (time-embedding projection)
  int rows_per_group, ldrb;
  const bf16* res; int ldres;   // residual or null
  void* out; int ldo; int out_f32; int geglu;
  int act;                                     // 1: exact (erf) GELU after the bias, before the residual (ViT MLP); bf16 coalesced epilogue only
  int force_tn, force_splitk;                  // 0 = automatic
  int group_m;                                 // m-tiles per rasterisation super-tile (0 = default 8)
  // folded LayerNorm: a producer GEMM emits per-row partial (sum, sumsq) of its rounded output, the consumer GEMM
  // multiplies the RAW rows by W' = W*diag(gamma) and finishes y = rstd*(acc - mean*c1[n]) + c2[n] in its epilogue
  float* rowstats_out;                         // producer: [tiles_n][M][2] or null
  const float* ln_stats; int ln_tiles;         // consumer: the producer's partials and how many n-tiles it had
  const float* ln_c1; const float* ln_c2;      // [N]: c1 = sum_k W'[n][k] ; c2 = sum_k beta[k]*W[n][k] (+ bias[n])
  int ln_C; float ln_eps;                      // normalised feature count (= K) and epsilon
  long long* timing;                           // optional per-block timeline (probe builds), normally null
  int dbg;                                     // measurement aid: bit0 skip the MFMA phase, bit1 skip the DMA refills (results invalid)
  // optional: up to two byte ranges (the weights of the launches that run NEXT) the blocks touch at their start - LDS-DMA into a dump
  // slot behind the instance's LDS (pf_dump_off, filled by the launcher; prefetch dropped where the slot does not fit)
  const void* pf[2]; int pf_bytes[2]; int pf_dump_off;
  float* partial; int splitk, kt_per_split;   // filled by the launcher
  // persistent stream-K launch (filled by the launcher): grid = one block per CU walking (tile, K-range) items; `partial` holds
  // one accumulator slab per block, `flags` one int per block, all zero at launch (see the kernel's work-item loop)
  int defer_reduce;        // split-K plans: 1 = the launcher skips the reduce pass (Exec defers it into the GroupNorm that reads the output: exec.hip PendRed)
  int persist; int* flags; int* err;      // err: dmx_dev_err_words() (set by the launcher)
  // GroupNorm statistics of the output for its consumer (norm.hip dmx_groupnorm_sums_launch, conv_halo.hip): per (sample, channel) a
  // DmxStat record (common.h; 4 x int64 fixed point) of the rounded outputs, ADDED into colstats[(sample*N + n)*4 ..] (zero before the launch);
  // cs_rows = rows per sample (a tile must not straddle samples: cs_rows % tile rows == 0).  Plans without a reduce pass only
  // (dmx_gemm_colstats_ok); bf16 coalesced epilogues.
  long long* colstats; int cs_rows;
  const bf16* zeros;                           // filled by the launcher
};
int dmx_gemm_launch(GemmArgs a, void* workspace, size_t workspace_bytes, hipStream_t stream);
size_t dmx_gemm_workspace_bytes(const GemmArgs& a);
void dmx_gemm_plan(const GemmArgs& a, int* tn, int* splitk, int* ktps);
int dmx_zero_page(const bf16** out);
void dmx_gemm_plan_override_set(int M, int N, int K, int st, int ups, int cfg, int sk);   // tuning aid; cfg < 0 clears all
bool dmx_gemm_colstats_ok(const GemmArgs& a);    // the plan this problem runs on can emit GemmArgs.colstats (a.cs_rows set)
int dmx_gemm_persist_blocks(const GemmArgs& a);   // grid of the persistent stream-K plan this problem will run on, 0 for a classic plan
int dmx_gemm_tiles_n(const GemmArgs& a);       // n-tiles of the plan that dmx_gemm_launch will pick (rowstats_out sizing)
// W' = bf16(W*gamma) and the c1 / c2 vectors of the folded LayerNorm, from the raw bf16 weights (rows may be GEGLU-packed)
// [4][N][4*Cin] phase weights of GemmArgs.ups2 from taps-major 3x3 weights [N][ldw3]
int dmx_ups_phase_weights_launch(const bf16* w3, int ldw3, bf16* wp, int N, int Cin, hipStream_t stream);
int dmx_ln_fold_launch(const bf16* w_raw, bf16* w_out, const float* gamma, const float* beta, const float* bias,
                       float* c1, float* c2, int N, int K, hipStream_t stream);

// ------------------------------------------------------------------ conv_halo.hip
// 3x3 stride-1 pad-1 convolution with a HALO input tile staged once per 64-channel chunk in LDS (the nine taps are shifted LDS
// fragment reads; only the weight tiles stream) and the preceding GroupNorm(+SiLU) applied to the staged tile in place, from the
// per-(sample, channel) statistics its producer(s) emitted (DmxStat below).  ResnetBlock2D = [GN -> SiLU -> conv3x3] x 2 becomes two launches.
struct HaloConvArgs {
  const bf16* x0; const bf16* x1; int ldx0, ldx1;   // NHWC input, two-source channel concat (x1 may be null)
  int cx0, Cin;                                      // channels [0, cx0) from x0, [cx0, Cin) from x1; both multiples of 64
  int B, H, W;                                       // input grid = output grid
  int gn, silu, groups; float eps;                   // gn = 1: y = GroupNorm(x)[SiLU] is what the conv sees (zero padding applies to y)
  const long long* st0; const long long* st1;        // DmxStat records [B][channels of x0 | x1][4]
  const float* gamma; const float* beta;             // [Cin]
  const bf16* s0; const bf16* s1; int lds0, lds1, cs0, Csc;   // fused 1x1 shortcut K segment on RAW tensors of the output grid (Csc = 0: none)
  const bf16* w; int ldw;                            // [N][9*Cin + Csc]: k = tap*Cin + c, then the shortcut channels
  int N;                                             // multiple of 160 or of 128
  const float* bias; const float* rowbias; int ldrb; // rowbias[b*ldrb + n] (time-embedding projection) or null
  const bf16* res; int ldres;                        // residual (output grid) or null
  bf16* out; int ldo;
  long long* colstats;                               // DmxStat records of the OUTPUT [B][N][4], added to (zero before the launch), or null
  int force_split;                                   // 0 = automatic K split (1 / 2 / 4 / 8 blocks per tile)
  int force_bn;                                      // 0 = automatic column tile (160 / 128 / 80 / 64)
  int force_waves;                                   // 0 = automatic; 8 = two-group ping-pong; 4 / 12 = warp-specialised, 4 compute + 4 / 8 loader waves (160 / 128-column tiles, bf16 build)
  int dbg; long long* timing;                        // measurement aids (0 / null in the product path)
  const void* pf[2]; int pf_bytes[2];                // optional: byte ranges (the next launches' weights) the blocks touch at their start (Exec::peek)
  // filled by the launcher
  int TH, TW, splits, xcd_tile_major; float* slabs; int* flags; const bf16* zeros; int* err;      // err: dmx_dev_err_words() (set by the launcher)
  int peers_local;          // launcher: 1 = the S blocks of a tile are dealt to ONE XCD (r fastest in the block decode) and may exchange their slabs through that
                            // XCD's L2 (plain stores) once every peer has confirmed its XCC id (words flags[blocks ..)); 0 = round-5 dealing, write-through always
  // (the block decode of the kernel without integer divisions: x / d = (x * magic) >> 32 for the dividends that occur, all < 2^20)
  int tiles_x, tiles_img, tiles_m, ncombo, cpg;
  unsigned mg_tiles_x, mg_tiles_img, mg_tiles_m, mg_ncombo, mg_pw, mg_cpg;
};
bool dmx_conv_halo_supported(const HaloConvArgs& a);
bool dmx_conv_halo_pays(const HaloConvArgs& a);        // supported AND at least as fast in situ as GroupNorm + implicit-GEMM conv (what the executors ask)
size_t dmx_conv_halo_workspace_bytes(const HaloConvArgs& a);
bool dmx_conv_halo_wants_stats(int H, int W, bool everywhere);   // geometry + level rule shared by the planner and the executors
int dmx_conv_halo_flag_count(const HaloConvArgs& a);   // ints of zeroed flags the launch needs (0: none)
int dmx_conv_halo_launch(HaloConvArgs a, void* workspace, size_t workspace_bytes, hipStream_t stream);
// statistics of a tensor nobody emitted them for: one streaming pass, DmxStat records [B][C][4] (zero before the launch)
int dmx_colstats_launch(const bf16* x, int ldx, int B, int HW, int C, long long* st, hipStream_t stream);


// ------------------------------------------------------------------ skinny.hip (weight-streaming conv / linear for M = B H W <= 256 rows)
constexpr int SK_MAX_CHUNKS = 24;          // staged chunks (<= 4 k-steps of 16 channels each) per K slice
constexpr int SK_COEF_CH = 448;            // GroupNorm'ed channels per K slice (the slice's (a, s) table in LDS)
struct SkinnySeg {                         // one K segment: a source tensor seen through `taps` filter taps
  const bf16* x; int ld;                   // NHWC rows = the pixels of the output grid (3x3 stride 1 pad 1, or 1x1)
  int C;                                   // channels, multiple of 16
  int taps;                                // 9 (3x3, pad 1) or 1
  // GroupNorm (+ SiLU, SkinnyArgs.silu) applied while the chunk is staged; null st: the tensor is used as it is
  const long long* st;                     // DmxStat records [B][C][4] of THIS tensor
  const float* gamma; const float* beta;   // of this tensor's channels (pointers already offset by gn_c0)
  int gn_c0;                               // first channel of this tensor inside the normalised (concatenated) tensor
};
struct SkinnyChunk { unsigned char seg, nks; unsigned short k0; int frag; unsigned short coef0, pad_; };
struct SkinnyPlanSlice {
  int nchunk;
  int g_count;                             // (sample-independent) GroupNorm group slots of this slice, <= 16: slot -> group in slot_group
  unsigned char slot_group[16];
  unsigned char g_first[4], slot0[4];      // per segment: first group its piece of the slice touches, and that group's slot
  SkinnyChunk ch[SK_MAX_CHUNKS];
};
struct SkinnyArgs {
  SkinnySeg seg[4]; int nseg;
  int B, H, W;                             // M = B H W in {64, 128, 256}; H, W powers of two; B <= 4
  int gn_groups, gn_Ctot; float gn_eps; int silu;   // GroupNorm over the concatenation of the segments that carry statistics (gn_groups = 0: none)
  const bf16* wp;                          // weights in fragment order (dmx_skinny_pack_launch): [N / 32][frags_per_nb][64 lanes][8]
  int N;                                   // multiple of 32
  const float* bias; const float* rowbias; int ldrb;     // rowbias[b * ldrb + n] (time-embedding projection) or null
  const bf16* res; int ldres;              // residual or null
  bf16* out; int ldo;
  long long* colstats;                     // DmxStat records of the OUTPUT [B][N][4], added to (zero before the launch), or null
  int force_S;                             // 0 = automatic number of K slices (tests)
  long long* timing;                       // measurement aid: [blocks][6] phase stamps (null in the product path)
  int dbg;                                 // measurement aid: bit 0 no MFMAs, bit 1 no weight refills, bit 2 no LDS fragment reads (results invalid)
  // filled by the launcher
  int S, frags_per_nb, gn_cpg, wsh, hwsh; float* slabs; int* flags; int* err; const bf16* zeros;
  SkinnyPlanSlice plan[8];
};
struct SkinnyPackDesc { int nseg, frags_per_nb; int frag0[4], taps[4], tap_stride[4], koff[4]; };    // source column of (segment, tap, channel c): tap * tap_stride + koff + c
bool dmx_skinny_enabled();
bool dmx_skinny_supported(const SkinnyArgs& a);
int dmx_skinny_frags_per_nb(const SkinnyArgs& a);
int dmx_skinny_flag_count(const SkinnyArgs& a);
size_t dmx_skinny_workspace_bytes(const SkinnyArgs& a);
int dmx_skinny_launch(SkinnyArgs a, void* workspace, size_t workspace_bytes, hipStream_t stream);
int dmx_skinny_pack_launch(const bf16* w, int ldw, bf16* wp, int N, const SkinnyPackDesc& d, hipStream_t stream);

// ------------------------------------------------------------------ wgrad.hip (training)
struct WgradArgs {
  const bf16* dy; int lddy;      // output gradient [M][N] (NHWC rows = output pixels)
  // the forward GEMM's activation operand, same gather semantics as GemmArgs (two sources, conv taps, upsample, stride)
  const bf16* x0; const bf16* x1; int ldx0, ldx1, cx0;
  int direct;
  int IH, IW, OH, OW, stride, pad, ups, ksize, Cin;
  int M, N, K;                   // K = ksize*ksize*Cin (conv) or the plain feature count (direct)
  float* out; int ldout;         // dW [N][ldout] fp32 (ldout 0 = K), packed k order (tap-major, channel-minor)
  int accumulate;                // out += instead of out =
  int splits, rows_per_split;    // filled by the launcher
  const bf16* zeros;             // filled by the launcher
};
size_t dmx_wgrad_workspace_bytes(const WgradArgs& a);
int dmx_wgrad_launch(WgradArgs a, void* workspace, size_t workspace_bytes, hipStream_t stream);
size_t dmx_colsum_ws_bytes(int groups, int rows_per_group, int N);
int dmx_colsum_launch(const bf16* dy, int lddy, int groups, int rows_per_group, int N, float* out, int ldo, int accumulate,
                      void* workspace, size_t workspace_bytes, hipStream_t stream);

// ------------------------------------------------------------------ norm.hip
struct GroupNormArgs {
  const bf16* x0; const bf16* x1; int ldx0, ldx1; int c0;   // two-source concat on channels
  int C, groups; int B, HW;
  const float* gamma; const float* beta; float eps; int silu;
  bf16* y; int ldy;
  float* partial;     // [B][nchunk][groups][2]
  float* stats_out;   // optional [B][groups][2] = (mean, rstd), kept for the backward pass (training)
  float* coef;        // [B][C][2] (filled by the launcher: lives behind partial in the workspace)
  // statistics from the producers of x0 / x1 (GemmArgs.colstats of the GEMM that wrote each tensor): [B][channels of that
  // tensor][4] DmxStat records (common.h); dmx_groupnorm_sums_launch only
  const long long* st0; const long long* st1;
  int nchunk, rows_per_chunk;
  // x0 = the NOT YET WRITTEN output of a split-K GEMM (slab path only): the slab load sums the partial planes in split order, adds bias / row bias /
  // residual exactly like dmx_splitk_reduce_kernel, writes x0 (its other consumers read it later) and keeps the rounded values for the statistics
  const float* red_partial; int red_splitk; long long red_mn; const float* red_bias; const float* red_rowbias; int red_ldrb, red_rpg;
  const bf16* red_res; int red_ldres;
};
bool dmx_gn_red_ok(GroupNormArgs a);          // the slab path can take this shape WITH the fused split-K reduce
size_t dmx_gn_workspace_bytes(int B, int HW, int groups);
int dmx_groupnorm_launch(GroupNormArgs a, hipStream_t stream);
int dmx_groupnorm_sums_launch(GroupNormArgs a, hipStream_t stream);   // statistics from the producers (a.st0 / a.st1): one apply-only launch
bool dmx_gn_single_launch(GroupNormArgs a);   // the register-resident one-launch path applies to this shape
int dmx_splitk_reduce_launch(const GemmArgs& a, hipStream_t stream);   // gemm.hip: the deferred reduce pass on its own
int dmx_layernorm_launch(const bf16* x, int ldx, bf16* y, int ldy, const float* gamma, const float* beta,
                         int rows, int C, float eps, hipStream_t stream);

// ------------------------------------------------------------------ norm_bwd.hip (training)
struct GroupNormBwdArgs {
  const bf16* x0; const bf16* x1; int ldx0, ldx1; int c0;   // forward input (two-source concat on channels)
  int C, groups; int B, HW;
  const float* gamma; const float* beta; int silu;
  const float* stats;            // [B][groups][2] = (mean, rstd) saved by the forward
  const bf16* dy; int lddy;
  bf16* dx0; int lddx0; bf16* dx1; int lddx1;               // input gradient, split like the input
  const bf16* res0; int ldres0; const bf16* res1; int ldres1;   // optional gradients to add (other consumers of x)
  float* dgamma; float* dbeta; int accumulate;
  float* part;                   // workspace, dmx_gn_bwd_workspace_bytes
  float* ab; int nchunk, rows_per_chunk;                    // filled by the launcher
};
size_t dmx_gn_bwd_workspace_bytes(int B, int HW, int C);
int dmx_groupnorm_bwd_launch(GroupNormBwdArgs a, hipStream_t stream);
size_t dmx_ln_bwd_workspace_bytes(int rows, int C);
int dmx_layernorm_bwd_launch(const bf16* x, int ldx, const bf16* dy, int lddy, const float* gamma, bf16* dx, int lddx,
                             const bf16* res, int ldres, float* dgamma, float* dbeta, int accumulate,
                             int rows, int C, float eps, void* workspace, size_t workspace_bytes, hipStream_t stream);
// packed = 1: h columns in the GEGLU weight-pack order (32 value columns, 32 gate columns, alternating)
int dmx_geglu_fwd_launch(const bf16* h, int ldh, bf16* y, int ldy, int rows, int C2, int packed, hipStream_t stream);
int dmx_geglu_bwd_launch(const bf16* h, int ldh, const bf16* dy, int lddy, bf16* dh, int lddh, int rows, int C2, int packed, hipStream_t stream);

int dmx_softmax_rows_launch(const float* s, int lds_, bf16* p, int ldp, int rows, int n, float scale, hipStream_t stream);

// ------------------------------------------------------------------ xf_chain.hip (row-local chains of the transformer block, C = 320)
struct XfChainArgs {
  int M, C;                              // rows (multiple of 64), channels (320)
  const bf16* x; int ldx;                // operand of the first GEMM: mode 0 attn1's output, mode 1 attn2's output
  const bf16* res; int ldres;            // residual of the first GEMM (the stream before that attention)
  const bf16* w0; const float* b0;       // to_out.0 of that attention [C][C] + bias
  bf16* h_out; int ldh;                  // the first GEMM's output (mode 0: the residual stream for mode 1; mode 1: a scratch copy the kernel reads back)
  const bf16* w1;                        // mode 0: attn2.to_q with norm2's gamma folded in [C][C]
  const float* c1; const float* c2;      // folded-LayerNorm vectors of the GEMM that consumes h: mode 0 to_q's [C], mode 1 FF1's [8C] (packed GEGLU order)
  bf16* y; int ldy;                      // mode 0: q2; mode 1: the block's output
  const bf16* wf1;                       // mode 1: ff.net.0.proj folded with norm3, packed GEGLU groups [8C][C]
  const bf16* wf2; const float* bf2;     //         ff.net.2 [C][4C] + bias
  const bf16* wpo; const float* bpo;     //         proj_out [C][C] + bias
  const bf16* xres; int ldxres;          //         the Transformer2DModel residual
  float eps;
  int dbg; long long* timing;            // measurement aids (0 / null in the product path): bit 0 no MFMA phase, bit 1 no DMA refills; [blocks][8] phase timestamps
  // mode 1: DmxStat records of the block output y for the GroupNorm that reads it next ([M / cs_rows][C][4], added to; cs_rows = rows per
  // sample, a multiple of 64), or null
  long long* colstats; int cs_rows;
  const void* pf[4]; int pf_bytes[4];      // optional: byte ranges (the next launches' weights) the blocks touch at their start (Exec::peek)
  // mode 2: x is the RAW tensor and the block normalises its rows on the way into the fragments: GroupNorm (no activation) from the DmxStat
  // records of x ([M / gn_rows][C][4]; gn_rows = rows per sample, a multiple of 64) - the same arithmetic, to the bit, as dmx_groupnorm's
  // apply pass followed by the plain mode 2.  Null gn_st: x is already normalised
  const long long* gn_st; const float* gn_gamma; const float* gn_beta; int gn_groups; int gn_rows; float gn_eps;
};
bool dmx_xf_chain_supported(int M, int C);
bool dmx_xf_chain_pays(int M, int C);      // what the model executors use: enough 64-row blocks to fill the chip
int dmx_xf_chain_launch(const XfChainArgs& a, int mode, hipStream_t stream);

// ------------------------------------------------------------------ attention.hip
struct AttnArgs {
  const bf16* q; int ldq;        // row (b*Sq + s), head h at column h*D
  const bf16* k; int ldk;        // row (b*kv_rows + s)
  int kv_rows;                   // rows per batch in k (>= Skv; Skv unless padded)
  const bf16* vt; int ldvt;      // V^T: row (h*D + d), column (b*skv_stride + s)   (used when v == nullptr)
  const bf16* v; int ldv;        // row-major V: row (b*kv_rows + s), head h at column h*D (LDS transpose-read path)
  int skv_stride;                // column offset between batches in vt
  bf16* o; int ldo;
  int B, H, Sq, Skv;
  float scale;
  float* lse;                    // optional [B][H][Sq]: log2-sum-exp of the scaled scores (kept for the backward pass)
  // optional: up to four byte ranges (the weights of the kernel that runs NEXT) every block touches a slice of at its start, so that they
  // sit in the memory-side cache when that kernel's blocks - which walk them in lock step - ask for them (row-major-V path only)
  const void* pf[4]; int pf_bytes[4];
  // balanced schedule (attention_sk.hip): partial (O, m, l) records of the helper parts [slots][36 KB], one zeroed flag per slot, the device-error words
  float* sk_part; int* sk_flags; int* err;
};
int dmx_attention_launch(const AttnArgs& a, hipStream_t stream);
// attention_sk.hip: stream-K over (query block, key tile) items on 3 x CUs block slots; slots = 0 where the plain grid is the plan
int dmx_attention_balanced_slots(const AttnArgs& a);
size_t dmx_attention_balanced_part_bytes(const AttnArgs& a);
int dmx_attention_balanced_launch(AttnArgs a, hipStream_t stream);

// ------------------------------------------------------------------ attention_wide.hip (single head, d = 128 / 256 / 512: the VAE mid block)
struct AttnWideArgs {
  const bf16* q; int ldq;        // row (b*Sq + s), d contiguous
  const bf16* k; int ldk;        // row (b*kv_rows + s)
  const bf16* v; int ldv;        // row-major V, row (b*kv_rows + s)
  int kv_rows;
  bf16* o; int ldo;
  int B, Sq, Skv, D;
  float scale;
  const bf16* zeros;             // filled by the launcher
};
bool dmx_attention_wide_supported(int D);
int dmx_attention_wide_launch(AttnWideArgs a, hipStream_t stream);

// ------------------------------------------------------------------ attention_bwd.hip (training)
struct AttnBwdArgs {
  const bf16* q; int ldq; const bf16* k; int ldk; const bf16* v; int ldv; int kv_rows;   // as the forward (row-major V)
  const bf16* o; const bf16* dout; int ldo;      // forward output and its gradient, same layout
  const float* lse;                              // [B][H][Sq] from the forward
  float* delta;                                  // workspace [B][H][Sq]: rowsum(dO * O)
  bf16* dq; int lddq; bf16* dk; int lddk; bf16* dv; int lddv;
  int B, H, Sq, Skv; float scale;
};
size_t dmx_attn_bwd_ws_bytes(int B, int H, int Sq);
int dmx_attention_bwd_launch(const AttnBwdArgs& a, hipStream_t stream);

// ------------------------------------------------------------------ ref_f32.hip (fp32 validation instantiation; tests only)
struct GemmF32Args {             // same gather semantics as GemmArgs, float operands, 3x3 taps never phase-decomposed
  const float* x0; const float* x1; int ldx0, ldx1, cx0, direct;
  int IH, IW, OH, OW, stride, pad, ups, ksize, Cin, Ktaps;
  const float* s0; const float* s1; int lds0, lds1, cs0;
  const float* w; int ldw; int M, N, K;
  const float* bias; const float* bias2; const float* rowbias; int rows_per_group, ldrb;
  const float* res; int ldres;
  float* out; int ldo; int geglu, act;
};
int dmx_gemm_f32_launch(const GemmF32Args& a, hipStream_t stream);
int dmx_groupnorm_f32_launch(const float* x0, int ldx0, int c0, const float* x1, int ldx1, int C, int groups, int B, int HW,
                             const float* gamma, const float* beta, float eps, int silu, float* y, int ldy, hipStream_t stream);
int dmx_layernorm_f32_launch(const float* x, int ldx, float* y, int ldy, const float* gamma, const float* beta, int rows, int C, float eps, hipStream_t stream);
int dmx_attention_f32_launch(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, int kv_rows, float* o, int ldo,
                             int B, int H, int Sq, int Skv, float scale, hipStream_t stream, int head_dim = 64);
int dmx_silu_f32_launch(float* x, size_t n, hipStream_t stream);
int dmx_concat_nchw_to_nhwc_f32_launch(const float* f0, int c0, const float* f1, int c1, const float* f2, int c2, float* out, int B, int HW, hipStream_t stream);

// ------------------------------------------------------------------ elementwise.hip
struct Im2colArgs {
  // sources: up to 3 NCHW fp32 tensors concatenated on channels, or one NHWC bf16 tensor
  const float* f0; const float* f1; const float* f2; int c0, c1, c2;
  const bf16* h; int ldh;
  int C;                    // total channels
  int B, IH, IW, OH, OW, ksize, stride, pad;
  bf16* out; int Kpad;      // [B*OH*OW][Kpad], k = tap*C + c, zero padded
};
int dmx_im2col_small_launch(const Im2colArgs& a, hipStream_t stream);
int dmx_nhwc_to_nchw_f32_launch(const float* in, int ldin, float* out, int B, int C, int HW, hipStream_t stream);
int dmx_nhwc_bf16_to_nchw_f32_launch(const bf16* in, int ldin, float* out, int B, int C, int HW, hipStream_t stream);
int dmx_nchw_f32_to_nhwc_bf16_launch(const float* in, bf16* out, int ldo, int B, int C, int HW, hipStream_t stream);
int dmx_cast_f32_to_bf16_launch(const float* in, bf16* out, size_t n, hipStream_t stream);
int dmx_pack_conv_weight_launch(const float* w, bf16* out, int Cout, int Cin, int ks, int ldk, int koff, hipStream_t stream);
int dmx_pack_rows_launch(const float* w, bf16* out, int rows, int cols, int ldo, int geglu, hipStream_t stream);
int dmx_pack_geglu_bias_launch(const float* b, float* out, int n, hipStream_t stream);
// training: transposed packs for data-gradient GEMMs, stride-2 / upsample adjoints
int dmx_pack_conv_weight_t_launch(const float* w, bf16* out, int Cout, int Cin, int ks, int ldk, int koff, hipStream_t stream);
int dmx_pack_rows_t_launch(const float* w, bf16* out, int rows, int cols, int ldo, hipStream_t stream);
int dmx_zero16_launch(void* p, size_t bytes, hipStream_t stream);
int dmx_zero_insert2_launch(const bf16* dy, int lddy, bf16* z, int B, int OH, int OW, int C, hipStream_t stream);
int dmx_sumpool2_launch(const void* du, int lddu, int du_f32, bf16* dx, int lddx, int B, int H, int W, int C, int accumulate, hipStream_t stream);
int dmx_cast_pad_rows_launch(const void* in, int in_is_bf16, bf16* out, int B, int S, int Spad, int C, hipStream_t stream);

// ------------------------------------------------------------------ train_small.hip (training)
int dmx_transpose_bf16_launch(const bf16* in, int ldin, bf16* out, int ldout, int R, int C, hipStream_t stream);
// Batched transposes (the W^T pass after every optimizer step is ~600 small transposes: one launch instead of 600).
// While a TrBatch is recording on this thread, dmx_transpose_bf16_launch only appends a job; run() uploads the job table to
// `table_dev` when it differs from `cache` (the previous call's table: steady state uploads nothing) and launches once.
struct TrJob { const unsigned short* in; unsigned short* out; int ldin, ldout, R, C, tiles_x, first; };
struct TrBatch {
  std::vector<TrJob> jobs; int tiles = 0;
  TrBatch(); ~TrBatch();
  int run(void* table_dev, size_t table_bytes, std::vector<TrJob>& cache, hipStream_t stream);
};
constexpr size_t DMX_TR_TABLE_BYTES = 2048 * sizeof(TrJob);
int dmx_add_bf16_launch(const bf16* a, int lda, const bf16* b, int ldb, bf16* o, int ldo, int rows, int C, hipStream_t stream);
size_t dmx_mse_workspace_bytes();
int dmx_mse_loss_launch(const float* pred, const float* target, size_t n, float* loss, float* dpred, float grad_scale,
                        void* workspace, size_t workspace_bytes, hipStream_t stream);
// 1x1 convs between <= 8 channels (VAE quant / post_quant), softmax backward over rows (VAE mid attention)
int dmx_pointwise_small_fwd_launch(const bf16* x, int ldx, const bf16* w, int ldw, const float* bias, void* y, int ldy, int M, int Cin, int Cout,
                                   int out_f32, hipStream_t stream);
size_t dmx_pointwise_small_bwd_ws_bytes(int M, int Cin, int Cout);
int dmx_pointwise_small_bwd_launch(const bf16* x, int ldx, const float* dy, int lddy, const bf16* w, int ldw, bf16* dx, int lddx,
                                   float* dw, int lddw, float* db, int M, int Cin, int Cout, void* workspace, size_t workspace_bytes, hipStream_t stream);
int dmx_softmax_bwd_rows_launch(const bf16* P, int ldp, const float* dP, int lddp, bf16* dS, int ldds, int rows, int n, float scale, hipStream_t stream);
// y = W act(x) + b with act = SiLU when silu_in (dmx_linear_small): dw (+)= dy^T act(x), db (+)= colsum(dy), dx = act'(x) * (dy W)
int dmx_linear_small_bwd_launch(const float* x, int ldx, const float* dy, int lddy, const bf16* w, int ldw,
                                float* dw, int lddw, float* db, int db_stride, float* dx, int lddx,
                                int B, int N, int K, int silu_in, int accumulate, hipStream_t stream);

// ------------------------------------------------------------------ temb.hip
int dmx_timestep_embedding_launch(const long long* t, int t_count, const float* freq, int B, int dim, float* out, hipStream_t stream);
int dmx_linear_small_launch(const float* x, int ldx, const bf16* w, int ldw, const float* bias, float* y, int ldy,
                            int B, int N, int K, int silu_in, hipStream_t stream);

// ------------------------------------------------------------------ sched.hip
int dmx_sched_ddim_launch(const float* x, const float* eps, const float* noise, float* out, size_t n,
                          float sqrt_bt, float sqrt_at, float sqrt_ap, float dir_coef, float std, int vpred, hipStream_t stream);
int dmx_sched_ddpm_launch(const float* x, const float* eps, const float* noise, float* out, size_t n,
                          float sqrt_bt, float sqrt_at, float c0, float c1, float sigma, int vpred, hipStream_t stream);
int dmx_add_noise_launch(const float* x0, const float* noise, const float* sa, const float* sb, float* out,
                         int B, size_t per, int velocity, hipStream_t stream);
int dmx_gaussian_sample_launch(const float* moments, const float* noise, float* out, int B, int C, int HW, float scale, hipStream_t stream);
