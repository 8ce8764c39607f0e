/*
 * diffute_hip.h - C-ABI of the MI355X-native (gfx950) DiffUTE hot path.
 *
 * The reference (chenhaoxing/DiffUTE) has no FFI of its own: its hot path is reached
 * through three Python classes of the un-vendored `diffusers` library.  Each entry point
 * below names the reference call site (file:line under /root/reference) whose arithmetic
 * it replaces; INTEGRATION.md shows the Python-side binding (ctypes) a maintainer adds.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer unless it is named
 *     `h_...` or is a `const char*` / descriptor struct (host memory);
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream);
 *   - activations between kernels are NHWC ("pixel-major") bf16 (fp16 in the -DDMX_F16 build, see dmx_element_type);
 *     model inputs / outputs at
 *     the Python boundary stay NCHW fp32 exactly as the reference passes them
 *     (app.ipynb:762, train_diffute_v1.py:731), converted inside the library;
 *   - functions return 0 on success, a negative DMX_ERR_* code otherwise; the message of
 *     the last failure on the calling thread is `dmx_last_error()`;
 *   - the library never allocates model memory: weights arena, context cache and workspace
 *     are caller-provided (sizes from the *_bytes queries).  The only internal allocation
 *     is one 4 KiB zero page used for convolution padding.
 */
#ifndef DIFFUTE_HIP_H
#define DIFFUTE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DMX_OK 0
#define DMX_ERR_ARG (-1)
#define DMX_ERR_HIP (-2)
#define DMX_ERR_UNSUPPORTED (-3)
#define DMX_ERR_WORKSPACE (-4)

typedef void* dmx_stream_t;
typedef struct dmx_unet dmx_unet;
typedef struct dmx_vae dmx_vae;

int dmx_version(void);
const char* dmx_last_error(void);
/* Device-side failures.  Kernels that wait for other blocks of their launch (the K-split slab exchange of dmx_conv3x3_gn, the
 * stream-K fix-up of the persistent GEMMs) bound the wait (~40 ms); a block that gives up writes a record into pinned host memory
 * and goes on (the GPU never hangs, the launch's result is invalid).  Every later launch of the library, every graph replay and this
 * poll read the record WITHOUT synchronising and return DMX_ERR_DEVICE (-5) once, with the detail in dmx_last_error().  Call it
 * after your own stream / device synchronisation to cover the last launches. */
int dmx_device_error(void);
/* test support for that channel: one thread raises `code`; `blocks` single-wave blocks that each fill a CU's LDS spin for ticks x 10 ns */
int dmx_test_raise_device_error(int code, dmx_stream_t stream);
int dmx_test_occupy_cus(int blocks, long long ticks, dmx_stream_t stream);
/* The 16-bit storage / MFMA operand type this build of the library computes in: "bf16" (libdiffute_hip.so) or "fp16"
 * (libdiffute_hip_f16.so, the same sources compiled with -DDMX_F16).  Wherever this header says "bf16" for an activation,
 * weight or context buffer it means this element type.  The host mirror loads the fp16 build for a model moved with
 * `.to(dtype=torch.float16)` (vae.to(device, dtype=weight_dtype), train_diffute_v1.py:789-797; BASELINE configs[4]). */
const char* dmx_element_type(void);

/* ------------------------------------------------------------------------------------
 * Operator level.  Used by the parity tests and by the executors.
 * ---------------------------------------------------------------------------------- */

/* K1/K1s/K2/K7/K8/K10: fused implicit-GEMM convolution / linear layer.
 * Replaces torch conv2d / linear / F.interpolate(nearest x2) / torch.cat(dim=1) inside
 * diffusers ResnetBlock2D, Downsample2D, Upsample2D, Attention and FeedForward, reached from
 * unet(...) app.ipynb:814, train_diffute_v1.py:913 and vae.encode/decode app.ipynb:793,819. */
typedef struct dmx_gemm_desc {
  const void* x0; const void* x1;   /* bf16 activations; x1 = second channel range or NULL   */
  int ldx0, ldx1, cx0;              /* row strides (elements); channels taken from x0        */
  int direct;                       /* 1: X row m = activation row m (linear / 1x1 stride 1) */
  int IH, IW, OH, OW;               /* source grid (before x2 upsample) and output grid      */
  int stride, pad, ups, ksize;      /* ksize 1 or 3; pad = top/left pad; ups = nearest x2    */
  int Cin;                          /* channels per tap (cx0 + channels of x1)               */
  int Ktaps;                        /* ksize*ksize*Cin                                       */
  const void* s0; const void* s1;   /* optional fused 1x1 shortcut sources (K - Ktaps chans) */
  int lds0, lds1, cs0;
  const void* w; int ldw;           /* bf16 weights [N][K], K = (tap, channel) contiguous    */
  int M, N, K;
  const float* bias;                /* [N] or NULL                                           */
  const float* rowbias;             /* [M/rows_per_group][ldrb] or NULL (time embedding)     */
  int rows_per_group, ldrb;
  const void* res; int ldres;       /* bf16 residual or NULL                                 */
  void* out; int ldo; int out_f32;  /* bf16 (default) or fp32 output                         */
  int geglu;                        /* 1: weights/bias GEGLU-packed, out[m][j] = a*gelu(b)   */
  int force_tn, force_splitk;       /* 0 = automatic plan; tuning / tests may pin the tile (1|2) and split */
  int group_m;                      /* m-tiles per L2 super-tile of the block rasterisation (0 = default) */
  long long* timing;                /* optional device buffer [blocks][4]: per-block start / prologue / loop / end
                                       timestamps in 10 ns ticks (measurement aid), normally NULL */
  int dbg;                          /* measurement aid, must be 0: bit0 skips the MFMA phase, bit1 the DMA refills */
  int act;                          /* 1: exact (erf) GELU after the bias (ViT MLP fc1), bf16 output only             */
  /* Folded LayerNorm (diffusers BasicTransformerBlock norm1/2/3 in front of attn1.to_q|k|v, attn2.to_q, ff.net.0.proj,
   * reached from unet(...) app.ipynb:814): the GEMM that PRODUCES the residual stream also emits, per output row, partial
   * (sum, sum of squares) of its rounded output - rowstats_out [dmx_conv_gemm_rowstats_tiles(d)][M][2] fp32 - and the GEMM
   * that CONSUMES LayerNorm(x) runs on the raw rows with w = W*diag(gamma) and finishes
   * y = rstd*(acc - mean*ln_c1[n]) + ln_c2[n]   (ln_c1[n] = sum_k w[n][k], ln_c2[n] = sum_k beta[k] W[n][k] + bias[n]). */
  float* rowstats_out;              /* producer side, or NULL                                                          */
  const float* ln_stats; int ln_tiles;   /* consumer side: the producer's partials and how many it wrote per row       */
  const float* ln_c1; const float* ln_c2; int ln_C; float ln_eps;   /* ln_C = normalised feature count (= K)           */
  /* GroupNorm statistics from the producer (ResnetBlock2D norm1 / norm2, Transformer2DModel.norm and conv_norm_out behind
   * unet(...), app.ipynb:814, and the same norms of AutoencoderKL): the conv / linear that WRITES a tensor also adds, per
   * (sample, channel), a statistics record of its rounded outputs - four int64 words {sum * 2^20, floor(sumsq * 2^8),
   * (sumsq - that) * 2^40, 0}, exact integer accumulation, no wrap below sumsq = 3.6e16 - into
   * colstats[(sample*N + n)*4 ..] (zero before the call; cs_rows = rows per sample, a multiple of the plan's tile
   * rows - dmx_conv_gemm_colstats_ok(d) says whether the plan this problem gets can do it).  The GroupNorm that READS the
   * tensor is then dmx_groupnorm_from_stats: one apply-only pass, no statistics pass over the tensor. */
  long long* colstats; int cs_rows;
} dmx_gemm_desc;
size_t dmx_conv_gemm_workspace_bytes(const dmx_gemm_desc* d);
int dmx_set_gn_producer_stats(int on);                      /* tuning aid: 0 makes the model executors compute GroupNorm statistics in the
                                                               consumer again (A/B runs inside one process); returns the old setting */
int dmx_conv_gemm_colstats_ok(const dmx_gemm_desc* d);      /* 1 when dmx_conv_gemm(d) can fill d->colstats (d->cs_rows set)      */
int dmx_conv_gemm_rowstats_tiles(const dmx_gemm_desc* d);   /* partials per row that dmx_conv_gemm(d) will write to rowstats_out */
int dmx_conv_gemm(const dmx_gemm_desc* d, void* workspace, size_t workspace_bytes, dmx_stream_t stream);

/* K10 (Upsample2D: F.interpolate(scale 2, nearest) + conv3x3, the diffusers call inside pipeline_diffute.py's unet(...),
 * SURVEY.md 8a) as four 2x2 convolutions on the SOURCE grid, one per output pixel parity, with the 3x3 taps that land on
 * the same source pixel summed beforehand: 4/9 of the multiply-adds of the direct form, same result up to one bf16
 * rounding of the summed taps.  phase_weights = [4][N][4*Cin] bf16 from dmx_pack_ups_phase_weights (w3 = taps-major
 * packed 3x3 weights [N][ldw3] as dmx_pack_conv_weight writes them).  x: NHWC bf16 [B*IH*IW][ldx]; out: NHWC bf16
 * [B*2IH*2IW][ldo]; bias fp32 [N] or NULL. */
int dmx_pack_ups_phase_weights(const void* w3, int ldw3, void* phase_weights, int N, int Cin, dmx_stream_t stream);
size_t dmx_conv_ups2x_workspace_bytes(int B, int IH, int IW, int Cin, int N, int force_tn, int force_splitk);
int dmx_conv_ups2x(const void* x, int ldx, int B, int IH, int IW, int Cin, const void* phase_weights, int N, const float* bias,
                   void* out, int ldo, int force_tn, int force_splitk, void* workspace, size_t workspace_bytes, dmx_stream_t stream);

/* Tuning aid (scripts/tune_in_situ.py): override the tile plan of every GEMM with this (M, N, K, stride, ups) signature
 * (ups: 0, 1, or 2 for the phase-decomposed upsample conv) by template instance `cfg` (0..11: gemm.hip tiles; 12..14: the persistent linear kernel of lin.hip) and split-K factor; cfg < 0
 * clears all overrides.  Not thread-safe; captured hipGraphs keep the plan they were captured with. */
int dmx_gemm_plan_override(int M, int N, int K, int stride, int ups, int cfg, int splitk);

/* Training: weight gradient of the conv / linear that `d` describes (the forward call's gather
 * fields x0/x1/cx0/direct/IH.../ksize/Cin and M, N; K taken as Ktaps - the fused shortcut is a separate direct call):
 *   dw[n][k] (+)= sum_m dy[m][n] * X[m][k]   fp32, k in the packed (tap, channel) order of the forward weights.
 * Replaces the conv/linear weight-gradient kernels autograd runs under `accelerator.backward(loss)`
 * (train_diffute_v1.py:925).  dmx_colsum gives the bias gradient (groups = 1) and the per-image gradient of the
 * time-embedding row bias (groups = B, rows_per_group = OH*OW). */
size_t dmx_conv_wgrad_workspace_bytes(const dmx_gemm_desc* d, int accumulate);
int dmx_conv_wgrad(const dmx_gemm_desc* d, const void* dy, int lddy, float* dw, int accumulate,
                   void* workspace, size_t workspace_bytes, dmx_stream_t stream);
size_t dmx_colsum_workspace_bytes(int groups, int rows_per_group, int N);
int dmx_colsum(const void* dy, int lddy, int groups, int rows_per_group, int N, float* out, int ldo, int accumulate,
               void* workspace, size_t workspace_bytes, dmx_stream_t stream);

/* K3: GroupNorm (+SiLU) over NHWC bf16; optional virtual channel concat of (x0 | x1).
 * Replaces torch.nn.GroupNorm + SiLU in ResnetBlock2D / Transformer2DModel / VAE blocks. */
size_t dmx_groupnorm_workspace_bytes(int B, int HW, int groups);
int dmx_groupnorm(const void* x0, int ldx0, const void* x1, int ldx1, int c0, int C, int groups,
                  int B, int HW, const float* gamma, const float* beta, float eps, int silu,
                  void* y, int ldy, void* workspace, size_t workspace_bytes, dmx_stream_t stream);
/* GroupNorm (+SiLU) with the statistics taken from the producers of x0 / x1: st0 = colstats of the GEMM that wrote x0
 * ([B][c0][4] int64 records), st1 likewise for x1 ([B][C - c0][4]) or NULL without a second source.  One launch. */
int dmx_groupnorm_from_stats(const void* x0, int ldx0, const void* x1, int ldx1, int c0, int C, int groups,
                             int B, int HW, const float* gamma, const float* beta, float eps, int silu,
                             const long long* st0, const long long* st1, void* y, int ldy, dmx_stream_t stream);

/* K1 + K3 as ONE launch (north_star: "NHWC conv2d with LDS-staged input tiles ... GroupNorm/SiLU fused per-channel in LDS"):
 * conv3x3 (stride 1, pad 1) over a halo tile that is staged ONCE per 64-channel chunk in LDS - the nine taps are shifted LDS
 * fragment reads - with the GroupNorm(32 groups)[+SiLU] in front of it applied to the staged tile in place, from the statistics
 * records of the tensor's producer(s).  Replaces `conv1(nonlinearity(norm1(x)))` / `conv2(nonlinearity(norm2(h)))` (+ conv_shortcut,
 * + the time-embedding add, + the residual add) of every diffusers ResnetBlock2D behind unet(...) /root/reference/app.ipynb:814,
 * train_diffute_v1.py:913 and vae.encode / vae.decode app.ipynb:793,819.
 *   x0 | x1: NHWC input (virtual channel concat; x1 NULL without one), cx0 / Cin multiples of 64; H x W a multiple of 8 x 32 or 16 x 16
 *   gn = 1: y = GroupNorm(x)[SiLU] feeds the conv (padding pixels are zeros of y); st0 / st1 = records [B][channels][4] of x0 / x1
 *   s0 | s1: optional 1x1 shortcut K segment on RAW tensors of the output grid (Csc channels, multiples of 64; 0 = none)
 *   w: [N][ldw] with k = tap*Cin + c, then the shortcut channels (dmx_pack_conv_weight); N a multiple of 160 or of 128
 *   out = conv + bias[n] + rowbias[b*ldrb + n] + res, NHWC; colstats: records of the OUTPUT [B][N][4], added to, or NULL
 * dmx_conv3x3_gn_supported: 1 when the kernel takes the problem (otherwise use dmx_groupnorm + dmx_conv_gemm).
 * dmx_colstats: the statistics records of a tensor whose producer emitted none (one streaming pass; st zero before the call). */
typedef struct dmx_halo_conv_desc {
  const void* x0; const void* x1; int ldx0, ldx1, cx0, Cin;
  int B, H, W;
  int gn, silu, groups; float eps;
  const long long* st0; const long long* st1;
  const float* gamma; const float* beta;
  const void* s0; const void* s1; int lds0, lds1, cs0, Csc;
  const void* w; int ldw; int N;
  const float* bias; const float* rowbias; int ldrb;
  const void* res; int ldres;
  void* out; int ldo;
  long long* colstats;
  int force_split;                  /* 0 = automatic; 1 / 2 / 4 / 8 blocks share the K range of a tile (tests, tuning) */
  int force_bn;                     /* 0 = automatic; 160 / 128 / 80 / 64 output columns per block (tests, tuning) */
  int force_waves;                  /* 0 = automatic; 8 = two-group ping-pong; 4 / 12 = warp-specialised, 4 compute + 4 / 8 loader waves (160 / 128-column tiles, bf16 build) */
  int dbg; long long* timing;       /* measurement aids, 0 / NULL */
} dmx_halo_conv_desc;
int dmx_conv3x3_gn_supported(const dmx_halo_conv_desc* d);
size_t dmx_conv3x3_gn_workspace_bytes(const dmx_halo_conv_desc* d);
int dmx_conv3x3_gn(const dmx_halo_conv_desc* d, void* workspace, size_t workspace_bytes, dmx_stream_t stream);
int dmx_colstats(const void* x, int ldx, int B, int HW, int C, long long* st, dmx_stream_t stream);
int dmx_set_halo_conv(int on);      /* tuning aid: 0 makes the model executors use GroupNorm + dmx_conv_gemm everywhere; returns the old setting */
int dmx_set_exclusive_device(int on); /* 1 (default): the library's launches have the GPU to themselves, one stream at a time.  0: other streams or other kernels (micro-batches on
                                       * several streams, a collective on a side stream) may hold CUs while a launch runs: dmx_conv3x3_gn then takes no in-kernel K split (its peers
                                       * must be co-resident; a starved launch raises DMX_ERR_DEVICE) and the executors use GroupNorm + dmx_conv_gemm where a split would be needed.
                                       * Returns the old setting; captured UNet steps are keyed on dmx_plan_epoch(), which it bumps.  The host mirror switches it
                                       * to 0 by itself wherever IT creates the concurrency (denoise(micro_batches > 1), set_gradient_sync with world > 1);
                                       * diffute_amd.set_exclusive_device() is the public setter for everything else (a second model / thread / stream). */
int dmx_get_exclusive_device(void); /* the current setting */
int dmx_plan_epoch(void);           /* counter bumped by every switch that changes which kernels / plans a graph walk uses (every dmx_set_* below and above,
                                     * dmx_gemm_plan_override): part of the key of the captured UNet steps, and what a caller that caches workspace sizes keys on */
int dmx_set_defer_reduce(int on);   /* tuning aid: 0 = a split-K convolution whose output is read first by a GroupNorm runs its own reduce pass (default 1: the GroupNorm's slab
                                     * kernel sums the partial planes in its load stage - bit-identical, one launch less; 2: only inside a resnet, conv1 -> norm2, not conv2 -> the next block's first GroupNorm); returns the old setting */
int dmx_set_halo_peers(int on);     /* 1 (default): the K-split blocks of a dmx_conv3x3_gn tile are dealt to ONE XCD and, once every peer has confirmed its XCC id, exchange their
                                       fp32 slabs through that XCD's L2 (plain stores); 0: the round-5 dealing with write-through slabs everywhere.  Same bits either way; returns the old setting */
int dmx_set_halo_ws(int on);        /* tuning aid: 0 keeps dmx_conv3x3_gn's planner off the warp-specialised instances (4 compute + 4 loader waves); returns the old setting */

/* Weight-streaming conv3x3 / conv1x1 / linear for the SKINNY levels (M = B H W in {64, 128, 256} output rows: ResnetBlock2D conv1 / conv2 at the 8x8 level at
 * batch 4 and at the 16x16 / 8x8 levels at batch 1, behind unet(...), /root/reference/app.ipynb:814).  K is a list of up to four SEGMENTS - a source tensor
 * [B H W][C] seen through 9 taps (3x3, stride 1, pad 1) or 1 tap (the fused 1x1 shortcut of a resnet; a linear) - and the weights come in FRAGMENT ORDER
 * (dmx_skinny_pack): every compute wave streams its 1-KB MFMA operands straight from memory into registers, the activations of a K slice are staged once in
 * LDS, the GroupNorm (+ SiLU) in front of the conv is applied to the staged chunk from the statistics records of its input(s) (seg.st; NULL = plain input),
 * N / 32 x S blocks ~ one per CU reduce their K slices inside the kernel in slice order (bit-reproducible).  Epilogue: + bias + rowbias[b] + residual, one
 * rounding, optional statistics records of the output.  Needs dmx_set_exclusive_device(1) (the S blocks of a tile wait for each other, bounded: a starved
 * launch raises DMX_ERR_DEVICE).  Round 6: the kernel is correct and exported, but it measures slower than the tiled split-K GEMM + GroupNorm launch it would
 * replace (EXPERIMENTS.md round 6), so the model executors do NOT take it; dmx_set_skinny is reserved for the day they do (returns the old setting). */
typedef struct {
  const void* x; int ld;            /* NHWC 16-bit rows = output pixels */
  int C;                            /* channels, multiple of 16 */
  int taps;                         /* 9 or 1 */
  const long long* st;              /* statistics records [B][C][4] of x, or NULL (no normalisation) */
  const float* gamma; const float* beta;   /* GroupNorm affine of THIS tensor's channels */
  int gn_c0;                        /* first channel of x inside the normalised (concatenated) tensor */
} dmx_skinny_seg;
typedef struct {
  dmx_skinny_seg seg[4]; int nseg;
  int B, H, W;
  int gn_groups, gn_Ctot; float gn_eps; int silu;    /* GroupNorm over the concatenation of the segments with st != NULL (gn_groups = 0: none) */
  const void* wp;                   /* dmx_skinny_pack output */
  int N;                            /* multiple of 32 */
  const float* bias; const float* rowbias; int ldrb;
  const void* res; int ldres;
  void* out; int ldo;
  long long* colstats;
  int force_S;                      /* 0 = automatic number of K slices (1 .. 8; tests) */
  long long* timing;                /* measurement aid: [blocks][6] phase stamps in 10 ns ticks, or NULL */
  int dbg;                          /* measurement aid, 0 */
} dmx_skinny_desc;
int dmx_skinny_conv_supported(const dmx_skinny_desc* d);
size_t dmx_skinny_conv_workspace_bytes(const dmx_skinny_desc* d);
int dmx_skinny_conv(const dmx_skinny_desc* d, void* workspace, size_t workspace_bytes, dmx_stream_t stream);
/* w [N][ldw] 16-bit, column of (segment s, tap t, channel c) = t * tap_stride[s] + koff[s] + c  ->  wp: N x sum_s(C[s] * taps[s]) elements in fragment order */
int dmx_skinny_pack(const void* w, int ldw, void* wp, int N, int nseg, const int* C, const int* taps, const int* tap_stride, const int* koff, dmx_stream_t stream);
int dmx_set_skinny(int on);

/* The row-local chains of diffusers' BasicTransformerBlock + Transformer2DModel.proj_out (the unet(...) call at
 * /root/reference/app.ipynb:814) at the C = 320 levels, ONE launch each (xf_chain.hip); rows M % 64 == 0:
 *   mode 0:  h_out = x w0^T + b0 + res ;  y = LayerNorm(h_out) folded into w1:  rstd * (h_out w1^T - mean * c1) + c2
 *            (attn1.to_out.0 + residual, then attn2.to_q behind norm2; w1 / c1 / c2 as dmx_pack_ln_fold writes them)
 *   mode 2:  h_out = x w0^T + b0 (res unused) ;  y[M][3C] = LayerNorm(h_out) folded into w1 [3C][C] with c1 / c2 [3C]
 *            (proj_in on the GroupNorm output, then attn1's stacked to_q | to_k | to_v behind norm1; ldy >= 3C)
 *   mode 1:  h_out = x w0^T + b0 + res ;  h3 = h_out + wf2 GEGLU(LayerNorm(h_out) folded into wf1) + bf2 ;
 *            y = h3 wpo^T + bpo + xres     (attn2.to_out.0 + residual, ff.net, proj_out + the block residual; wf1 [8C][C] in
 *            the packed GEGLU order of dmx_pack_geglu_weight with c1 / c2 [8C] in the same order; wf2 [C][4C])
 * All 16-bit operands in the build's element type, biases / c1 / c2 fp32.  dmx_xf_chain_supported: 1 for (M, C) it takes.
 * The UNet executor takes the chains when the M / 64 row blocks fill their last round of CUs at least half; dmx_set_xf_chain(0) makes it use the separate
 * GEMMs always, 2 the chains at every supported size (A/B runs in one process); returns the old setting. */
typedef struct {
  int M, C;
  const void* x; int ldx;
  const void* res; int ldres;
  const void* w0; const float* b0;
  void* h_out; int ldh;
  const void* w1;
  const float* c1; const float* c2;
  void* y; int ldy;
  const void* wf1;
  const void* wf2; const float* bf2;
  const void* wpo; const float* bpo;
  const void* xres; int ldxres;
  float eps;
  int dbg; long long* timing;   /* measurement aids, 0 / NULL: ablation bits (results invalid), [M/64][8] phase timestamps in 10 ns ticks */
  /* mode 1, optional: statistics records of y for the GroupNorm that reads it next, [M / cs_rows][C][4] int64, ADDED to (zero them first);
   * cs_rows = rows per sample, a multiple of 64 */
  long long* colstats; int cs_rows;
  /* mode 2, optional: x is the RAW tensor and GroupNorm(gn_groups, gn_eps, gn_gamma, gn_beta; no activation) is applied in the operand
   * load from the statistics records gn_st of x ([M / gn_rows][C][4], as dmx_colstats / a producer's colstats writes them; gn_rows = rows
   * per sample, a multiple of 64) - bit-identical to dmx_groupnorm_from_stats followed by the plain mode 2 */
  const long long* gn_st; const float* gn_gamma; const float* gn_beta; int gn_groups; int gn_rows; float gn_eps;
} dmx_xf_chain_desc;
int dmx_xf_chain_ok(int M, int C);
int dmx_xf_chain(const dmx_xf_chain_desc* d, int mode, dmx_stream_t stream);
int dmx_set_xf_chain(int on);
int dmx_set_weight_prefetch(int on); /* tuning aid: 0 = the launches of dmx_unet_forward* do not touch the weights of the launches that follow them
                                      * (default 1: every GEMM / attention / halo-conv / chain launch prefetches up to 4 MB of them into the memory-side
                                      * cache - the plan comes from a dry walk of the same graph); returns the old setting */

/* Training (P5 over K3/K4/K8): forward GroupNorm that also keeps (mean, rstd) per (image, group), and the backward
 * kernels of GroupNorm(+SiLU), LayerNorm and the unfused GEGLU.  All deterministic.  `res*` is an optional gradient
 * added into dx (the tensor's other consumer, e.g. the residual branch); dgamma/dbeta fp32, `accumulate` adds.
 * Replace autograd's native_group_norm_backward / native_layer_norm_backward / gelu_backward under
 * accelerator.backward (train_diffute_v1.py:925). */
int dmx_groupnorm_train(const void* x0, int ldx0, const void* x1, int ldx1, int c0, int C, int groups,
                        int B, int HW, const float* gamma, const float* beta, float eps, int silu,
                        void* y, int ldy, float* stats, void* workspace, size_t workspace_bytes, dmx_stream_t stream);
size_t dmx_groupnorm_bwd_workspace_bytes(int B, int HW, int C);
int dmx_groupnorm_bwd(const void* x0, int ldx0, const void* x1, int ldx1, int c0, int C, int groups, int B, int HW,
                      const float* gamma, const float* beta, int silu, const float* stats,
                      const void* dy, int lddy, void* dx0, int lddx0, void* dx1, int lddx1,
                      const void* res0, int ldres0, const void* res1, int ldres1,
                      float* dgamma, float* dbeta, int accumulate,
                      void* workspace, size_t workspace_bytes, dmx_stream_t stream);
size_t dmx_layernorm_bwd_workspace_bytes(int rows, int C);
int dmx_layernorm_bwd(const void* x, int ldx, const void* dy, int lddy, const float* gamma, void* dx, int lddx,
                      const void* res, int ldres, float* dgamma, float* dbeta, int accumulate,
                      int rows, int C, float eps, void* workspace, size_t workspace_bytes, dmx_stream_t stream);
int dmx_geglu_fwd(const void* h, int ldh, void* y, int ldy, int rows, int C2, dmx_stream_t stream);
int dmx_geglu_bwd(const void* h, int ldh, const void* dy, int lddy, void* dh, int lddh, int rows, int C2, dmx_stream_t stream);

/* K4: LayerNorm over the channel axis of [rows][C] bf16 (BasicTransformerBlock norm1/2/3). */
int dmx_layernorm(const void* x, int ldx, void* y, int ldy, const float* gamma, const float* beta,
                  int rows, int C, float eps, dmx_stream_t stream);

/* K5/K6: fused attention core softmax(Q K^T * scale) V, head dim 64, no mask.
 * q: row (b*Sq+s), head h at column 64h.  k: row (b*kv_rows+s).  vt: V transposed,
 * row (64h+d), column (b*skv_stride+s); columns [Skv, round_up(Skv,8)) must be finite.
 * Replaces diffusers Attention (optionally xformers, train_diffute_v1.py:648-659). */
int dmx_attention_fwd(const void* q, int ldq, const void* k, int ldk, int kv_rows,
                      const void* vt, int ldvt, int skv_stride, void* o, int ldo,
                      int B, int H, int Sq, int Skv, float scale, dmx_stream_t stream);

/* Same attention with V given row-major (row b*kv_rows+s, head h at column 64h - e.g. a slice of a fused
 * q|k|v projection): the P.V operand is fetched with gfx950 LDS transpose reads, no V^T tensor is needed. */
/* K6b: single-head attention with a wide head, d = 128 / 256 / 512 (AutoencoderKL mid block: diffusers `Attention` inside
 * the VAE's UNetMidBlock2D, reached from vae.encode / vae.decode, app.ipynb:793,819; train_diffute_v1.py:875,886).
 * q / k / v / o: bf16 row-major, row (b*Sq + s) resp. (b*kv_rows + s), d contiguous (column slices of one fused q|k|v buffer
 * are fine: pass the row stride).  Fused flash-style: no S x S buffer, no workspace. */
int dmx_attention_wide(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, int kv_rows,
                       void* o, int ldo, int B, int Sq, int Skv, int D, float scale, dmx_stream_t stream);
int dmx_attention_fwd_v(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, int kv_rows,
                        void* o, int ldo, int B, int H, int Sq, int Skv, float scale, dmx_stream_t stream);
/* The same op on the BALANCED schedule (stream-K over (128-query block, 64-key tile) items on 3 x CUs block slots; a split row's (O, m, l) halves are
 * folded in a fixed order: bit-repeatable, equal to dmx_attention_fwd_v within the fp32 rounding of the fold).  The executors take it where the plain
 * grid fills the slots unevenly (the 4096 x 4096 self-attention of the 64x64 level at batch 4).  workspace_bytes = 0: the plan keeps the plain grid. */
size_t dmx_attention_fwd_v_balanced_workspace_bytes(int B, int H, int Sq, int Skv);
int dmx_attention_fwd_v_balanced(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, int kv_rows,
                                 void* o, int ldo, int B, int H, int Sq, int Skv, float scale,
                                 void* workspace, size_t workspace_bytes, dmx_stream_t stream);
int dmx_set_attn_balanced(int mode);   /* 0: never; 1 (default): where the plan says it pays; 2: wherever the kernel takes the problem (tests, A/B); returns the old setting */

/* Training (P5 over K5/K6): forward that also keeps each row's log2-sum-exp, and the flash-style backward (dQ, dK, dV;
 * deterministic, nothing of size Sq x Skv is stored).  workspace: [B][H][Sq] floats.  Replace autograd's scaled-dot-
 * product-attention backward under accelerator.backward (train_diffute_v1.py:925). */
int dmx_attention_fwd_train(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, int kv_rows,
                            void* o, int ldo, float* lse, int B, int H, int Sq, int Skv, float scale, dmx_stream_t stream);
size_t dmx_attention_bwd_workspace_bytes(int B, int H, int Sq);
int dmx_attention_bwd(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, int kv_rows,
                      const void* o, const void* d_o, int ldo, const float* lse,
                      void* dq, int lddq, void* dk, int lddk, void* dv, int lddv,
                      int B, int H, int Sq, int Skv, float scale,
                      void* workspace, size_t workspace_bytes, dmx_stream_t stream);

/* K9: sinusoidal timestep embedding (flip_sin_to_cos) and the small-M fp32 linear used by the
 * time-embedding MLP.  t: int64 [t_count] (t_count 1 or B); freq: fp32 [dim/2] table. */
int dmx_timestep_embedding(const int64_t* t, int t_count, const float* freq, int B, int dim, float* out, dmx_stream_t stream);
int dmx_linear_small(const float* x, int ldx, const void* w_bf16, int ldw, const float* bias, float* y, int ldy,
                     int B, int N, int K, int silu_in, dmx_stream_t stream);

/* small-channel im2col (conv_in 9->320, VAE 3->128 / 4->512, quant convs): NCHW fp32 sources
 * (f0|f1|f2 concatenated on channels = torch.cat([latents, mask, masked_latents], 1),
 * app.ipynb:811) or one NHWC bf16 source -> bf16 [B*OH*OW][Kpad]. */
int dmx_im2col_small(const float* f0, int c0, const float* f1, int c1, const float* f2, int c2,
                     const void* h_nhwc, int ldh, int C, int B, int IH, int IW, int OH, int OW,
                     int ksize, int stride, int pad, void* out, int Kpad, dmx_stream_t stream);

/* weight packing: fp32 torch layouts -> bf16 GEMM rows */
int dmx_pack_conv_weight(const float* w, void* out, int Cout, int Cin, int ksize, int ldk, int koff, dmx_stream_t stream);
int dmx_pack_linear_weight(const float* w, void* out, int rows, int cols, int ldo, int geglu, dmx_stream_t stream);
int dmx_pack_geglu_bias(const float* b, float* out, int n, dmx_stream_t stream);
/* training: the data gradient dX = dY * W runs through dmx_conv_gemm with a transposed pack of W
 * (conv: out[ci][flip(tap)*Cout + n], linear: out[k][n]); a stride-2 conv first zero-inserts dY to the input grid,
 * a conv behind a nearest x2 upsample sum-pools its data gradient 2x2.  (autograd conv/linear backward,
 * train_diffute_v1.py:925) */
int dmx_pack_conv_weight_t(const float* w, void* out, int Cout, int Cin, int ksize, int ldk, int koff, dmx_stream_t stream);
int dmx_pack_linear_weight_t(const float* w, void* out, int rows, int cols, int ldo, dmx_stream_t stream);
int dmx_zero_insert2(const void* dy, int lddy, void* z, int B, int OH, int OW, int C, dmx_stream_t stream);
int dmx_sumpool2(const void* du, int lddu, int du_f32, void* dx, int lddx, int B, int H, int W, int C, int accumulate, dmx_stream_t stream);

/* layout helpers */
int dmx_cast_f32_to_bf16(const float* in, void* out, size_t n, dmx_stream_t stream);
int dmx_nhwc_bf16_to_nchw_f32(const void* in, int ldin, float* out, int B, int C, int HW, dmx_stream_t stream);
int dmx_nhwc_f32_to_nchw_f32(const float* in, int ldin, float* out, int B, int C, int HW, dmx_stream_t stream);
int dmx_nchw_f32_to_nhwc_bf16(const float* in, void* out, int ldo, int B, int C, int HW, dmx_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Scheduler.  Scalar coefficients are computed by the host-side
 * scheduler classes; these are the elementwise updates over [B,4,h,w] fp32 latents.
 * ---------------------------------------------------------------------------------- */
/* DDIMScheduler.step(...).prev_sample  (north_star; same call shape as app.ipynb:816) */
int dmx_sched_step_ddim(const float* sample, const float* model_output, const float* noise, float* prev_sample, size_t n,
                        float sqrt_beta_prod_t, float sqrt_alpha_prod_t, float sqrt_alpha_prod_prev,
                        float dir_coef, float std_dev, int v_prediction, dmx_stream_t stream);
/* DDPMScheduler.step(...).prev_sample  (app.ipynb:816); noise = NULL when t == 0 */
int dmx_sched_step_ddpm(const float* sample, const float* model_output, const float* noise, float* prev_sample, size_t n,
                        float sqrt_beta_prod_t, float sqrt_alpha_prod_t, float coef_x0, float coef_xt,
                        float sigma, int v_prediction, dmx_stream_t stream);
/* scheduler.add_noise / get_velocity (train_diffute_v1.py:897,907): per-sample coefficients */
int dmx_sched_add_noise(const float* x0, const float* noise, const float* sqrt_alpha_prod, const float* sqrt_one_minus,
                        float* out, int B, size_t per_sample, dmx_stream_t stream);
int dmx_sched_get_velocity(const float* x0, const float* noise, const float* sqrt_alpha_prod, const float* sqrt_one_minus,
                           float* out, int B, size_t per_sample, dmx_stream_t stream);
/* latent_dist.sample() * scaling_factor (app.ipynb:793-794; train_diffute_v1.py:875-876);
 * noise = NULL gives latent_dist.mode() (train_vae.py:721). moments NCHW fp32 [B][2C][HW]. */
int dmx_gaussian_sample(const float* moments, const float* noise, float* out, int B, int C, int HW, float scale, dmx_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Whole-model executors.
 * ---------------------------------------------------------------------------------- */
typedef struct dmx_unet_config {
  int in_channels, out_channels;
  int block_out_channels[4];
  int layers_per_block;
  int heads[4];                 /* diffusers `attention_head_dim` (= head counts)            */
  int cross_attention_dim;
  int norm_num_groups;
  int down_has_attn[4], up_has_attn[4];
} dmx_unet_config;

/* UNet2DConditionModel  (from_pretrained: train_diffute_v1.py:633-635, app.ipynb:551-553) */
dmx_unet* dmx_unet_create(const dmx_unet_config* cfg);
void dmx_unet_destroy(dmx_unet* u);
int dmx_unet_param_count(const dmx_unet* u);
/* name = diffusers state-dict key; shape up to 4 dims (unused = 0) */
int dmx_unet_param_info(const dmx_unet* u, int index, const char** name, int shape[4]);
size_t dmx_unet_arena_bytes(const dmx_unet* u);
int dmx_unet_bind_arena(dmx_unet* u, void* arena, size_t bytes);
int dmx_unet_load_param(dmx_unet* u, const char* name, const float* src_f32, dmx_stream_t stream);
int dmx_unet_finalize(dmx_unet* u, const float* h_freq_table, dmx_stream_t stream);
size_t dmx_unet_context_bytes(const dmx_unet* u, int B, int ctx_len);
size_t dmx_unet_workspace_bytes(dmx_unet* u, int B, int H, int W, int ctx_len);
/* cross-attention K / V^T of the glyph context, constant across denoise steps (app.ipynb:776,814) */
int dmx_unet_set_context(dmx_unet* u, const void* ctx, int ctx_is_bf16, int B, int ctx_len,
                         void* context_cache, size_t context_bytes, void* workspace, size_t workspace_bytes,
                         dmx_stream_t stream);
/* unet(sample, timestep, encoder_hidden_states).sample  (app.ipynb:814, train_diffute_v1.py:913).
 * The 9-channel sample is given as up to three NCHW fp32 tensors (c0+c1+c2 = in_channels):
 * pass the concatenated tensor as f0 with c0 = 9, or latents/mask/masked latents separately
 * (fuses torch.cat of app.ipynb:811).  timesteps: device int64 [t_count], t_count 1 or B.
 * out: NCHW fp32 [B][out_channels][H][W]. */
int dmx_unet_forward(dmx_unet* u, const float* f0, int c0, const float* f1, int c1, const float* f2, int c2,
                     const int64_t* timesteps, int t_count, const void* context_cache, int ctx_len,
                     float* out, int B, int H, int W, void* workspace, size_t workspace_bytes, dmx_stream_t stream);

/* The time-embedding MLP and the stacked time_emb_proj of every resnet depend on the timestep only: a denoise loop
 * (/root/reference/app.ipynb:806-816) can compute them for ALL its timesteps in one batched pass and let each step fetch its row
 * - one tiny launch instead of four per step, and the 49 MB of projection weights are streamed once per loop instead of once
 * per step.  Rows are bit-identical to what the per-step path computes.  dmx_unet_temb_table fills table[T][tproj] (fp32;
 * dmx_unet_temb_table_floats(T) elements); dmx_unet_use_temb_table(u, table, step_index) makes the following scalar-timestep
 * forwards read row *step_index (an int on the device, updated by the caller between steps); (NULL, NULL) switches back. */
size_t dmx_unet_temb_table_floats(dmx_unet* u, int T);
size_t dmx_unet_temb_table_workspace_bytes(dmx_unet* u, int T);
int dmx_unet_temb_table(dmx_unet* u, const int64_t* timesteps, int T, float* table, void* workspace, size_t workspace_bytes, dmx_stream_t stream);
int dmx_unet_use_temb_table(dmx_unet* u, const float* table, const int* step_index);
/* Same as dmx_unet_forward; the launch sequence is captured into a hipGraph the second time an identical argument
 * tuple is seen and replayed afterwards.  Requires a non-NULL stream (falls back to eager launches otherwise). */
int dmx_unet_forward_graph(dmx_unet* u, const float* f0, int c0, const float* f1, int c1, const float* f2, int c2,
                           const int64_t* timesteps, int t_count, const void* context_cache, int ctx_len,
                           float* out, int B, int H, int W, void* workspace, size_t workspace_bytes, dmx_stream_t stream);

/* Validation / debugging entry points (tests only; the product path never calls them).
 *   dmx_unet_forward_taps   dmx_unet_forward that also copies out the block outputs conv_in, down0..3, mid, up0..3 (the points the
 *                           oracle taps, oracle/unet.py) as NCHW fp32, back to back, into `taps`; tap_shapes[4*i..] = (B, C, H, W).
 *   dmx_unet_forward_f32    the same graph walker on fp32 activations, the fp32 master copy of the parameters (`masters`:
 *                           dmx_unet_grad_bytes(u) bytes filled by dmx_unet_master_import for every parameter) and plain fp32
 *                           kernels: north_star's "within 1e-3 rel fp32" against the fp32 reference path (app.ipynb:560 runs
 *                           inference in fp32).  `context` = raw glyph context [B][ctx_len][cross_attention_dim] fp32; taps optional. */
int dmx_unet_forward_taps(dmx_unet* u, const float* f0, int c0, const float* f1, int c1, const float* f2, int c2,
                          const int64_t* timesteps, int t_count, const void* context_cache, int ctx_len, float* out, int B, int H, int W,
                          void* workspace, size_t workspace_bytes, float* taps, size_t tap_floats, int* tap_shapes, int* n_taps, dmx_stream_t stream);
size_t dmx_unet_workspace_bytes_f32(dmx_unet* u, int B, int H, int W, int ctx_len);
int dmx_unet_forward_f32(dmx_unet* u, const void* masters, const float* f0, int c0, const float* f1, int c1, const float* f2, int c2,
                         const int64_t* timesteps, int t_count, const float* context, int ctx_len, float* out, int B, int H, int W,
                         void* workspace, size_t workspace_bytes, float* taps, size_t tap_floats, int* tap_shapes, int* n_taps, dmx_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Training of the denoiser: train_diffute_v1.py:913-925.
 *   dmx_unet_train_prepare   W^T copies of every GEMM weight (data-gradient operands) into `wt`; after each weight update
 *   dmx_unet_train_forward   pred = unet(cat(f0,f1,f2), t, ctx), keeping what the backward needs inside `workspace`
 *   dmx_unet_train_backward  all parameter gradients (fp32, packed like the weights arena: the gradient of the element
 *                            at arena byte offset o sits at byte offset 2*o of `grads`); records event i when gradient
 *                            bucket i is final, so the gradient exchange (RCCL) can overlap the rest of the backward
 *   dmx_unet_grad_export     one parameter's gradient in its torch layout
 *   dmx_mse_loss             loss = mean((pred - target)^2) and dpred = 2 (pred - target) / n * grad_scale
 * ---------------------------------------------------------------------------------- */
size_t dmx_unet_train_workspace_bytes(dmx_unet* u, int B, int H, int W, int ctx_len);
size_t dmx_unet_train_wt_bytes(const dmx_unet* u);
int dmx_unet_train_prepare(dmx_unet* u, void* wt_arena, size_t wt_bytes, dmx_stream_t stream);
size_t dmx_unet_grad_bytes(const dmx_unet* u);
int dmx_unet_train_forward(dmx_unet* u, const void* wt_arena,
                           const float* f0, int c0, const float* f1, int c1, const float* f2, int c2,
                           const int64_t* timesteps, int t_count, const void* ctx, int ctx_is_bf16, int ctx_len,
                           float* pred, int B, int H, int W, void* workspace, size_t workspace_bytes, dmx_stream_t stream);
int dmx_unet_train_bucket_count(const dmx_unet* u);
int dmx_unet_train_bucket_range(const dmx_unet* u, int i, size_t* begin, size_t* end);
int dmx_unet_train_tail_range(const dmx_unet* u, size_t* begin, size_t* end);
int dmx_unet_train_backward(dmx_unet* u, void* grads, const float* dpred, void* const* events, int n_events, dmx_stream_t stream);
int dmx_unet_grad_export(const dmx_unet* u, const void* grads, const char* name, float* dst, dmx_stream_t stream);
int dmx_unet_grad_range(const dmx_unet* u, const char* name, size_t* begin, size_t* end);
/* Training of the autoencoder (train_vae.py:716-736): recon = decode(encode(x).latent_dist.mode()),
 * backward for dLoss/drecon.  Same protocol and gradient-arena convention as the UNet. */
size_t dmx_vae_train_workspace_bytes(dmx_vae* v, int B, int H, int W);
size_t dmx_vae_train_wt_bytes(const dmx_vae* v);
int dmx_vae_train_prepare(dmx_vae* v, void* wt_arena, size_t wt_bytes, dmx_stream_t stream);
size_t dmx_vae_grad_bytes(const dmx_vae* v);
int dmx_vae_train_forward(dmx_vae* v, const void* wt_arena, const float* x, float* recon, int B, int H, int W,
                          void* workspace, size_t workspace_bytes, dmx_stream_t stream);
int dmx_vae_train_backward(dmx_vae* v, void* grads, const float* drecon, dmx_stream_t stream);
int dmx_vae_grad_export(const dmx_vae* v, const void* grads, const char* name, float* dst, dmx_stream_t stream);

/* On-device pre/post-processing either side of the denoise loop.  Reference (host, PIL / numpy / cv2 /
 * albumentations): generate_mask app.ipynb:370-378; prepare_mask_and_masked_image :380-383; the crop + alb.Resize(512,512)
 * + alb.Normalize(0.5, 0.5) + ToTensorV2 pipelines :332-344,:722-745; F.interpolate(mask, latent size) :776-779;
 * (image_vae / 2 + 0.5) * 255, cv2.resize to the crop, paste inside the text box, round to uint8 :825-846.
 * Resize semantics: OpenCV INTER_LINEAR as published (uint8 fixed point / fp32; exact 2x downscale = 2x2 area mean).
 *   mask_rasterize:   mask[y][x] = 1 inside the inclusive rectangle (x0,y0)-(x1,y1), else 0 (PIL draw.rectangle, fill=1).
 *   preprocess_crop:  image_hwc uint8 [H][W][3], mask uint8 [H][W]; crop [y_s:y_s+crop_scale, x_s:x_s+crop_scale] (clipped at
 *                     the border like a numpy slice) -> S x S: out_image / out_masked_image fp32 [3][S][S] normalised to
 *                     [-1,1] (masked = image * (mask < 0.5) before the resize), out_mask uint8 [S][S],
 *                     out_mask_latent fp32 [S/8][S/8] (nearest; may be NULL).
 *   postprocess_paste: image_vae fp32 [3][S][S] in [-1,1] -> resized to the crop extent, written over original_hwc inside
 *                     the box [y1:y2, x1:x2] only; out_hwc uint8 [H][W][3] (everything else copied).  Values outside
 *                     [0,255] are clamped (numpy's cast is undefined there). */
int dmx_mask_rasterize(unsigned char* mask, int H, int W, int x0, int y0, int x1, int y1, dmx_stream_t stream);
int dmx_preprocess_crop(const unsigned char* image_hwc, const unsigned char* mask, int H, int W, int x_s, int y_s, int crop_scale,
                        int S, float* out_image, float* out_masked_image, unsigned char* out_mask, float* out_mask_latent,
                        dmx_stream_t stream);
int dmx_postprocess_paste(const float* image_vae, int S, const unsigned char* original_hwc, unsigned char* out_hwc, int H, int W,
                          int x_s, int y_s, int crop_scale, int x1, int y1, int x2, int y2, dmx_stream_t stream);

/* Fused AdamW + global-norm clipping over packed fp32 arenas (torch.optim.AdamW + clip_grad_norm_,
 * train_diffute_v1.py:721-727,927-930).  masters / exp_avg / exp_avg_sq / grads: dmx_unet_grad_bytes each.  The step also
 * rewrites the weights arena (bf16 weights, fp32 vectors) in place; afterwards call dmx_unet_refresh_derived (folded
 * LayerNorm / bias copies of the inference graph) and dmx_unet_train_prepare (transposed weights).
 * scalars: device float[2] = (gradient norm before clipping, clip coefficient).
 * ema (optional, may be NULL): fp32 shadow arena of the same layout, updated in the same pass with
 * ema -= (1 - ema_decay) * (ema - p_new)  (diffusers EMAModel.step; `ema_unet.step(unet.parameters())`, :934-935). */
int dmx_unet_optim_chunks(const dmx_unet* u);
size_t dmx_unet_optim_table_bytes(const dmx_unet* u);
size_t dmx_unet_optim_elements(const dmx_unet* u);
int dmx_unet_optim_table(const dmx_unet* u, void* table_dev, size_t bytes, dmx_stream_t stream);
int dmx_unet_master_import(const dmx_unet* u, void* masters, const char* name, const float* src, dmx_stream_t stream);
int dmx_unet_adamw_step(dmx_unet* u, const void* table_dev, int nchunks, void* masters, void* exp_avg, void* exp_avg_sq, const void* grads,
                        float lr, float beta1, float beta2, float eps, float weight_decay, int step, float max_grad_norm,
                        float* scalars, void* workspace, size_t workspace_bytes, void* ema, float ema_decay, dmx_stream_t stream);
/* The same step on a LOSS-SCALED gradient (accelerate's `--mixed_precision fp16`, train_diffute_v1.py:267,583,925: GradScaler.scale(loss)
 * .backward(), then unscale_ -> clip_grad_norm_ -> step inside accelerator.clip_grad_norm_ / optimizer.step()): grads holds g * S and
 * grad_inv_scale = 1 / S.  Norm, clipping and the update use the unscaled gradient; scalars is device float[3] = (norm of the unscaled
 * gradient, factor applied to the arena's values, found_inf): when the gradient holds an inf or NaN, found_inf = 1 and the step changes
 * NOTHING (masters, moments, EMA shadow, weights arena) - the caller then must not count it (GradScaler.step / update). */
int dmx_unet_adamw_step_scaled(dmx_unet* u, const void* table_dev, int nchunks, void* masters, void* exp_avg, void* exp_avg_sq, const void* grads,
                               float lr, float beta1, float beta2, float eps, float weight_decay, int step, float max_grad_norm,
                               float* scalars, void* workspace, size_t workspace_bytes, void* ema, float ema_decay, float grad_inv_scale,
                               dmx_stream_t stream);
int dmx_unet_refresh_derived(dmx_unet* u, dmx_stream_t stream);
size_t dmx_mse_loss_workspace_bytes(void);
int dmx_mse_loss(const float* pred, const float* target, size_t n, float* loss, float* dpred, float grad_scale,
                 void* workspace, size_t workspace_bytes, dmx_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Glyph encoder: the ViT encoder of TrOCR, `trocr_model(pixel_values).last_hidden_state`
 * (app.ipynb:773-776, train_diffute_v1.py:868-871).  Same handle protocol as the UNet / VAE: create, enumerate the
 * parameters (transformers ViTModel state-dict keys), bind an arena, load fp32 tensors, finalize, forward.
 * ---------------------------------------------------------------------------------- */
typedef struct dmx_vit dmx_vit;
typedef struct dmx_vit_config {
  int image_size, patch_size, num_channels;     /* 384, 16, 3 */
  int hidden_size, num_layers, num_heads;       /* 1024, 24, 16 (head dim must be 64) */
  int intermediate_size;                        /* 4096 */
  int qkv_bias;                                 /* 0 for TrOCR (BEiT-style), 1 for plain ViT checkpoints */
  float layer_norm_eps;                         /* 1e-12 */
} dmx_vit_config;
dmx_vit* dmx_vit_create(const dmx_vit_config* cfg);
void dmx_vit_destroy(dmx_vit* v);
int dmx_vit_param_count(const dmx_vit* v);
int dmx_vit_param_info(const dmx_vit* v, int index, const char** name, int shape[4]);
size_t dmx_vit_arena_bytes(const dmx_vit* v);
int dmx_vit_bind_arena(dmx_vit* v, void* arena, size_t bytes);
int dmx_vit_load_param(dmx_vit* v, const char* name, const float* src_f32, dmx_stream_t stream);
int dmx_vit_finalize(dmx_vit* v, dmx_stream_t stream);
size_t dmx_vit_workspace_bytes(dmx_vit* v, int B);
int dmx_vit_forward(dmx_vit* v, const float* pixel_values, float* last_hidden_state, int B,
                    void* workspace, size_t workspace_bytes, dmx_stream_t stream);
/* fp32 VALIDATION instantiation (tests only): the same walker on fp32 activations, the fp32 master copy of the parameters
 * (`masters`: dmx_vit_master_bytes(v) bytes, zero-filled, then dmx_vit_master_import for every parameter) and the plain fp32
 * kernels - compared with transformers' ViTModel output (tests/golden/vit_transformers.npz) at north_star's 1e-3. */
size_t dmx_vit_master_bytes(const dmx_vit* v);
int dmx_vit_master_import(const dmx_vit* v, void* masters, const char* name, const float* src, dmx_stream_t stream);
size_t dmx_vit_workspace_bytes_f32(dmx_vit* v, int B);
int dmx_vit_forward_f32(dmx_vit* v, const void* masters, const float* pixel_values, float* last_hidden_state, int B,
                        void* workspace, size_t workspace_bytes, dmx_stream_t stream);

typedef struct dmx_vae_config {
  int in_channels, out_channels, latent_channels;
  int block_out_channels[4];
  int layers_per_block;
  int norm_num_groups;
} dmx_vae_config;

/* AutoencoderKL (from_pretrained: train_diffute_v1.py:632, app.ipynb:550, train_vae.py:516) */
dmx_vae* dmx_vae_create(const dmx_vae_config* cfg);
void dmx_vae_destroy(dmx_vae* v);
int dmx_vae_param_count(const dmx_vae* v);
int dmx_vae_param_info(const dmx_vae* v, int index, const char** name, int shape[4]);
size_t dmx_vae_arena_bytes(const dmx_vae* v);
int dmx_vae_bind_arena(dmx_vae* v, void* arena, size_t bytes);
int dmx_vae_load_param(dmx_vae* v, const char* name, const float* src_f32, dmx_stream_t stream);
int dmx_vae_finalize(dmx_vae* v, dmx_stream_t stream);
size_t dmx_vae_workspace_bytes(dmx_vae* v, int B, int H, int W, int decode);
/* vae.encode(x) -> moments NCHW fp32 [B][2*latent][H/8][W/8]  (app.ipynb:781,793) */
int dmx_vae_encode(dmx_vae* v, const float* x, float* moments, int B, int H, int W,
                   void* workspace, size_t workspace_bytes, dmx_stream_t stream);
/* vae.decode(z).sample: z NCHW fp32 [B][latent][h][w] -> image [B][3][8h][8w]  (app.ipynb:819) */
/* fp32 VALIDATION instantiation of the two graphs above (tests only; north_star's "within 1e-3 rel fp32" at model level for
 * app.ipynb:793,819): fp32 activations, the fp32 master copy of the parameters (`masters`: dmx_vae_grad_bytes(v) bytes,
 * filled by dmx_vae_master_import for every parameter) and the plain fp32 kernels.  The mid-block attention keeps the
 * scores of 16 queries in LDS: images up to ~384 px. */
int dmx_vae_master_import(const dmx_vae* v, void* masters, const char* name, const float* src, dmx_stream_t stream);
size_t dmx_vae_workspace_bytes_f32(dmx_vae* v, int B, int H, int W, int decode);
int dmx_vae_encode_f32(dmx_vae* v, const void* masters, const float* x, float* moments, int B, int H, int W,
                       void* workspace, size_t workspace_bytes, dmx_stream_t stream);
int dmx_vae_decode_f32(dmx_vae* v, const void* masters, const float* z, float* image, int B, int h, int w,
                       void* workspace, size_t workspace_bytes, dmx_stream_t stream);
int dmx_vae_decode(dmx_vae* v, const float* z, float* image, int B, int h, int w,
                   void* workspace, size_t workspace_bytes, dmx_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Measurement aid (bench.py roofline leg): while enabled, every executor launch is bracketed
 * by hipEvents on its own stream.  dmx_profile_end synchronises and fills, per kernel class
 * (0 conv/linear GEMM, 1 attention, 2 GroupNorm, 3 LayerNorm, 4 other), four doubles:
 * launches, total milliseconds, algorithmic FLOPs, algorithmic bytes.
 * ---------------------------------------------------------------------------------- */
int dmx_profile_begin(void);
int dmx_profile_end(double* h_out, int n_out);
/* optional: write one CSV row per launch (class, ms, flops, bytes, shape tag) at the next dmx_profile_end */
int dmx_profile_dump_path(const char* h_path);
size_t dmx_profile_symbols(char* buf, size_t cap);   /* per kernel SYMBOL of the region dmx_profile_end closed last: lines "class\tlaunches\tms\tflops\tbytes\tsymbol" (rocprofv3's spelling of the name); returns bytes written, 0 = buffer too small */

#ifdef __cplusplus
}
#endif
#endif /* DIFFUTE_HIP_H */
