// Host-side executor infrastructure shared by the UNet and VAE graphs:
//   ParamTable  - diffusers state-dict key -> packed location in the caller's weights arena
//   Workspace   - first-fit sub-allocator over the caller's activation workspace; a dry run
//                 of the same alloc/free sequence gives the exact peak (workspace_bytes query)
//   Exec        - op wrappers that launch (or, in a dry run, only account for) the kernels
#pragma once
#include <map>
#include <string>
#include <unordered_map>
#include <vector>
#include "kernels.h"

struct PackRule {
  enum Kind { COPY_F32, CONV, LINEAR, GEGLU_W, GEGLU_B } kind = COPY_F32;
  size_t dst = 0;          // byte offset into the arena
  int rows = 0, cols = 0;  // LINEAR/GEGLU: [rows][cols] -> ld ; CONV: Cout, Cin
  int ks = 0, ld = 0, koff = 0;
};
struct ParamEntry { std::string name; int shape[4]; PackRule rule; };

class ParamTable {
 public:
  size_t reserve(size_t bytes) { size_t o = total_; total_ += align_up(bytes, 256); return o; }
  void add(const std::string& name, std::initializer_list<int> shape, const PackRule& r) {
    ParamEntry e; e.name = name; int i = 0; for (int k = 0; k < 4; ++k) e.shape[k] = 0;
    for (int s : shape) e.shape[i++] = s;
    e.rule = r; index_[name] = (int)entries_.size(); entries_.push_back(e);
  }
  // helpers returning the destination offset
  size_t f32(const std::string& name, int n) {
    PackRule r; r.kind = PackRule::COPY_F32; r.dst = reserve((size_t)n * 4); r.rows = n; add(name, {n}, r); return r.dst;
  }
  void f32_at(const std::string& name, int n, size_t dst) {
    PackRule r; r.kind = PackRule::COPY_F32; r.dst = dst; r.rows = n; add(name, {n}, r);
  }
  size_t linear(const std::string& name, int rows, int cols) {
    PackRule r; r.kind = PackRule::LINEAR; r.dst = reserve((size_t)rows * cols * 2); r.rows = rows; r.cols = cols; r.ld = cols;
    add(name, {rows, cols}, r); return r.dst;
  }
  void linear_at(const std::string& name, int rows, int cols, size_t dst, int ld) {
    PackRule r; r.kind = PackRule::LINEAR; r.dst = dst; r.rows = rows; r.cols = cols; r.ld = ld; add(name, {rows, cols}, r);
  }
  void conv_at(const std::string& name, int cout, int cin, int ks, size_t dst, int ldk, int koff) {
    PackRule r; r.kind = PackRule::CONV; r.dst = dst; r.rows = cout; r.cols = cin; r.ks = ks; r.ld = ldk; r.koff = koff;
    add(name, {cout, cin, ks, ks}, r);
  }
  const std::vector<ParamEntry>& entries() const { return entries_; }
  const ParamEntry* find(const std::string& name) const {
    auto it = index_.find(name); return it == index_.end() ? nullptr : &entries_[it->second];
  }
  size_t total() const { return total_; }
  int load(char* arena, const char* name, const float* src, hipStream_t stream) const;
 private:
  std::vector<ParamEntry> entries_;
  std::unordered_map<std::string, int> index_;
  size_t total_ = 0;
};

class Workspace {
 public:
  void reset(void* base, size_t cap, bool dry) {
    base_ = dry ? (char*)4096 : (char*)base; cap_ = dry ? ((size_t)1 << 46) : cap; dry_ = dry;
    free_.clear(); live_.clear(); free_[0] = cap_; used_ = 0; peak_ = 0; failed_ = false;
  }
  void* alloc(size_t bytes);
  void release(const void* p);
  size_t peak() const { return peak_; }
  bool failed() const { return failed_; }
 private:
  char* base_ = nullptr; size_t cap_ = 0; bool dry_ = false;
  std::map<size_t, size_t> free_;                 // offset -> size
  std::unordered_map<size_t, size_t> live_;       // offset -> size
  size_t used_ = 0, peak_ = 0; bool failed_ = false;
};

// optim.hip: one parameter (torch layout, fp32) into an fp32 master arena laid out like the packed weights arena
// (weights-arena byte o <-> master byte 2*o); shared by the UNet / autoencoder / ViT handles
int dmx_master_import(const ParamTable& pt, void* masters, const char* name, const float* src, hipStream_t stream, const char* who);

struct Tn {                 // NHWC bf16 activation [B*H*W][C] with row stride ld
  bf16* p = nullptr; int B = 0, H = 0, W = 0, C = 0, ld = 0;
  // GroupNorm statistics of this tensor, written by the epilogue of the GEMM that produced it (GemmArgs.colstats / HaloConvArgs.colstats): [B][C][4]
  // DmxStat records (common.h); null when the producer could not emit them (split-K plans, fp32 mode, ...)
  const long long* cst = nullptr;
  // >= 0: the tensor is the output of a split-K GEMM whose reduce pass is still pending (Exec::pend_[pend]): its memory is written by the GroupNorm that
  // reads it first (fused: norm.hip GroupNormArgs.red_*) or by Exec::flush
  int pend = -1;
  int rows() const { return B * H * W; }
};
// a split-K GEMM launched without its reduce pass (ConvOpts.defer): what the reduce needs, and the partial planes in the workspace
struct PendRed { GemmArgs a; void* wsp = nullptr; bool done = false; };

// Debug taps: block outputs of a forward pass copied out as NCHW fp32 (tests compare them with the oracle's per-block tensors)
struct TapSink {
  float* buf = nullptr; size_t cap = 0, used = 0;     // floats
  int n = 0; int shape[16][4];                         // (B, C, H, W) of tap i, written back to back in `buf`
};

struct ConvOpts {
  int ksize = 3, stride = 1, pad = 1, ups = 0;
  const float* bias2 = nullptr;   // fp32 validation mode only: a second bias vector (the raw conv_shortcut bias; the product path folds it)
  int ldw = 0;                    // weight row stride when it is not K (fp32 validation mode: the K-padded conv_in matrix)
  int ups2 = 0;             // with ups: `w` holds the [4][Cout][4*Cin] phase weights (GemmArgs.ups2), not the 3x3 taps
  const float* bias = nullptr;
  const float* rowbias = nullptr; int ldrb = 0;
  const Tn* res = nullptr;
  const Tn* sc0 = nullptr; const Tn* sc1 = nullptr;   // fused 1x1 shortcut sources
  int out_f32 = 0;
  int stats = 0;            // 1: emit the GroupNorm statistics of the output when the plan allows it (its next consumer is a GroupNorm)
  int defer = 0;            // 1: the caller promises that the output's FIRST consumer is Exec::groupnorm / conv_gn (as x0): a split-K plan then leaves its
                            // reduce pass to that GroupNorm (one launch and one round trip of the tensor less); any other consumer must Exec::flush first
};

class Exec {
 public:
  hipStream_t stream = nullptr;
  bool dry = false;
  int rc = 0;
  Workspace ws;
  // fp32 VALIDATION mode (ref_f32.hip): activations are floats (Tn::p points to float data), weights come from the fp32
  // master arena (callers pass float pointers typed as bf16*), every op runs the plain fp32 kernel.  Tests only.
  bool f32 = false;
  TapSink* taps = nullptr;
  size_t esz() const { return f32 ? 4 : 2; }
  // column offset inside an activation row, in the active element type
  bf16* col(const Tn& t, int c) const { return (bf16*)((char*)t.p + (size_t)c * esz()); }
  void tap(const Tn& t);           // copy a block output to the tap sink (no-op without one)

  Tn make(int B, int H, int W, int C) {
    Tn t; t.B = B; t.H = H; t.W = W; t.C = C; t.ld = C;
    t.p = (bf16*)ws.alloc((size_t)B * H * W * C * esz());
    if (ws.failed() && !rc) { dmx_set_error("workspace too small"); rc = DMX_ERR_WORKSPACE; }
    return t;
  }
  void* raw(size_t bytes) {
    void* p = ws.alloc(bytes);
    if (ws.failed() && !rc) { dmx_set_error("workspace too small"); rc = DMX_ERR_WORKSPACE; }
    return p;
  }
  // Persistent stream-K GEMMs (gemm.hip) need one zeroed int per block.  The first such GEMM of a forward takes a pool from
  // the workspace and zeroes it with ONE memset node; every launch gets its own slice (a launch never reuses flags).
  int* flag_pool = nullptr; size_t flag_cap = 0, flag_used = 0;
  // Statistics slices (Tn::cst) come from a pool zeroed once per forward as well (the producers ADD into them).
  long long* cs_pool = nullptr; size_t cs_cap = 0, cs_used = 0;
  bool zero_pool(void* p, size_t bytes, int which);
  std::vector<PendRed> pend_;
  void flush(const Tn& t);         // run the pending reduce pass of t (no-op when there is none)
  void want_stats(GemmArgs& a, Tn& y, int rows_per_sample, int B);
  long long* stat_slice(int B, int C);
  // statistics records of a tensor whose producer emitted none (one streaming pass, conv_halo.hip dmx_colstats_launch); no-op when
  // the tensor has them, in fp32 validation mode or when the halo conv is switched off
  void ensure_stats(Tn& t);
  // y = conv3x3(SiLU(GroupNorm(x0 | x1))) [+ bias + row bias + residual | fused 1x1 shortcut] as ONE launch (conv_halo.hip) when the
  // kernel takes the problem and both sources carry statistics records; otherwise GroupNorm + conv as two ops
  Tn conv_gn(const Tn& x0, const Tn* x1, const float* gamma, const float* beta, int groups, float eps, const bf16* w, int Cout, const ConvOpts& o);
  void drop(const void* p) { ws.release(p); }
  void drop(const Tn& t) {
    if (t.pend >= 0 && t.pend < (int)pend_.size() && !pend_[t.pend].done) {      // dropped before anybody read it: the partial planes go, no reduce pass needed
      if (pend_[t.pend].wsp) ws.release(pend_[t.pend].wsp);
      pend_[t.pend].wsp = nullptr; pend_[t.pend].done = true;
    }
    // t may be the RESIDUAL a still pending reduce pass has to add (conv2 of a resnet whose input dies before the next block's GroupNorm runs): that
    // pass runs now, while t's memory is still t's.  (Offsets compare equal in the dry walk too: the same allocator decides both walks.)
    for (size_t i = 0; i < pend_.size(); ++i)
      if (!pend_[i].done && pend_[i].a.res && (const void*)pend_[i].a.res == (const void*)t.p) { Tn q; q.pend = (int)i; flush(q); }
    drop((const void*)t.p);
  }

  // y = GroupNorm(x0|x1) [SiLU]
  Tn groupnorm(const Tn& x0, const Tn* x1, const float* gamma, const float* beta, int groups, float eps, bool silu);
  // conv (3x3 / 1x1, optional stride-2, upsample, concat input, fused shortcut/residual/temb)
  Tn conv(const Tn& x0, const Tn* x1, const bf16* w, int Cout, const ConvOpts& o, void* f32_out = nullptr);
  // y[rows][N] = x[rows][K] W[N][K]^T (+bias)(+res) ; geglu -> N/2 columns
  // rowstats: if non-null, *rowstats receives a workspace buffer [tiles_n][rows][2] with per-row partial (sum, sumsq)
  // of the output (for a following folded LayerNorm); ln: statistics of x from its producer + folded vectors.
  struct LnIn { const float* stats = nullptr; int tiles = 0; const float* c1 = nullptr; const float* c2 = nullptr; float eps = 1e-5f; };
  struct RowStats { float* buf = nullptr; int tiles = 0; };
  Tn linear(const Tn& x, const bf16* w, int N, const float* bias, const Tn* res, bool geglu,
            RowStats* rowstats = nullptr, const LnIn* ln = nullptr, bool gn_stats = false);
  // y = gelu(x W^T + b), exact erf GELU in the GEMM epilogue (ViT MLP)
  Tn linear_gelu(const Tn& x, const bf16* w, int N, const float* bias);
  // generic gemm on raw pointers (swapped-role V^T projection etc.)
  void gemm_raw(const bf16* x, int ldx, int M, const bf16* w, int ldw, int N, int K, const float* bias,
                void* out, int ldo, int out_f32);
  Tn layernorm(const Tn& x, const float* gamma, const float* beta, float eps);
  // xf_chain.hip: the row-local chains of a transformer block as one launch each (mode 0 / 1); chain_ok = this shape and mode of
  // operation take them (C = 320, rows % 64 == 0, bf16 path, dmx_set_xf_chain(1))
  bool chain_ok(const Tn& x) const;
  void xf_chain(int mode, XfChainArgs& a);
  bool chain_gn_fold(const Tn& x) const;              // mode-2 chain: normalise the raw x in its operand load (x carries statistics records)
  void chain_stats(XfChainArgs& a, Tn& y);            // mode-1 chain: also emit the statistics records of its output y (when a fused GroupNorm -> conv can use them)
  // fused attention core; V row-major (LDS transpose-read path)
  // Weight prefetch plan: a dry walk of the graph records the weight ranges of its launches in order (note()); in the real walk a
  // launch asks for the ranges of the launches that FOLLOW it (peek()) and its blocks touch them at their start, so that they sit in
  // the memory-side cache when the next kernel's blocks - which walk them in lock step - ask for them (gemm.hip / attention.hip)
  struct PfPlan { std::vector<std::pair<const void*, long>> w; };
  PfPlan* plan = nullptr; bool plan_rec = false, plan_bad = false; int plan_i = 0;
  void note(const void* w, long bytes);
  void peek(const void** p, int* n, int slots);
  void attention(const bf16* q, int ldq, const bf16* k, int ldk, const bf16* v, int ldv, int kv_rows,
                 bf16* o, int ldo, int B, int H, int Sq, int Skv, float scale, bool kv_static = false);
 private:
  void run_gemm(GemmArgs& a, Tn* defer_to = nullptr);
};

// ResnetBlock2D weights (offsets into the arena) shared by the UNet and VAE graphs
struct ResW {
  int cin = 0, cout = 0, temb_off = -1; bool shortcut = false;
  size_t n1g = 0, n1b = 0, n2g = 0, n2b = 0, w1 = 0, b1 = 0, w2 = 0, b2 = 0, b2raw = 0, bscraw = 0;
};
// registers norm1/conv1/norm2/conv2/conv_shortcut keys under prefix p; the 1x1 shortcut is packed
// as extra K columns of conv2 and its bias is folded into conv2's by resnet_finalize.
void resnet_build(ParamTable& pt, ResW& r, const std::string& p, int cin, int cout);
int resnet_finalize(const ResW& r, char* arena, hipStream_t stream);
// wmul: 1 = `arena` is the packed bf16 weights arena; 2 = fp32 validation mode, `arena` is the fp32 master arena (byte offsets double)
Tn resnet_run(Exec& ex, const char* arena, const ResW& r, const Tn& x0, const Tn* x1, int groups, float eps,
              const float* tproj, int tproj_total, int wmul = 1);
