// Executor infrastructure: error state, parameter packing, workspace allocator, op wrappers.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include "exec.h"
#include <vector>

static thread_local char g_err[512] = "";
void dmx_set_error(const char* fmt, ...) {
  va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap);
}
const char* dmx_get_error() { return g_err; }
static int* g_dev_err = nullptr;        // pinned + mapped (hipHostMalloc): host and device use the same address
int* dmx_dev_err_words() {
  static bool tried = false;
  if (!tried) {
    tried = true;
    void* p = nullptr;
    if (hipHostMalloc(&p, 64, hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess && p) { memset(p, 0, 64); g_dev_err = (int*)p; }
    else (void)hipGetLastError();
  }
  return g_dev_err;
}
int dmx_poll_device_error() {
  volatile int* e = g_dev_err;
  if (!e || !e[0]) return DMX_OK;
  const int k = e[0], blk = e[2], d0 = e[3], d1 = e[4], d2 = e[5];
  if (k == DMX_DEVK_HALO_PEER)
    dmx_set_error("device error: conv3x3_gn (halo conv) block %d gave up after 40 ms waiting for the K-split slab of peer %d of tile slot %d (splits %d) - "
                  "the blocks of a tile were not co-resident (CUs taken by another stream?); the result of that launch is invalid", blk, d1, d0, d2);
  else if (k == DMX_DEVK_STREAMK_HELPER)
    dmx_set_error("device error: stream-K GEMM block %d (tile %d) gave up after 40 ms waiting for the partial sums of helper block %d - "
                  "the blocks of the launch were not co-resident; the result of that launch is invalid", blk, d0, d1);
  else if (k == DMX_DEVK_SKINNY_PEER)
    dmx_set_error("device error: weight-streaming conv (skinny.hip) block %d gave up after 40 ms waiting for the partial tile of K slice %d of output tile %d (slices %d) - "
                  "the blocks of a tile were not co-resident (CUs taken by another stream?); the result of that launch is invalid", blk, d1, d0, d2);
  else if (k == DMX_DEVK_ATTN_PEER)
    dmx_set_error("device error: attention (balanced schedule) slot %d gave up after 40 ms waiting for the partial (O, m, l) record of slot %d for query block %d (%d slots) - "
                  "the slots in front of it did not run; the result of that launch is invalid", blk, d0, d1, d2);
  else
    dmx_set_error("device error %d raised by block %d (%d, %d, %d)", k, blk, d0, d1, d2);
  // cleared so that the process may go on after handling it - code word first, the claim word LAST (behind a fence): a block that gives up while
  // the host is clearing either finds the claim still taken (its raise is dropped: the host is already reporting an error of this launch wave)
  // or wins a claim that nobody zeroes afterwards.  Clearing the claim before the code (round 5) let a raiser win the CAS and have its code
  // overwritten with 0 - claim taken, no code: every later raise of the process was dropped
  e[0] = 0;
  for (int i = 2; i < 8; ++i) e[i] = 0;
  __atomic_thread_fence(__ATOMIC_SEQ_CST);
  e[1] = 0;
  return DMX_ERR_DEVICE;
}
extern "C" int dmx_device_error(void) { return dmx_poll_device_error(); }
int dmx_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { dmx_set_error("launch of %s failed: %s", what, hipGetErrorString(e)); return DMX_ERR_HIP; }
  const int rc = dmx_poll_device_error();
  if (rc) {                                            // the record belongs to an EARLIER launch: say so, and say where it surfaced
    char first[400]; strncpy(first, dmx_get_error(), sizeof(first) - 1); first[sizeof(first) - 1] = 0;
    dmx_set_error("%s [surfaced at the launch check of %s, which itself was enqueued correctly; the walk it belongs to is abandoned]", first, what ? what : "?");
  }
  return rc;
}

// --------------------------------------------------------------------------- plan signature
// Every process-global switch that changes which kernels / plans a graph walk uses (dmx_set_*, dmx_gemm_plan_override) records its value here; the
// hash of the values is part of the key of the captured hipGraphs (unet_model.h GraphKey) and of the host mirror's workspace-size cache, so toggling
// a switch after the first forward can neither replay a graph captured under another setting nor run a walk in a workspace sized for another one -
// and switching BACK finds the graphs of the old setting again (a counter would strand them).
static int g_plan_sw[DMX_SW_COUNT] = {1, 1, 1, 1, 1, 1, 1, 0, 1, 1, 1};      // the defaults of the switches, in DmxPlanSwitch order
void dmx_plan_switch(int slot, int value) { if (slot >= 0 && slot < DMX_SW_COUNT) g_plan_sw[slot] = value; }
void dmx_plan_epoch_bump() { ++g_plan_sw[DMX_SW_OVERRIDES]; }
extern "C" int dmx_plan_epoch(void) {
  unsigned h = 2166136261u;
  for (int i = 0; i < DMX_SW_COUNT; ++i) h = (h ^ (unsigned)g_plan_sw[i]) * 16777619u;
  return (int)(h & 0x7fffffffu);
}

// --------------------------------------------------------------------------- profiler
namespace {
struct ProfRec { hipEvent_t a, b; int cls; double flops, bytes; char tag[96]; char sym[112]; };
struct ProfSym { char sym[112]; int cls; double n, ms, flops, bytes; };
std::vector<ProfSym> g_prof_syms;                     // per kernel symbol, filled by dmx_profile_end
bool g_prof_on = false;
std::vector<ProfRec> g_prof;
char g_prof_path[512] = "";
}
ProfScope::ProfScope(ProfClass c, hipStream_t s, double flops, double bytes, const char* tag) {
  if (!g_prof_on) return;
  ProfRec r; r.cls = c; r.flops = flops; r.bytes = bytes; r.tag[0] = 0; r.sym[0] = 0;
  if (tag) { strncpy(r.tag, tag, sizeof(r.tag) - 1); r.tag[sizeof(r.tag) - 1] = 0; }
  if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return;
  (void)hipEventRecord(r.a, s);
  slot = (int)g_prof.size(); g_prof.push_back(r);
  stream_ = s;
}
ProfScope::~ProfScope() { if (slot >= 0) (void)hipEventRecord(g_prof[slot].b, stream_); }

bool dmx_profile_active() { return g_prof_on; }
// the kernel symbol (rocprofv3's spelling) of the launch the innermost open ProfScope brackets: called by the launch helpers, which know
// their template arguments.  The last note of a scope wins.
void dmx_profile_note_symbol(const char* sym) {
  if (!g_prof_on || g_prof.empty() || !sym) return;
  strncpy(g_prof.back().sym, sym, sizeof(g_prof.back().sym) - 1); g_prof.back().sym[sizeof(g_prof.back().sym) - 1] = 0;
}
extern "C" int dmx_profile_begin(void) {
  for (auto& r : g_prof) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
  g_prof.clear(); g_prof_on = true; return DMX_OK;
}
// out[cls*4 + {0,1,2,3}] = launches, total ms, algorithmic flops, algorithmic bytes ; cls in ProfClass order
extern "C" int dmx_profile_end(double* out, int n_out) {
  g_prof_on = false;
  DMX_HIP(hipDeviceSynchronize());
  for (int i = 0; i < n_out; ++i) out[i] = 0.0;
  FILE* f = g_prof_path[0] ? fopen(g_prof_path, "w") : nullptr;
  if (f) fprintf(f, "class,ms,flops,bytes,tag,symbol\n");
  g_prof_syms.clear();
  for (auto& r : g_prof) {
    float ms = 0.f; (void)hipEventElapsedTime(&ms, r.a, r.b);
    if (f) fprintf(f, "%d,%.6f,%.0f,%.0f,%s,\"%s\"\n", r.cls, ms, r.flops, r.bytes, r.tag, r.sym);
    if (r.sym[0]) {
      ProfSym* e = nullptr;
      for (auto& q : g_prof_syms) if (!strcmp(q.sym, r.sym)) { e = &q; break; }
      if (!e) { ProfSym q{}; strcpy(q.sym, r.sym); q.cls = r.cls; g_prof_syms.push_back(q); e = &g_prof_syms.back(); }
      e->n += 1; e->ms += ms; e->flops += r.flops; e->bytes += r.bytes;
    }
    if (r.cls * 4 + 3 < n_out) { out[r.cls * 4] += 1; out[r.cls * 4 + 1] += ms; out[r.cls * 4 + 2] += r.flops; out[r.cls * 4 + 3] += r.bytes; }
    (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b);
  }
  if (f) fclose(f);
  g_prof.clear();
  return DMX_OK;
}
// per kernel SYMBOL of the profiled region that dmx_profile_end closed last: one line "class<TAB>launches<TAB>ms<TAB>flops<TAB>bytes<TAB>symbol" each
// (launches whose helper notes its symbol: the GEMM, halo-conv, chain and attention instances); returns the bytes written (0: buffer too small)
extern "C" size_t dmx_profile_symbols(char* buf, size_t cap) {
  size_t n = 0;
  for (auto& q : g_prof_syms) {
    const int w = snprintf(buf + n, n < cap ? cap - n : 0, "%d\t%.0f\t%.6f\t%.0f\t%.0f\t%s\n", q.cls, q.n, q.ms, q.flops, q.bytes, q.sym);
    if (w < 0 || n + (size_t)w >= cap) return 0;
    n += (size_t)w;
  }
  return n;
}
// optional: per-launch CSV written by the next dmx_profile_end (empty path disables)
extern "C" int dmx_profile_dump_path(const char* path) {
  g_prof_path[0] = 0;
  if (path) { strncpy(g_prof_path, path, sizeof(g_prof_path) - 1); g_prof_path[sizeof(g_prof_path) - 1] = 0; }
  return DMX_OK;
}

// --------------------------------------------------------------------------- ParamTable
int ParamTable::load(char* arena, const char* name, const float* src, hipStream_t stream) const {
  const ParamEntry* e = find(name);
  DMX_REQUIRE(e != nullptr, "load_param: unknown parameter '%s'", name);
  DMX_REQUIRE(arena != nullptr, "load_param: arena not bound");
  DMX_REQUIRE(src != nullptr, "load_param: null source for '%s'", name);
  const PackRule& r = e->rule;
  switch (r.kind) {
    case PackRule::COPY_F32:
      DMX_HIP(hipMemcpyAsync(arena + r.dst, src, (size_t)r.rows * 4, hipMemcpyDeviceToDevice, stream));
      return DMX_OK;
    case PackRule::CONV:
      return dmx_pack_conv_weight_launch(src, (bf16*)(arena + r.dst), r.rows, r.cols, r.ks, r.ld, r.koff, stream);
    case PackRule::LINEAR:
      return dmx_pack_rows_launch(src, (bf16*)(arena + r.dst), r.rows, r.cols, r.ld, 0, stream);
    case PackRule::GEGLU_W:
      return dmx_pack_rows_launch(src, (bf16*)(arena + r.dst), r.rows, r.cols, r.ld, 1, stream);
    case PackRule::GEGLU_B:
      return dmx_pack_geglu_bias_launch(src, (float*)(arena + r.dst), r.rows, stream);
  }
  return DMX_ERR_ARG;
}

// --------------------------------------------------------------------------- Workspace
void* Workspace::alloc(size_t bytes) {
  bytes = align_up(bytes ? bytes : 1, 256);
  for (auto it = free_.begin(); it != free_.end(); ++it) {
    if (it->second >= bytes) {
      const size_t off = it->first, sz = it->second;
      free_.erase(it);
      if (sz > bytes) free_[off + bytes] = sz - bytes;
      live_[off] = bytes;
      used_ += bytes;
      if (off + bytes > peak_) peak_ = off + bytes;
      return base_ + off;
    }
  }
  failed_ = true;
  return nullptr;
}
void Workspace::release(const void* p) {
  if (!p) return;
  const size_t off = (size_t)((const char*)p - base_);
  auto it = live_.find(off);
  if (it == live_.end()) return;
  size_t sz = it->second;
  live_.erase(it);
  used_ -= sz;
  size_t o = off;
  auto nxt = free_.lower_bound(o);
  if (nxt != free_.end() && o + sz == nxt->first) { sz += nxt->second; nxt = free_.erase(nxt); }
  if (nxt != free_.begin()) {
    auto prv = std::prev(nxt);
    if (prv->first + prv->second == o) { o = prv->first; sz += prv->second; free_.erase(prv); }
  }
  free_[o] = sz;
}

// --------------------------------------------------------------------------- Exec ops
// tuning aid (A/B inside one process, like dmx_gemm_plan_override): 0 switches the producer-side GroupNorm statistics off
static int g_gn_producer_stats = 1;
extern "C" int dmx_set_gn_producer_stats(int on) { const int old = g_gn_producer_stats; g_gn_producer_stats = on; dmx_plan_switch(DMX_SW_GN_STATS, on); return old; }

// tuning aid (A/B inside one process): 0 = split-K convolutions always run their own reduce pass (ConvOpts.defer ignored); 2 = only the in-block deferrals
// (conv1 -> norm2), conv2 keeps its own reduce pass
static int g_defer_reduce = 1;
extern "C" int dmx_set_defer_reduce(int on) { const int old = g_defer_reduce; g_defer_reduce = on; dmx_plan_switch(DMX_SW_DEFER, on); return old; }

static int g_halo_conv = 1;
extern "C" int dmx_set_halo_conv(int on) { const int old = g_halo_conv; g_halo_conv = on; dmx_plan_switch(DMX_SW_HALO, on); return old; }
bool dmx_halo_conv_enabled() { return g_halo_conv != 0; }

// The per-forward pools (statistics records, stream-K / halo flags) are zeroed by a KERNEL node (dmx_zero16_launch), never by
// hipMemsetAsync: a memset node of a captured hipGraph is not safe to replay on this runtime (ROCm 7.2, gfx950).  Measured in round 5
// (EXPERIMENTS.md "non-finite latents"): when a pass replays a graph that an EARLIER pass captured (the loop's buffers alternate between
// two address sets, so two instantiated graphs take turns), the memset node fills the whole 8-MB statistics pool with a 64-bit POINTER
// value (0x73ebd9840000: the address of the loop's latents buffer) instead of zeros - its fill arguments are not owned by the graph
// instance.  The producers then add onto garbage: wrong GroupNorm statistics, negative variances, NaN (BENCH_r04).
#ifdef DMX_PROBES
// probe builds only: dmx_set_pool_memset_nodes(1) goes back to memset nodes; with (3) a counting kernel before and after each memset
// node records the non-zero 16-byte units it found and a sample of what the node wrote (scripts/soak.py --pool-counts)
static int g_pool_memset_nodes = 0;
static unsigned long long* g_pool_counts = nullptr;    // [stats, flags][before, after], [4..7] sample values
extern "C" int dmx_set_pool_memset_nodes(int mode) { const int old = g_pool_memset_nodes; g_pool_memset_nodes = mode; return old; }
__global__ __launch_bounds__(256) void dmx_probe_count_nonzero_kernel(const u32x4* p, size_t n16, unsigned long long* dst) {
  unsigned long long c = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) { const u32x4 v = p[i]; c += (v.x | v.y | v.z | v.w) != 0u; }
  if (c) atomicAdd(dst, c);
  if (blockIdx.x == 0 && threadIdx.x == 0 && p[n16 - 1].x != 0u) { dst[4] = p[n16 - 1].x | ((unsigned long long)p[n16 - 1].y << 32); dst[5] = p[n16 / 2].x | ((unsigned long long)p[n16 / 2].y << 32); }   // units no producer ever adds to
}
extern "C" int dmx_probe_pool_counts(unsigned long long* host8, int reset) {
  if (!g_pool_counts) { DMX_HIP(hipMalloc(&g_pool_counts, 128)); DMX_HIP(hipMemset(g_pool_counts, 0, 128)); }
  DMX_HIP(hipDeviceSynchronize());
  if (host8) DMX_HIP(hipMemcpy(host8, g_pool_counts, 64, hipMemcpyDeviceToHost));
  if (reset) DMX_HIP(hipMemset(g_pool_counts, 0, 128));
  return DMX_OK;
}
#endif
bool Exec::zero_pool(void* p, size_t bytes, int which) {
  if (dry || rc) return !rc;
#ifdef DMX_PROBES
  if (g_pool_memset_nodes & 1) {
    const bool count = (g_pool_memset_nodes & 2) && g_pool_counts;
    if (count) hipLaunchKernelGGL(dmx_probe_count_nonzero_kernel, dim3(256), dim3(256), 0, stream, (const u32x4*)p, bytes / 16, g_pool_counts + (which - 1) * 2);
    if (hipMemsetAsync(p, 0, bytes, stream) != hipSuccess) { dmx_set_error("pool memset failed"); rc = DMX_ERR_HIP; return false; }
    if (count) hipLaunchKernelGGL(dmx_probe_count_nonzero_kernel, dim3(256), dim3(256), 0, stream, (const u32x4*)p, bytes / 16, g_pool_counts + (which - 1) * 2 + 1);
    return true;
  }
#endif
  (void)which;
  rc = dmx_zero16_launch(p, bytes, stream);
  return !rc;
}

// a zeroed slice of DmxStat records [B][C][4] from the per-forward pool (null when it is exhausted)
long long* Exec::stat_slice(int B, int C) {
  if (rc) return nullptr;
  if (!cs_pool) {
    cs_cap = (size_t)B * 64 * 1024;                    // 64 k channels per sample over the forward (SD2 UNet: ~42 k)
    cs_pool = (long long*)raw(cs_cap * DMX_STAT_WORDS * sizeof(long long)); cs_used = 0;
    if (!zero_pool(cs_pool, cs_cap * DMX_STAT_WORDS * sizeof(long long), 1)) return nullptr;
  }
  const size_t n = (size_t)B * C;
  if (cs_used + n > cs_cap) return nullptr;
  long long* s = cs_pool + DMX_STAT_WORDS * cs_used; cs_used += n;
  return s;
}

void Exec::want_stats(GemmArgs& a, Tn& y, int rows_per_sample, int B) {
  if (rc || f32 || !g_gn_producer_stats) return;
  a.cs_rows = rows_per_sample;
  if (!dmx_gemm_colstats_ok(a)) { a.cs_rows = 0; return; }
  a.colstats = stat_slice(B, a.N);
  if (!a.colstats) { a.cs_rows = 0; return; }                 // pool exhausted: the consumer computes its own statistics
  y.cst = a.colstats;
}

static int g_weight_prefetch = 1;
extern "C" int dmx_set_weight_prefetch(int on) { const int old = g_weight_prefetch; g_weight_prefetch = on; dmx_plan_switch(DMX_SW_PREFETCH, on); return old; }
void Exec::note(const void* w, long bytes) {
  if (!plan) return;
  if (plan_rec) plan->w.push_back({w, bytes});
  else if (plan_i >= (int)plan->w.size() || plan->w[plan_i].first != w) plan_bad = true;      // the real walk left the dry walk's launch order: no more ranges by index
  ++plan_i;
}
void Exec::peek(const void** p, int* n, int slots) {
  for (int i = 0; i < slots; ++i) { p[i] = nullptr; n[i] = 0; }
  if (!plan || plan_rec || plan_bad || !g_weight_prefetch) return;
#ifdef DMX_PROBES
  static const long cap = [] { const char* e = getenv("DMX_PF_CAP_MB"); long c = e ? atol(e) : 4; if (c <= 0 || c > 1024) c = 4; return c << 20; }();   // (probe builds: tuning aid)
#else
  constexpr long cap = 4L << 20;
#endif
  long left = cap;                                     // per launch: a few DMA instructions per wave, not a second weight stream
  for (int i = 0, k = plan_i; i < slots && k < (int)plan->w.size() && left > 0; ++k) {
    if (!plan->w[k].first || plan->w[k].second <= 0) continue;
    const long nb = plan->w[k].second < left ? plan->w[k].second : left;
    if (nb < 1024) break;                              // (the kernels clamp the last unit to nb - 16: never a range shorter than a unit)
    p[i] = plan->w[k].first; n[i] = (int)nb; left -= nb; ++i;
  }
}
void Exec::flush(const Tn& t) {
  if (t.pend < 0 || t.pend >= (int)pend_.size() || pend_[t.pend].done) return;
  PendRed& P = pend_[t.pend];
  if (!dry && !rc) rc = dmx_splitk_reduce_launch(P.a, stream);
  if (P.wsp) ws.release(P.wsp);
  P.wsp = nullptr; P.done = true;
}
void Exec::run_gemm(GemmArgs& a, Tn* defer_to) {
  if (rc) return;
  note(a.w, (long)a.N * a.ldw * 2 * (a.ups2 ? 4 : 1));
  peek(a.pf, a.pf_bytes, 2);
  if (const int pg = dmx_gemm_persist_blocks(a)) {
    constexpr size_t POOL = 64 * 1024;                 // ints: 256 launches of 256 blocks
    if (!flag_pool) {
      flag_pool = (int*)raw(POOL * sizeof(int)); flag_cap = POOL; flag_used = 0;
      if (!zero_pool(flag_pool, POOL * sizeof(int), 2)) return;
    }
    const size_t n = align_up((size_t)pg, 64);
    if (flag_used + n <= flag_cap) { a.flags = flag_pool + flag_used; flag_used += n; }   // else: the launcher zeroes a slice of its own workspace
  }
  const size_t wsb = dmx_gemm_workspace_bytes(a);
  void* w = wsb ? raw(wsb) : nullptr;
  int c = 0, sk = 1, ktps = 0;
  if (defer_to) dmx_gemm_plan(a, &c, &sk, &ktps);
  const bool defer = defer_to && sk > 1 && !dmx_gemm_persist_blocks(a) && !a.out_f32 && wsb > 0;      // (no pointer values in this decision: the dry walk has none)
  if (defer) a.defer_reduce = 1;
  if (!dry && !rc) rc = dmx_gemm_launch(a, w, wsb, stream);
  if (defer) {                                         // the partial planes stay in the workspace until the GroupNorm (or a flush) has read them
    PendRed P; P.a = a; P.a.partial = (float*)w; P.a.splitk = sk; P.a.kt_per_split = ktps; P.wsp = w;
    defer_to->pend = (int)pend_.size();
    pend_.push_back(P);
  } else if (w) ws.release(w);
}

Tn Exec::groupnorm(const Tn& x0, const Tn* x1, const float* gamma, const float* beta, int groups, float eps, bool silu) {
  const int C = x0.C + (x1 ? x1->C : 0);
  Tn y = make(x0.B, x0.H, x0.W, C);
  if (f32) {
    if (!dry && !rc)
      rc = dmx_groupnorm_f32_launch((const float*)x0.p, x0.ld, x0.C, x1 ? (const float*)x1->p : nullptr, x1 ? x1->ld : 0, C, groups, x0.B, x0.H * x0.W,
                                    gamma, beta, eps, silu ? 1 : 0, (float*)y.p, y.ld, stream);
    return y;
  }
  if (x1) flush(*x1);                                  // (only x0 can hand its reduce pass to this GroupNorm)
  if (x0.cst && (!x1 || x1->cst)) {                    // statistics came with the tensor(s): one apply-only launch
    flush(x0);
    if (!dry && !rc) {
      GroupNormArgs a{};
      a.x0 = x0.p; a.ldx0 = x0.ld; a.c0 = x0.C;
      a.x1 = x1 ? x1->p : nullptr; a.ldx1 = x1 ? x1->ld : 0;
      a.C = C; a.groups = groups; a.B = x0.B; a.HW = x0.H * x0.W;
      a.gamma = gamma; a.beta = beta; a.eps = eps; a.silu = silu ? 1 : 0;
      a.y = y.p; a.ldy = y.ld; a.st0 = x0.cst; a.st1 = x1 ? x1->cst : x0.cst;
      char tag[96]; snprintf(tag, sizeof(tag), "rows=%d C=%d sums", x0.rows(), C);
      ProfScope ps(PROF_GNORM, stream, 0.0, 4.0 * (double)x0.rows() * C, tag);
      rc = dmx_groupnorm_sums_launch(a, stream);
    }
    return y;
  }
  const size_t pb = dmx_gn_workspace_bytes(x0.B, x0.H * x0.W, groups);
  void* part = raw(pb);
  // x0 may be the output of a split-K GEMM whose reduce pass was left to this GroupNorm (ConvOpts.defer): the slab kernel sums the partial planes in
  // its load stage where it takes the shape, otherwise the reduce pass runs now
  PendRed* P = (x0.pend >= 0 && x0.pend < (int)pend_.size() && !pend_[x0.pend].done) ? &pend_[x0.pend] : nullptr;
  GroupNormArgs a{};
  a.x0 = x0.p; a.ldx0 = x0.ld; a.c0 = x0.C;
  a.x1 = x1 ? x1->p : nullptr; a.ldx1 = x1 ? x1->ld : 0;
  a.C = C; a.groups = groups; a.B = x0.B; a.HW = x0.H * x0.W;
  a.gamma = gamma; a.beta = beta; a.eps = eps; a.silu = silu ? 1 : 0;
  a.y = y.p; a.ldy = y.ld; a.partial = (float*)part;
  if (P) {
    a.red_partial = P->a.partial ? P->a.partial : (const float*)8;      // (dry walk: no addresses - any non-null value asks the same question)
    a.red_splitk = P->a.splitk; a.red_mn = (long long)P->a.M * P->a.N;
    a.red_bias = P->a.bias; a.red_rowbias = P->a.rowbias; a.red_ldrb = P->a.ldrb; a.red_rpg = P->a.rows_per_group;
    a.red_res = P->a.res; a.red_ldres = P->a.ldres;
    if (P->a.N != x0.C || P->a.ldo != x0.ld || !dmx_gn_red_ok(a)) {
      flush(x0); P = nullptr;
      a.red_partial = nullptr; a.red_splitk = 0;
    }
  }
  if (!dry && !rc) {
    const bool one = dmx_gn_single_launch(a);
    char tag[96]; snprintf(tag, sizeof(tag), "rows=%d C=%d%s%s", x0.rows(), C, one ? " slab" : "", P ? " + split-K reduce" : "");
    // bf16: read x + write y when the slab stays in registers, otherwise x is read twice (stats, apply)
    ProfScope ps(PROF_GNORM, stream, 0.0, (one ? 4.0 : 6.0) * (double)x0.rows() * C, tag);
    if (!rc) rc = dmx_groupnorm_launch(a, stream);
  }
  if (P) { if (P->wsp) ws.release(P->wsp); P->wsp = nullptr; P->done = true; }
  ws.release(part);
  return y;
}

Tn Exec::conv(const Tn& x0, const Tn* x1, const bf16* w, int Cout, const ConvOpts& o, void* f32_out) {
  // a source whose split-K reduce pass is still pending (ConvOpts.defer of ITS producer: the promised GroupNorm did not come first) is completed now
  flush(x0); if (x1) flush(*x1); if (o.sc0) flush(*o.sc0); if (o.sc1) flush(*o.sc1); if (o.res) flush(*o.res);
  int OH = x0.H, OW = x0.W;
  if (o.ups) { OH *= 2; OW *= 2; }
  if (o.stride > 1) { OH /= o.stride; OW /= o.stride; }          // (2 in the UNet / autoencoder; the ViT patch conv of the fp32 path: patch size)
  Tn y; y.B = x0.B; y.H = OH; y.W = OW; y.C = Cout; y.ld = Cout;
  if (!o.out_f32) y = make(x0.B, OH, OW, Cout);
  if (f32) {
    GemmF32Args a{};
    a.x0 = (const float*)x0.p; a.ldx0 = x0.ld; a.cx0 = x0.C;
    a.x1 = x1 ? (const float*)x1->p : (const float*)x0.p; a.ldx1 = x1 ? x1->ld : x0.ld;
    a.Cin = x0.C + (x1 ? x1->C : 0);
    a.ksize = o.ksize; a.stride = o.stride; a.pad = o.pad; a.ups = o.ups;
    a.direct = (o.ksize == 1 && o.stride == 1 && !o.ups) ? 1 : 0;
    a.IH = x0.H; a.IW = x0.W; a.OH = OH; a.OW = OW;
    a.Ktaps = o.ksize * o.ksize * a.Cin; a.K = a.Ktaps;
    if (o.sc0) {
      a.s0 = (const float*)o.sc0->p; a.lds0 = o.sc0->ld; a.cs0 = o.sc0->C;
      a.s1 = o.sc1 ? (const float*)o.sc1->p : a.s0; a.lds1 = o.sc1 ? o.sc1->ld : o.sc0->ld;
      a.K += o.sc0->C + (o.sc1 ? o.sc1->C : 0);
    }
    a.w = (const float*)w; a.ldw = o.ldw ? o.ldw : a.K;
    a.M = x0.B * OH * OW; a.N = Cout;
    a.bias = o.bias; a.bias2 = o.bias2; a.rowbias = o.rowbias; a.rows_per_group = OH * OW; a.ldrb = o.ldrb;
    if (o.res) { a.res = (const float*)o.res->p; a.ldres = o.res->ld; }
    a.out = o.out_f32 ? (float*)f32_out : (float*)y.p; a.ldo = Cout;
    if (!dry && !rc) rc = dmx_gemm_f32_launch(a, stream);
    return y;
  }
  GemmArgs a{};
  a.x0 = x0.p; a.ldx0 = x0.ld; a.cx0 = x0.C;
  a.x1 = x1 ? x1->p : x0.p; a.ldx1 = x1 ? x1->ld : x0.ld;
  a.Cin = x0.C + (x1 ? x1->C : 0);
  a.ksize = o.ksize; a.stride = o.stride; a.pad = o.pad; a.ups = o.ups;
  a.direct = (o.ksize == 1 && o.stride == 1 && !o.ups) ? 1 : 0;
  a.IH = x0.H; a.IW = x0.W; a.OH = OH; a.OW = OW;
  a.Ktaps = o.ksize * o.ksize * a.Cin;
  a.K = a.Ktaps;
  if (o.sc0) {
    a.s0 = o.sc0->p; a.lds0 = o.sc0->ld; a.cs0 = o.sc0->C;
    a.s1 = o.sc1 ? o.sc1->p : o.sc0->p; a.lds1 = o.sc1 ? o.sc1->ld : o.sc0->ld;
    a.K += o.sc0->C + (o.sc1 ? o.sc1->C : 0);
  }
  a.w = w; a.ldw = a.K;
  a.M = x0.B * OH * OW; a.N = Cout;
  if (o.ups && o.ups2) {                               // four 2x2 phase convolutions on the source grid
    a.ups = 0; a.ups2 = 1; a.ksize = 2; a.pad = 0;
    a.OH = x0.H; a.OW = x0.W; a.M4 = x0.rows();
    a.Ktaps = a.K = 4 * a.Cin; a.ldw = a.K; a.w_phase_stride = (long long)Cout * a.K;
  }
  a.bias = o.bias; a.rowbias = o.rowbias; a.rows_per_group = OH * OW; a.ldrb = o.ldrb;
  if (o.res) { a.res = o.res->p; a.ldres = o.res->ld; }
  a.out = o.out_f32 ? f32_out : (void*)y.p; a.ldo = Cout; a.out_f32 = o.out_f32;
  if (o.stats && !o.out_f32) want_stats(a, y, (o.ups && o.ups2) ? x0.H * x0.W : OH * OW, x0.B);
  // (o.defer == 2: the consumer is the NEXT block's GroupNorm - dmx_set_defer_reduce(2) keeps only the in-block deferrals of round 5: A/B aid)
  run_gemm(a, (o.defer && g_defer_reduce && !(o.defer == 2 && g_defer_reduce == 2) && !o.out_f32 && !a.ups2) ? &y : nullptr);
  return y;
}

bool dmx_halo_conv_enabled();
void Exec::ensure_stats(Tn& t) {
  if (rc || f32 || t.cst || !dmx_halo_conv_enabled() || !g_gn_producer_stats || (t.C & 7) || (t.ld & 7)) return;
  // only where a fused GroupNorm -> conv launch can consume them (the tile geometries of conv_halo.hip, levels where it pays)
  if (!dmx_conv_halo_wants_stats(t.H, t.W, g_halo_conv != 1)) return;
  flush(t);
  long long* st = stat_slice(t.B, t.C);
  if (!st) return;
  if (!dry && !rc) rc = dmx_colstats_launch(t.p, t.ld, t.B, t.H * t.W, t.C, st, stream);
  t.cst = st;
}

Tn Exec::conv_gn(const Tn& x0, const Tn* x1, const float* gamma, const float* beta, int groups, float eps, const bf16* w, int Cout, const ConvOpts& o) {
  HaloConvArgs a{};
  bool fused = !f32 && dmx_halo_conv_enabled() && o.ksize == 3 && o.stride == 1 && o.pad == 1 && !o.ups && !o.out_f32 && x0.cst && (!x1 || x1->cst);
  if (fused) {
    a.x0 = x0.p; a.ldx0 = x0.ld; a.cx0 = x0.C; a.Cin = x0.C + (x1 ? x1->C : 0);
    if (x1) { a.x1 = x1->p; a.ldx1 = x1->ld; }
    a.B = x0.B; a.H = x0.H; a.W = x0.W;
    a.gn = 1; a.silu = 1; a.groups = groups; a.eps = eps; a.st0 = x0.cst; a.st1 = x1 ? x1->cst : nullptr; a.gamma = gamma; a.beta = beta;
    if (o.sc0) {
      a.s0 = o.sc0->p; a.lds0 = o.sc0->ld; a.cs0 = o.sc0->C; a.Csc = o.sc0->C;
      if (o.sc1) { a.s1 = o.sc1->p; a.lds1 = o.sc1->ld; a.Csc += o.sc1->C; }
    }
    a.w = w; a.ldw = 9 * a.Cin + a.Csc; a.N = Cout;
    a.bias = o.bias; a.rowbias = o.rowbias; a.ldrb = o.ldrb;
    if (o.res) { a.res = o.res->p; a.ldres = o.res->ld; }
    a.ldo = Cout;
    fused = g_halo_conv > 1 ? dmx_conv_halo_supported(a) : dmx_conv_halo_pays(a);      // (dmx_set_halo_conv(2): wherever the kernel runs - tests, A/B)
  }
  if (!fused) {
    Tn t = groupnorm(x0, x1, gamma, beta, groups, eps, true);
    Tn y = conv(t, nullptr, w, Cout, o);
    drop(t);
    return y;
  }
  flush(x0); if (x1) flush(*x1); if (o.sc0) flush(*o.sc0); if (o.sc1) flush(*o.sc1); if (o.res) flush(*o.res);
  Tn y = make(x0.B, x0.H, x0.W, Cout);
  a.out = y.p;
  if (o.stats && g_gn_producer_stats) { a.colstats = stat_slice(x0.B, Cout); y.cst = a.colstats; }
  const int nflags = dmx_conv_halo_flag_count(a);
  if (nflags) {                                        // one zeroed int per block from the pool the stream-K GEMMs use (zeroed once per forward)
    constexpr size_t POOL = 64 * 1024;
    if (!flag_pool) {
      flag_pool = (int*)raw(POOL * sizeof(int)); flag_cap = POOL; flag_used = 0;
      if (!zero_pool(flag_pool, POOL * sizeof(int), 2)) return y;
    }
    const size_t n = align_up((size_t)nflags, 64);
    if (flag_used + n <= flag_cap) { a.flags = flag_pool + flag_used; flag_used += n; }   // else: the launcher zeroes a slice of its own workspace
  }
  const size_t wsb = dmx_conv_halo_workspace_bytes(a);
  void* wsp = wsb ? raw(wsb) : nullptr;
  note(a.w, (long)a.N * a.ldw * 2);
  peek(a.pf, a.pf_bytes, 2);
  if (!dry && !rc) rc = dmx_conv_halo_launch(a, wsp, wsb, stream);
  if (wsp) ws.release(wsp);
  return y;
}

Tn Exec::linear(const Tn& x, const bf16* w, int N, const float* bias, const Tn* res, bool geglu,
                RowStats* rowstats, const LnIn* ln, bool gn_stats) {
  flush(x); if (res) flush(*res);
  const int Nout = geglu ? N / 2 : N;
  Tn y = make(x.B, x.H, x.W, Nout);
  if (f32) {                                           // (the folded-LayerNorm / row-statistics protocol is a bf16-path fusion: callers normalise explicitly)
    GemmF32Args a{};
    a.x0 = (const float*)x.p; a.x1 = a.x0; a.ldx0 = a.ldx1 = x.ld; a.cx0 = a.Cin = x.C; a.direct = 1; a.ksize = 1; a.stride = 1;
    a.Ktaps = a.K = x.C; a.w = (const float*)w; a.ldw = x.C; a.M = x.rows(); a.N = N; a.bias = bias; a.rows_per_group = 1;
    if (res) { a.res = (const float*)res->p; a.ldres = res->ld; }
    a.out = (float*)y.p; a.ldo = Nout; a.geglu = geglu ? 1 : 0;
    if (!dry && !rc) rc = dmx_gemm_f32_launch(a, stream);
    return y;
  }
  GemmArgs a{};
  a.x0 = x.p; a.x1 = x.p; a.ldx0 = x.ld; a.ldx1 = x.ld; a.cx0 = x.C; a.Cin = x.C;
  a.direct = 1; a.ksize = 1; a.stride = 1; a.IH = a.OH = x.H; a.IW = a.OW = x.W;
  a.Ktaps = x.C; a.K = x.C;
  a.w = w; a.ldw = x.C; a.M = x.rows(); a.N = N;
  a.bias = bias; a.rows_per_group = 1;
  if (res) { a.res = res->p; a.ldres = res->ld; }
  a.out = y.p; a.ldo = Nout; a.geglu = geglu ? 1 : 0;
  if (ln) { a.ln_stats = ln->stats; a.ln_tiles = ln->tiles; a.ln_c1 = ln->c1; a.ln_c2 = ln->c2; a.ln_C = x.C; a.ln_eps = ln->eps; }
  if (rowstats) {
    a.rowstats_out = (float*)8;                      // non-null marker so the plan is the one the launch will use
    rowstats->tiles = dmx_gemm_tiles_n(a);
    rowstats->buf = (float*)raw((size_t)rowstats->tiles * a.M * 2 * sizeof(float));
    a.rowstats_out = rowstats->buf;
  }
  if (gn_stats && !geglu) want_stats(a, y, x.H * x.W, x.B);
  run_gemm(a);
  return y;
}

void Exec::gemm_raw(const bf16* x, int ldx, int M, const bf16* w, int ldw, int N, int K, const float* bias,
                    void* out, int ldo, int out_f32) {
  GemmArgs a{};
  a.x0 = x; a.x1 = x; a.ldx0 = ldx; a.ldx1 = ldx; a.cx0 = K; a.Cin = K;
  a.direct = 1; a.ksize = 1; a.stride = 1; a.IH = a.OH = 1; a.IW = a.OW = M;
  a.Ktaps = K; a.K = K; a.w = w; a.ldw = ldw; a.M = M; a.N = N;
  a.bias = bias; a.rows_per_group = 1; a.out = out; a.ldo = ldo; a.out_f32 = out_f32;
  run_gemm(a);
}

Tn Exec::linear_gelu(const Tn& x, const bf16* w, int N, const float* bias) {
  flush(x);
  Tn y = make(x.B, x.H, x.W, N);
  if (f32) {
    GemmF32Args a{};
    a.x0 = (const float*)x.p; a.x1 = a.x0; a.ldx0 = a.ldx1 = x.ld; a.cx0 = a.Cin = x.C; a.direct = 1; a.ksize = 1; a.stride = 1;
    a.Ktaps = a.K = x.C; a.w = (const float*)w; a.ldw = x.C; a.M = x.rows(); a.N = N; a.bias = bias; a.rows_per_group = 1;
    a.out = (float*)y.p; a.ldo = N; a.act = 1;
    if (!dry && !rc) rc = dmx_gemm_f32_launch(a, stream);
    return y;
  }
  GemmArgs a{};
  a.x0 = x.p; a.x1 = x.p; a.ldx0 = x.ld; a.ldx1 = x.ld; a.cx0 = x.C; a.Cin = x.C;
  a.direct = 1; a.ksize = 1; a.stride = 1; a.IH = a.OH = x.H; a.IW = a.OW = x.W;
  a.Ktaps = x.C; a.K = x.C;
  a.w = w; a.ldw = x.C; a.M = x.rows(); a.N = N;
  a.bias = bias; a.rows_per_group = 1;
  a.out = y.p; a.ldo = N; a.act = 1;
  run_gemm(a);
  return y;
}

Tn Exec::layernorm(const Tn& x, const float* gamma, const float* beta, float eps) {
  flush(x);
  Tn y = make(x.B, x.H, x.W, x.C);
  if (f32) {
    if (!dry && !rc) rc = dmx_layernorm_f32_launch((const float*)x.p, x.ld, (float*)y.p, y.ld, gamma, beta, x.rows(), x.C, eps, stream);
    return y;
  }
  if (!dry && !rc) {
    char tag[96]; snprintf(tag, sizeof(tag), "rows=%d C=%d", x.rows(), x.C);
    ProfScope ps(PROF_LNORM, stream, 0.0, 4.0 * (double)x.rows() * x.C, tag);
    rc = dmx_layernorm_launch(x.p, x.ld, y.p, y.ld, gamma, beta, x.rows(), x.C, eps, stream);
  }
  return y;
}

static int g_xf_chain = 1;
extern "C" int dmx_set_xf_chain(int on) { const int old = g_xf_chain; g_xf_chain = on; dmx_plan_switch(DMX_SW_XF_CHAIN, on); return old; }
bool Exec::chain_ok(const Tn& x) const { return (g_xf_chain & 3) && !f32 && x.ld == x.C && ((g_xf_chain & 3) > 1 ? dmx_xf_chain_supported(x.rows(), x.C) : dmx_xf_chain_pays(x.rows(), x.C)); }
// (dmx_set_xf_chain bit 2: chains without the folded entry GroupNorm - A/B aid)
bool Exec::chain_gn_fold(const Tn& x) const { return !(g_xf_chain & 4) && !f32 && x.cst != nullptr && (x.H * x.W) % 64 == 0 && x.ld == x.C; }
void Exec::chain_stats(XfChainArgs& a, Tn& y) {
  if (rc || f32 || !dmx_halo_conv_enabled() || !g_gn_producer_stats || (y.H * y.W) % 64) return;
  if (!dmx_conv_halo_wants_stats(y.H, y.W, g_halo_conv != 1)) return;
  a.colstats = stat_slice(y.B, y.C);
  if (!a.colstats) return;
  a.cs_rows = y.H * y.W; y.cst = a.colstats;
}
void Exec::xf_chain(int mode, XfChainArgs& a) {
  const long cc = (long)a.C * a.C * 2;
  note(a.w0, cc);
  if (mode == 0) note(a.w1, cc);
  else if (mode == 2) note(a.w1, 3 * cc);
  else { note(a.wf1, 8 * cc); note(a.wf2, 4 * cc); note(a.wpo, cc); }
  peek(a.pf, a.pf_bytes, 4);
  if (dry || rc) return;
  rc = dmx_xf_chain_launch(a, mode, stream);
}

void Exec::attention(const bf16* q, int ldq, const bf16* k, int ldk, const bf16* v, int ldv, int kv_rows,
                     bf16* o, int ldo, int B, int H, int Sq, int Skv, float scale, bool kv_static) {
  // (kv_static: K | V live in the per-image context cache, written long before this launch - cold, and every block of a head walks them
  // in lock step: they join the prefetch plan like a weight matrix.  One interleaved buffer: ldk elements per key row.)
#ifdef DMX_PROBES
  static const int kv_pf = [] { const char* e = getenv("DMX_PF_KV"); return e ? atoi(e) : 1; }();      // (probe builds: tuning aid)
#else
  constexpr int kv_pf = 1;
#endif
  if (kv_static && kv_pf) note(k, (long)B * kv_rows * ldk * 2);
  // balanced schedule (attention_sk.hip) where its plan takes the problem: partial records in the workspace, flags from the zeroed pool
  AttnArgs a{};
  a.q = q; a.ldq = ldq; a.k = k; a.ldk = ldk; a.kv_rows = kv_rows; a.v = v; a.ldv = ldv;
  a.o = o; a.ldo = ldo; a.B = B; a.H = H; a.Sq = Sq; a.Skv = Skv; a.scale = scale;
  const int sk_slots = f32 ? 0 : dmx_attention_balanced_slots(a);
  void* sk_ws = nullptr;
  if (sk_slots) {
    constexpr size_t POOL = 64 * 1024;
    if (!flag_pool) {
      flag_pool = (int*)raw(POOL * sizeof(int)); flag_cap = POOL; flag_used = 0;
      if (!zero_pool(flag_pool, POOL * sizeof(int), 2)) return;
    }
    const size_t n = align_up((size_t)sk_slots, 64);
    const size_t pb = dmx_attention_balanced_part_bytes(a);
    if (flag_used + n <= flag_cap) { a.sk_flags = flag_pool + flag_used; flag_used += n; sk_ws = raw(pb); a.sk_part = (float*)sk_ws; }
    // (pool exhausted: the plain grid - the same decision in the dry walk, which counts the same slices)
  }
  struct Rel { Workspace& w; void* p; ~Rel() { if (p) w.release(p); } } rel{ws, sk_ws};
  if (dry || rc) return;
  if (f32) {
    rc = dmx_attention_f32_launch((const float*)q, ldq, (const float*)k, ldk, (const float*)v, ldv, kv_rows, (float*)o, ldo, B, H, Sq, Skv, scale, stream);
    return;
  }
  {
#ifdef DMX_PROBES      // (probe builds, DMX_PF_ATTN=0: the attention launches issue no prefetch units - the launches in front of them already cover the next ranges)
    static const int attn_pf = [] { const char* e = getenv("DMX_PF_ATTN"); return e ? atoi(e) : 1; }();
#else
    constexpr int attn_pf = 1;
#endif
    if (attn_pf) peek(a.pf, a.pf_bytes, 4);
  }
  char tag[96]; snprintf(tag, sizeof(tag), "B=%d H=%d Sq=%d Skv=%d", B, H, Sq, Skv);
  ProfScope ps(PROF_ATTN, stream, 4.0 * B * H * (double)Sq * Skv * 64.0, 2.0 * 64.0 * B * H * (2.0 * Sq + 2.0 * Skv), tag);
  rc = a.sk_part ? dmx_attention_balanced_launch(a, stream) : dmx_attention_launch(a, stream);
}

void Exec::tap(const Tn& t) {
  if (!taps) return;
  flush(t);
  if (dry || rc) return;
  const size_t n = (size_t)t.rows() * t.C;
  if (taps->n >= 16) return;                           // (only reachable with the DMX_TAPS_FINE debugging switch)
  if (taps->used + n > taps->cap) { dmx_set_error("tap buffer too small"); rc = DMX_ERR_WORKSPACE; return; }
  float* dst = taps->buf + taps->used;
  rc = f32 ? dmx_nhwc_to_nchw_f32_launch((const float*)t.p, t.ld, dst, t.B, t.C, t.H * t.W, stream)
           : dmx_nhwc_bf16_to_nchw_f32_launch(t.p, t.ld, dst, t.B, t.C, t.H * t.W, stream);
  int* sh = taps->shape[taps->n++]; sh[0] = t.B; sh[1] = t.C; sh[2] = t.H; sh[3] = t.W;
  taps->used += n;
}

// --------------------------------------------------------------------------- ResnetBlock2D
void resnet_build(ParamTable& pt, ResW& r, const std::string& p, int cin, int cout) {
  r.cin = cin; r.cout = cout; r.shortcut = (cin != cout);
  r.n1g = pt.f32(p + "norm1.weight", cin); r.n1b = pt.f32(p + "norm1.bias", cin);
  r.w1 = pt.reserve((size_t)cout * 9 * cin * 2);
  pt.conv_at(p + "conv1.weight", cout, cin, 3, r.w1, 9 * cin, 0);
  r.b1 = pt.f32(p + "conv1.bias", cout);
  r.n2g = pt.f32(p + "norm2.weight", cout); r.n2b = pt.f32(p + "norm2.bias", cout);
  const int k2 = 9 * cout + (r.shortcut ? cin : 0);
  r.w2 = pt.reserve((size_t)cout * k2 * 2);
  pt.conv_at(p + "conv2.weight", cout, cout, 3, r.w2, k2, 0);
  if (r.shortcut) {
    r.b2raw = pt.f32(p + "conv2.bias", cout);
    pt.conv_at(p + "conv_shortcut.weight", cout, cin, 1, r.w2, k2, 9 * cout);
    r.bscraw = pt.f32(p + "conv_shortcut.bias", cout);
    r.b2 = pt.reserve((size_t)cout * 4);
  } else {
    r.b2 = pt.f32(p + "conv2.bias", cout); r.b2raw = r.b2; r.bscraw = 0;
  }
}

__global__ void dmx_add_vec_kernel(const float* a, const float* b, float* o, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) o[i] = a[i] + b[i];
}
int resnet_finalize(const ResW& r, char* arena, hipStream_t stream) {
  if (!r.shortcut) return DMX_OK;
  hipLaunchKernelGGL(dmx_add_vec_kernel, dim3(cdiv(r.cout, 256)), dim3(256), 0, stream,
                     (const float*)(arena + r.b2raw), (const float*)(arena + r.bscraw), (float*)(arena + r.b2), r.cout);
  return dmx_check_launch("dmx_add_vec_kernel");
}

Tn resnet_run(Exec& ex, const char* arena, const ResW& r, const Tn& x0, const Tn* x1, int groups, float eps,
              const float* tproj, int tproj_total, int wmul) {
  auto F = [&](size_t off) { return (const float*)(arena + off * (size_t)wmul); };
  auto H = [&](size_t off) { return (const bf16*)(arena + off * (size_t)wmul); };
  // [GroupNorm -> SiLU -> conv3x3] twice; each pair is ONE launch where the halo conv takes it (Exec::conv_gn)
  ConvOpts o1; o1.bias = F(r.b1); o1.stats = 1;        // norm2 reads conv1's output
  o1.defer = 1;                                        // ... FIRST (and only): a split-K conv1 leaves its reduce pass to norm2
  if (r.temb_off >= 0 && tproj) { o1.rowbias = tproj + r.temb_off; o1.ldrb = tproj_total; }
  Tn t2 = ex.conv_gn(x0, x1, F(r.n1g), F(r.n1b), groups, eps, H(r.w1), r.cout, o1);
  ConvOpts o2; o2.bias = F(r.b2); o2.stats = 1;        // a GroupNorm comes next in every graph (next resnet / transformer / out norm)
  o2.defer = 2;                                        // ... and where it reads y as its x0 FIRST, a split-K conv2 leaves its reduce pass to it (any other first
                                                       // consumer - down / up-sampling conv, tap, linear - completes the tensor itself: Exec::flush in each of them)
  if (ex.f32 && r.shortcut) { o2.bias = F(r.b2raw); o2.bias2 = F(r.bscraw); }     // the folded bias is derived data: not in the master arena
  if (r.shortcut) { o2.sc0 = &x0; o2.sc1 = x1; } else { o2.res = &x0; }
  Tn y = ex.conv_gn(t2, nullptr, F(r.n2g), F(r.n2b), groups, eps, H(r.w2), r.cout, o2);
  ex.drop(t2);
  return y;
}
