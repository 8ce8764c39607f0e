// Fused AdamW over the packed fp32 arenas (reference: torch.optim.AdamW(unet.parameters(), lr, betas,
// weight_decay, eps) + accelerator.clip_grad_norm_(unet.parameters(), max_grad_norm), train_diffute_v1.py:721-727,927-930).
//
// State lives in four fp32 arenas with the layout of the gradient arena (element of the weights arena at byte offset o
// <-> byte offset 2*o): master parameters P, Adam moments M and V, gradients G.  One step is
//   1. global gradient norm over the trainable elements: per-chunk partial sums of squares (fixed order), one block folds
//      them and leaves clip = min(1, max_norm / (norm + 1e-6)) on the device (no host round trip);
//   2. one pass: g *= clip; p *= 1 - lr*wd; m,v updated; p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps)   (torch's AdamW);
//      the new parameter is also written to the weights arena in its compute type (bf16 weights, fp32 norm/bias vectors),
//      so no re-pack of 866 M parameters follows;
//   3. the derived data of the inference graph (folded LayerNorm copies, folded shortcut biases) and the transposed
//      weights of the training graph are refreshed by the caller (dmx_unet_refresh_derived, dmx_unet_train_prepare).
// The work list is a table of <= 64 Ki-element chunks of the trainable ranges built once from the parameter table.
#include <algorithm>
#include <vector>
#include "unet_model.h"
#include "vae_model.h"

namespace {
struct OptChunk { unsigned long long index; unsigned int count; unsigned int is_bf16; unsigned long long arena_off; };
constexpr unsigned CHUNK = 65536;

std::vector<OptChunk> build_chunks(const dmx_unet* u) {
  struct R { size_t lo, hi; int bf; size_t dst; };
  std::vector<R> rs;
  for (const ParamEntry& e : u->pt.entries()) {
    const PackRule& r = e.rule;
    size_t lo = 2 * r.dst, n = 0; int bf = 1;
    switch (r.kind) {
      case PackRule::COPY_F32: case PackRule::GEGLU_B: n = (size_t)r.rows * 4; bf = 0; break;
      case PackRule::LINEAR: case PackRule::GEGLU_W: n = ((size_t)(r.rows - 1) * r.ld + r.cols) * 4; break;
      case PackRule::CONV: lo += (size_t)r.koff * 4; n = ((size_t)(r.rows - 1) * r.ld + (size_t)r.ks * r.ks * r.cols) * 4; break;
    }
    rs.push_back({lo, lo + n, bf, r.dst});
  }
  std::sort(rs.begin(), rs.end(), [](const R& a, const R& b) { return a.lo < b.lo; });
  std::vector<OptChunk> out;
  size_t covered = 0;
  for (const R& r : rs) {
    size_t lo = std::max(r.lo, covered), hi = r.hi;
    if (lo >= hi) continue;
    covered = hi;
    for (size_t b = lo; b < hi; b += (size_t)CHUNK * 4) {
      OptChunk c; c.index = b / 4; c.count = (unsigned)(std::min(hi, b + (size_t)CHUNK * 4) - b) / 4; c.is_bf16 = (unsigned)r.bf;
      // gradient byte b <-> weights-arena byte: bf16 entries b/2; an fp32 entry at arena byte dst keeps 4-byte elements: dst + (b - 2*dst)
      c.arena_off = r.bf ? b / 2 : r.dst + (b - 2 * r.dst);
      out.push_back(c);
    }
  }
  return out;
}

__global__ __launch_bounds__(256) void dmx_sqnorm_part_kernel(const OptChunk* tab, const float* g, float* part) {
  __shared__ float red[256];
  const OptChunk c = tab[blockIdx.x];
  float s = 0.f;
  for (unsigned i = threadIdx.x; i < c.count; i += 256) { const float v = g[c.index + i]; s += v * v; }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int k = 128; k >= 1; k >>= 1) { if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k]; __syncthreads(); }
  if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}
// scal[0] = ||g|| (of the UNSCALED gradient g * inv_scale), scal[1] = the factor the update multiplies g by (clip coefficient * inv_scale);
// with `found` (loss-scaled fp16 training, torch GradScaler's contract): scal[2] = 1 when the gradient holds an inf / NaN - the update
// kernel then leaves every arena untouched - else 0
__global__ __launch_bounds__(1024) void dmx_clip_coef_kernel(const float* part, int n, float max_norm, float* scal, float inv_scale, int found) {
  __shared__ double red[1024];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 1024) s += (double)part[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int k = 512; k >= 1; k >>= 1) { if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k]; __syncthreads(); }
  if (threadIdx.x == 0) {
    const float norm = (float)sqrt(red[0]) * inv_scale;
    scal[0] = norm;
    float c = 1.0f;
    if (max_norm > 0.f) { c = max_norm / (norm + 1e-6f); if (c > 1.0f) c = 1.0f; }     // torch.nn.utils.clip_grad_norm_
    scal[1] = c * inv_scale;
    if (found) scal[2] = (norm - norm == 0.f) ? 0.f : 1.f;          // (inf - inf and NaN - NaN are NaN)
  }
}
__global__ __launch_bounds__(256) void dmx_adamw_kernel(const OptChunk* tab, float* p, float* m, float* v, const float* g, char* arena,
                                                        const float* scal, float lr, float b1, float b2, float eps, float wd, float bc1, float bc2_sqrt,
                                                        float* ema, float ema_omd, int found) {
  const OptChunk c = tab[blockIdx.x];
  if (found && scal[2] != 0.f) return;                         // GradScaler.step: an overflowed step is skipped as a whole
  const float clip = scal[1];
  const float step_size = lr / bc1;
  for (unsigned i = threadIdx.x; i < c.count; i += 256) {
    const size_t j = c.index + i;
    const float gr = g[j] * clip;
    float pv = p[j] * (1.0f - lr * wd);
    const float mv = m[j] * b1 + gr * (1.0f - b1);          // exp_avg.lerp_(grad, 1 - beta1)
    const float vv = v[j] * b2 + gr * gr * (1.0f - b2);
    const float denom = sqrtf(vv) / bc2_sqrt + eps;
    pv -= step_size * (mv / denom);
    p[j] = pv; m[j] = mv; v[j] = vv;
    if (ema) { const float e = ema[j]; ema[j] = e - ema_omd * (e - pv); }          // EMAModel.step: s.sub_((1 - decay) * (s - p))
    if (c.is_bf16) ((unsigned short*)(arena + c.arena_off))[i] = f2bf_bits(pv);
    else ((float*)(arena + c.arena_off))[i] = pv;
  }
}

// torch-layout fp32 parameter -> packed fp32 (the inverse of dmx_grad_unpack_kernel in unet_train.hip)
__global__ __launch_bounds__(256) void dmx_master_pack_kernel(const float* src, float* g, int kind, int rows, int cols, int ks, int ld, int koff) {
  const size_t total = (kind == 1) ? (size_t)rows * cols * ks * ks : (size_t)rows * (cols > 0 ? cols : 1);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    if (kind == 0) {
      g[i] = src[i];
    } else if (kind == 1) {
      const int kk = ks * ks;
      const int tap = (int)(i % kk); const size_t r = i / kk;
      const int ci = (int)(r % cols); const int n = (int)(r / cols);
      g[(size_t)n * ld + koff + (size_t)tap * cols + ci] = src[i];
    } else if (kind == 2) {
      const int c = (int)(i % cols); const size_t r = i / cols;
      g[r * ld + c] = src[i];
    } else {
      const int c = (kind == 3) ? (int)(i % cols) : 0;
      const int r = (kind == 3) ? (int)(i / cols) : (int)i;
      const int J = r >> 6, w = r & 63;
      const int s0 = (w < 32) ? (32 * J + w) : (rows / 2 + 32 * J + (w - 32));
      if (kind == 3) g[(size_t)r * ld + c] = src[(size_t)s0 * cols + c];
      else g[r] = src[s0];
    }
  }
}
}  // namespace

extern "C" int dmx_unet_optim_chunks(const dmx_unet* u) { return u ? (int)build_chunks(u).size() : 0; }
extern "C" size_t dmx_unet_optim_table_bytes(const dmx_unet* u) { return u ? build_chunks(u).size() * sizeof(OptChunk) : 0; }
// number of trainable elements the table covers (row paddings of packed matrices included)
extern "C" size_t dmx_unet_optim_elements(const dmx_unet* u) {
  size_t n = 0; if (u) for (const OptChunk& c : build_chunks(u)) n += c.count; return n;
}
extern "C" int dmx_unet_optim_table(const dmx_unet* u, void* table_dev, size_t bytes, dmx_stream_t stream) {
  DMX_REQUIRE(u && table_dev, "unet_optim_table: null argument");
  const std::vector<OptChunk> t = build_chunks(u);
  DMX_REQUIRE(bytes >= t.size() * sizeof(OptChunk), "unet_optim_table: need %zu bytes", t.size() * sizeof(OptChunk));
  DMX_HIP(hipMemcpyAsync(table_dev, t.data(), t.size() * sizeof(OptChunk), hipMemcpyHostToDevice, (hipStream_t)stream));
  DMX_HIP(hipStreamSynchronize((hipStream_t)stream));      // `t` is a host temporary
  return DMX_OK;
}

// master[name] <- src (fp32, torch layout)
int dmx_master_import(const ParamTable& pt, void* masters, const char* name, const float* src, hipStream_t stream, const char* who) {
  const ParamEntry* e = pt.find(name);
  DMX_REQUIRE(e != nullptr, "%s: unknown parameter %s", who, name);
  const PackRule& r = e->rule;
  float* g = (float*)((char*)masters + 2 * r.dst);
  int kind = 0, rows = r.rows, cols = r.cols;
  switch (r.kind) {
    case PackRule::COPY_F32: kind = 0; cols = 0; break;
    case PackRule::CONV: kind = 1; break;
    case PackRule::LINEAR: kind = 2; break;
    case PackRule::GEGLU_W: kind = 3; break;
    case PackRule::GEGLU_B: kind = 4; cols = 0; break;
  }
  const size_t total = (kind == 1) ? (size_t)rows * cols * r.ks * r.ks : (size_t)rows * (cols > 0 ? cols : 1);
  int blocks = (int)((total + 255) / 256); if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(dmx_master_pack_kernel, dim3(blocks), dim3(256), 0, stream, src, g, kind, rows, cols, r.ks, r.ld, r.koff);
  return dmx_check_launch("dmx_master_pack_kernel");
}
extern "C" int dmx_unet_master_import(const dmx_unet* u, void* masters, const char* name, const float* src, dmx_stream_t stream) {
  DMX_REQUIRE(u && masters && name && src, "unet_master_import: null argument");
  return dmx_master_import(u->pt, masters, name, src, (hipStream_t)stream, "unet_master_import");
}
// the autoencoder's fp32 master copy (same layout rule: weights-arena byte o <-> master byte 2*o); used by the fp32
// validation path dmx_vae_encode_f32 / dmx_vae_decode_f32
extern "C" int dmx_vae_master_import(const dmx_vae* v, void* masters, const char* name, const float* src, dmx_stream_t stream) {
  DMX_REQUIRE(v && masters && name && src, "vae_master_import: null argument");
  return dmx_master_import(v->pt, masters, name, src, (hipStream_t)stream, "vae_master_import");
}

// scalars: device float[2] = (gradient norm before clipping, clip coefficient); workspace: nchunks floats.
// ema (optional, fp32 arena with the masters' layout): shadow parameters updated in the same pass with
// ema -= (1 - ema_decay) * (ema - p_new)   (diffusers EMAModel.step, the reference's `ema_unet.step(unet.parameters())`,
// train_diffute_v1.py:934-935); the caller owns the decay schedule.
// grad_inv_scale / check_finite (dmx_unet_adamw_step_scaled): the gradient arena holds grad * loss_scale (accelerate's fp16 mixed precision,
// `--mixed_precision fp16`, train_diffute_v1.py:267,583: GradScaler.scale(loss).backward()); norm, clipping and the update use
// g * grad_inv_scale, scalars is float[3] and scalars[2] = 1 marks a gradient with inf / NaN - that step changes nothing (GradScaler.step).
static int adamw_step_(dmx_unet* u, const void* table_dev, int nchunks, void* masters, void* exp_avg, void* exp_avg_sq, const void* grads,
                       float lr, float beta1, float beta2, float eps, float weight_decay, int step, float max_grad_norm,
                       float* scalars, void* workspace, size_t workspace_bytes, void* ema, float ema_decay, float grad_inv_scale, int check_finite,
                       dmx_stream_t stream) {
  DMX_REQUIRE(u && u->arena && table_dev && masters && exp_avg && exp_avg_sq && grads && scalars, "unet_adamw_step: null argument");
  DMX_REQUIRE(nchunks > 0 && step >= 1, "unet_adamw_step: bad chunk count / step");
  DMX_REQUIRE(grad_inv_scale > 0.f && grad_inv_scale - grad_inv_scale == 0.f, "unet_adamw_step: grad_inv_scale must be positive and finite");
  DMX_REQUIRE(workspace && workspace_bytes >= (size_t)nchunks * sizeof(float), "unet_adamw_step: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  const OptChunk* tab = (const OptChunk*)table_dev;
  hipLaunchKernelGGL(dmx_sqnorm_part_kernel, dim3(nchunks), dim3(256), 0, s, tab, (const float*)grads, (float*)workspace);
  int rc = dmx_check_launch("dmx_sqnorm_part_kernel");
  if (rc) return rc;
  hipLaunchKernelGGL(dmx_clip_coef_kernel, dim3(1), dim3(1024), 0, s, (const float*)workspace, nchunks, max_grad_norm, scalars, grad_inv_scale, check_finite);
  rc = dmx_check_launch("dmx_clip_coef_kernel");
  if (rc) return rc;
  const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
  hipLaunchKernelGGL(dmx_adamw_kernel, dim3(nchunks), dim3(256), 0, s, tab, (float*)masters, (float*)exp_avg, (float*)exp_avg_sq, (const float*)grads,
                     u->arena, (const float*)scalars, lr, beta1, beta2, eps, weight_decay, (float)bc1, (float)sqrt(bc2),
                     (float*)ema, 1.0f - ema_decay, check_finite);
  rc = dmx_check_launch("dmx_adamw_kernel");
  if (rc) return rc;
  u->drop_graphs();                      // captured inference graphs hold no weights, but folded copies are stale until refreshed
  return DMX_OK;
}
extern "C" int dmx_unet_adamw_step(dmx_unet* u, const void* table_dev, int nchunks, void* masters, void* exp_avg, void* exp_avg_sq, const void* grads,
                                   float lr, float beta1, float beta2, float eps, float weight_decay, int step, float max_grad_norm,
                                   float* scalars, void* workspace, size_t workspace_bytes, void* ema, float ema_decay, dmx_stream_t stream) {
  return adamw_step_(u, table_dev, nchunks, masters, exp_avg, exp_avg_sq, grads, lr, beta1, beta2, eps, weight_decay, step, max_grad_norm, scalars,
                     workspace, workspace_bytes, ema, ema_decay, 1.0f, 0, stream);
}
extern "C" int dmx_unet_adamw_step_scaled(dmx_unet* u, const void* table_dev, int nchunks, void* masters, void* exp_avg, void* exp_avg_sq,
                                          const void* grads, float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                                          float max_grad_norm, float* scalars, void* workspace, size_t workspace_bytes, void* ema, float ema_decay,
                                          float grad_inv_scale, dmx_stream_t stream) {
  return adamw_step_(u, table_dev, nchunks, masters, exp_avg, exp_avg_sq, grads, lr, beta1, beta2, eps, weight_decay, step, max_grad_norm, scalars,
                     workspace, workspace_bytes, ema, ema_decay, grad_inv_scale, 1, stream);
}
