// Glyph encoder: the ViT encoder of TrOCR (reference call sites app.ipynb:773-776,
// train_diffute_v1.py:868-871: `trocr_model(pixel_values).last_hidden_state` -> [B, 577, 1024], fed to the UNet as
// encoder_hidden_states).  Module structure = the public transformers ViTModel (pre-LayerNorm blocks):
//   x = [cls | patch_conv16x16/16(pixels)] + position_embeddings
//   24 x { x += out(attn(qkv(LN_before(x)))) ; x += fc2(gelu(fc1(LN_after(x)))) } ; last_hidden_state = LN(x)
// Every op is one of the gfx950 kernels of the UNet path: patch embedding = im2col (NCHW fp32 -> [B*576][768] bf16) + GEMM,
// q|k|v one GEMM (N = 3D), flash attention d=64 (16 heads, S = 577, ragged tail), fc1 with the exact-erf GELU in the GEMM
// epilogue, residual adds in the GEMM epilogues, LayerNorm kernel (eps 1e-12).
#include <memory>
#include <string>
#include <vector>
#include "exec.h"
#include "../../include/diffute_hip.h"

namespace {
struct VitLayer { size_t l1g, l1b, wqkv, bqkv, wo, bo, l2g, l2b, w1, b1, w2, b2; };

__global__ __launch_bounds__(256) void dmx_vit_assemble_kernel(const bf16* emb, const float* cls, const float* pos, bf16* x, int B, int NP, int D) {
  // x[b][0] = cls + pos[0] ; x[b][1+i] = emb[b][i] + pos[1+i]
  const int d8 = D / 8;
  const size_t total = (size_t)B * (NP + 1) * d8;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % d8) * 8; size_t r = i / d8;
    const int tkn = (int)(r % (NP + 1)); const int b = (int)(r / (NP + 1));
    float v[8];
    if (tkn == 0) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = cls[c + e];
    } else {
      unpack_bf8(*(const u32x4*)(emb + ((size_t)b * NP + (tkn - 1)) * D + c), v);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] += pos[(size_t)tkn * D + c + e];
    *(u32x4*)(x + r * D + c) = pack_bf8(v);
  }
}
__global__ __launch_bounds__(256) void dmx_vit_assemble_f32_kernel(const float* emb, const float* cls, const float* pos, float* x, int B, int NP, int D) {
  const size_t total = (size_t)B * (NP + 1) * D;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % D); const size_t r = i / D;
    const int tkn = (int)(r % (NP + 1)); const int b = (int)(r / (NP + 1));
    x[i] = (tkn == 0 ? cls[c] : emb[((size_t)b * NP + (tkn - 1)) * D + c]) + pos[(size_t)tkn * D + c];
  }
}
__global__ __launch_bounds__(256) void dmx_bf16_to_f32_kernel(const bf16* in, float* out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    out[i] = bf_bits2f(((const unsigned short*)in)[i]);
}
}  // namespace

struct dmx_vit {
  dmx_vit_config cfg;
  ParamTable pt;
  char* arena = nullptr;
  bool finalized = false;
  size_t cls, pos, pw, pb, lng, lnb; int kpad = 0, np = 0;
  std::vector<VitLayer> layers;
  template <typename T> T* at(size_t off) const { return (T*)(arena + off); }
  // fp32 VALIDATION mode (dmx_vit_forward_f32, tests only): parameters from the caller's fp32 master arena (byte offsets double)
  const char* masters_f32 = nullptr;
  template <typename T> const T* W(size_t off) const { return masters_f32 ? (const T*)(masters_f32 + 2 * off) : (const T*)(arena + off); }
};

extern "C" dmx_vit* dmx_vit_create(const dmx_vit_config* cfg) {
  if (!cfg) { dmx_set_error("vit_create: null config"); return nullptr; }
  const int D = cfg->hidden_size, H = cfg->num_heads;
  if (D % 64 != 0 || H <= 0 || D / H != 64) { dmx_set_error("vit_create: hidden_size=%d heads=%d: head dim must be 64", D, H); return nullptr; }
  if (cfg->image_size % cfg->patch_size != 0 || cfg->intermediate_size % 64 != 0) { dmx_set_error("vit_create: bad image/patch/intermediate size"); return nullptr; }
  auto v = std::make_unique<dmx_vit>();
  v->cfg = *cfg;
  ParamTable& pt = v->pt;
  const int g = cfg->image_size / cfg->patch_size; v->np = g * g;
  const int C = cfg->num_channels, P = cfg->patch_size, I = cfg->intermediate_size;
  auto f32n = [&](const std::string& name, std::initializer_list<int> shape, int n) {
    PackRule r; r.kind = PackRule::COPY_F32; r.dst = pt.reserve((size_t)n * 4); r.rows = n; pt.add(name, shape, r); return r.dst;
  };
  v->cls = f32n("embeddings.cls_token", {1, 1, D}, D);
  v->pos = f32n("embeddings.position_embeddings", {1, v->np + 1, D}, (v->np + 1) * D);
  v->kpad = (int)align_up((size_t)P * P * C, 64);
  v->pw = pt.reserve((size_t)D * v->kpad * 2);
  pt.conv_at("embeddings.patch_embeddings.projection.weight", D, C, P, v->pw, v->kpad, 0);
  v->pb = pt.f32("embeddings.patch_embeddings.projection.bias", D);
  v->layers.resize(cfg->num_layers);
  for (int i = 0; i < cfg->num_layers; ++i) {
    VitLayer& L = v->layers[i];
    const std::string p = "encoder.layer." + std::to_string(i) + ".";
    L.l1g = pt.f32(p + "layernorm_before.weight", D); L.l1b = pt.f32(p + "layernorm_before.bias", D);
    L.wqkv = pt.reserve((size_t)3 * D * D * 2);
    pt.linear_at(p + "attention.attention.query.weight", D, D, L.wqkv, D);
    pt.linear_at(p + "attention.attention.key.weight", D, D, L.wqkv + (size_t)D * D * 2, D);
    pt.linear_at(p + "attention.attention.value.weight", D, D, L.wqkv + (size_t)2 * D * D * 2, D);
    L.bqkv = pt.reserve((size_t)3 * D * 4);                    // zero (bind_arena) unless the checkpoint has q/k/v biases
    if (cfg->qkv_bias) {
      pt.f32_at(p + "attention.attention.query.bias", D, L.bqkv);
      pt.f32_at(p + "attention.attention.key.bias", D, L.bqkv + (size_t)D * 4);
      pt.f32_at(p + "attention.attention.value.bias", D, L.bqkv + (size_t)2 * D * 4);
    }
    L.wo = pt.linear(p + "attention.output.dense.weight", D, D); L.bo = pt.f32(p + "attention.output.dense.bias", D);
    L.l2g = pt.f32(p + "layernorm_after.weight", D); L.l2b = pt.f32(p + "layernorm_after.bias", D);
    L.w1 = pt.linear(p + "intermediate.dense.weight", I, D); L.b1 = pt.f32(p + "intermediate.dense.bias", I);
    L.w2 = pt.linear(p + "output.dense.weight", D, I); L.b2 = pt.f32(p + "output.dense.bias", D);
  }
  v->lng = pt.f32("layernorm.weight", D); v->lnb = pt.f32("layernorm.bias", D);
  return v.release();
}
extern "C" void dmx_vit_destroy(dmx_vit* v) { delete v; }
extern "C" int dmx_vit_param_count(const dmx_vit* v) { return v ? (int)v->pt.entries().size() : 0; }
extern "C" int dmx_vit_param_info(const dmx_vit* v, int index, const char** name, int shape[4]) {
  DMX_REQUIRE(v && index >= 0 && index < (int)v->pt.entries().size(), "vit_param_info: bad index %d", index);
  const ParamEntry& e = v->pt.entries()[index];
  if (name) *name = e.name.c_str();
  if (shape) for (int k = 0; k < 4; ++k) shape[k] = e.shape[k];
  return DMX_OK;
}
extern "C" size_t dmx_vit_arena_bytes(const dmx_vit* v) { return v ? v->pt.total() : 0; }
extern "C" int dmx_vit_bind_arena(dmx_vit* v, void* arena, size_t bytes) {
  DMX_REQUIRE(v && arena && bytes >= v->pt.total(), "vit_bind_arena: need %zu bytes", v ? v->pt.total() : (size_t)0);
  v->arena = (char*)arena; v->finalized = false;
  DMX_HIP(hipMemset(arena, 0, v->pt.total()));      // K padding of the patch filter, absent q/k/v biases
  return DMX_OK;
}
extern "C" int dmx_vit_load_param(dmx_vit* v, const char* name, const float* src, dmx_stream_t stream) {
  DMX_REQUIRE(v != nullptr, "vit_load_param: null handle");
  v->finalized = false;
  return v->pt.load(v->arena, name, src, (hipStream_t)stream);
}
extern "C" int dmx_vit_finalize(dmx_vit* v, dmx_stream_t stream) {
  DMX_REQUIRE(v && v->arena, "vit_finalize: arena not bound");
  DMX_HIP(hipStreamSynchronize((hipStream_t)stream));
  const bf16* zp = nullptr;
  int rc = dmx_zero_page(&zp);
  v->finalized = (rc == 0);
  return rc;
}

namespace {
int vit_run(dmx_vit* v, Exec& ex, const float* pixels, float* out, int B) {
  const dmx_vit_config& c = v->cfg;
  const int D = c.hidden_size, H = c.num_heads, S = v->np + 1, g = c.image_size / c.patch_size;
  // ---- patch embedding
  Tn emb;
  if (ex.f32) {                                        // NCHW -> NHWC, then the generic strided conv on the K-padded filter matrix
    Tn xn = ex.make(B, c.image_size, c.image_size, c.num_channels);
    if (!ex.dry && !ex.rc) ex.rc = dmx_concat_nchw_to_nhwc_f32_launch(pixels, c.num_channels, nullptr, 0, nullptr, 0, (float*)xn.p, B, c.image_size * c.image_size, ex.stream);
    ConvOpts op; op.ksize = c.patch_size; op.stride = c.patch_size; op.pad = 0; op.bias = v->W<float>(v->pb); op.ldw = v->kpad;
    emb = ex.conv(xn, nullptr, v->W<bf16>(v->pw), D, op);
    ex.drop(xn);
  } else {
    Tn col = ex.make(B, g, g, v->kpad);
    if (!ex.dry && !ex.rc) {
      Im2colArgs a{}; a.f0 = pixels; a.c0 = c.num_channels; a.C = c.num_channels;
      a.B = B; a.IH = a.IW = c.image_size; a.OH = a.OW = g; a.ksize = c.patch_size; a.stride = c.patch_size; a.pad = 0; a.out = col.p; a.Kpad = v->kpad;
      ex.rc = dmx_im2col_small_launch(a, ex.stream);
    }
    emb = ex.linear(col, v->W<bf16>(v->pw), D, v->W<float>(v->pb), nullptr, false);
    ex.drop(col);
  }
  Tn x = ex.make(1, 1, B * S, D);
  if (ex.f32) {
    if (!ex.dry && !ex.rc) {
      hipLaunchKernelGGL(dmx_vit_assemble_f32_kernel, dim3(4096), dim3(256), 0, ex.stream, (const float*)emb.p, v->W<float>(v->cls), v->W<float>(v->pos), (float*)x.p, B, v->np, D);
      ex.rc = dmx_check_launch("dmx_vit_assemble_f32_kernel");
    }
  } else if (!ex.dry && !ex.rc) {
    const size_t total = (size_t)B * S * (D / 8);
    int blocks = (int)((total + 255) / 256); if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(dmx_vit_assemble_kernel, dim3(blocks), dim3(256), 0, ex.stream, emb.p, v->W<float>(v->cls), v->W<float>(v->pos), x.p, B, v->np, D);
    ex.rc = dmx_check_launch("dmx_vit_assemble_kernel");
  }
  ex.drop(emb);
  for (const VitLayer& L : v->layers) {
    Tn n1 = ex.layernorm(x, v->W<float>(L.l1g), v->W<float>(L.l1b), c.layer_norm_eps);
    const float* bqkv = v->W<float>(L.bqkv);
    float* btmp = nullptr;
    if (ex.f32) {                                      // the q / k / v bias vectors are not adjacent in the fp32 master arena
      btmp = (float*)ex.raw((size_t)3 * D * 4);
      if (!ex.dry && !ex.rc)
        for (int i = 0; i < 3 && !ex.rc; ++i)
          if (hipMemcpyAsync(btmp + (size_t)i * D, v->W<float>(L.bqkv + (size_t)i * D * 4), (size_t)D * 4, hipMemcpyDeviceToDevice, ex.stream) != hipSuccess) {
            dmx_set_error("hipMemcpyAsync failed (ViT q/k/v biases)"); ex.rc = DMX_ERR_HIP;
          }
      bqkv = btmp;
    }
    Tn qkv = ex.linear(n1, v->W<bf16>(L.wqkv), 3 * D, bqkv, nullptr, false);
    ex.drop(n1);
    if (btmp) ex.drop(btmp);
    Tn a = ex.make(1, 1, B * S, D);
    ex.attention(qkv.p, 3 * D, ex.col(qkv, D), 3 * D, ex.col(qkv, 2 * D), 3 * D, S, a.p, D, B, H, S, S, 0.125f);
    ex.drop(qkv);
    Tn x2 = ex.linear(a, v->W<bf16>(L.wo), D, v->W<float>(L.bo), &x, false);
    ex.drop(a); ex.drop(x);
    Tn n2 = ex.layernorm(x2, v->W<float>(L.l2g), v->W<float>(L.l2b), c.layer_norm_eps);
    Tn h = ex.linear_gelu(n2, v->W<bf16>(L.w1), c.intermediate_size, v->W<float>(L.b1));
    ex.drop(n2);
    x = ex.linear(h, v->W<bf16>(L.w2), D, v->W<float>(L.b2), &x2, false);
    ex.drop(h); ex.drop(x2);
  }
  Tn y = ex.layernorm(x, v->W<float>(v->lng), v->W<float>(v->lnb), c.layer_norm_eps);
  ex.drop(x);
  if (ex.f32) {
    if (!ex.dry && !ex.rc && hipMemcpyAsync(out, y.p, (size_t)B * S * D * 4, hipMemcpyDeviceToDevice, ex.stream) != hipSuccess) {
      dmx_set_error("hipMemcpyAsync failed (ViT output)"); ex.rc = DMX_ERR_HIP;
    }
  } else if (!ex.dry && !ex.rc) {
    const size_t n = (size_t)B * S * D;
    int blocks = (int)((n + 255) / 256); if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(dmx_bf16_to_f32_kernel, dim3(blocks), dim3(256), 0, ex.stream, y.p, out, n);
    ex.rc = dmx_check_launch("dmx_bf16_to_f32_kernel");
  }
  ex.drop(y);
  return ex.rc;
}
}  // namespace

extern "C" size_t dmx_vit_workspace_bytes(dmx_vit* v, int B) {
  if (!v) return 0;
  Exec ex; ex.dry = true; ex.ws.reset(nullptr, 0, true);
  vit_run(v, ex, nullptr, nullptr, B);
  return ex.ws.peak() + 4096;
}
// ---- fp32 VALIDATION instantiation (tests only): the same walker on fp32 activations, the caller's fp32 master copy of the
// parameters (dmx_vit_master_bytes, filled by dmx_vit_master_import) and the plain fp32 kernels of ref_f32.hip
extern "C" size_t dmx_vit_master_bytes(const dmx_vit* v) { return v ? 2 * v->pt.total() : 0; }
extern "C" int dmx_vit_master_import(const dmx_vit* v, void* masters, const char* name, const float* src, dmx_stream_t stream) {
  DMX_REQUIRE(v && masters && name && src, "vit_master_import: null argument");
  return dmx_master_import(v->pt, masters, name, src, (hipStream_t)stream, "vit_master_import");
}
extern "C" size_t dmx_vit_workspace_bytes_f32(dmx_vit* v, int B) {
  if (!v) return 0;
  Exec ex; ex.dry = true; ex.f32 = true; ex.ws.reset(nullptr, 0, true);
  v->masters_f32 = (const char*)4096;
  vit_run(v, ex, nullptr, nullptr, B);
  v->masters_f32 = nullptr;
  return ex.ws.peak() + 4096;
}
extern "C" int dmx_vit_forward_f32(dmx_vit* v, const void* masters, const float* pixel_values, float* last_hidden_state, int B,
                                   void* workspace, size_t workspace_bytes, dmx_stream_t stream) {
  DMX_REQUIRE(v && masters && pixel_values && last_hidden_state && workspace && B > 0, "vit_forward_f32: null argument");
  Exec ex; ex.stream = (hipStream_t)stream; ex.f32 = true; ex.ws.reset(workspace, workspace_bytes, false);
  v->masters_f32 = (const char*)masters;
  const int rc = vit_run(v, ex, pixel_values, last_hidden_state, B);
  v->masters_f32 = nullptr;
  return rc;
}
// last_hidden_state [B][num_patches + 1][hidden] fp32 from pixel_values [B][C][image][image] fp32 (NCHW, already
// resized / normalised by the processor)
extern "C" int dmx_vit_forward(dmx_vit* v, const float* pixel_values, float* last_hidden_state, int B,
                               void* workspace, size_t workspace_bytes, dmx_stream_t stream) {
  DMX_REQUIRE(v && v->finalized, "vit_forward: weights not finalized (bind_arena, load_param*, finalize)");
  DMX_REQUIRE(pixel_values && last_hidden_state && workspace && B > 0, "vit_forward: null argument");
  Exec ex; ex.stream = (hipStream_t)stream; ex.ws.reset(workspace, workspace_bytes, false);
  return vit_run(v, ex, pixel_values, last_hidden_state, B);
}
