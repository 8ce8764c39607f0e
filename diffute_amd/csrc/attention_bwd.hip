// Flash-style attention backward, head dim 64, bf16 MFMA (training rows of SURVEY.md 8a: P5 over K5 / K6).
//
// With P = softmax(Q K^T * scale) recomputed from the forward's per-row log2-sum-exp (no S or P is ever stored):
//   delta_q = sum_d dO[q][d] * O[q][d]
//   dV = P^T dO          dP = dO V^T          dS = P o (dP - delta) * scale
//   dQ = dS K            dK = dS^T Q
// Two kernels, both deterministic (no atomics; each output element is produced by exactly one wave):
//   dq  : grid (ceil(Sq/128), H, B), a wave owns 32 queries and streams 64-key K / V tiles through LDS - the forward
//         kernel's structure: S^T = K Q^T and dP^T = V dO^T put all scores of ONE query in a lane (lse / delta are lane
//         scalars), the bf16 dS^T fragment is the B operand of dQ^T += K^T dS^T with K^T fetched by LDS transpose reads.
//   dkv : grid (ceil(Skv/128), H, B), a wave owns 32 keys (K, V fragments live in registers) and the block streams
//         32-query Q / dO tiles through LDS: S = Q K^T and dP = dO V^T put all scores of ONE key in a lane, P and dS are
//         the B operands of dV^T += dO^T P and dK^T += Q^T dS with dO^T / Q^T fetched by LDS transpose reads.
// Every MFMA k-slot mapping is the forward's: slot e of k-step s <-> index 16 s + 4 (lane>>5) + (e&3) + 8 (e>>2).
#include "common.h"
#include "kernels.h"

#define KROW 72      // direct-read tile row stride in elements (144 B: conflict-free ds_read_b128)
#define VRS 192      // transpose-read tile row stride in BYTES (64 data + 32 pad elements)
#define DMX_TR8(V, A, O0, O1, O2, O3, O4, O5, O6, O7)                                                     \
  asm volatile("ds_read_b64_tr_b16 %0, %8 offset:%9\n\tds_read_b64_tr_b16 %1, %8 offset:%10\n\t"          \
               "ds_read_b64_tr_b16 %2, %8 offset:%11\n\tds_read_b64_tr_b16 %3, %8 offset:%12\n\t"         \
               "ds_read_b64_tr_b16 %4, %8 offset:%13\n\tds_read_b64_tr_b16 %5, %8 offset:%14\n\t"         \
               "ds_read_b64_tr_b16 %6, %8 offset:%15\n\tds_read_b64_tr_b16 %7, %8 offset:%16"              \
               : "=&v"(V[0]), "=&v"(V[1]), "=&v"(V[2]), "=&v"(V[3]), "=&v"(V[4]), "=&v"(V[5]), "=&v"(V[6]), "=&v"(V[7]) \
               : "v"(A), "i"(O0), "i"(O1), "i"(O2), "i"(O3), "i"(O4), "i"(O5), "i"(O6), "i"(O7) : "memory")
#define DMX_TR4(V, A, O0, O1, O2, O3)                                                                     \
  asm volatile("ds_read_b64_tr_b16 %0, %4 offset:%5\n\tds_read_b64_tr_b16 %1, %4 offset:%6\n\t"          \
               "ds_read_b64_tr_b16 %2, %4 offset:%7\n\tds_read_b64_tr_b16 %3, %4 offset:%8"              \
               : "=&v"(V[0]), "=&v"(V[1]), "=&v"(V[2]), "=&v"(V[3])                                       \
               : "v"(A), "i"(O0), "i"(O1), "i"(O2), "i"(O3) : "memory")

namespace {
__device__ __forceinline__ bf16x8 frag_from_tr(unsigned long long a, unsigned long long b) {
  const u32x4 v = {(unsigned)a, (unsigned)(a >> 32), (unsigned)b, (unsigned)(b >> 32)};
  return __builtin_bit_cast(bf16x8, v);
}
// 16 accumulator registers of one 32-row block -> two bf16 B-operand fragments (k-steps of 16)
__device__ __forceinline__ void pack_frags(const float* v, bf16x8* out) {
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const u32x4 w = {pack_bf2(v[8 * u], v[8 * u + 1]), pack_bf2(v[8 * u + 2], v[8 * u + 3]),
                     pack_bf2(v[8 * u + 4], v[8 * u + 5]), pack_bf2(v[8 * u + 6], v[8 * u + 7])};
    out[u] = __builtin_bit_cast(bf16x8, w);
  }
}

// delta[b][h][q] = sum_d dO * O : 8 lanes per (row, head)
__global__ __launch_bounds__(256) void dmx_attn_delta_kernel(const AttnBwdArgs p) {
  const size_t idx = (size_t)blockIdx.x * 32 + (threadIdx.x >> 3);      // (b*Sq + q)*H + h
  const int part = threadIdx.x & 7;
  const size_t total = (size_t)p.B * p.Sq * p.H;
  float s = 0.f;
  size_t row = 0; int h = 0;
  if (idx < total) {
    row = idx / p.H; h = (int)(idx - row * p.H);
    float a[8], b[8];
    unpack_bf8(*(const u32x4*)(p.o + row * p.ldo + h * 64 + part * 8), a);
    unpack_bf8(*(const u32x4*)(p.dout + row * p.ldo + h * 64 + part * 8), b);
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i] * b[i];
  }
  s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4);
  if (idx < total && part == 0) {
    const size_t bb = row / p.Sq, q = row - bb * p.Sq;
    p.delta[(bb * p.H + h) * p.Sq + q] = s;
  }
}

#define DQ_STAGE (64 * KROW * 2 + 64 * VRS + 64 * KROW * 2)      // K direct | K transpose-read layout | V direct
__global__ __launch_bounds__(256, 2) void dmx_attn_bwd_dq_kernel(const AttnBwdArgs p) {
  __shared__ __attribute__((aligned(16))) char smem[2 * DQ_STAGE];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  const int b = blockIdx.z, h = blockIdx.y;
  const int q0 = blockIdx.x * 128 + wave * 32;
  const float sl2 = p.scale * 1.4426950408889634f;

  bf16x8 qf[4], dof[4];
  float l2q, dlq;
  {
    int qrow = q0 + lr; if (qrow >= p.Sq) qrow = p.Sq - 1;
    const bf16* qp = p.q + ((size_t)b * p.Sq + qrow) * p.ldq + h * 64 + 8 * lh;
    const bf16* dp = p.dout + ((size_t)b * p.Sq + qrow) * p.ldo + h * 64 + 8 * lh;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) { qf[kk] = *(const bf16x8*)(qp + 16 * kk); dof[kk] = *(const bf16x8*)(dp + 16 * kk); }
    l2q = p.lse[((size_t)b * p.H + h) * p.Sq + qrow];
    dlq = p.delta[((size_t)b * p.H + h) * p.Sq + qrow];
  }
  const int srow0 = t >> 3, spc = t & 7;
  const bf16* kbase = p.k + (size_t)b * p.kv_rows * p.ldk + h * 64 + spc * 8;
  const bf16* vbase = p.v + (size_t)b * p.kv_rows * p.ldv + h * 64 + spc * 8;
  u32x4 kreg[2], vreg[2];
  auto load_tile = [&](int kv0) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int key = kv0 + srow0 + 32 * i; if (key >= p.Skv) key = p.Skv - 1;
      kreg[i] = *(const u32x4*)(kbase + (size_t)key * p.ldk);
      vreg[i] = *(const u32x4*)(vbase + (size_t)key * p.ldv);
    }
  };
  auto write_tile = [&](int buf) {
    char* ks = smem + buf * DQ_STAGE;
    char* kt = ks + 64 * KROW * 2;
    char* vs = kt + 64 * VRS;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = srow0 + 32 * i;
      *(u32x4*)(ks + r * (KROW * 2) + spc * 16) = kreg[i];
      *(u32x4*)(kt + r * VRS + spc * 16) = kreg[i];
      *(u32x4*)(vs + r * (KROW * 2) + spc * 16) = vreg[i];
    }
  };
  f32x16 dq[2];
#pragma unroll
  for (int i = 0; i < 16; ++i) { dq[0][i] = 0.f; dq[1][i] = 0.f; }

  const int ntiles = (p.Skv + 63) / 64;
  load_tile(0);
  write_tile(0);
  __syncthreads();
  for (int it = 0; it < ntiles; ++it) {
    const int kv0 = it * 64;
    if (it + 1 < ntiles) load_tile(kv0 + 64);
    const char* ks = smem + (it & 1) * DQ_STAGE;
    const char* kt = ks + 64 * KROW * 2;
    const char* vs = kt + 64 * VRS;
    // S^T and dP^T: [r] <-> (query lr, key kv0 + 32kt + (r&3) + 8(r>>2) + 4lh)
    f32x16 s[2], dp[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
#pragma unroll
      for (int i = 0; i < 16; ++i) { s[c][i] = 0.f; dp[c][i] = 0.f; }
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const bf16x8 kf = *(const bf16x8*)(ks + (32 * c + lr) * (KROW * 2) + (2 * kk + lh) * 16);
        const bf16x8 vf = *(const bf16x8*)(vs + (32 * c + lr) * (KROW * 2) + (2 * kk + lh) * 16);
        s[c] = DMX_MFMA_32x32x16(kf, qf[kk], s[c]);
        dp[c] = DMX_MFMA_32x32x16(vf, dof[kk], dp[c]);
      }
    }
    bf16x8 pf[4];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      float ds[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = kv0 + 32 * c + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const float pr = key < p.Skv ? __builtin_amdgcn_exp2f(__builtin_fmaf(s[c][r], sl2, -l2q)) : 0.f;
        ds[r] = pr * (dp[c][r] - dlq) * p.scale;
      }
      pack_frags(ds, pf + 2 * c);
    }
    // dQ^T += K^T dS^T (K^T rows d from the transpose-read layout)
    {
      const int p16 = lane & 15, g = lane >> 4;
      const unsigned ka = (unsigned)(unsigned long long)(const void*)kt +
                          (unsigned)((4 * (g >> 1) + (p16 >> 2)) * VRS + (16 * (g & 1) + 4 * (p16 & 3)) * 2);
      unsigned long long v0[8], v1[8];
      DMX_TR8(v0, ka, 0 * VRS, 8 * VRS, 16 * VRS, 24 * VRS, 32 * VRS, 40 * VRS, 48 * VRS, 56 * VRS);
      DMX_TR8(v1, ka, 64 + 0 * VRS, 64 + 8 * VRS, 64 + 16 * VRS, 64 + 24 * VRS, 64 + 32 * VRS, 64 + 40 * VRS, 64 + 48 * VRS, 64 + 56 * VRS);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        dq[0] = DMX_MFMA_32x32x16(frag_from_tr(v0[2 * s4], v0[2 * s4 + 1]), pf[s4], dq[0]);
        dq[1] = DMX_MFMA_32x32x16(frag_from_tr(v1[2 * s4], v1[2 * s4 + 1]), pf[s4], dq[1]);
      }
    }
    if (it + 1 < ntiles) write_tile((it + 1) & 1);
    __syncthreads();
  }
  // store: lane holds query lr, d = 32dt + 8g + 4lh + e
  const int qrow = q0 + lr;
  if (qrow < p.Sq) {
    bf16* op = p.dq + ((size_t)b * p.Sq + qrow) * p.lddq + h * 64 + 4 * lh;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const u32x2 pk = {pack_bf2(dq[dt][4 * g], dq[dt][4 * g + 1]), pack_bf2(dq[dt][4 * g + 2], dq[dt][4 * g + 3])};
        *(u32x2*)(op + 32 * dt + 8 * g) = pk;
      }
  }
}

// per 32-query tile: Q direct | Q transpose-read layout | dO direct | dO transpose-read layout | lse[32] | delta[32]
#define KV_QD 0
#define KV_QT (32 * KROW * 2)
#define KV_DD (KV_QT + 32 * VRS)
#define KV_DT (KV_DD + 32 * KROW * 2)
#define KV_LS (KV_DT + 32 * VRS)
#define KV_STAGE (KV_LS + 256)
__global__ __launch_bounds__(256, 2) void dmx_attn_bwd_dkv_kernel(const AttnBwdArgs p) {
  __shared__ __attribute__((aligned(16))) char smem[2 * KV_STAGE];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  const int b = blockIdx.z, h = blockIdx.y;
  const int k0 = blockIdx.x * 128 + wave * 32;
  const float sl2 = p.scale * 1.4426950408889634f;
  const int key = k0 + lr;
  const bool key_ok = key < p.Skv;

  bf16x8 kf[4], vf[4];                 // B operands: row = this lane's key, k-slots d = 16kk + 8lh..
  {
    const int kc = key_ok ? key : p.Skv - 1;
    const bf16* kp = p.k + ((size_t)b * p.kv_rows + kc) * p.ldk + h * 64 + 8 * lh;
    const bf16* vp = p.v + ((size_t)b * p.kv_rows + kc) * p.ldv + h * 64 + 8 * lh;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) { kf[kk] = *(const bf16x8*)(kp + 16 * kk); vf[kk] = *(const bf16x8*)(vp + 16 * kk); }
  }
  // staging: chunk t -> query row t>>3, 16-byte piece t&7 (one Q and one dO chunk per thread); t<32: lse, 32<=t<64: delta
  const int srow = t >> 3, spc = t & 7;
  const bf16* qbase = p.q + (size_t)b * p.Sq * p.ldq + h * 64 + spc * 8;
  const bf16* dbase = p.dout + (size_t)b * p.Sq * p.ldo + h * 64 + spc * 8;
  const float* lbase = p.lse + ((size_t)b * p.H + h) * p.Sq;
  const float* ebase = p.delta + ((size_t)b * p.H + h) * p.Sq;
  u32x4 qreg, dreg; float sreg = 0.f;
  auto load_tile = [&](int q0) {
    int qrow = q0 + srow; if (qrow >= p.Sq) qrow = p.Sq - 1;
    qreg = *(const u32x4*)(qbase + (size_t)qrow * p.ldq);
    dreg = *(const u32x4*)(dbase + (size_t)qrow * p.ldo);
    if (t < 64) {
      int qq = q0 + (t & 31); if (qq >= p.Sq) qq = p.Sq - 1;
      sreg = (t < 32) ? lbase[qq] : ebase[qq];
    }
  };
  auto write_tile = [&](int buf) {
    char* st = smem + buf * KV_STAGE;
    *(u32x4*)(st + KV_QD + srow * (KROW * 2) + spc * 16) = qreg;
    *(u32x4*)(st + KV_QT + srow * VRS + spc * 16) = qreg;
    *(u32x4*)(st + KV_DD + srow * (KROW * 2) + spc * 16) = dreg;
    *(u32x4*)(st + KV_DT + srow * VRS + spc * 16) = dreg;
    if (t < 64) ((float*)(st + KV_LS))[t] = sreg;
  };
  f32x16 dv[2], dk[2];
#pragma unroll
  for (int i = 0; i < 16; ++i) { dv[0][i] = 0.f; dv[1][i] = 0.f; dk[0][i] = 0.f; dk[1][i] = 0.f; }

  const int ntiles = (p.Sq + 31) / 32;
  load_tile(0);
  write_tile(0);
  __syncthreads();
  for (int it = 0; it < ntiles; ++it) {
    const int q0 = it * 32;
    if (it + 1 < ntiles) load_tile(q0 + 32);
    const char* st = smem + (it & 1) * KV_STAGE;
    // S and dP: [r] <-> (query q0 + (r&3) + 8(r>>2) + 4lh, key = this lane's)
    f32x16 s, dp;
#pragma unroll
    for (int i = 0; i < 16; ++i) { s[i] = 0.f; dp[i] = 0.f; }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const bf16x8 qa = *(const bf16x8*)(st + KV_QD + lr * (KROW * 2) + (2 * kk + lh) * 16);
      const bf16x8 da = *(const bf16x8*)(st + KV_DD + lr * (KROW * 2) + (2 * kk + lh) * 16);
      s = DMX_MFMA_32x32x16(qa, kf[kk], s);
      dp = DMX_MFMA_32x32x16(da, vf[kk], dp);
    }
    float pr[16], ds[16];
    const float* ls = (const float*)(st + KV_LS);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 l4 = *(const f32x4*)(ls + 8 * g + 4 * lh);
      const f32x4 e4 = *(const f32x4*)(ls + 32 + 8 * g + 4 * lh);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r = 4 * g + e;
        const bool ok = key_ok && (q0 + 8 * g + 4 * lh + e < p.Sq);
        const float pv = ok ? __builtin_amdgcn_exp2f(__builtin_fmaf(s[r], sl2, -l4[e])) : 0.f;
        pr[r] = pv; ds[r] = pv * (dp[r] - e4[e]) * p.scale;
      }
    }
    bf16x8 pfp[2], pfs[2];
    pack_frags(pr, pfp);
    pack_frags(ds, pfs);
    // dV^T += dO^T P ; dK^T += Q^T dS  (A operands: rows d, k-slots = queries, via transpose reads)
    {
      const int p16 = lane & 15, g = lane >> 4;
      const unsigned off = (unsigned)((4 * (g >> 1) + (p16 >> 2)) * VRS + (16 * (g & 1) + 4 * (p16 & 3)) * 2);
      const unsigned base = (unsigned)(unsigned long long)(const void*)st;
      unsigned long long d0[4], d1[4], q0v[4], q1v[4];
      DMX_TR4(d0, base + KV_DT + off, 0 * VRS, 8 * VRS, 16 * VRS, 24 * VRS);
      DMX_TR4(d1, base + KV_DT + off, 64 + 0 * VRS, 64 + 8 * VRS, 64 + 16 * VRS, 64 + 24 * VRS);
      DMX_TR4(q0v, base + KV_QT + off, 0 * VRS, 8 * VRS, 16 * VRS, 24 * VRS);
      DMX_TR4(q1v, base + KV_QT + off, 64 + 0 * VRS, 64 + 8 * VRS, 64 + 16 * VRS, 64 + 24 * VRS);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        dv[0] = DMX_MFMA_32x32x16(frag_from_tr(d0[2 * ks], d0[2 * ks + 1]), pfp[ks], dv[0]);
        dv[1] = DMX_MFMA_32x32x16(frag_from_tr(d1[2 * ks], d1[2 * ks + 1]), pfp[ks], dv[1]);
        dk[0] = DMX_MFMA_32x32x16(frag_from_tr(q0v[2 * ks], q0v[2 * ks + 1]), pfs[ks], dk[0]);
        dk[1] = DMX_MFMA_32x32x16(frag_from_tr(q1v[2 * ks], q1v[2 * ks + 1]), pfs[ks], dk[1]);
      }
    }
    if (it + 1 < ntiles) write_tile((it + 1) & 1);
    __syncthreads();
  }
  if (key_ok) {
    bf16* kp = p.dk + ((size_t)b * p.kv_rows + key) * p.lddk + h * 64 + 4 * lh;
    bf16* vp = p.dv + ((size_t)b * p.kv_rows + key) * p.lddv + h * 64 + 4 * lh;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const u32x2 a = {pack_bf2(dk[dt][4 * g], dk[dt][4 * g + 1]), pack_bf2(dk[dt][4 * g + 2], dk[dt][4 * g + 3])};
        const u32x2 c = {pack_bf2(dv[dt][4 * g], dv[dt][4 * g + 1]), pack_bf2(dv[dt][4 * g + 2], dv[dt][4 * g + 3])};
        *(u32x2*)(kp + 32 * dt + 8 * g) = a;
        *(u32x2*)(vp + 32 * dt + 8 * g) = c;
      }
  }
}
}  // namespace

size_t dmx_attn_bwd_ws_bytes(int B, int H, int Sq) { return (size_t)B * H * Sq * sizeof(float); }

int dmx_attention_bwd_launch(const AttnBwdArgs& a, hipStream_t stream) {
  DMX_REQUIRE(a.B > 0 && a.H > 0 && a.Sq > 0 && a.Skv > 0, "attention_bwd: empty problem");
  DMX_REQUIRE(a.kv_rows >= a.Skv, "attention_bwd: kv_rows=%d < Skv=%d", a.kv_rows, a.Skv);
  DMX_REQUIRE(a.q && a.k && a.v && a.o && a.dout && a.lse && a.delta && a.dq && a.dk && a.dv, "attention_bwd: null argument");
  DMX_REQUIRE(a.ldq % 8 == 0 && a.ldk % 8 == 0 && a.ldv % 8 == 0 && a.ldo % 8 == 0 && a.lddq % 4 == 0 && a.lddk % 4 == 0 && a.lddv % 4 == 0,
              "attention_bwd: strides must be multiples of 8");
  const double fl = 4.0 * a.B * a.H * (double)a.Sq * a.Skv * 64;          // one forward's worth of FLOPs
  const size_t total = (size_t)a.B * a.Sq * a.H;
  hipLaunchKernelGGL(dmx_attn_delta_kernel, dim3((unsigned)((total + 31) / 32)), dim3(256), 0, stream, a);
  int rc = dmx_check_launch("dmx_attn_delta_kernel");
  if (rc) return rc;
  {
    ProfScope ps(PROF_ATTN, stream, 1.5 * fl, 0.0, "attn_bwd_dq");
    hipLaunchKernelGGL(dmx_attn_bwd_dq_kernel, dim3(cdiv(a.Sq, 128), a.H, a.B), dim3(256), 0, stream, a);
  }
  rc = dmx_check_launch("dmx_attn_bwd_dq_kernel");
  if (rc) return rc;
  {
    ProfScope ps(PROF_ATTN, stream, 2.0 * fl, 0.0, "attn_bwd_dkv");
    hipLaunchKernelGGL(dmx_attn_bwd_dkv_kernel, dim3(cdiv(a.Skv, 128), a.H, a.B), dim3(256), 0, stream, a);
  }
  return dmx_check_launch("dmx_attn_bwd_dkv_kernel");
}
