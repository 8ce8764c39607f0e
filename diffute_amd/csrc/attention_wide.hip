// Fused single-head attention with a WIDE head (d = C = 128 / 256 / 512): the mid-block attention of AutoencoderKL
// (diffusers `Attention` inside UNetMidBlock2D of the VAE, reached from vae.encode / vae.decode at
// app.ipynb:793,819 and train_diffute_v1.py:875,886):   O[b, q, :] = softmax(Q K^T / sqrt(d)) V   over S = H*W tokens.
//
// Flash-style, nothing of size S x S is ever materialised (round 1 ran QK^T GEMM -> fp32 scores in HBM (64 MB per sample at
// 512 px) -> row softmax -> PV GEMM).  What makes d = 512 different from the d = 64 kernel is the per-query state: a wave
// that owns 32 query rows carries O^T (512 x 32 fp32 = 256 accumulator registers) and its Q fragments (32 k-steps x 4 =
// 128 registers), so a block is 4 waves at ONE wave per SIMD (the 512-entry unified register file of gfx950) and the
// contraction over d runs over MFMA k-steps:
//   S^T[32 keys x 32 q]  = K[32 x d] Q^T            d/16 x v_mfma_f32_32x32x16_bf16, K fragments by ds_read_b128
//   online softmax on the raw scores (exp2 domain, deferred max), P rounded to bf16, lane-local (a lane owns one query)
//   O^T[d x 32 q]       += V^T[d x 32 keys] P^T      d/32 x 2 MFMAs, V^T fragments by LDS transpose reads of row-major V
// K / V tiles (32 keys x d, 64 KB per tile at d = 512) go HBM/L2 -> LDS by LDS-DMA (no staging registers are left), two
// buffers: tile t+1 is requested at the top of iteration t and waited for at its end.  Both images are conflict-free
// through XOR swizzles applied on the DMA SOURCE address (the DMA writes lane-linear): K 16-byte chunks ^ (key & 15)
// for the ds_read_b128 fragment reads, V 64-byte slots ^ (key & 3) for the transpose reads.
#include "common.h"
#include "kernels.h"

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((address_space(3))) s16x4* lds_s16x4;

template <int D>
__global__ __launch_bounds__(256, 1) void dmx_attn_wide_kernel(const AttnWideArgs p) {
  constexpr int RS = D * 2;                           // LDS row bytes (K and V rows are stored back to back: swizzled, not padded)
  constexpr int LPRW = D / 8;                         // lanes (16-byte chunks) per row
  constexpr int RPI = 64 / LPRW;                      // rows per DMA instruction (1 at d = 512)
  constexpr int TILE = 32 * RS;                       // one operand tile
  constexpr int NI = 32 / RPI / 4;                    // DMA instructions per wave per operand per tile
  constexpr int KS = D / 16, DT = D / 32;
  static_assert(D % 128 == 0 && LPRW <= 64 && NI >= 1, "head width must be a multiple of 128 (<= 512)");
  extern __shared__ __attribute__((aligned(16))) char smem[];       // [2 buffers][K tile | V tile]
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int lr = lane & 31, lh = lane >> 5;
  const int b = blockIdx.y;
  const int q0 = blockIdx.x * 128 + wave * 32;
  const float sl2 = p.scale * 1.4426950408889634f;
  const float thr = 8.0f / sl2;

  // ---- DMA assignment: instruction i of this wave covers rows (4*i + wave) * RPI .. + RPI of the tile
  // Addresses are a wave-uniform 64-bit base (the tile's first row) plus a 32-bit per-lane byte offset recomputed per
  // instruction - no pointer arrays live across the loop (the register file is full of O and Q).  Keys past Skv re-read the
  // last valid row: their scores are masked to -inf, so P = 0 multiplies finite V data.
  const int drow = lane / LPRW, dchk = lane % LPRW;   // row within the instruction, 16-byte chunk within the row
  const char* kb = (const char*)(p.k + (size_t)b * p.kv_rows * p.ldk);
  const char* vb = (const char*)(p.v + (size_t)b * p.kv_rows * p.ldv);
  const unsigned ldkb = (unsigned)p.ldk * 2u, ldvb = (unsigned)p.ldv * 2u;
  auto stage = [&](int buf, int kv0) {
    char* ks = smem + buf * 2 * TILE;
    char* vs = ks + TILE;
    const char* kt = kb + (size_t)kv0 * ldkb;         // uniform
    const char* vt = vb + (size_t)kv0 * ldvb;
    const int last = p.Skv - 1 - kv0;                 // last valid row of this tile (>= 31 except in the final tile)
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int r0 = (4 * i + wave) * RPI;            // wave-uniform first row of the instruction
      const int r = r0 + drow;
      const unsigned rc = (unsigned)min(r, last);
      const unsigned kc = (unsigned)(dchk ^ (r & 15));                             // K: chunk XOR (row & 15)
      const unsigned vc = (unsigned)((((dchk >> 2) ^ (r & 3)) << 2) | (dchk & 3));  // V: 64-byte slot XOR (row & 3)
      dmx_dma16(kt + (rc * ldkb + kc * 16u), DMX_LDS_ADDR(ks + r0 * RS));      // (asm, not the builtin: common.h dmx_dma16)
      dmx_dma16(vt + (rc * ldvb + vc * 16u), DMX_LDS_ADDR(vs + r0 * RS));
    }
  };
  const int ntiles = (p.Skv + 31) / 32;
  stage(0, 0);                                         // in flight while the Q fragments are fetched

  // ---- Q fragments (MFMA B operand): query lr, d = 16*kk + 8*lh .. +8
  bf16x8 qf[KS];
  {
    int qrow = q0 + lr; if (qrow >= p.Sq) qrow = p.Sq - 1;
    const bf16* qp = p.q + ((size_t)b * p.Sq + qrow) * p.ldq + 8 * lh;
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) qf[kk] = *(const bf16x8*)(qp + 16 * kk);
  }
  f32x16 o[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
#pragma unroll
    for (int i = 0; i < 16; ++i) o[dt][i] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  // ---- per-lane fragment addresses
  const int ksw = (lh ^ (lr & 15)) << 4;              // K: row lr, chunk (2kk + lh) ^ (lr & 15)  ==  (2kk << 4) ^ ksw
  const int p16 = lane & 15, g = lane >> 4;
  const int vrow = 4 * (g >> 1) + (p16 >> 2);         // V: key row of this lane's transpose-read address (+ 16*s4 + 8*half)
  const int vcol = 32 * (g & 1) + 8 * (p16 & 3);      //    byte offset inside the 64-byte slot
  int vslot[4];                                       //    swizzled slot offset for dt & 3 = 0..3 (row & 3 == (p16 >> 2) & 3 for all of this lane's rows)
#pragma unroll
  for (int j = 0; j < 4; ++j) vslot[j] = ((j ^ ((p16 >> 2) & 3)) << 6) + vcol;

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  // the Q fragments must not look like pending loads inside the loop: hipcc would re-wait for them there (vmcnt retires in
  // order, so that wait would also drain the K/V tile DMA issued at the top of each iteration) - see attention.hip
#pragma unroll
  for (int kk = 0; kk < KS; ++kk) asm volatile("" : "+v"(qf[kk]));
  __syncthreads();
  for (int it = 0; it < ntiles; ++it) {
    const int kv0 = it * 32;
    if (it + 1 < ntiles) stage((it + 1) & 1, kv0 + 32);
    const char* ks = smem + (it & 1) * 2 * TILE;
    const char* vs = ks + TILE;

    // ---- S^T = K Q^T over the d/16 k-steps; two accumulators so consecutive MFMAs are independent.  Fragment reads are
    // software-pipelined by hand in stages of 4 k-steps (next stage's ds_reads, then this stage's MFMAs) with the scheduler
    // fenced per stage: left alone it hoists all d/16 fragment reads (128 registers at d = 512) and spills Q.
    f32x16 s0, s1;
    {
      const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      const char* kr = ks + lr * RS;
      constexpr int NST = KS / 4;
      bf16x8 kf[2][4];
#pragma unroll
      for (int j = 0; j < 4; ++j) kf[0][j] = *(const bf16x8*)(kr + (((2 * j) << 4) ^ ksw));
#pragma unroll
      for (int st = 0; st < NST; ++st) {
        if (st + 1 < NST) {
#pragma unroll
          for (int j = 0; j < 4; ++j) kf[(st + 1) & 1][j] = *(const bf16x8*)(kr + (((2 * (4 * (st + 1) + j)) << 4) ^ ksw));
        }
#pragma unroll
        for (int j = 0; j < 4; j += 2) {
          s0 = DMX_MFMA_32x32x16(kf[st & 1][j], qf[4 * st + j], (st == 0 && j == 0) ? zero : s0);
          s1 = DMX_MFMA_32x32x16(kf[st & 1][j + 1], qf[4 * st + j + 1], (st == 0 && j == 0) ? zero : s1);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    f32x16& s = s0;                                   // s[r] = score(query lr, key kv0 + (r&3) + 8(r>>2) + 4lh)
    s0 += s1;
    if (kv0 + 32 > p.Skv) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (kv0 + (r & 3) + 8 * (r >> 2) + 4 * lh >= p.Skv) s[r] = -INFINITY;
    }
    float mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    if (!__all(mx <= m_run + thr)) {                  // wave-uniform: some query row needs a higher reference max
      const float m_new = fmaxf(m_run, mx);
      const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * sl2);
      m_run = m_new; l_run *= alpha;
      // O lives in the accumulator half of the register file (no VALU access there): rescale IN PLACE, one element at a time
      // through a scratch VGPR.  Written as asm because a plain `o *= alpha` under this branch makes the allocator keep a
      // second, VGPR-resident copy of all d/2 accumulators for the join and spill Q.  (The MFMAs that wrote / will read these
      // registers are hundreds of cycles away on both sides: no MFMA <-> accvgpr hazard window is open here.)
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          float tmp;
          asm volatile("v_accvgpr_read_b32 %1, %0\n\tv_mul_f32 %1, %1, %2\n\ts_nop 0\n\tv_accvgpr_write_b32 %0, %1" : "+a"(o[dt][i]), "=&v"(tmp) : "v"(alpha));
        }
    }
    const float mc = -m_run * sl2;
    float psum = 0.f;
    bf16x8 pf[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      unsigned int w[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float p0 = __builtin_amdgcn_exp2f(__builtin_fmaf(s[8 * u + 2 * e], sl2, mc));
        const float p1 = __builtin_amdgcn_exp2f(__builtin_fmaf(s[8 * u + 2 * e + 1], sl2, mc));
        psum += p0 + p1;
        w[e] = pack_bf2(p0, p1);
      }
      u32x4 wv = {w[0], w[1], w[2], w[3]};
      pf[u] = __builtin_bit_cast(bf16x8, wv);
    }
    l_run += psum;

    // ---- O^T += V^T P^T : k-slot e of step u <-> key 16u + 4lh + (e&3) + 8(e>>2); V^T fragments by transpose reads,
    // pipelined like the K fragments (stage = two d-tiles = 8 reads / 4 MFMAs)
    {
      const char* va = vs + vrow * RS;
      auto vread = [&](int dt, int u) {
        const char* vd = va + 256 * (dt >> 2) + vslot[dt & 3];
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(vd + (16 * u) * RS));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(vd + (16 * u + 8) * RS));
        const s16x8 vv = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        return __builtin_bit_cast(bf16x8, vv);
      };
      constexpr int NSV = DT / 2;
      bf16x8 vf[2][4];
#pragma unroll
      for (int j = 0; j < 4; ++j) vf[0][j] = vread(j >> 1, j & 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int st = 0; st < NSV; ++st) {
        if (st + 1 < NSV) {
#pragma unroll
          for (int j = 0; j < 4; ++j) vf[(st + 1) & 1][j] = vread(2 * (st + 1) + (j >> 1), j & 1);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int dt = 2 * st + (j >> 1);
          o[dt] = DMX_MFMA_32x32x16(vf[st & 1][j], pf[j & 1], o[dt]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // tile it+1 has landed (requested a whole iteration ago)
    __syncthreads();
  }

  // ---- normalise and store: lane holds query lr, d = 32dt + 8g' + 4lh + e
  const float l_tot = l_run + __shfl_xor(l_run, 32);
  const float inv = 1.0f / l_tot;
  const int qrow = q0 + lr;
  if (qrow < p.Sq) {
    bf16* op = p.o + ((size_t)b * p.Sq + qrow) * p.ldo + 4 * lh;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int gg = 0; gg < 4; ++gg) {
        u32x2 pk = {pack_bf2(o[dt][4 * gg] * inv, o[dt][4 * gg + 1] * inv), pack_bf2(o[dt][4 * gg + 2] * inv, o[dt][4 * gg + 3] * inv)};
        *(u32x2*)(op + 32 * dt + 8 * gg) = pk;
      }
  }
}

template <int D>
static int launch_wide(const AttnWideArgs& a, hipStream_t stream) {
  const size_t lds = (size_t)2 * 2 * 32 * D * 2;
  DMX_LDS_OPT_IN((dmx_attn_wide_kernel<D>), lds);
  hipLaunchKernelGGL((dmx_attn_wide_kernel<D>), dim3(cdiv(a.Sq, 128), a.B), dim3(256), lds, stream, a);
  return dmx_check_launch("dmx_attn_wide_kernel");
}

bool dmx_attention_wide_supported(int D) { return D == 128 || D == 256 || D == 512; }

int dmx_attention_wide_launch(AttnWideArgs a, hipStream_t stream) {
  DMX_REQUIRE(a.B > 0 && a.Sq > 0 && a.Skv > 0 && a.kv_rows >= a.Skv, "attention_wide: bad problem B=%d Sq=%d Skv=%d kv_rows=%d", a.B, a.Sq, a.Skv, a.kv_rows);
  DMX_REQUIRE(dmx_attention_wide_supported(a.D), "attention_wide: head width %d unsupported (128, 256, 512)", a.D);
  DMX_REQUIRE(a.ldq % 8 == 0 && a.ldk % 8 == 0 && a.ldv % 8 == 0 && a.ldo % 4 == 0, "attention_wide: strides must be multiples of 8");
  int rc = dmx_zero_page(&a.zeros);
  if (rc) return rc;
  if (a.D == 512) return launch_wide<512>(a, stream);
  if (a.D == 256) return launch_wide<256>(a, stream);
  return launch_wide<128>(a, stream);
}
