// Balanced schedule of the head-dim-64 attention forward (attention.hip dmx_attn_d64_kernel<true, 1, 4>): stream-K over (query block, key tile) items.
//
// Why: the 64x64-level self-attention of the headline pass (B = 4, 5 heads, 4096 x 4096) is 640 blocks of 128 query rows on 768 block slots (three
// 4-wave blocks per CU): half the SIMDs run three waves, half two, and the launch ends with the three-wave SIMDs (112.9 us against 121.0 us for
// 768 blocks and 92.7 us for 512: EXPERIMENTS.md round 5).  Here the grid IS the 768 slots; slot k takes the global (query block, 64-key tile)
// iterations [T k / 768, T (k + 1) / 768) - 53 or 54 of a query block's 64 tiles - so every SIMD carries the same 2.5 waves' worth of work.
//
// A slot's range touches at most two query blocks (in general: a head part, whole blocks, a tail part).  The part that does NOT reach its query block's
// last key tile is a HELPER part: it runs FIRST and publishes (O, m, l) of its keys - fp32, write-through, then a flag.  The parts that reach the last
// tile are OWNER parts: they run afterwards, fold the helpers' partials in (fixed order: nearest slot first), normalise and store.  The helper of a
// query block is the slot(s) just in front of the owner, and they published at the START of their lives, so the owner does not wait in practice; it
// cannot deadlock while the grid is resident (3 blocks per CU: the launcher's grid) and, short of that, only the 8 slots at an XCD boundary wait for a block
// dispatched after them (the others wait for lower-numbered hardware blocks, which wait for nobody before they publish) - the spin is bounded all the same
// and RAISES (DMX_DEVK_ATTN_PEER).
//
// Arithmetic: the tile loop is dmx_attn_d64_kernel's (same fragments, same optimistic reference max); a split row's (O, m, l) halves are combined with
// exp2 factors as the online softmax combines tiles, so results differ from the unsplit kernel in the last bits (not bit-equal to it) and are
// bit-repeatable run to run (fixed split points, fixed fold order).
#include "common.h"
#include "kernels.h"

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((address_space(3))) s16x4* lds_s16x4;

#define SK_PART_QUADS 9                     // per wave: 8 quads of O (32 floats per lane) + (m, l, -, -)
#define SK_PART_BYTES (4 * SK_PART_QUADS * 64 * 16)      // per slot: 4 waves x 9 quads x 64 lanes x 16 B = 36 KB

__global__ __launch_bounds__(256, 3) void dmx_attn_d64_sk_kernel(const AttnArgs p) {
  constexpr int NW = 4, KTB = 64 * 128, VTB = 64 * 128;
  __shared__ __attribute__((aligned(16))) char smem[2 * (KTB + VTB)];
  __shared__ __attribute__((aligned(16))) char pf_dump[1024];
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int lr = lane & 31, lh = lane >> 5;
  const float sl2 = p.scale * 1.4426950408889634f;
  const int nt = (p.Skv + 63) >> 6, qpb = p.Sq >> 7;   // key tiles per query block; query blocks per (image, head)   (Sq % 128 == 0: launcher)
  const long long T = (long long)p.B * p.H * qpb * nt;
  // XCD-contiguous slots: hardware block s runs on XCD s % 8 (observed placement; only speed depends on it), so XCD x takes the slots [x ns / 8, (x + 1) ns / 8) -
  // one eighth of the (image, head) pairs, whose K / V (2.5 heads x 1 MB at the headline shape) then stay in that XCD's 4 MB L2 whatever tile each slot is at.
  // (The plain grid gets its L2 hits from lock step - all query blocks of a head start at tile 0 together; the slots of a stream-K schedule are at 768 different phases.)
  const int ns = gridDim.x;
  const int slot = ((ns & 7) == 0) ? (blockIdx.x & 7) * (ns >> 3) + (blockIdx.x >> 3) : blockIdx.x;
  auto lo_of = [&](int k) { return (int)(T * k / ns); };
  const int lo = lo_of(slot), hi = lo_of(slot + 1);
  if (hi <= lo) return;                                // (launcher: ns <= T)

  // LDS-DMA geometry: as in dmx_attn_d64_kernel (K: 16-byte chunk ^ ((row >> 1) & 7); V: 64-byte half ^ ((row >> 1) & 1), on the SOURCE address)
  const int drow = lane >> 3, dchk = lane & 7;
  const unsigned ldkb = (unsigned)p.ldk * 2u, ldvb = (unsigned)p.ldv * 2u;
  const unsigned kcb = (unsigned)((dchk ^ (((drow >> 1) + 4 * (wave & 1)) & 7)) << 4);
  const unsigned vcb = (unsigned)((((((dchk >> 2) ^ ((drow >> 1) & 1)) << 2) | (dchk & 3))) << 4);
  const unsigned koff = (unsigned)drow * ldkb + kcb, voff = (unsigned)drow * ldvb + vcb;
  const int ksw = (lh ^ ((lr >> 1) & 7)) << 4;
  int npf = 0; bool pf_open = true;                    // weight prefetch units (AttnArgs.pf): behind the second tile of the slot's first part only

  f32x16 o[2];
  float m_run, l_run;

  // one part: query block qb, key tiles [t0, t1)  ->  (o, m_run, l_run) of this wave's 32 query rows over those keys
  auto part = [&](const int qb, const int t0, const int t1) {
    const int bh = qb / qpb, qblk = qb - bh * qpb, b = bh / p.H, h = bh - b * p.H;
    const int q0 = qblk * 128 + wave * 32;
    bf16x8 qf[4];
    {
      const bf16* qp = p.q + ((size_t)b * p.Sq + q0 + lr) * p.ldq + h * 64 + 8 * lh;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) qf[kk] = *(const bf16x8*)(qp + 16 * kk);
    }
    const char* kdma = (const char*)(p.k + (size_t)b * p.kv_rows * p.ldk + h * 64);
    const char* vdma = (const char*)(p.v + (size_t)b * p.kv_rows * p.ldv + h * 64);
    auto stage = [&](int buf, int kv0) {
      char* ksd = smem + buf * (KTB + VTB);
      char* vsd = ksd + KTB;
      if (kv0 + 64 <= p.Skv) {
#pragma unroll
        for (int i = 0; i < 8 / NW; ++i) {
          const int r0 = 8 * (wave + NW * i);
          dmx_dma16((kdma + (size_t)(kv0 + r0) * ldkb + koff), DMX_LDS_ADDR((ksd + r0 * 128)));
          dmx_dma16((vdma + (size_t)(kv0 + r0) * ldvb + voff), DMX_LDS_ADDR((vsd + r0 * 128)));
        }
      } else {                                         // last, ragged tile: keys past Skv re-read the last valid row (their P is 0)
        const int last = p.Skv - 1 - kv0;
#pragma unroll
        for (int i = 0; i < 8 / NW; ++i) {
          const int r0 = 8 * (wave + NW * i);
          const unsigned rc = (unsigned)min(r0 + drow, last);
          dmx_dma16((kdma + (size_t)kv0 * ldkb + (rc * ldkb + kcb)), DMX_LDS_ADDR((ksd + r0 * 128)));
          dmx_dma16((vdma + (size_t)kv0 * ldvb + (rc * ldvb + vcb)), DMX_LDS_ADDR((vsd + r0 * 128)));
        }
      }
    };
#pragma unroll
    for (int i = 0; i < 16; ++i) { o[0][i] = 0.f; o[1][i] = 0.f; }
    m_run = -INFINITY; l_run = 0.f;
    stage(0, t0 * 64);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (covers the Q fragment loads as well: see the note in dmx_attn_d64_kernel)
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) asm volatile("" : "+v"(qf[kk]));
    __syncthreads();
    for (int it = t0; it < t1; ++it) {
      const int kv0 = it * 64, par = (it - t0) & 1;
      if (it + 1 < t1) stage(par ^ 1, kv0 + 64);
      const bool pf_now = pf_open && it == t0;
      if (pf_now) {
        pf_open = false;
        const int nblk = ns, blk = slot;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int nb = p.pf_bytes[r];
          for (int u = blk * NW + wave; u * 1024 < nb && npf < 3; u += nblk * NW, ++npf) {
            int off = u * 1024 + lane * 16; if (off > nb - 16) off = nb - 16;
            dmx_dma16(((const char*)p.pf[r] + off), DMX_LDS_ADDR(pf_dump));
          }
        }
      }
      const char* ks = smem + par * (KTB + VTB);
      const char* vs = ks + KTB;
      f32x16 s[2];
      {
        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        bf16x8 kf[2][4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
          for (int kt = 0; kt < 2; ++kt) kf[kt][kk] = *(const bf16x8*)(ks + (32 * kt + lr) * 128 + (((2 * kk) << 4) ^ ksw));
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
          for (int kt = 0; kt < 2; ++kt) s[kt] = DMX_MFMA_32x32x16(kf[kt][kk], qf[kk], kk == 0 ? zero : s[kt]);
        __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
      }
      if (kv0 + 64 > p.Skv) {
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int key = kv0 + 32 * kt + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (key >= p.Skv) s[kt][r] = -INFINITY;
          }
      }
      bf16x8 pf[4];
      auto exponentiate = [&](float mc) {
        float q0s = 0.f, q1s = 0.f;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            unsigned int w[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float a0 = __builtin_fmaf(s[kt][8 * u + 2 * e], sl2, mc), a1 = __builtin_fmaf(s[kt][8 * u + 2 * e + 1], sl2, mc);
              const float p0 = __builtin_amdgcn_exp2f(a0), p1 = __builtin_amdgcn_exp2f(a1);
              q0s += p0; q1s += p1;
              w[e] = pack_bf2(p0, p1);
            }
            u32x4 wv = {w[0], w[1], w[2], w[3]};
            pf[2 * kt + u] = __builtin_bit_cast(bf16x8, wv);
          }
        return q0s + q1s;
      };
      float psum = exponentiate(-m_run * sl2);
      if (__any(!(psum <= 8192.0f))) {
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
          for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[kt][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * sl2);
        m_run = m_new;
        l_run *= alpha;
#pragma unroll
        for (int i = 0; i < 16; ++i) { o[0][i] *= alpha; o[1][i] *= alpha; }
        psum = exponentiate(-m_run * sl2);
      }
      l_run += psum;
      {
        const int p16 = lane & 15, g = lane >> 4;
        const char* va = vs + (4 * (g >> 1) + (p16 >> 2)) * 128 + (16 * (g & 1) + 4 * (p16 & 3)) * 2;
        const int vsw = (p16 >> 3) & 1;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
          for (int s4 = 0; s4 < 4; ++s4) {
            const s16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(va + 64 * (dt ^ vsw) + (2 * s4) * 8 * 128));
            const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(va + 64 * (dt ^ vsw) + (2 * s4 + 1) * 8 * 128));
            const s16x8 vv = __builtin_shufflevector(lo4, hi4, 0, 1, 2, 3, 4, 5, 6, 7);
            o[dt] = DMX_MFMA_32x32x16(__builtin_bit_cast(bf16x8, vv), pf[s4], o[dt]);
          }
      }
      if (!pf_now || npf == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else if (npf == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
      else if (npf == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
      __syncthreads();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (a one-tile part leaves its prefetch units in flight behind its only wait)
  };

  // partial of slot k, wave w: quads [k][w][9][64 lanes]; buffer descriptors so that the cache policy can be named (sc1: write-through / L1 bypass)
  auto part_rsrc = [&](int k) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)((char*)p.sk_part + (size_t)k * SK_PART_BYTES), 0, SK_PART_BYTES, 0x00020000);
  };
  const int poff = (wave * SK_PART_QUADS * 64 + lane) * 16;

  const int qb_first = lo / nt, qb_last = (hi - 1) / nt;
  const bool has_helper = (hi % nt) != 0;              // the range stops inside query block qb_last
  if (has_helper) {
    const int g0 = qb_last * nt;
    part(qb_last, (lo > g0 ? lo : g0) - g0, hi - g0);
    const __amdgpu_buffer_rsrc_t rs = part_rsrc(slot);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const f32x4 v = {o[q >> 2][4 * (q & 3)], o[q >> 2][4 * (q & 3) + 1], o[q >> 2][4 * (q & 3) + 2], o[q >> 2][4 * (q & 3) + 3]};
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, poff + q * 1024, 0, 16);
    }
    {
      const f32x4 v = {m_run, l_run, 0.f, 0.f};
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, poff + 8 * 1024, 0, 16);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // write-through stores drained -> block barrier -> flag (relaxed, agent scope)
    __syncthreads();
    if (t == 0) __hip_atomic_store(p.sk_flags + slot, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  const int qb_own_last = has_helper ? qb_last - 1 : qb_last;
  for (int qb = qb_first; qb <= qb_own_last; ++qb) {
    const int g0 = qb * nt;
    const int t0 = (lo > g0 ? lo : g0) - g0;
    part(qb, t0, nt);
    if (t0 > 0) {
      // the slots in front of this one that hold the other keys of query block qb: slot - 1 down to the slot that holds tile g0
      int kf = slot - 1;
      while (kf > 0 && lo_of(kf) > g0) --kf;
      if (t == 0) {
        const long long tw0 = __builtin_amdgcn_s_memrealtime();
        for (int k = slot - 1; k >= kf; --k)
          while (__hip_atomic_load(p.sk_flags + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
            __builtin_amdgcn_s_sleep(2);
            if (__builtin_amdgcn_s_memrealtime() - tw0 > 4000000) { dmx_dev_raise(p.err, DMX_DEVK_ATTN_PEER, slot, k, qb, ns); break; }
          }
      }
      __syncthreads();
      for (int k = slot - 1; k >= kf; --k) {
        const __amdgpu_buffer_rsrc_t rs = part_rsrc(k);
        f32x4 hv[9];
#pragma unroll
        for (int q = 0; q < 9; ++q) hv[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, poff + q * 1024, 0, 16));
        const float mh = hv[8][0], lhh = hv[8][1];
        const float M = fmaxf(m_run, mh);
        const float a = __builtin_amdgcn_exp2f((m_run - M) * sl2), ah = __builtin_amdgcn_exp2f((mh - M) * sl2);
        l_run = l_run * a + lhh * ah;
        m_run = M;
#pragma unroll
        for (int q = 0; q < 8; ++q)
#pragma unroll
          for (int e = 0; e < 4; ++e) o[q >> 2][4 * (q & 3) + e] = o[q >> 2][4 * (q & 3) + e] * a + hv[q][e] * ah;
      }
    }
    // normalise and store: lane holds query q0 + lr, d = 32dt + 8g + 4lh + e
    {
      const int bh = qb / qpb, qblk = qb - bh * qpb, b = bh / p.H, h = bh - b * p.H;
      const int qrow = qblk * 128 + wave * 32 + lr;
      const float l_tot = l_run + __shfl_xor(l_run, 32);
      const float inv = 1.0f / l_tot;
      bf16* op = p.o + ((size_t)b * p.Sq + qrow) * p.ldo + h * 64 + 4 * lh;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          u32x2 pk = {pack_bf2(o[dt][4 * g] * inv, o[dt][4 * g + 1] * inv), pack_bf2(o[dt][4 * g + 2] * inv, o[dt][4 * g + 3] * inv)};
          *(u32x2*)(op + 32 * dt + 8 * g) = pk;
        }
    }
  }
}

// ---- host side
static int g_attn_balanced = 1;            // dmx_set_attn_balanced: 0 never, 1 where the plan says it pays, 2 wherever the kernel takes the problem (tests)
extern "C" int dmx_set_attn_balanced(int mode) { const int old = g_attn_balanced; g_attn_balanced = mode; dmx_plan_switch(DMX_SW_ATTN_BALANCED, mode); return old; }

static int sk_n_cus() {
  static int n = 0;
  if (!n) { int dev = 0; hipDeviceProp_t pr; if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) n = pr.multiProcessorCount; if (n <= 0) n = 256; }
  return n;
}
// slots of the balanced schedule for this problem, 0 = the plain grid.  Measured (EXPERIMENTS.md round 6, profiles/r06_attn_balanced_probe.txt): a slot pays ~7 us
// for its second prologue, the publish and the fold, and the slots lose the lock step that keeps a head's K / V tiles in L2.  That is a win where the plain
// grid leaves CUs idle or with half the work of their neighbours (at most two blocks per CU: 4096 x 4096 at batch 1, 59.6 -> 40.7 us, batch 2, 82 -> 64 us), a wash at the headline shape
// (2.5 blocks per CU on 3 slots: 130 vs 130 us stand-alone, 334.2 vs 334.3 ms per pass in situ) and a loss where the slots are evenly filled (batch 8: +7 %).
int dmx_attention_balanced_slots(const AttnArgs& a) {
  if (!g_attn_balanced || !a.v || a.lse || a.Sq % 128 || a.B <= 0) return 0;
  const long long nqb = (long long)a.B * a.H * (a.Sq / 128), nt = (a.Skv + 63) / 64, T = nqb * nt;
  const int ns = 3 * sk_n_cus();
  if (T < ns) return 0;
  if (g_attn_balanced >= 2) return ns;
  // (with other streams sharing the CUs - dmx_set_exclusive_device(0): micro-batches, a collective next to the step - the slots of a launch are not all
  // resident and the 8 slots at an XCD boundary would wait for blocks dispatched after them: correct (bounded, others retire), but not a plan to choose)
  if (!dmx_exclusive_device()) return 0;
  if (nqb > 2 * sk_n_cus() || T / ns < 12) return 0;
  return ns;
}
size_t dmx_attention_balanced_part_bytes(const AttnArgs& a) {
  const int ns = dmx_attention_balanced_slots(a);
  return ns ? (size_t)ns * SK_PART_BYTES : 0;
}
int dmx_attention_balanced_launch(AttnArgs a, hipStream_t stream) {
  const int ns = dmx_attention_balanced_slots(a);
  DMX_REQUIRE(ns > 0 && a.sk_part && a.sk_flags, "attention (balanced schedule): not planned for this problem / no workspace");
  DMX_REQUIRE(a.ldq % 8 == 0 && a.ldk % 8 == 0 && a.ldv % 8 == 0 && a.ldo % 4 == 0 && a.kv_rows >= a.Skv, "attention (balanced schedule): strides");
  a.err = dmx_dev_err_words();
  hipLaunchKernelGGL(dmx_attn_d64_sk_kernel, dim3(ns), dim3(256), 0, stream, a);
  dmx_profile_note_symbol("dmx_attn_d64_sk_kernel(AttnArgs)");
  return dmx_check_launch("dmx_attn_d64_sk_kernel");
}
