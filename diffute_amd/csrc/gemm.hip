// Fused implicit-GEMM conv / linear kernel for gfx950 (bf16 MFMA, fp32 accumulate).
//
// One kernel covers SURVEY.md 8a rows K1 (conv3x3 s1), K1s (stride 2, incl. the VAE's
// asymmetric pad), K2 (conv1x1 / resnet shortcut), K7 (linear), K8 (GEGLU feed-forward)
// and K10 (nearest x2 upsample and skip-concat folded into the operand gather).
//
//   out[m][n] = epilogue( sum_k X[m][k] * W[n][k] )
//
// X is never materialised: row m is an output pixel of an NHWC tensor and k walks
// (tap, channel) of the 3x3 window, read straight from the activation tensor(s) with
// `global_load_lds` (16 B per lane, DMA into LDS).  Padding / tails read a zero page.
// A second source splits the channel range (UNet skip concat) and an optional third
// K-segment appends the 1x1 shortcut of a ResnetBlock2D so conv2 + shortcut + residual
// is one launch.
//
// Tiling: block = 256 threads = 4 waves (2 along m x 2 along n); block tile 128(m) x
// {128,64}(n) x 64(k); per wave 2 x TN MFMA tiles of v_mfma_f32_32x32x16_bf16.  The
// weight tile is the MFMA A operand (rows = n) and the activation tile the B operand
// (cols = m) so each lane ends up with 4 consecutive output channels of one pixel
// (8-byte bf16 / 16-byte fp32 stores).  LDS tiles are [rows][64] bf16 with the 16-byte
// k-chunk XOR-swizzled by ((row>>1)&7) on the SOURCE side (glds writes lane-linear), which
// makes the ds_read_b128 fragment reads bank-conflict free.  Two LDS stages (64 KB at
// BN=128) -> 2 blocks per CU; the next K-tile's DMA is in flight while the current one
// is multiplied.
#include "common.h"
#include "kernels.h"

#define BM 128
#define BK 64

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

struct RowSrc {          // per-thread state for one staged activation row
  int bbase;             // b * IH * IW   (pixel index base) or plain row index
  int iy0, ix0;          // oy*stride - pad_t, ox*stride - pad_l
  int valid;             // m < M
};

template <int TN>
__global__ __launch_bounds__(256, 2) void dmx_gemm_kernel(const GemmArgs p) {
  constexpr int BN = 64 * TN;
  constexpr int X_BYTES = BM * BK * 2;
  constexpr int W_BYTES = BN * BK * 2;
  constexpr int STAGE = X_BYTES + W_BYTES;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave & 1, wn = wave >> 1;

  // ---- block -> tile mapping (XCD-aware: each XCD's L2 sees one n-tile at a time)
  int bid = blockIdx.x;
  const int nblk = gridDim.x;
  if ((nblk & 7) == 0) bid = (bid & 7) * (nblk >> 3) + (bid >> 3);
  const int tiles_m = (p.M + BM - 1) / BM;
  const int tile_n = bid / tiles_m;
  const int tile_m = bid - tile_n * tiles_m;
  const int m0 = tile_m * BM;
  const int n0 = tile_n * BN;

  const int nkt_total = p.K / BK;
  int kt_begin = 0, kt_end = nkt_total;
  if (p.splitk > 1) {
    kt_begin = blockIdx.y * p.kt_per_split;
    kt_end = min(kt_begin + p.kt_per_split, nkt_total);
  }

  // ---- per-thread staging rows: chunk q = t + 256*i -> row q>>3, slot q&7
  const int slot = t & 7;
  RowSrc xr[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + (t >> 3) + 32 * i;
    xr[i].valid = m < p.M;
    if (p.direct) {
      xr[i].bbase = m; xr[i].iy0 = 0; xr[i].ix0 = 0;
    } else {
      const int ohw = p.OH * p.OW;
      const int b = m / ohw;
      const int rem = m - b * ohw;
      const int oy = rem / p.OW;
      const int ox = rem - oy * p.OW;
      xr[i].bbase = b * p.IH * p.IW;
      xr[i].iy0 = oy * p.stride - p.pad;
      xr[i].ix0 = ox * p.stride - p.pad;
    }
  }
  const bf16* wrow[2 * TN];
  int kcw[2 * TN];
#pragma unroll
  for (int i = 0; i < 2 * TN; ++i) {
    const int r = (t >> 3) + 32 * i;
    const int n = n0 + r;
    wrow[i] = (n < p.N) ? (p.w + (size_t)n * p.ldw) : nullptr;
    kcw[i] = (slot ^ ((r >> 1) & 7)) * 8;
  }
  int kcx[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) kcx[i] = (slot ^ ((((t >> 3) + 32 * i) >> 1) & 7)) * 8;

  const int eh = p.ups ? 2 * p.IH : p.IH;   // extent of the (virtually upsampled) input grid
  const int ew = p.ups ? 2 * p.IW : p.IW;

  auto stage = [&](int kt, int buf) {
    char* xs = smem + buf * STAGE;
    char* ws = xs + X_BYTES;
    const int k0 = kt * BK;
    // ---- activation tile
    const bf16* src; int ld; int ci; int dy = 0, dx = 0; bool sc = false;
    if (k0 < p.Ktaps) {
      int tap = 0; ci = k0;
      if (p.ksize == 3) { tap = k0 / p.Cin; ci = k0 - tap * p.Cin; dy = tap / 3; dx = tap - dy * 3; }
      if (ci < p.cx0) { src = p.x0 + ci; ld = p.ldx0; } else { src = p.x1 + (ci - p.cx0); ld = p.ldx1; }
    } else {
      sc = true; ci = k0 - p.Ktaps;
      if (ci < p.cs0) { src = p.s0 + ci; ld = p.lds0; } else { src = p.s1 + (ci - p.cs0); ld = p.lds1; }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bf16* g = p.zeros;
      if (xr[i].valid) {
        if (p.direct) {
          g = src + (size_t)xr[i].bbase * ld + kcx[i];
        } else if (sc) {       // shortcut: 1x1 at the output pixel (shortcut source has the output grid)
          const int opix = m0 + (t >> 3) + 32 * i;
          g = src + (size_t)opix * ld + kcx[i];
        } else {
          const int iy = xr[i].iy0 + dy, ix = xr[i].ix0 + dx;
          if (iy >= 0 && iy < eh && ix >= 0 && ix < ew) {
            const int sy = p.ups ? (iy >> 1) : iy, sx = p.ups ? (ix >> 1) : ix;
            g = src + (size_t)(xr[i].bbase + sy * p.IW + sx) * ld + kcx[i];
          }
        }
      }
      char* l = xs + (wave * 64 + 256 * i) * 16;
      __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)l, 16, 0, 0);
    }
    // ---- weight tile
#pragma unroll
    for (int i = 0; i < 2 * TN; ++i) {
      const bf16* g = wrow[i] ? (wrow[i] + k0 + kcw[i]) : p.zeros;
      char* l = ws + (wave * 64 + 256 * i) * 16;
      __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)l, 16, 0, 0);
    }
  };

  f32x16 acc[TN][2];
#pragma unroll
  for (int a = 0; a < TN; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  // fragment read offsets (bytes) for this lane: row part and the swizzle key
  const int lr = lane & 31, lh = lane >> 5;
  int xoff[2], xkey[2], woff[TN], wkey[TN];
#pragma unroll
  for (int b = 0; b < 2; ++b) { const int r = wm * 64 + b * 32 + lr; xoff[b] = r * 128; xkey[b] = (r >> 1) & 7; }
#pragma unroll
  for (int a = 0; a < TN; ++a) { const int r = wn * 32 * TN + a * 32 + lr; woff[a] = r * 128; wkey[a] = (r >> 1) & 7; }

  if (kt_begin < kt_end) stage(kt_begin, 0);
  for (int kt = kt_begin; kt < kt_end; ++kt) {
    const int buf = (kt - kt_begin) & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (kt + 1 < kt_end) stage(kt + 1, buf ^ 1);
    const char* xs = smem + buf * STAGE;
    const char* ws = xs + X_BYTES;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int kc = 2 * kk + lh;
      bf16x8 xf[2], wf[TN];
#pragma unroll
      for (int b = 0; b < 2; ++b) xf[b] = *(const bf16x8*)(xs + xoff[b] + ((kc ^ xkey[b]) << 4));
#pragma unroll
      for (int a = 0; a < TN; ++a) wf[a] = *(const bf16x8*)(ws + woff[a] + ((kc ^ wkey[a]) << 4));
#pragma unroll
      for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[a], xf[b], acc[a][b], 0, 0, 0);
    }
  }

  // ---------------------------------------------------------------- epilogue
  // acc[a][b][4g+e] = out[m = m0 + wm*64 + b*32 + lr][n = n0 + wn*32*TN + a*32 + 8g + 4lh + e]
  if (p.splitk > 1) {
    float* part = p.partial + (size_t)blockIdx.y * p.M * p.N;
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int m = m0 + wm * 64 + b * 32 + lr;
        if (m >= p.M) continue;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int n = n0 + wn * 32 * TN + a * 32 + 8 * g + 4 * lh;
          float* o = part + (size_t)m * p.N + n;
          if (n + 3 < p.N && (p.N & 3) == 0) {
            f32x4 v = {acc[a][b][4 * g], acc[a][b][4 * g + 1], acc[a][b][4 * g + 2], acc[a][b][4 * g + 3]};
            *(f32x4*)o = v;
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) if (n + e < p.N) o[e] = acc[a][b][4 * g + e];
          }
        }
      }
    return;
  }

  if (p.geglu) {
    if constexpr (TN == 2) {
      // packed weight rows: [32 'a' rows | 32 matching 'b' rows] per 64-row group
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int m = m0 + wm * 64 + b * 32 + lr;
        if (m >= p.M) continue;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int nl = 8 * g + 4 * lh;
          const int na = n0 + wn * 64 + nl;          // packed row of the 'a' half
          const int j = (n0 + wn * 64) / 2 + nl;     // output column
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float av = acc[0][b][4 * g + e] + p.bias[na + e];
            const float bv = acc[1][b][4 * g + e] + p.bias[na + 32 + e];
            v[e] = av * gelu_erf_f(bv);
          }
          u32x2 pk = {pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
          *(u32x2*)((bf16*)p.out + (size_t)m * p.ldo + j) = pk;
        }
      }
    }
    return;
  }

  const bool vec_ok = ((p.N & 3) == 0) && ((p.ldo & 3) == 0);
#pragma unroll
  for (int a = 0; a < TN; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int m = m0 + wm * 64 + b * 32 + lr;
      if (m >= p.M) continue;
      const float* rb = p.rowbias ? (p.rowbias + (size_t)(m / p.rows_per_group) * p.ldrb) : nullptr;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = n0 + wn * 32 * TN + a * 32 + 8 * g + 4 * lh;
        if (n >= p.N) continue;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[a][b][4 * g + e];
        if (vec_ok && n + 3 < p.N) {
          if (p.bias) { const f32x4 bv = *(const f32x4*)(p.bias + n);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += bv[e]; }
          if (rb) { const f32x4 bv = *(const f32x4*)(rb + n);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += bv[e]; }
          if (p.res) {
            const u32x2 rv = *(const u32x2*)(p.res + (size_t)m * p.ldres + n);
            v[0] += __uint_as_float(rv[0] << 16); v[1] += __uint_as_float(rv[0] & 0xffff0000u);
            v[2] += __uint_as_float(rv[1] << 16); v[3] += __uint_as_float(rv[1] & 0xffff0000u);
          }
          if (p.out_f32) {
            f32x4 o = {v[0], v[1], v[2], v[3]};
            *(f32x4*)((float*)p.out + (size_t)m * p.ldo + n) = o;
          } else {
            u32x2 pk = {pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
            *(u32x2*)((bf16*)p.out + (size_t)m * p.ldo + n) = pk;
          }
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if (n + e >= p.N) continue;
            float x = v[e];
            if (p.bias) x += p.bias[n + e];
            if (rb) x += rb[n + e];
            if (p.res) x += bf_bits2f(*(const unsigned short*)(p.res + (size_t)m * p.ldres + n + e));
            if (p.out_f32) ((float*)p.out)[(size_t)m * p.ldo + n + e] = x;
            else ((unsigned short*)p.out)[(size_t)m * p.ldo + n + e] = f2bf_bits(x);
          }
        }
      }
    }
}

// split-K second pass: sum partials in fixed order, then the same epilogue.
__global__ __launch_bounds__(256) void dmx_splitk_reduce_kernel(const GemmArgs p) {
  const size_t total4 = (size_t)p.M * p.N / 4;
  const size_t MN = (size_t)p.M * p.N;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (size_t)gridDim.x * blockDim.x) {
    const size_t e0 = i * 4;
    const int m = (int)(e0 / p.N);
    const int n = (int)(e0 - (size_t)m * p.N);
    f32x4 s = *(const f32x4*)(p.partial + e0);
    for (int k = 1; k < p.splitk; ++k) {
      const f32x4 q = *(const f32x4*)(p.partial + (size_t)k * MN + e0);
      s += q;
    }
    float v[4] = {s[0], s[1], s[2], s[3]};
    if (p.bias) { const f32x4 bv = *(const f32x4*)(p.bias + n); for (int e = 0; e < 4; ++e) v[e] += bv[e]; }
    if (p.rowbias) { const f32x4 bv = *(const f32x4*)(p.rowbias + (size_t)(m / p.rows_per_group) * p.ldrb + n); for (int e = 0; e < 4; ++e) v[e] += bv[e]; }
    if (p.res) {
      const u32x2 rv = *(const u32x2*)(p.res + (size_t)m * p.ldres + n);
      v[0] += __uint_as_float(rv[0] << 16); v[1] += __uint_as_float(rv[0] & 0xffff0000u);
      v[2] += __uint_as_float(rv[1] << 16); v[3] += __uint_as_float(rv[1] & 0xffff0000u);
    }
    if (p.out_f32) {
      f32x4 o = {v[0], v[1], v[2], v[3]};
      *(f32x4*)((float*)p.out + (size_t)m * p.ldo + n) = o;
    } else {
      u32x2 pk = {pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
      *(u32x2*)((bf16*)p.out + (size_t)m * p.ldo + n) = pk;
    }
  }
}

// ------------------------------------------------------------------------- host side
static const bf16* g_zero_page = nullptr;

int dmx_zero_page(const bf16** out) {
  if (!g_zero_page) {
    void* z = nullptr;
    DMX_HIP(hipMalloc(&z, 4096));
    DMX_HIP(hipMemset(z, 0, 4096));
    g_zero_page = (const bf16*)z;
  }
  *out = g_zero_page;
  return DMX_OK;
}

static int pick_tn(const GemmArgs& a) {
  if (a.geglu) return 2;
  if (a.N <= 64) return 1;
  const int r = a.N % 128;
  if (r != 0 && r <= 64) {                       // e.g. 320, 960: 64-wide tiles waste nothing
    return 1;
  }
  // occupancy: prefer 64-wide tiles when 128-wide ones cannot fill the chip
  const long blocks128 = (long)cdiv(a.M, BM) * cdiv(a.N, 128);
  if (blocks128 < 192 && a.K <= 1024) return 1;
  return 2;
}

void dmx_gemm_plan(const GemmArgs& a, int* tn_out, int* splitk_out, int* ktps_out) {
  const int tn = pick_tn(a);
  const int bn = 64 * tn;
  const long blocks = (long)cdiv(a.M, BM) * cdiv(a.N, bn);
  const int nkt = a.K / BK;
  int splitk = 1;
  if (!a.geglu && (a.N % 4) == 0 && blocks < 160 && nkt >= 8) {
    splitk = (int)((384 + blocks - 1) / blocks);
    if (splitk > nkt / 4) splitk = nkt / 4;
    if (splitk > 16) splitk = 16;
    if (splitk < 1) splitk = 1;
  }
  int ktps = cdiv(nkt, splitk);
  splitk = cdiv(nkt, ktps);
  *tn_out = tn; *splitk_out = splitk; *ktps_out = ktps;
}

size_t dmx_gemm_workspace_bytes(const GemmArgs& a) {
  int tn, sk, ktps;
  dmx_gemm_plan(a, &tn, &sk, &ktps);
  return sk > 1 ? (size_t)sk * a.M * a.N * sizeof(float) : 0;
}

int dmx_gemm_launch(GemmArgs a, void* workspace, size_t workspace_bytes, hipStream_t stream) {
  DMX_REQUIRE(a.M > 0 && a.N > 0 && a.K > 0, "gemm: empty problem M=%d N=%d K=%d", a.M, a.N, a.K);
  DMX_REQUIRE(a.K % BK == 0, "gemm: K=%d must be a multiple of %d", a.K, BK);
  DMX_REQUIRE(a.ldw % 8 == 0 && a.ldx0 % 8 == 0, "gemm: leading dimensions must be multiples of 8 (ldw=%d ldx0=%d)", a.ldw, a.ldx0);
  DMX_REQUIRE(a.cx0 % BK == 0 && a.Cin % BK == 0, "gemm: channel splits must be multiples of %d (Cin=%d cx0=%d)", BK, a.Cin, a.cx0);
  DMX_REQUIRE(a.Ktaps % BK == 0 && a.Ktaps <= a.K, "gemm: bad Ktaps=%d K=%d", a.Ktaps, a.K);
  if (a.Ktaps < a.K) DMX_REQUIRE(a.s0 != nullptr && a.cs0 % BK == 0, "gemm: shortcut segment needs s0 and aligned cs0");
  if (a.geglu) DMX_REQUIRE(a.bias && a.N % 128 == 0 && !a.out_f32 && !a.res && !a.rowbias, "gemm: GEGLU needs bias, N%%128==0, bf16 out");
  int rc = dmx_zero_page(&a.zeros);
  if (rc) return rc;
  int tn, sk, ktps;
  dmx_gemm_plan(a, &tn, &sk, &ktps);
  a.splitk = sk; a.kt_per_split = ktps;
  if (sk > 1) {
    const size_t need = (size_t)sk * a.M * a.N * sizeof(float);
    if (workspace == nullptr || workspace_bytes < need) {
      dmx_set_error("gemm: split-K needs %zu bytes of workspace, got %zu", need, workspace_bytes);
      return DMX_ERR_WORKSPACE;
    }
    a.partial = (float*)workspace;
  }
  const int bn = 64 * tn;
  dim3 grid(cdiv(a.M, BM) * cdiv(a.N, bn), sk, 1);
  const size_t lds = 2 * (size_t)(BM * BK * 2 + bn * BK * 2);
  // algorithmic work of this launch: 2*M*N*K flops; bytes = activations read once + weights + output
  const double flops = 2.0 * a.M * (double)a.N * a.K;
  const double bytes = 2.0 * ((double)a.M * (a.K / (a.direct ? 1 : (a.ksize * a.ksize))) + (double)a.N * a.K + (double)a.M * (a.geglu ? a.N / 2 : a.N));
  if (tn == 2) {
    ProfScope ps(PROF_GEMM128, stream, flops, bytes);
    static bool attr2 = false;
    if (!attr2) { (void)hipFuncSetAttribute((const void*)dmx_gemm_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr2 = true; }
    hipLaunchKernelGGL(dmx_gemm_kernel<2>, grid, dim3(256), lds, stream, a);
  } else {
    ProfScope ps(PROF_GEMM64, stream, flops, bytes);
    static bool attr1 = false;
    if (!attr1) { (void)hipFuncSetAttribute((const void*)dmx_gemm_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr1 = true; }
    hipLaunchKernelGGL(dmx_gemm_kernel<1>, grid, dim3(256), lds, stream, a);
  }
  rc = dmx_check_launch("dmx_gemm_kernel");
  if (rc) return rc;
  if (sk > 1) {
    const size_t total4 = (size_t)a.M * a.N / 4;
    int blocks = (int)((total4 + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    ProfScope ps(PROF_SPLITK, stream, 0.0, 4.0 * sk * (double)a.M * a.N);
    hipLaunchKernelGGL(dmx_splitk_reduce_kernel, dim3(blocks), dim3(256), 0, stream, a);
    rc = dmx_check_launch("dmx_splitk_reduce_kernel");
  }
  return rc;
}
