// Weight-gradient GEMM for gfx950:  dW[n][k] = sum_m dY[m][n] * X[m][k]   (training rows P5/K1/K7 of SURVEY.md 8a)
//
// The reduction runs over the ROWS of both operands (output pixels m), i.e. both MFMA operands are "transposed"
// relative to how they sit in memory.  Tiles of dY [32 m][128 n] and of the gathered activations X [32 m][64 c] are
// DMA'd into LDS row-major exactly like the forward kernel's tiles (same implicit-GEMM gather: filter tap, two-source
// channel concat, nearest x2 upsample, stride 2; padding reads a zero page), and the MFMA fragments are fetched with
// `ds_read_b64_tr_b16`, the LDS transpose read: a 16-lane group reads a [4 m][16 col] block and every lane receives
// the 4 m-values of ONE column, so two reads give a lane its 8 k-slots of v_mfma_f32_32x32x16_bf16.  Both operands use
// the same m -> k-slot mapping (m = 16 ks + 4 (lane>>5) + 8 i + j for read i, element j), so the contraction is
// consistent without any data movement.
//   block = 256 threads = 2 (n) x 2 (c) waves, wave tile 64 n x 32 c, block tile 128 n x 64 c; grid.z = split over m
//   (fp32 partials, fixed-order reduce -> deterministic).  A column tile never straddles a tap or a source tensor
//   (channel counts are multiples of 64).  The X fragment is the MFMA A operand, so a lane ends with 4 consecutive k
//   of one n and stores 16 B.
// LDS bank layout: the transpose read touches 4 rows x 32 B per 16 lanes, so the 16-byte chunks are XOR-swizzled by
// the row (on the DMA's SOURCE address, the DMA itself writes lane-linear): dY rows (256 B) chunk ^= 4*(row&3);
// X rows (128 B) chunk ^= 4*((row>>1)&1)  ->  each 32-lane phase covers all 64 banks once.
#include "kernels.h"

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

namespace {
constexpr int MC = 32;                       // m rows per stage
constexpr int BNW = 128, BCW = 64;           // block tile: n x c
constexpr int DY_ROWB = BNW * 2, X_ROWB = BCW * 2;
constexpr int DY_BYTES = MC * DY_ROWB, X_BYTES = MC * X_ROWB, STAGE = DY_BYTES + X_BYTES;   // 8 KB + 4 KB
constexpr int NST = 4;

#define DMX_TR4(V, A, O0, O1, O2, O3)                                                                     \
  asm volatile("ds_read_b64_tr_b16 %0, %4 offset:%5\n\tds_read_b64_tr_b16 %1, %4 offset:%6\n\t"          \
               "ds_read_b64_tr_b16 %2, %4 offset:%7\n\tds_read_b64_tr_b16 %3, %4 offset:%8"              \
               : "=&v"(V[0]), "=&v"(V[1]), "=&v"(V[2]), "=&v"(V[3])                                       \
               : "v"(A), "i"(O0), "i"(O1), "i"(O2), "i"(O3) : "memory")

__global__ __launch_bounds__(256, 2) void dmx_wgrad_kernel(const WgradArgs p) {
  __shared__ __attribute__((aligned(16))) char smem[NST * STAGE];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wn = wave & 1, wc = wave >> 1;
  const int n0 = blockIdx.x * BNW;
  const int k0 = blockIdx.y * BCW;                       // first output column (packed k index)
  const int m_begin = blockIdx.z * p.rows_per_split;
  const int m_end = min(m_begin + p.rows_per_split, p.M);
  const int nchunks = (m_end - m_begin + MC - 1) / MC;

  // ---- which tap / source tensor this column tile reads
  const int tap = p.direct ? 0 : k0 / p.Cin;
  const int cin_off = p.direct ? k0 : k0 - tap * p.Cin;
  const bool second = cin_off >= p.cx0;
  const bf16* xsrc = second ? p.x1 : p.x0;
  const int ldx = second ? p.ldx1 : p.ldx0;
  const int xcol = second ? cin_off - p.cx0 : cin_off;
  const int dyt = tap / p.ksize, dxt = tap - dyt * p.ksize;
  const int eh = p.ups ? 2 * p.IH : p.IH, ew = p.ups ? 2 * p.IW : p.IW;
  const int ohw = p.OH * p.OW;

  // ---- staging roles: dY tile = 512 chunks (2 per thread), X tile = 256 chunks (1 per thread)
  const int xr = t >> 3, xch = (t & 7) ^ (4 * ((xr >> 1) & 1));          // X: LDS row, SOURCE chunk
  int dr[2], dch[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int q = t + 256 * i;
    dr[i] = q >> 4;
    dch[i] = (q & 15) ^ (4 * (dr[i] & 3));
  }
  auto stage = [&](int buf, int mc) {                     // issues exactly 3 DMA loads per thread
    char* ds = smem + buf * STAGE;
    char* xs = ds + DY_BYTES;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int m = mc + dr[i];
      const int n = n0 + dch[i] * 8;
      const bf16* g = (m < m_end && n < p.N) ? p.dy + (size_t)m * p.lddy + n : p.zeros;
      __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(ds + (wave * 64 + 256 * i) * 16), 16, 0, 0);
    }
    {
      const int m = mc + xr;
      const bf16* g = p.zeros;
      if (m < m_end) {
        if (p.direct) {
          g = xsrc + (size_t)m * ldx + xcol + xch * 8;
        } else {
          const int b = m / ohw;
          const int rem = m - b * ohw;
          const int oy = rem / p.OW, ox = rem - oy * p.OW;
          const int iy = oy * p.stride - p.pad + dyt, ix = ox * p.stride - p.pad + dxt;
          if (iy >= 0 && iy < eh && ix >= 0 && ix < ew) {
            const int sy = p.ups ? (iy >> 1) : iy, sx = p.ups ? (ix >> 1) : ix;
            g = xsrc + ((size_t)b * p.IH * p.IW + (size_t)sy * p.IW + sx) * ldx + xcol + xch * 8;
          }
        }
      }
      __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(xs + wave * 64 * 16), 16, 0, 0);
    }
  };

  // ---- fragment read addresses (see header): group g16 = lane>>4 covers MFMA rows 16*(g16&1) + p16, m-half g16>>1
  const int p16 = lane & 15, g16 = lane >> 4;
  const int rsub = 4 * (g16 >> 1) + (p16 >> 2);           // row inside an 8-row slab; rsub & 3 == p16 >> 2
  unsigned dya[2], xa;
  {
    const unsigned base = (unsigned)(unsigned long long)(const void*)smem;
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      const int col = 64 * wn + 32 * nb + 16 * (g16 & 1) + 4 * (p16 & 3);
      dya[nb] = base + rsub * DY_ROWB + (((col >> 3) ^ (4 * (rsub & 3))) << 4) + (col & 7) * 2;
    }
    const int col = 32 * wc + 16 * (g16 & 1) + 4 * (p16 & 3);
    xa = base + DY_BYTES + rsub * X_ROWB + (((col >> 3) ^ (4 * ((rsub >> 1) & 1))) << 4) + (col & 7) * 2;
  }

  f32x16 acc[2];
#pragma unroll
  for (int nb = 0; nb < 2; ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;

  auto compute = [&](const int buf) {
    // rows 0..31 of the stage = k-steps 0,1; read i of a k-step starts at row 16 ks + 8 i
    unsigned long long xv[4], d0[4], d1[4];
    DMX_TR4(xv, xa + buf * STAGE, 0 * X_ROWB, 8 * X_ROWB, 16 * X_ROWB, 24 * X_ROWB);
    DMX_TR4(d0, dya[0] + buf * STAGE, 0 * DY_ROWB, 8 * DY_ROWB, 16 * DY_ROWB, 24 * DY_ROWB);
    DMX_TR4(d1, dya[1] + buf * STAGE, 0 * DY_ROWB, 8 * DY_ROWB, 16 * DY_ROWB, 24 * DY_ROWB);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const u32x4 av = {(unsigned)xv[2 * ks], (unsigned)(xv[2 * ks] >> 32), (unsigned)xv[2 * ks + 1], (unsigned)(xv[2 * ks + 1] >> 32)};
      const u32x4 b0 = {(unsigned)d0[2 * ks], (unsigned)(d0[2 * ks] >> 32), (unsigned)d0[2 * ks + 1], (unsigned)(d0[2 * ks + 1] >> 32)};
      const u32x4 b1 = {(unsigned)d1[2 * ks], (unsigned)(d1[2 * ks] >> 32), (unsigned)d1[2 * ks + 1], (unsigned)(d1[2 * ks + 1] >> 32)};
      acc[0] = DMX_MFMA_32x32x16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, b0), acc[0]);
      acc[1] = DMX_MFMA_32x32x16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, b1), acc[1]);
    }
  };

  // ---- ring: NST-1 stages of DMA in flight, counted vmcnt (3 loads per thread per stage), raw barriers
  int issued = 0;
#pragma unroll
  for (int s = 0; s < NST - 1; ++s) { stage(s, m_begin + issued * MC); ++issued; }   // past the end: zero-page loads
  for (int c = 0; c < nchunks; ++c) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * 3) : "memory");
    __builtin_amdgcn_s_barrier();
    stage((c + NST - 1) % NST, m_begin + issued * MC); ++issued;
    compute(c % NST);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  // ---- store: acc[nb][4g+e] = dW[n = n0 + 64wn + 32nb + (lane&31)][k = k0 + 32wc + 8g + 4(lane>>5) + e]
  float* out = p.out + (size_t)blockIdx.z * p.N * p.ldout;
  const int lr = lane & 31, lh = lane >> 5;
#pragma unroll
  for (int nb = 0; nb < 2; ++nb) {
    const int n = n0 + 64 * wn + 32 * nb + lr;
    if (n >= p.N) continue;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 v = {acc[nb][4 * g], acc[nb][4 * g + 1], acc[nb][4 * g + 2], acc[nb][4 * g + 3]};
      *(f32x4*)(out + (size_t)n * p.ldout + k0 + 32 * wc + 8 * g + 4 * lh) = v;
    }
  }
}

// out[n][k] (+)= sum_s partial[s][n][k], fixed order; partials are compact [N][K], out has row stride ldo
__global__ void dmx_sum_partials_kernel(const float* __restrict__ part, float* __restrict__ out, int N, int K, int ldo, int splits, int accumulate) {
  const int k4 = K / 4;
  const size_t n4 = (size_t)N * k4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    f32x4 s = ((const f32x4*)part)[i];
    for (int k = 1; k < splits; ++k) {
      const f32x4 v = ((const f32x4*)part)[(size_t)k * n4 + i];
      s[0] += v[0]; s[1] += v[1]; s[2] += v[2]; s[3] += v[3];
    }
    const size_t n = i / k4, c = (i - n * k4) * 4;
    f32x4* o = (f32x4*)(out + n * ldo + c);
    if (accumulate) { const f32x4 ov = *o; s[0] += ov[0]; s[1] += ov[1]; s[2] += ov[2]; s[3] += ov[3]; }
    *o = s;
  }
}

// Column sums of dY (bias gradient; per-image groups give the time-embedding row-bias gradient), two deterministic
// levels: partial[g*cpg + j][n] over CR-row chunks (256 threads = 4 row phases x 64 columns), then a fixed-order sum.
constexpr int CR = 256;
__global__ __launch_bounds__(256) void dmx_colsum_part_kernel(const bf16* __restrict__ dy, int lddy, int rows_per_group, int cpg, int N,
                                                              float* __restrict__ part) {
  // 256 threads = 32 row lanes x 8 column octets (64 columns, 16-byte loads); fixed-order fold through LDS
  __shared__ float red[32][65];
  const int oc = threadIdx.x & 7, rl = threadIdx.x >> 3;
  const int c = blockIdx.x * 64 + oc * 8;
  const int grp = blockIdx.y / cpg, j = blockIdx.y - grp * cpg;
  const int r0 = j * CR, r1 = min(r0 + CR, rows_per_group);
  float s[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) s[e] = 0.f;
  if (c < N) {
    const bf16* base = dy + (size_t)grp * rows_per_group * lddy + c;
    for (int r = r0 + rl; r < r1; r += 32) {
      float f[8]; unpack_bf8(*(const u32x4*)(base + (size_t)r * lddy), f);
#pragma unroll
      for (int e = 0; e < 8; ++e) s[e] += f[e];
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[rl][oc * 8 + e] = s[e];
  __syncthreads();
  if (threadIdx.x < 64) {
    const int cc = blockIdx.x * 64 + threadIdx.x;
    if (cc < N) {
      float v = 0.f;
#pragma unroll
      for (int k = 0; k < 32; ++k) v += red[k][threadIdx.x];
      part[(size_t)blockIdx.y * N + cc] = v;
    }
  }
}
// 256 threads = 8 partial lanes x 32 columns: the cpg partials of a column are independent loads spread over 8 lanes (a
// single thread walking up to 128 of them serially made this trivial kernel cost 12 us), folded in lane order
__global__ __launch_bounds__(256) void dmx_colsum_final_kernel(const float* __restrict__ part, int cpg, int N, float* __restrict__ out, int ldo, int accumulate) {
  __shared__ float red[8][33];
  const int cl = threadIdx.x & 31, l = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl, grp = blockIdx.y;
  float s = 0.f;
  if (c < N)
    for (int j = l; j < cpg; j += 8) s += part[((size_t)grp * cpg + j) * N + c];
  red[l][cl] = s;
  __syncthreads();
  if (threadIdx.x < 32 && c < N) {
    float v = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) v += red[k][cl];
    float* o = out + (size_t)grp * ldo + c;
    *o = accumulate ? *o + v : v;
  }
}
}  // namespace

static int wgrad_splits(const WgradArgs& a) {
  const long tiles = (long)cdiv(a.N, BNW) * (a.K / BCW);
  int s = (int)((1024 + tiles - 1) / tiles);                 // ~4 blocks per CU in flight
  const int max_s = cdiv(a.M, 8 * MC);                        // at least 8 chunks per split
  if (s > max_s) s = max_s;
  if (s > 64) s = 64;
  if (s < 1) s = 1;
  return s;
}

size_t dmx_wgrad_workspace_bytes(const WgradArgs& a) {
  const int s = wgrad_splits(a);
  return (s > 1 || a.accumulate) ? (size_t)s * a.N * a.K * sizeof(float) : 0;
}

int dmx_wgrad_launch(WgradArgs a, void* workspace, size_t workspace_bytes, hipStream_t stream) {
  DMX_REQUIRE(a.M > 0 && a.N > 0 && a.K > 0, "wgrad: empty problem M=%d N=%d K=%d", a.M, a.N, a.K);
  DMX_REQUIRE(a.K % BCW == 0 && a.N % 8 == 0 && a.lddy % 8 == 0 && a.ldx0 % 8 == 0 && a.ldout % 4 == 0, "wgrad: K %% 64, N %% 8 and leading dimensions %% 8 required (N=%d K=%d)", a.N, a.K);
  if (!a.direct) DMX_REQUIRE(a.Cin % BCW == 0 && a.cx0 % BCW == 0 && a.K == a.ksize * a.ksize * a.Cin, "wgrad: conv channel splits must be multiples of 64 (Cin=%d cx0=%d)", a.Cin, a.cx0);
  else DMX_REQUIRE(a.cx0 % BCW == 0 || a.cx0 >= a.K, "wgrad: source split must be a multiple of 64");
  if (a.cx0 < (a.direct ? a.K : a.Cin)) DMX_REQUIRE(a.x1 != nullptr && a.ldx1 % 8 == 0, "wgrad: second source missing");
  int rc = dmx_zero_page(&a.zeros);
  if (rc) return rc;
  const int splits = wgrad_splits(a);
  const size_t need = dmx_wgrad_workspace_bytes(a);
  float* final_out = a.out;
  if (need) {
    if (workspace == nullptr || workspace_bytes < need) {
      dmx_set_error("wgrad: needs %zu bytes of workspace, got %zu", need, workspace_bytes);
      return DMX_ERR_WORKSPACE;
    }
    a.out = (float*)workspace;
  }
  a.splits = splits;
  const int ld_final = a.ldout > 0 ? a.ldout : a.K;
  a.ldout = need ? a.K : ld_final;
  a.rows_per_split = cdiv(cdiv(a.M, splits), MC) * MC;
  dim3 grid(cdiv(a.N, BNW), a.K / BCW, cdiv(a.M, a.rows_per_split));
  a.splits = grid.z;
  {
    char tag[96];
    snprintf(tag, sizeof(tag), "wgrad M=%d N=%d K=%d splits=%d", a.M, a.N, a.K, a.splits);
    ProfScope ps(PROF_WGRAD, stream, 2.0 * a.M * (double)a.N * a.K, 2.0 * ((double)a.M * a.N + (double)a.M * a.K / (a.direct ? 1 : a.ksize * a.ksize)) + 4.0 * a.N * a.K, tag);
    hipLaunchKernelGGL(dmx_wgrad_kernel, grid, dim3(256), 0, stream, a);
  }
  rc = dmx_check_launch("dmx_wgrad_kernel");
  if (rc) return rc;
  if (need) {
    const size_t n4 = (size_t)a.N * a.K / 4;
    int blocks = (int)((n4 + 255) / 256); if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(dmx_sum_partials_kernel, dim3(blocks), dim3(256), 0, stream, (const float*)workspace, final_out, a.N, a.K, ld_final, a.splits, a.accumulate);
    rc = dmx_check_launch("dmx_sum_partials_kernel");
  }
  return rc;
}

size_t dmx_colsum_ws_bytes(int groups, int rows_per_group, int N) {
  return (size_t)groups * cdiv(rows_per_group, CR) * N * sizeof(float);
}

int dmx_colsum_launch(const bf16* dy, int lddy, int groups, int rows_per_group, int N, float* out, int ldo, int accumulate,
                      void* workspace, size_t workspace_bytes, hipStream_t stream) {
  DMX_REQUIRE(groups > 0 && rows_per_group > 0 && N > 0, "colsum: empty problem");
  DMX_REQUIRE(N % 8 == 0 && lddy % 8 == 0, "colsum: N=%d and lddy=%d must be multiples of 8", N, lddy);
  const int cpg = cdiv(rows_per_group, CR);
  const size_t need = dmx_colsum_ws_bytes(groups, rows_per_group, N);
  if (workspace == nullptr || workspace_bytes < need) {
    dmx_set_error("colsum: needs %zu bytes of workspace, got %zu", need, workspace_bytes);
    return DMX_ERR_WORKSPACE;
  }
  ProfScope ps(PROF_OTHER, stream, 0.0, 2.0 * groups * (double)rows_per_group * N, "colsum");
  hipLaunchKernelGGL(dmx_colsum_part_kernel, dim3(cdiv(N, 64), groups * cpg), dim3(256), 0, stream, dy, lddy, rows_per_group, cpg, N, (float*)workspace);
  int rc = dmx_check_launch("dmx_colsum_part_kernel");
  if (rc) return rc;
  hipLaunchKernelGGL(dmx_colsum_final_kernel, dim3(cdiv(N, 32), groups), dim3(256), 0, stream, (const float*)workspace, cpg, N, out, ldo, accumulate);
  return dmx_check_launch("dmx_colsum_final_kernel");
}
