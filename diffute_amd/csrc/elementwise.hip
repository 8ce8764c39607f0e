// Layout / glue kernels around the MFMA path.
// All HBM-bound and tiny next to the convs; written for coalesced 16-byte accesses where
// the data is wide enough to matter.
#include "common.h"
#include "kernels.h"

// ---- small-channel im2col (conv_in 9->320, VAE 3->128 / 4->512, quant convs).
// out[m][k], k = tap*C + c, zero padded to Kpad; reads NCHW fp32 model inputs (up to three
// tensors concatenated on channels: latents | mask | masked-image latents, app.ipynb:811)
// or one NHWC bf16 tensor.  One thread per 8 consecutive k (one 16-byte store).
__global__ __launch_bounds__(256) void dmx_im2col_small_kernel(const Im2colArgs p) {
  const int ko = p.Kpad >> 3;
  const size_t total = (size_t)p.B * p.OH * p.OW * ko;
  const int kreal = p.ksize * p.ksize * p.C;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const size_t m = idx / ko;
    const int k8 = (int)(idx - m * ko) * 8;
    const int ohw = p.OH * p.OW;
    const int b = (int)(m / ohw);
    const int rem = (int)(m - (size_t)b * ohw);
    const int oy = rem / p.OW, ox = rem - oy * p.OW;
    float f[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int k = k8 + i;
      float v = 0.f;
      if (k < kreal) {
        const int tap = k / p.C, c = k - tap * p.C;
        const int dy = tap / p.ksize, dx = tap - dy * p.ksize;
        const int iy = oy * p.stride + dy - p.pad, ix = ox * p.stride + dx - p.pad;
        if (iy >= 0 && iy < p.IH && ix >= 0 && ix < p.IW) {
          if (p.h) {
            v = bf_bits2f(*(const unsigned short*)(p.h + ((size_t)b * p.IH * p.IW + (size_t)iy * p.IW + ix) * p.ldh + c));
          } else {
            const float* src; int cc, cn;
            if (c < p.c0) { src = p.f0; cc = c; cn = p.c0; }
            else if (c < p.c0 + p.c1) { src = p.f1; cc = c - p.c0; cn = p.c1; }
            else { src = p.f2; cc = c - p.c0 - p.c1; cn = p.c2; }
            v = src[(((size_t)b * cn + cc) * p.IH + iy) * p.IW + ix];
          }
        }
      }
      f[i] = v;
    }
    *(u32x4*)(p.out + m * p.Kpad + k8) = pack_bf8(f);
  }
}

int dmx_im2col_small_launch(const Im2colArgs& a, hipStream_t stream) {
  DMX_REQUIRE(a.Kpad % 64 == 0 && a.Kpad >= a.ksize * a.ksize * a.C, "im2col: Kpad=%d too small / unaligned", a.Kpad);
  DMX_REQUIRE(a.h != nullptr || (a.f0 != nullptr && a.c0 + a.c1 + a.c2 == a.C), "im2col: bad sources");
  const size_t total = (size_t)a.B * a.OH * a.OW * (a.Kpad / 8);
  int blocks = (int)((total + 255) / 256); if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(dmx_im2col_small_kernel, dim3(blocks), dim3(256), 0, stream, a);
  return dmx_check_launch("dmx_im2col_small_kernel");
}

// ---- NHWC -> NCHW fp32 (model outputs: eps [B,4,h,w], image [B,3,H,W], moments [B,8,h,w])
template <typename T>
__global__ __launch_bounds__(256) void dmx_to_nchw_kernel(const T* in, int ldin, float* out, int B, int C, int HW) {
  const size_t total = (size_t)B * C * HW;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int pix = (int)(i % HW);
    const size_t bc = i / HW;
    const int c = (int)(bc % C);
    const size_t b = bc / C;
    const T v = in[(b * HW + pix) * ldin + c];
    if constexpr (sizeof(T) == 2) out[i] = bf_bits2f(*(const unsigned short*)&v);
    else out[i] = (float)v;
  }
}
int dmx_nhwc_to_nchw_f32_launch(const float* in, int ldin, float* out, int B, int C, int HW, hipStream_t stream) {
  const size_t total = (size_t)B * C * HW;
  int blocks = (int)((total + 255) / 256); if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(dmx_to_nchw_kernel<float>, dim3(blocks), dim3(256), 0, stream, in, ldin, out, B, C, HW);
  return dmx_check_launch("dmx_to_nchw_kernel<float>");
}
int dmx_nhwc_bf16_to_nchw_f32_launch(const bf16* in, int ldin, float* out, int B, int C, int HW, hipStream_t stream) {
  const size_t total = (size_t)B * C * HW;
  int blocks = (int)((total + 255) / 256); if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(dmx_to_nchw_kernel<bf16>, dim3(blocks), dim3(256), 0, stream, in, ldin, out, B, C, HW);
  return dmx_check_launch("dmx_to_nchw_kernel<bf16>");
}

// ---- fp32 -> bf16 cast
__global__ __launch_bounds__(256) void dmx_cast_kernel(const float* in, bf16* out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    ((unsigned short*)out)[i] = f2bf_bits(in[i]);
}
int dmx_cast_f32_to_bf16_launch(const float* in, bf16* out, size_t n, hipStream_t stream) {
  int blocks = (int)((n + 255) / 256); if (blocks > 4096) blocks = 4096; if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(dmx_cast_kernel, dim3(blocks), dim3(256), 0, stream, in, out, n);
  return dmx_check_launch("dmx_cast_kernel");
}

// ---- [B][S][C] (fp32 or bf16) -> bf16 [B][Spad][C], rows >= S zero (glyph context, 577 -> 640)
__global__ __launch_bounds__(256) void dmx_cast_pad_rows_kernel(const void* in, int in_is_bf16, bf16* out, int B, int S, int Spad, int C) {
  const size_t total = (size_t)B * Spad * C;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const size_t br = i / C;
    const int r = (int)(br % Spad);
    const size_t b = br / Spad;
    unsigned short v = 0;
    if (r < S) {
      const size_t src = (b * S + r) * C + c;
      v = in_is_bf16 ? ((const unsigned short*)in)[src] : f2bf_bits(((const float*)in)[src]);
    }
    ((unsigned short*)out)[i] = v;
  }
}
int dmx_cast_pad_rows_launch(const void* in, int in_is_bf16, bf16* out, int B, int S, int Spad, int C, hipStream_t stream) {
  const size_t total = (size_t)B * Spad * C;
  int blocks = (int)((total + 255) / 256); if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(dmx_cast_pad_rows_kernel, dim3(blocks), dim3(256), 0, stream, in, in_is_bf16, out, B, S, Spad, C);
  return dmx_check_launch("dmx_cast_pad_rows_kernel");
}

// ---- weight packing (fp32 torch layouts -> bf16 K-contiguous GEMM rows)
// conv [Cout][Cin][ks][ks] -> out[n][koff + tap*Cin + ci], row stride ldk
__global__ __launch_bounds__(256) void dmx_pack_conv_w_kernel(const float* w, bf16* out, int Cout, int Cin, int ks, int ldk, int koff) {
  const size_t total = (size_t)Cout * Cin * ks * ks;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int kk = ks * ks;
    const int ci = (int)(i % Cin);
    const size_t r = i / Cin;
    const int tap = (int)(r % kk);
    const int n = (int)(r / kk);
    const float v = w[((size_t)n * Cin + ci) * kk + tap];
    ((unsigned short*)out)[(size_t)n * ldk + koff + (size_t)tap * Cin + ci] = f2bf_bits(v);
  }
}
int dmx_pack_conv_weight_launch(const float* w, bf16* out, int Cout, int Cin, int ks, int ldk, int koff, hipStream_t stream) {
  const size_t total = (size_t)Cout * Cin * ks * ks;
  int blocks = (int)((total + 255) / 256); if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(dmx_pack_conv_w_kernel, dim3(blocks), dim3(256), 0, stream, w, out, Cout, Cin, ks, ldk, koff);
  return dmx_check_launch("dmx_pack_conv_w_kernel");
}
// ---- training: transposed packs for the data-gradient GEMMs (dX = dY * W runs through the forward kernel with the
// roles of the channel axes swapped), conv [Cout][Cin][ks][ks] -> out[ci][koff + flip(tap)*Cout + n]: the data
// gradient of a stride-1 conv is the conv of dY with the spatially flipped, channel-transposed filter.
__global__ __launch_bounds__(256) void dmx_pack_conv_w_t_kernel(const float* w, bf16* out, int Cout, int Cin, int ks, int ldk, int koff) {
  const size_t total = (size_t)Cout * Cin * ks * ks;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int kk = ks * ks;
    const int n = (int)(i % Cout);
    const size_t r = i / Cout;
    const int tapf = (int)(r % kk);
    const int ci = (int)(r / kk);
    const float v = w[((size_t)n * Cin + ci) * kk + (kk - 1 - tapf)];
    ((unsigned short*)out)[(size_t)ci * ldk + koff + (size_t)tapf * Cout + n] = f2bf_bits(v);
  }
}
int dmx_pack_conv_weight_t_launch(const float* w, bf16* out, int Cout, int Cin, int ks, int ldk, int koff, hipStream_t stream) {
  const size_t total = (size_t)Cout * Cin * ks * ks;
  int blocks = (int)((total + 255) / 256); if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(dmx_pack_conv_w_t_kernel, dim3(blocks), dim3(256), 0, stream, w, out, Cout, Cin, ks, ldk, koff);
  return dmx_check_launch("dmx_pack_conv_w_t_kernel");
}
// linear [rows][cols] -> bf16 [cols][ldo] (transpose), 32x32 tiles through LDS so both sides are coalesced
__global__ __launch_bounds__(256) void dmx_pack_rows_t_kernel(const float* w, bf16* out, int rows, int cols, int ldo) {
  __shared__ float tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int r = r0 + ty + 8 * j, c = c0 + tx;
    tile[ty + 8 * j][tx] = (r < rows && c < cols) ? w[(size_t)r * cols + c] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int c = c0 + ty + 8 * j, r = r0 + tx;
    if (c < cols && r < rows) ((unsigned short*)out)[(size_t)c * ldo + r] = f2bf_bits(tile[tx][ty + 8 * j]);
  }
}
int dmx_pack_rows_t_launch(const float* w, bf16* out, int rows, int cols, int ldo, hipStream_t stream) {
  hipLaunchKernelGGL(dmx_pack_rows_t_kernel, dim3(cdiv(cols, 32), cdiv(rows, 32)), dim3(256), 0, stream, w, out, rows, cols, ldo);
  return dmx_check_launch("dmx_pack_rows_t_kernel");
}
// stride-2 conv data gradient: dX = conv_s1(zero-inserted dY, flipped filter); z[b][2y][2x] = dy[b][y][x], 0 elsewhere
__global__ __launch_bounds__(256) void dmx_zero_insert2_kernel(const bf16* dy, int lddy, bf16* z, int B, int OH, int OW, int C) {
  const int c8 = C / 8;
  const size_t total = (size_t)B * 4 * OH * OW * c8;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % c8);
    size_t r = i / c8;
    const int x = (int)(r % (2 * OW)); r /= 2 * OW;
    const int y = (int)(r % (2 * OH));
    const int b = (int)(r / (2 * OH));
    u32x4 v = {0u, 0u, 0u, 0u};
    if (!(x & 1) && !(y & 1)) v = *(const u32x4*)(dy + ((size_t)(b * OH + (y >> 1)) * OW + (x >> 1)) * lddy + c * 8);
    *(u32x4*)(z + i * 8) = v;
  }
}
int dmx_zero_insert2_launch(const bf16* dy, int lddy, bf16* z, int B, int OH, int OW, int C, hipStream_t stream) {
  DMX_REQUIRE(C % 8 == 0 && lddy % 8 == 0, "zero_insert2: C %% 8");
  const size_t total = (size_t)B * 4 * OH * OW * (C / 8);
  int blocks = (int)((total + 255) / 256); if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(dmx_zero_insert2_kernel, dim3(blocks), dim3(256), 0, stream, dy, lddy, z, B, OH, OW, C);
  return dmx_check_launch("dmx_zero_insert2_kernel");
}
// zero `bytes` (a multiple of 16, 16-byte aligned) of device memory as an ordinary KERNEL node: the per-forward flag / statistics pools of the
// executors (exec.hip).  Not hipMemsetAsync: see Exec::zero_pool.
__global__ __launch_bounds__(256) void dmx_zero16_kernel(u32x4* p, size_t n16) {
  const u32x4 z = {0u, 0u, 0u, 0u};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) p[i] = z;
}
int dmx_zero16_launch(void* p, size_t bytes, hipStream_t stream) {
  DMX_REQUIRE(bytes % 16 == 0 && ((size_t)p & 15) == 0, "zero16: %zu bytes at %p are not 16-byte units", bytes, p);
  if (!bytes) return DMX_OK;
  const size_t n16 = bytes / 16;
  int blocks = (int)((n16 + 255) / 256); if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(dmx_zero16_kernel, dim3(blocks), dim3(256), 0, stream, (u32x4*)p, n16);
  return dmx_check_launch("dmx_zero16_kernel");
}
// nearest x2 upsample backward: dx[b][y][x] (+)= sum of the 2x2 block of du (bf16 or fp32 in; fp32 sum, one rounding)
template <bool F32IN>
__global__ __launch_bounds__(256) void dmx_sumpool2_kernel(const void* du_, int lddu, bf16* dx, int lddx, int B, int H, int W, int C, int accumulate) {
  const int c8 = C / 8;
  const size_t total = (size_t)B * H * W * c8;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % c8);
    size_t r = i / c8;
    const int x = (int)(r % W); r /= W;
    const int y = (int)(r % H);
    const int b = (int)(r / H);
    float s[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const size_t off = ((size_t)(b * 2 * H + 2 * y + (q >> 1)) * (2 * W) + 2 * x + (q & 1)) * lddu + c * 8;
      float f[8];
      if (F32IN) {
        const f32x4 a = *(const f32x4*)((const float*)du_ + off), bq = *(const f32x4*)((const float*)du_ + off + 4);
        f[0] = a[0]; f[1] = a[1]; f[2] = a[2]; f[3] = a[3]; f[4] = bq[0]; f[5] = bq[1]; f[6] = bq[2]; f[7] = bq[3];
      } else {
        unpack_bf8(*(const u32x4*)((const bf16*)du_ + off), f);
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) s[e] += f[e];
    }
    bf16* o = dx + ((size_t)(b * H + y) * W + x) * lddx + c * 8;
    if (accumulate) {
      float f[8]; unpack_bf8(*(const u32x4*)o, f);
#pragma unroll
      for (int e = 0; e < 8; ++e) s[e] += f[e];
    }
    *(u32x4*)o = pack_bf8(s);
  }
}
int dmx_sumpool2_launch(const void* du, int lddu, int du_f32, bf16* dx, int lddx, int B, int H, int W, int C, int accumulate, hipStream_t stream) {
  DMX_REQUIRE(C % 8 == 0 && lddu % 8 == 0 && lddx % 8 == 0, "sumpool2: C %% 8");
  const size_t total = (size_t)B * H * W * (C / 8);
  int blocks = (int)((total + 255) / 256); if (blocks > 8192) blocks = 8192;
  if (du_f32) hipLaunchKernelGGL(dmx_sumpool2_kernel<true>, dim3(blocks), dim3(256), 0, stream, du, lddu, dx, lddx, B, H, W, C, accumulate);
  else hipLaunchKernelGGL(dmx_sumpool2_kernel<false>, dim3(blocks), dim3(256), 0, stream, du, lddu, dx, lddx, B, H, W, C, accumulate);
  return dmx_check_launch("dmx_sumpool2_kernel");
}
// linear [rows][cols] -> bf16 [rows][ldo]; geglu=1 interleaves 32-row groups of the value and
// gate halves ([a0..a31 | b0..b31 | a32..a63 | ...]) so a 64-row MFMA wave tile holds matching pairs.
__device__ __forceinline__ int geglu_src_row(int r, int rows) {
  const int J = r >> 6, w = r & 63;
  return (w < 32) ? (32 * J + w) : (rows / 2 + 32 * J + (w - 32));
}
__global__ __launch_bounds__(256) void dmx_pack_rows_kernel(const float* w, bf16* out, int rows, int cols, int ldo, int geglu) {
  const size_t total = (size_t)rows * cols;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % cols);
    const int r = (int)(i / cols);
    const int sr = geglu ? geglu_src_row(r, rows) : r;
    ((unsigned short*)out)[(size_t)r * ldo + c] = f2bf_bits(w[(size_t)sr * cols + c]);
  }
}
int dmx_pack_rows_launch(const float* w, bf16* out, int rows, int cols, int ldo, int geglu, hipStream_t stream) {
  const size_t total = (size_t)rows * cols;
  int blocks = (int)((total + 255) / 256); if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(dmx_pack_rows_kernel, dim3(blocks), dim3(256), 0, stream, w, out, rows, cols, ldo, geglu);
  return dmx_check_launch("dmx_pack_rows_kernel");
}
__global__ void dmx_pack_geglu_bias_kernel(const float* b, float* out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = b[geglu_src_row(i, n)];
}
int dmx_pack_geglu_bias_launch(const float* b, float* out, int n, hipStream_t stream) {
  hipLaunchKernelGGL(dmx_pack_geglu_bias_kernel, dim3(cdiv(n, 256)), dim3(256), 0, stream, b, out, n);
  return dmx_check_launch("dmx_pack_geglu_bias_kernel");
}

// ---- NCHW fp32 -> NHWC bf16 (test helper / generic input conversion)
__global__ __launch_bounds__(256) void dmx_to_nhwc_kernel(const float* in, bf16* out, int ldo, int B, int C, int HW) {
  const size_t total = (size_t)B * C * HW;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const size_t bp = i / C;
    const int pix = (int)(bp % HW);
    const size_t b = bp / HW;
    ((unsigned short*)out)[bp * ldo + c] = f2bf_bits(in[(b * C + c) * HW + pix]);
  }
}
int dmx_nchw_f32_to_nhwc_bf16_launch(const float* in, bf16* out, int ldo, int B, int C, int HW, hipStream_t stream) {
  const size_t total = (size_t)B * C * HW;
  int blocks = (int)((total + 255) / 256); if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(dmx_to_nhwc_kernel, dim3(blocks), dim3(256), 0, stream, in, out, ldo, B, C, HW);
  return dmx_check_launch("dmx_to_nhwc_kernel");
}
