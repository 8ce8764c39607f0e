// On-device pre/post-processing around the denoise loop (reference app.ipynb:370-383 mask,
// :332-344 + :722-745 resize / normalise pipelines, :776-779 mask to latent resolution, :825-846 paste-back).
// The reference does all of this on the host with PIL / numpy / cv2 / albumentations and crosses PCIe twice; here the
// uint8 source image stays in HBM and the three 512x512 network inputs (image, masked image, mask) come out of ONE kernel,
// the result is pasted back by another.  Pure HBM-bound byte work: one thread per destination pixel, coalesced writes.
//
// Resize semantics follow OpenCV's cv::resize(INTER_LINEAR) as published (imgproc/resize.cpp), which is what
// albumentations.Resize and the notebook's cv2.resize call:
//   source coordinate fx = (dx + 0.5) * scale - 0.5, sx = floor(fx), a horizontal tap off the border is moved onto it with its weight reset, vertical taps only clamp the row;
//   uint8 images: fixed point - weights cvRound(w * 2048) as int16, horizontal sums kept as int32, vertical
//     dst = (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2;
//   float images: horizontal r = s0*a0 + s1*a1, vertical dst = r0*b0 + r1*b1 in fp32;
//   an exact 2x downscale is taken by the INTER_AREA fast path instead (2x2 mean; uint8 (a+b+c+d+2)>>2).
// cv2 / albumentations are not installed in the build image, so these rules are restated, not pinned against the library
// ("parity unpinned" in oracle/prepost.py, which the tests compare against bit for bit).
#include "common.h"
#include "kernels.h"
#include "../../include/diffute_hip.h"
#include <math.h>

namespace {
struct Tap { int s0, s1; short a0, a1; float f0, f1; };

// OpenCV's tap for destination index d: `n` source samples, `scale` = n / dst_size (double, like cv::resize).
//   fx = (float)((d + 0.5) * scale - 0.5); s = cvFloor(fx); fx -= s;
// Horizontally a tap that falls off the border is moved onto it and its weight reset (s < 0 -> s = 0, fx = 0;
// s >= n-1 -> s = n-1, fx = 0); vertically only the row indices are clamped and the weights are kept.
__device__ __forceinline__ Tap tap_for(int d, int n, double scale, bool horizontal) {
  float f = (float)(((double)d + 0.5) * scale - 0.5);
  int s = (int)floorf(f);
  float w = f - (float)s;
  Tap t;
  if (horizontal) {
    if (s < 0) { s = 0; w = 0.f; }
    if (s >= n - 1) { s = n - 1; w = 0.f; }
    t.s0 = s; t.s1 = min(s + 1, n - 1);
  } else {
    t.s0 = min(max(s, 0), n - 1); t.s1 = min(max(s + 1, 0), n - 1);
  }
  t.f0 = 1.f - w; t.f1 = w;
  t.a0 = (short)__float2int_rn(t.f0 * 2048.f);       // saturate_cast<short>(cvRound(.)); |value| <= 2048
  t.a1 = (short)__float2int_rn(t.f1 * 2048.f);
  return t;
}
__device__ __forceinline__ int vert_u8(int r0, int r1, short b0, short b1) {
  return ((((int)b0 * (r0 >> 4)) >> 16) + (((int)b1 * (r1 >> 4)) >> 16) + 2) >> 2;
}

struct PreArgs {
  const unsigned char* img; const unsigned char* mask; int H, W;       // HWC uint8 image, [H][W] mask of {0,1}
  int xs, ys, cw, ch;                                                  // crop origin and (clipped) extent
  int S;                                                               // network resolution (512)
  float* out_img; float* out_masked; unsigned char* out_mask; float* out_mask_lat;   // [3][S][S], [3][S][S], [S][S], [S/8][S/8]
  double sx, sy; int area2;
};

__global__ __launch_bounds__(256) void dmx_preprocess_kernel(const PreArgs p) {
  const int dx = blockIdx.x * blockDim.x + threadIdx.x, dy = blockIdx.y;
  if (dx >= p.S) return;
  int vi[3], vm[3], vk;
  auto px = [&](int y, int x, int c) -> int { return p.img[((size_t)(p.ys + y) * p.W + p.xs + x) * 3 + c]; };
  auto mk = [&](int y, int x) -> int { return p.mask[(size_t)(p.ys + y) * p.W + p.xs + x]; };
  if (p.area2) {
    const int x0 = 2 * dx, y0 = 2 * dy;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      int a = 0, b = 0;
      for (int j = 0; j < 2; ++j)
        for (int i = 0; i < 2; ++i) { const int v = px(y0 + j, x0 + i, c); a += v; b += mk(y0 + j, x0 + i) ? 0 : v; }
      vi[c] = (a + 2) >> 2; vm[c] = (b + 2) >> 2;
    }
    vk = (mk(y0, x0) + mk(y0, x0 + 1) + mk(y0 + 1, x0) + mk(y0 + 1, x0 + 1) + 2) >> 2;
  } else {
    const Tap tx = tap_for(dx, p.cw, p.sx, true), ty = tap_for(dy, p.ch, p.sy, false);
    const int m00 = mk(ty.s0, tx.s0), m01 = mk(ty.s0, tx.s1), m10 = mk(ty.s1, tx.s0), m11 = mk(ty.s1, tx.s1);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int s00 = px(ty.s0, tx.s0, c), s01 = px(ty.s0, tx.s1, c), s10 = px(ty.s1, tx.s0, c), s11 = px(ty.s1, tx.s1, c);
      vi[c] = vert_u8(s00 * tx.a0 + s01 * tx.a1, s10 * tx.a0 + s11 * tx.a1, ty.a0, ty.a1);
      // masked_image = image * (mask < 0.5) before the resize (prepare_mask_and_masked_image, app.ipynb:380-383)
      vm[c] = vert_u8((m00 ? 0 : s00) * tx.a0 + (m01 ? 0 : s01) * tx.a1, (m10 ? 0 : s10) * tx.a0 + (m11 ? 0 : s11) * tx.a1, ty.a0, ty.a1);
    }
    vk = vert_u8(m00 * tx.a0 + m01 * tx.a1, m10 * tx.a0 + m11 * tx.a1, ty.a0, ty.a1);
  }
  const size_t plane = (size_t)p.S * p.S, o = (size_t)dy * p.S + dx;
  // albumentations.Normalize(mean 0.5, std 0.5, max_pixel_value 255): (x - 127.5) * (1 / 127.5) in fp32
  const float mean = 0.5f * 255.f, inv = 1.0f / (0.5f * 255.f);
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    p.out_img[c * plane + o] = ((float)vi[c] - mean) * inv;
    p.out_masked[c * plane + o] = ((float)vm[c] - mean) * inv;
  }
  p.out_mask[o] = (unsigned char)vk;
  // F.interpolate(mask, size = S/8) (nearest): source index = floor(dst * 8)
  if (p.out_mask_lat && (dx & 7) == 0 && (dy & 7) == 0) p.out_mask_lat[(size_t)(dy >> 3) * (p.S >> 3) + (dx >> 3)] = (float)vk;
}

__global__ __launch_bounds__(256) void dmx_mask_rasterize_kernel(unsigned char* mask, int H, int W, int x0, int y0, int x1, int y1) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x < W) mask[(size_t)y * W + x] = (x >= x0 && x <= x1 && y >= y0 && y <= y1) ? 1 : 0;   // PIL rectangles include both corners
}

struct PostArgs {
  const float* vae; int S;                     // decoder output [3][S][S] in [-1, 1]
  const unsigned char* ori; unsigned char* out; int H, W;
  int xs, ys, rw, rh;                          // paste origin and the extent the S x S image is resized to
  int x1, y1, x2, y2;                          // text box: only these pixels are replaced
  double sx, sy; int area2;
};
__device__ __forceinline__ float post_src(const PostArgs& p, int c, int y, int x) {
  return (p.vae[((size_t)c * p.S + y) * p.S + x] / 2.f + 0.5f) * 255.0f;      // (image_vae / 2 + 0.5) * 255.0
}
__global__ __launch_bounds__(256) void dmx_postprocess_kernel(const PostArgs p) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= p.W) return;
  const size_t o = ((size_t)y * p.W + x) * 3;
  const bool in_box = x >= p.x1 && x < p.x2 && y >= p.y1 && y < p.y2;
  const int dx = x - p.xs, dy = y - p.ys;
  const bool in_crop = dx >= 0 && dx < p.rw && dy >= 0 && dy < p.rh;
  if (!(in_box && in_crop)) { p.out[o] = p.ori[o]; p.out[o + 1] = p.ori[o + 1]; p.out[o + 2] = p.ori[o + 2]; return; }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float v;
    if (p.area2) {
      v = (post_src(p, c, 2 * dy, 2 * dx) + post_src(p, c, 2 * dy, 2 * dx + 1) + post_src(p, c, 2 * dy + 1, 2 * dx) + post_src(p, c, 2 * dy + 1, 2 * dx + 1)) * 0.25f;
    } else {
      const Tap tx = tap_for(dx, p.S, p.sx, true), ty = tap_for(dy, p.S, p.sy, false);
      const float r0 = post_src(p, c, ty.s0, tx.s0) * tx.f0 + post_src(p, c, ty.s0, tx.s1) * tx.f1;
      const float r1 = post_src(p, c, ty.s1, tx.s0) * tx.f0 + post_src(p, c, ty.s1, tx.s1) * tx.f1;
      v = r0 * ty.f0 + r1 * ty.f1;
    }
    // inf_res.round().astype("uint8"): round half to even; values outside [0, 255] are clamped here (numpy's cast of an
    // out-of-range float is undefined behaviour - the one deliberate deviation)
    v = rintf(v);
    v = v < 0.f ? 0.f : (v > 255.f ? 255.f : v);
    p.out[o + c] = (unsigned char)v;
  }
}
}  // namespace

extern "C" int dmx_mask_rasterize(unsigned char* mask, int H, int W, int x0, int y0, int x1, int y1, dmx_stream_t stream) {
  DMX_REQUIRE(mask && H > 0 && W > 0 && H <= 65535, "mask_rasterize: bad arguments");
  hipLaunchKernelGGL(dmx_mask_rasterize_kernel, dim3(cdiv(W, 256), H), dim3(256), 0, (hipStream_t)stream, mask, H, W, x0, y0, x1, y1);
  return dmx_check_launch("dmx_mask_rasterize_kernel");
}

extern "C" int dmx_preprocess_crop(const unsigned char* image_hwc, const unsigned char* mask, int H, int W, int x_s, int y_s, int crop_scale,
                                   int S, float* out_image, float* out_masked_image, unsigned char* out_mask, float* out_mask_latent,
                                   dmx_stream_t stream) {
  DMX_REQUIRE(image_hwc && mask && out_image && out_masked_image && out_mask, "preprocess_crop: null argument");
  DMX_REQUIRE(H > 0 && W > 0 && S > 0 && S % 8 == 0 && S <= 65535 && crop_scale > 0, "preprocess_crop: bad sizes");
  DMX_REQUIRE(x_s >= 0 && y_s >= 0 && x_s < W && y_s < H, "preprocess_crop: crop origin (%d, %d) outside the %dx%d image", x_s, y_s, W, H);
  PreArgs p{};
  p.img = image_hwc; p.mask = mask; p.H = H; p.W = W; p.xs = x_s; p.ys = y_s;
  p.cw = crop_scale < W - x_s ? crop_scale : W - x_s;                  // numpy slicing clips the crop at the border
  p.ch = crop_scale < H - y_s ? crop_scale : H - y_s;
  p.S = S; p.out_img = out_image; p.out_masked = out_masked_image; p.out_mask = out_mask; p.out_mask_lat = out_mask_latent;
  p.sx = (double)p.cw / S; p.sy = (double)p.ch / S;
  p.area2 = (p.cw == 2 * S && p.ch == 2 * S) ? 1 : 0;
  hipLaunchKernelGGL(dmx_preprocess_kernel, dim3(cdiv(S, 256), S), dim3(256), 0, (hipStream_t)stream, p);
  return dmx_check_launch("dmx_preprocess_kernel");
}

extern "C" int dmx_postprocess_paste(const float* image_vae, int S, const unsigned char* original_hwc, unsigned char* out_hwc, int H, int W,
                                     int x_s, int y_s, int crop_scale, int x1, int y1, int x2, int y2, dmx_stream_t stream) {
  DMX_REQUIRE(image_vae && original_hwc && out_hwc, "postprocess_paste: null argument");
  DMX_REQUIRE(H > 0 && W > 0 && H <= 65535 && S > 0 && crop_scale > 0 && x_s >= 0 && y_s >= 0 && x_s < W && y_s < H, "postprocess_paste: bad sizes");
  PostArgs p{};
  p.vae = image_vae; p.S = S; p.ori = original_hwc; p.out = out_hwc; p.H = H; p.W = W; p.xs = x_s; p.ys = y_s;
  p.rh = (y_s + crop_scale > H) ? H - y_s : crop_scale;                // app.ipynb:831-839
  p.rw = (x_s + crop_scale > W) ? W - x_s : crop_scale;
  p.x1 = x1; p.y1 = y1; p.x2 = x2; p.y2 = y2;
  p.sx = (double)S / p.rw; p.sy = (double)S / p.rh;
  p.area2 = (S == 2 * p.rw && S == 2 * p.rh) ? 1 : 0;
  hipLaunchKernelGGL(dmx_postprocess_kernel, dim3(cdiv(W, 256), H), dim3(256), 0, (hipStream_t)stream, p);
  return dmx_check_launch("dmx_postprocess_kernel");
}
