"""TEST INFRASTRUCTURE ONLY (oracle): numpy restatement of the reference's host-side pre/post-processing around the
denoise loop.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

PARITY UNPINNED for the resize rules: the reference calls albumentations.Resize / cv2.resize (app.ipynb:332-344,:840),
neither library is installed in this image, so `cv_resize_linear_*` restate OpenCV's published INTER_LINEAR algorithm
(imgproc/resize.cpp: coordinate rule, border handling, the 11-bit fixed-point path for uint8, the INTER_AREA switch on an
exact 2x downscale) and cannot be checked against the library here.  Everything else is pinned: the mask is drawn by PIL
itself (the reference's own call), normalisation / masking / nearest downsample / paste / rounding are plain numpy /
torch expressions copied in meaning from the cited lines.
"""
import numpy as np
import torch
from PIL import Image, ImageDraw


def generate_mask(im_shape_wh, ocr_locate):
    """app.ipynb:370-378 (PIL: both corners inclusive)"""
    mask = Image.new("L", tuple(int(v) for v in im_shape_wh), 0)
    ImageDraw.Draw(mask).rectangle(tuple(int(v) for v in ocr_locate[:4]), fill=1)
    return np.array(mask)


def prepare_mask_and_masked_image(image, mask):
    """app.ipynb:380-383"""
    return np.multiply(image, np.stack([mask < 0.5, mask < 0.5, mask < 0.5]).transpose(1, 2, 0)).astype(image.dtype)


def crop_scale_for(location, h, w):
    """the crop-size ladder, app.ipynb:674-695"""
    char_height = int(location[3] - location[1]); char_lenth = int(location[2] - location[0])
    short_side = min(h, w)
    for bound, size in ((128, 128), (256, 256), (384, 384), (512, 512), (640, 640), (784, 784), (1000, 1000)):
        if 6 * char_height < bound:
            crop_lenth = max(size, char_lenth); break
    else:
        crop_lenth = 6 * char_height
    return min(crop_lenth, short_side) if char_lenth < crop_lenth else short_side


def crop_origin(location, crop_scale, w, rng):
    """app.ipynb:701-720 (the y branch tests against w, as the reference does)"""
    x1, y1, x2, y2 = (int(v) for v in location[:4])
    if x2 - x1 < crop_scale:
        x_s = x2 - crop_scale if x2 - crop_scale > 0 else (x1 if x1 + crop_scale < w else 0)
    else:
        x_s = int(rng.randint(x1, max(0, x2 - crop_scale - 1)))
    if y2 - y1 < crop_scale:
        y_s = y2 - crop_scale if y2 - crop_scale > 0 else (y1 if y1 + crop_scale < w else 0)
    else:
        y_s = int(rng.randint(y1, max(0, y2 - crop_scale - 1)))
    return x_s, y_s


def _taps(dst, n, horizontal):
    """OpenCV: fx = (float)((d + 0.5) * scale - 0.5); s = floor(fx); fx -= s; horizontal taps off the border move onto it with
    weight 0, vertical taps only clamp the row index"""
    scale = float(n) / float(dst)
    f = ((np.arange(dst, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    w = (f - s.astype(np.float32)).astype(np.float32)
    if horizontal:
        lo = s < 0; s = np.where(lo, 0, s); w = np.where(lo, np.float32(0), w)
        hi = s >= n - 1; s = np.where(hi, n - 1, s); w = np.where(hi, np.float32(0), w)
        s0, s1 = s, np.minimum(s + 1, n - 1)
    else:
        s0, s1 = np.clip(s, 0, n - 1), np.clip(s + 1, 0, n - 1)
    f0 = (np.float32(1) - w).astype(np.float32); f1 = w.astype(np.float32)
    a0 = np.rint(f0 * np.float32(2048)).astype(np.int32); a1 = np.rint(f1 * np.float32(2048)).astype(np.int32)
    return s0, s1, f0, f1, a0, a1


def cv_resize_linear_u8(src, dsize_wh):
    """cv2.resize(src uint8 [h][w][(c)], (dw, dh), INTER_LINEAR) restated"""
    src = np.asarray(src); squeeze = src.ndim == 2
    if squeeze:
        src = src[:, :, None]
    h, w, _ = src.shape; dw, dh = dsize_wh
    s = src.astype(np.int32)
    if w == 2 * dw and h == 2 * dh:                       # INTER_AREA fast path on an exact 2x downscale
        out = (s[0::2, 0::2] + s[0::2, 1::2] + s[1::2, 0::2] + s[1::2, 1::2] + 2) >> 2
    else:
        x0, x1, _, _, ax0, ax1 = _taps(dw, w, True)
        y0, y1, _, _, ay0, ay1 = _taps(dh, h, False)
        rows = s[:, x0] * ax0[None, :, None] + s[:, x1] * ax1[None, :, None]              # [h][dw][c] int32
        r0, r1 = rows[y0], rows[y1]
        out = (((ay0[:, None, None] * (r0 >> 4)) >> 16) + ((ay1[:, None, None] * (r1 >> 4)) >> 16) + 2) >> 2
    out = out.astype(np.uint8)
    return out[:, :, 0] if squeeze else out


def cv_resize_linear_f32(src, dsize_wh):
    """cv2.resize(src float32 [h][w][c], (dw, dh)) (INTER_LINEAR) restated"""
    src = np.asarray(src, dtype=np.float32)
    h, w, _ = src.shape; dw, dh = dsize_wh
    if w == 2 * dw and h == 2 * dh:
        return ((((src[0::2, 0::2] + src[0::2, 1::2]) + src[1::2, 0::2]) + src[1::2, 1::2]) * np.float32(0.25)).astype(np.float32)
    x0, x1, fx0, fx1, _, _ = _taps(dw, w, True)
    y0, y1, fy0, fy1, _, _ = _taps(dh, h, False)
    rows = (src[:, x0] * fx0[None, :, None]).astype(np.float32) + (src[:, x1] * fx1[None, :, None]).astype(np.float32)
    return ((rows[y0] * fy0[:, None, None]).astype(np.float32) + (rows[y1] * fy1[:, None, None]).astype(np.float32)).astype(np.float32)


def alb_normalize(img_u8):
    """albumentations.Normalize(mean=.5, std=.5, max_pixel_value=255): fp32 (x - 127.5) * (1 / 127.5)"""
    mean = np.float32(0.5) * np.float32(255); denom = np.float32(1) / (np.float32(0.5) * np.float32(255))
    return ((img_u8.astype(np.float32) - mean) * denom).astype(np.float32)


def preprocess(instance_image, location, x_s, y_s, crop_scale, S=512):
    """app.ipynb:699,:700,:722-745 and :776-779: returns the network inputs (CHW fp32 image / masked image, uint8 mask,
    mask at latent resolution) for one sample"""
    h, w, _ = instance_image.shape
    mask = generate_mask((w, h), location)
    masked = prepare_mask_and_masked_image(instance_image, mask)
    sl = (slice(y_s, y_s + crop_scale), slice(x_s, x_s + crop_scale))
    img_c, mask_c, masked_c = instance_image[sl], mask[sl], masked[sl]
    image = alb_normalize(cv_resize_linear_u8(img_c, (S, S))).transpose(2, 0, 1)
    masked_image = alb_normalize(cv_resize_linear_u8(masked_c, (S, S))).transpose(2, 0, 1)
    mask_s = cv_resize_linear_u8(mask_c, (S, S))
    mask_lat = torch.nn.functional.interpolate(torch.from_numpy(mask_s)[None, None].float(), size=(S // 8, S // 8))[0, 0].numpy()
    return dict(mask_full=mask, image=np.ascontiguousarray(image), masked_image=np.ascontiguousarray(masked_image), mask=mask_s, mask_latent=mask_lat)


def postprocess(image_vae, instance_image, location, x_s, y_s, crop_scale):
    """app.ipynb:825-846: image_vae fp32 [3][S][S] in [-1,1] -> uint8 [h][w][3] with the text box replaced.  Values outside
    [0,255] are clamped before the cast (numpy's astype is undefined there; the HIP kernel clamps too)."""
    h, w, _ = instance_image.shape
    x1, y1, x2, y2 = (int(v) for v in location[:4])
    image = ((np.asarray(image_vae, dtype=np.float32) / np.float32(2) + np.float32(0.5)) * np.float32(255.0)).astype(np.float32).transpose(1, 2, 0)
    r_h = h - y_s if y_s + crop_scale > h else crop_scale
    r_w = w - x_s if x_s + crop_scale > w else crop_scale
    inf_res = instance_image.astype(np.float32).copy()
    mid = instance_image.astype(np.float32).copy()
    mid[y_s:y_s + crop_scale, x_s:x_s + crop_scale, :] = cv_resize_linear_f32(image, (r_w, r_h))
    inf_res[y1:y2, x1:x2, :] = mid[y1:y2, x1:x2, :]
    return np.clip(np.rint(inf_res), 0, 255).astype(np.uint8)
