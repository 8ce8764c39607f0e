"""On-device pre/post-processing around the denoise loop: the host mirror of the notebook's helper
functions (app.ipynb:370-383 mask, :674-720 crop ladder / origin, :722-745 resize + normalise, :776-779 latent mask,
:825-846 paste-back) over the HIP kernels in csrc/prepost.hip.  The uint8 image is uploaded once; the three network inputs
come out of one kernel and the result is pasted back by another - no PIL / cv2 / albumentations and no second PCIe hop."""
import numpy as np
import torch

from . import _cabi


def crop_scale_for(location, h, w):
    """crop-size ladder (app.ipynb:674-695)"""
    char_height = int(location[3] - location[1]); char_lenth = int(location[2] - location[0])
    short_side = min(h, w)
    crop_lenth = 6 * char_height
    for bound in (128, 256, 384, 512, 640, 784, 1000):
        if 6 * char_height < bound:
            crop_lenth = max(bound, char_lenth)
            break
    return min(crop_lenth, short_side) if char_lenth < crop_lenth else short_side


def crop_origin(location, crop_scale, w, rng=np.random):
    """crop origin (app.ipynb:701-720), including the reference's use of the image WIDTH in the y branch"""
    x1, y1, x2, y2 = (int(v) for v in location[:4])

    def pick(a1, a2):
        if a2 - a1 < crop_scale:
            if a2 - crop_scale > 0:
                return a2 - crop_scale
            return a1 if a1 + crop_scale < w else 0
        return int(rng.randint(a1, max(0, a2 - crop_scale - 1)))
    return pick(x1, x2), pick(y1, y2)


def _u8(t, name):
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.uint8 and t.is_contiguous()):
        raise TypeError(f"{name}: expected a contiguous uint8 CUDA tensor")
    return t


def generate_mask(h, w, location, device):
    """generate_mask (app.ipynb:370-378) on the device: uint8 [h][w], 1 inside the inclusive box"""
    mask = torch.empty(h, w, dtype=torch.uint8, device=device)
    x0, y0, x1, y1 = (int(v) for v in location[:4])
    _cabi.check(_cabi.lib().dmx_mask_rasterize(_cabi.ptr(mask), h, w, x0, y0, x1, y1, _cabi.current_stream()), "mask_rasterize")
    return mask


def preprocess(instance_image, location, x_s, y_s, crop_scale, size=512):
    """instance_image: uint8 CUDA tensor [h][w][3].  Returns dict(image, masked_image: fp32 [1,3,S,S] in [-1,1];
    mask: uint8 [1,1,S,S]; mask_latent: fp32 [1,1,S/8,S/8]; mask_full: uint8 [h][w])."""
    img = _u8(instance_image, "instance_image")
    h, w, c = img.shape
    if c != 3:
        raise ValueError("instance_image must be HWC with 3 channels")
    dev = img.device
    mask_full = generate_mask(h, w, location, dev)
    S = int(size)
    image = torch.empty(1, 3, S, S, dtype=torch.float32, device=dev)
    masked = torch.empty(1, 3, S, S, dtype=torch.float32, device=dev)
    mask = torch.empty(1, 1, S, S, dtype=torch.uint8, device=dev)
    mask_lat = torch.empty(1, 1, S // 8, S // 8, dtype=torch.float32, device=dev)
    _cabi.check(_cabi.lib().dmx_preprocess_crop(_cabi.ptr(img), _cabi.ptr(mask_full), h, w, int(x_s), int(y_s), int(crop_scale), S,
                                               _cabi.ptr(image), _cabi.ptr(masked), _cabi.ptr(mask), _cabi.ptr(mask_lat),
                                               _cabi.current_stream()), "preprocess_crop")
    return dict(image=image, masked_image=masked, mask=mask, mask_latent=mask_lat, mask_full=mask_full)


def postprocess(image_vae, instance_image, location, x_s, y_s, crop_scale):
    """image_vae: fp32 CUDA [1,3,S,S] (or [3,S,S]) decoder output in [-1,1]; returns the uint8 [h][w][3] result with the
    text box replaced (app.ipynb:825-846)."""
    img = _u8(instance_image, "instance_image")
    h, w, _ = img.shape
    v = image_vae.reshape(-1, image_vae.shape[-2], image_vae.shape[-1])
    if v.shape[0] != 3 or v.shape[1] != v.shape[2]:
        raise ValueError("image_vae must be one square 3-channel image")
    v = v.to(torch.float32).contiguous()
    out = torch.empty_like(img)
    x1, y1, x2, y2 = (int(t) for t in location[:4])
    _cabi.check(_cabi.lib().dmx_postprocess_paste(_cabi.ptr(v), int(v.shape[-1]), _cabi.ptr(img), _cabi.ptr(out), h, w, int(x_s), int(y_s),
                                                 int(crop_scale), x1, y1, x2, y2, _cabi.current_stream()), "postprocess_paste")
    return out
