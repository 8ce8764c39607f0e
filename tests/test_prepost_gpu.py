"""on-device pre/post-processing (csrc/prepost.hip through the C-ABI) against the numpy restatement of the
notebook's host code (oracle/prepost.py) - bit-exact, since it is integer / byte work plus a handful of ordered fp32
operations.  The oracle's resize rules are restated from OpenCV's published algorithm (cv2 is not installed: "parity
unpinned" for that part); the mask comes from PIL itself."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

# name, (h, w), box (x1, y1, x2, y2), crop origin, crop_scale
PRE_CASES = [
    ("up_128", (300, 400), (150, 120, 230, 138), None, None),           # ladder -> 128, upscale x4
    ("clipped_256", (300, 400), (120, 100, 260, 140), None, None),      # ladder -> 256, crop clipped at the bottom (non-square)
    ("identity_512", (700, 900), (300, 300, 520, 380), (200, 150), 512),
    ("down_640", (800, 1000), (300, 300, 600, 400), (100, 60), 640),
    ("down_784", (1100, 1300), (400, 500, 900, 620), (250, 200), 784),
    ("area_1024", (1100, 1300), (400, 500, 900, 620), (70, 40), 1024),   # exact 2x downscale -> INTER_AREA path
    ("odd_333", (500, 500), (10, 10, 300, 60), (0, 0), 333),
    ("box_at_border", (256, 320), (0, 0, 319, 255), (0, 0), 256),
]


def _case(case):
    from oracle import prepost as OP
    name, (h, w), box, origin, crop = case
    rng = np.random.RandomState(abs(hash(name)) % (2 ** 31))
    img = rng.randint(0, 256, (h, w, 3), dtype=np.uint8)
    if crop is None:
        crop = OP.crop_scale_for(box, h, w)
        origin = OP.crop_origin(box, crop, w, np.random.RandomState(1))
    return img, list(box), origin[0], origin[1], crop


@pytest.mark.parametrize("case", PRE_CASES, ids=[c[0] for c in PRE_CASES])
def test_preprocess_matches_host_pipeline(cuda, case):
    import diffute_amd as D
    from oracle import prepost as OP
    img, box, x_s, y_s, crop = _case(case)
    want = OP.preprocess(img, box, x_s, y_s, crop)
    got = D.prepost.preprocess(torch.from_numpy(img).to(cuda), box, x_s, y_s, crop)
    assert np.array_equal(got["mask_full"].cpu().numpy(), want["mask_full"]), "rasterised mask differs from PIL's"
    assert np.array_equal(got["mask"][0, 0].cpu().numpy(), want["mask"]), "resized mask"
    assert np.array_equal(got["mask_latent"][0, 0].cpu().numpy(), want["mask_latent"]), "mask at latent resolution"
    assert np.array_equal(got["image"][0].cpu().numpy(), want["image"]), "resized + normalised image"
    assert np.array_equal(got["masked_image"][0].cpu().numpy(), want["masked_image"]), "resized + normalised masked image"
    # properties that do not depend on the resize rule
    assert got["image"].abs().max() <= 1.0 and set(np.unique(got["mask"].cpu().numpy())) <= {0, 1}
    m = got["mask"][0, 0].bool()
    inside = got["masked_image"][0][:, m]
    if crop <= 512 and inside.numel():                    # upscaling: the interior of the box is exactly "black" (-1)
        assert float(inside.min()) == -1.0


@pytest.mark.parametrize("case", PRE_CASES, ids=[c[0] for c in PRE_CASES])
def test_postprocess_matches_host_pipeline(cuda, case):
    import diffute_amd as D
    from oracle import prepost as OP
    img, box, x_s, y_s, crop = _case(case)
    g = torch.Generator().manual_seed(5)
    vae = (torch.randn(1, 3, 512, 512, generator=g) * 0.6).clamp(-1.3, 1.3)          # some values leave [-1, 1]: the clamp path
    want = OP.postprocess(vae[0].numpy(), img, box, x_s, y_s, crop)
    got = D.prepost.postprocess(vae.to(cuda), torch.from_numpy(img).to(cuda), box, x_s, y_s, crop).cpu().numpy()
    assert got.shape == img.shape and got.dtype == np.uint8
    assert np.array_equal(got, want)
    x1, y1, x2, y2 = box
    outside = np.ones(img.shape[:2], bool); outside[y1:y2, x1:x2] = False
    assert np.array_equal(got[outside], img[outside]), "pixels outside the text box must be untouched"


def test_preprocess_feeds_the_pipeline_inputs(cuda):
    """round trip of the two halves: a decoder output that equals the normalised crop pastes the original pixels back
    (identity resize, crop 512)"""
    import diffute_amd as D
    img, box, x_s, y_s, crop = _case(PRE_CASES[2])
    dev_img = torch.from_numpy(img).to(cuda)
    pre = D.prepost.preprocess(dev_img, box, x_s, y_s, crop)
    out = D.prepost.postprocess(pre["image"], dev_img, box, x_s, y_s, crop)
    assert torch.equal(out, dev_img)
    with pytest.raises(TypeError):
        D.prepost.preprocess(torch.from_numpy(img), box, x_s, y_s, crop)             # host tensor: no CPU fallback
