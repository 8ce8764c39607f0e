"""Per-launch hipEvent profile of AutoencoderKL encode + decode (cfg3: 32 x 512 px) - shapes ranked by time.  Measurement aid."""
import collections
import csv
import ctypes
import os
import sys
import tempfile

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diffute_amd as D  # noqa: E402
from diffute_amd import _cabi  # noqa: E402
from diffute_amd.synthetic import text_crop_images  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda")
lib = _cabi.lib()
vae = D.AutoencoderKL(device=dev).requires_grad_(False)
img = text_crop_images(B, 512, 512, device=dev)
z = torch.randn(B, 4, 64, 64, device=dev)
with torch.no_grad():
    vae.encode(img); vae.decode(z); torch.cuda.synchronize()
    for what, fn in (("encode", lambda: vae.encode(img)), ("decode", lambda: vae.decode(z))):
        path = os.path.join(tempfile.gettempdir(), "vae_launches.csv")
        lib.dmx_profile_dump_path(path.encode()); lib.dmx_profile_begin()
        fn()
        buf = (ctypes.c_double * 100)()
        _cabi.check(lib.dmx_profile_end(buf, 100), "profile_end")
        agg = collections.OrderedDict()
        for r in csv.DictReader(open(path)):
            a = agg.setdefault((r["class"], r["tag"]), [0, 0.0, 0.0]); a[0] += 1; a[1] += float(r["ms"]); a[2] += float(r["flops"])
        tot = sum(a[1] for a in agg.values())
        print(f"--- {what}: {tot:.2f} ms over {sum(a[0] for a in agg.values())} profiled launches")
        for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
            print(f"  cls {k[0]:>2s} {k[1]:58s} x{a[0]:3d} {a[1]:7.2f} ms  {a[2] / a[1] / 1e9 if a[1] else 0:7.1f} TF/s")
