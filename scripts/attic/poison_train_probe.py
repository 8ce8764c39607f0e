"""Does the training step read memory it never wrote?  Runs the cfg4-shaped step (B = 8 and B = 1, 512 px) on a fresh process, then again after the torch
caching allocator's free blocks were POISONED (filled with NaN bit patterns and released): every torch.empty() the step performs then returns NaN-filled
memory.  A difference (or a NaN) means some kernel consumed uninitialised memory.  Measurement / debugging aid."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
import diffute_amd as D
from diffute_amd.models import mse_loss
from diffute_amd.synthetic import synth_inputs
from oracle import prng
dev = torch.device("cuda")
unet = D.UNet2DConditionModel(device=dev)
lat, mask, mlat, ctx = synth_inputs(8, 64, 64, 577, 1024, device=dev)
x = torch.cat([lat, mask, mlat], 1)
t = torch.tensor([437, 12, 999, 650, 3, 800, 250, 501], device=dev)
tgt = torch.from_numpy(prng.normal(9, 43, 8 * 4 * 64 * 64).reshape(8, 4, 64, 64)).to(dev)

def step(xs, ts, cs, tg, sel=None):
    unet.zero_grad(set_to_none=True)
    pred = unet(xs, ts, cs).sample
    loss = mse_loss(pred if sel is None else pred[sel], tg if sel is None else tg[sel])
    loss.backward()
    torch.cuda.synchronize()
    return float(loss.detach()), pred.detach().clone(), {k: p.grad.clone() for k, p in unet.named_parameters()}

def poison(gb=96, val=float("nan")):
    """ONE block of `gb` GiB (the arenas of the training step are several GiB each: they must be carved out of poisoned memory, not malloc'ed fresh)"""
    torch.cuda.empty_cache()
    blk = torch.empty((gb << 28,), dtype=torch.float32, device=dev)
    blk.fill_(val)
    torch.cuda.synchronize()
    del blk
    return gb

def run(tag):
    l8, p8, g8 = step(x, t, ctx, tgt, sel=slice(0, 1))
    l1, p1, g1 = step(x[:1].contiguous(), t[:1].contiguous(), ctx[:1].contiguous(), tgt[:1].contiguous())
    num = den = 0.0; worst = (0.0, "")
    nonfinite = [k for k in g1 if not (torch.isfinite(g1[k]).all() and torch.isfinite(g8[k]).all())]
    for k in g1:
        a, b = g8[k].float(), g1[k].float()
        num += float((a - b).pow(2).sum()); den += float(b.pow(2).sum())
        r = abs(float(a.norm()) / max(float(b.norm()), 1e-30) - 1)
        if r > worst[0]: worst = (r, k)
    print(f"{tag}: loss B8-sel {l8:.5f} B1 {l1:.5f}; in-batch vs alone rel-L2 {(num / max(den, 1e-30)) ** 0.5:.3e}, worst norm ratio {worst[0]:.3f} ({worst[1]}); non-finite gradients: {nonfinite[:4]}", flush=True)
    return g8, g1

ga8, ga1 = run("fresh process      ")
# drop the model's cached training buffers so that they are re-allocated from poisoned memory
for name in ("_tb",):
    if hasattr(unet, name): delattr(unet, name)
n = poison()
gb8, gb1 = run(f"after NaN poison ({n} GiB)")
same8 = all(torch.equal(ga8[k], gb8[k]) for k in ga8); same1 = all(torch.equal(ga1[k], gb1[k]) for k in ga1)
print("B=8 gradients bit-equal across poison:", same8, "| B=1:", same1)
if not (same8 and same1):
    bad = [k for k in ga1 if not torch.equal(ga1[k], gb1[k])][:6] + [k for k in ga8 if not torch.equal(ga8[k], gb8[k])][:6]
    print("differ:", bad)
for name in ("_tb",):
    if hasattr(unet, name): delattr(unet, name)
n = poison(val=3.0e4)
gc8, gc1 = run(f"after 3e4 poison ({n} GiB)")
print("B=8 bit-equal:", all(torch.equal(ga8[k], gc8[k]) for k in ga8), "| B=1:", all(torch.equal(ga1[k], gc1[k]) for k in ga1))
