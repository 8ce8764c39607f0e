"""Autoencoder training step timing (train_vae.py:716-736 on the product classes): forward that keeps activations + HIP backward."""
import sys, time, torch
sys.path.insert(0, ".")
import diffute_amd as D
from diffute_amd.models import mse_loss
dev = torch.device("cuda")
vae = D.AutoencoderKL(device=dev)
for B, px in ((1, 256), (4, 512)):
    x = torch.rand(B, 3, px, px, device=dev) * 2 - 1
    for it in range(3):
        if it == 1:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        vae.zero_grad(set_to_none=True)
        loss = mse_loss(vae(x)["sample"], x)
        loss.backward()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 2
    gn = float(torch.sqrt(sum(p.grad.float().pow(2).sum() for p in vae.parameters())))
    print(f"VAE train fwd+bwd B={B} {px}px: {dt*1e3:.1f} ms, loss {float(loss.detach()):.5f}, |grad| {gn:.4f}, finite {all(torch.isfinite(p.grad).all() for p in vae.parameters())}, mem {torch.cuda.max_memory_allocated()/2**30:.1f} GB")
