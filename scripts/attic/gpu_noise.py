"""Background GPU load for race hunting: tiny-UNet training steps in a loop for N seconds (a second process on the same GPU, like the world-2 workers of
tests/conftest.py).  python scripts/gpu_noise.py [seconds]"""
import sys, time
import torch
sys.path.insert(0, ".")
import diffute_amd as D
from diffute_amd.models import mse_loss
from diffute_amd.synthetic import synth_inputs
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
m = D.UNet2DConditionModel(block_out_channels=(64, 128, 256, 256), attention_head_dim=(1, 2, 4, 4), cross_attention_dim=128).cuda()
lat, mask, mlat, ctx = synth_inputs(2, 16, 16, 20, 128, device=dev)
x = torch.cat([lat, mask, mlat], 1); t = torch.tensor([321, 5], device=dev); tgt = torch.zeros(2, 4, 16, 16, device=dev)
t0 = time.time(); n = 0
while time.time() - t0 < float(sys.argv[1] if len(sys.argv) > 1 else 60):
    m.zero_grad(set_to_none=True)
    mse_loss(m(x, t, ctx).sample, tgt).backward()
    n += 1
    if n % 50 == 0: torch.cuda.synchronize()
torch.cuda.synchronize()
print("noise steps", n)
