"""One rank of the world-2 composition check of D1 (train_diffute_v1.py:780,925): the in-backward, bucketed
gradient exchange (UNet2DConditionModel.set_gradient_sync) running through the REAL HIP backward with world_size = 2.

Both ranks share cuda:0 (the box has one GPU), so the group is gloo (RCCL refuses two ranks on one device) and the exchange
schedule is mode="all_reduce" (gloo has no CUDA reduce-scatter).  Started by tests/conftest.py at session start - before the
pytest process has touched the GPU - and judged by tests/test_dist_gpu.py from the JSON / tensor files written here.

Each rank: the single-rank gradients of BOTH ranks' inputs (no exchange), then one synchronised backward on its own input.
Expected: synchronised gradient == mean of the two single-rank gradients (to fp32 rounding: 1/world is folded into dLoss/dpred
before the backward instead of applied after the sum), bit-identical on both ranks, exposed_exchange_ms() populated.
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
TINY_UNET = dict(block_out_channels=(64, 128, 256, 256), attention_head_dim=(1, 2, 4, 4), cross_attention_dim=128)


def main():
    rank, world, port, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    res = {"rank": rank, "ok": False}
    try:
        import torch
        import torch.distributed as dist
        import diffute_amd as D
        from diffute_amd import dist as DD
        from diffute_amd.models import mse_loss
        from diffute_amd.synthetic import synth_inputs
        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        model = D.UNet2DConditionModel(**TINY_UNET).cuda()              # seeded init: the same weights on every rank
        DD.broadcast_parameters(model.parameters(), dist, src=0, module=model)     # D3 anyway (train_diffute_v1.py:780)

        def inputs(r):
            lat, mask, mlat, ctx = synth_inputs(1, 8, 8, 20, 128, seed=10 * r, device=dev)
            return torch.cat([lat, mask, mlat], 1), torch.tensor([321 + 111 * r], device=dev), ctx, torch.full((1, 4, 8, 8), 1.0 - 0.5 * r, device=dev)

        def grads(r):
            x, t, ctx, target = inputs(r)
            model.zero_grad(set_to_none=True)
            loss = mse_loss(model(x, t, ctx).sample, target)
            loss.backward()
            torch.cuda.synchronize()
            return float(loss.detach()), torch.cat([p.grad.reshape(-1).double() for p in model.parameters()])

        l0, g0 = grads(0)
        l1, g1 = grads(1)
        want = (g0 + g1) / 2
        model.set_gradient_sync(dist, mode="all_reduce")
        lr, gs = grads(rank)
        res["exposed_ms"] = model.exposed_exchange_ms()
        model.set_gradient_sync(None)
        res["loss_mean"] = DD.gather_scalar(lr, dist, world)       # D2
        res["loss_expected"] = (l0 + l1) / 2
        res["rel_err"] = float((gs - want).norm() / want.norm())
        res["max_abs_err"] = float((gs - want).abs().max())
        res["differs_from_own"] = float((gs - (g0 if rank == 0 else g1)).norm() / want.norm())
        res["grad_norm"] = float(want.norm())
        torch.save(gs.float().cpu(), out + f".rank{rank}.pt")
        res["ok"] = True
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:                                                # noqa: BLE001 - reported to the judging test
        import traceback
        res["error"] = f"{type(e).__name__}: {e}\n{traceback.format_exc()[-1500:]}"
    with open(out + f".rank{rank}.json", "w") as f:
        json.dump(res, f)


if __name__ == "__main__":
    main()
