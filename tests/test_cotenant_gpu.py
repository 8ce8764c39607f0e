"""The training step must never hand back a silently different gradient because something else is on the GPU (the one anomalous
lease of round 5, EXPERIMENTS.md): with CUs held by another stream of this process, and with a second PROCESS running the headline denoise loop on the
same GPU, every step of the cfg4 shape (train_diffute_v1.py:913-925 at 8 x 512 px) is either BIT-EQUAL to the quiet run or raises DMX_ERR_DEVICE.
Plus the quiet 10-step soak: ten identical B = 8 steps give ten bit-identical gradients.  scripts/cotenant_repro.py is the long form (seven regimes,
separate processes, per-tensor checksums); its logs of round 6 are under profiles/."""
import json
import os
import time

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_train_step_under_cotenant(cuda):
    from conftest import COTENANT
    import diffute_amd as D
    from diffute_amd import _cabi
    from diffute_amd.models import mse_loss
    from diffute_amd.synthetic import synth_inputs
    from oracle import prng
    lib = _cabi.lib()
    unet = D.UNet2DConditionModel(device=cuda)
    params = [p for _, p in unet.named_parameters()]
    lat, mask, mlat, ctx = synth_inputs(8, 64, 64, 577, 1024, device=cuda)
    x = torch.cat([lat, mask, mlat], 1)
    t = torch.tensor([437, 12, 999, 650, 3, 800, 250, 501], device=cuda)
    tgt = torch.from_numpy(prng.normal(9, 43, 8 * 4 * 64 * 64).reshape(8, 4, 64, 64)).to(cuda)
    kinds = {"full": (x, t, ctx, tgt, None), "sel0": (x, t, ctx, tgt, slice(0, 1)),
             "b1": (x[:1].contiguous(), t[:1].contiguous(), ctx[:1].contiguous(), tgt[:1].contiguous(), None)}

    def step(kind):
        """-> (loss, pred, flat gradient); RuntimeError when a kernel gave up on an in-kernel wait (the only acceptable alternative to equal bits)"""
        xs, ts, cs, tg, sel = kinds[kind]
        unet.zero_grad(set_to_none=True)
        pred = unet(xs, ts, cs).sample
        loss = mse_loss(pred if sel is None else pred[sel], tg if sel is None else tg[sel])
        loss.backward()
        D.synchronize()                                 # host sync + device-error poll: the public sync point
        return float(loss.detach()), pred.detach().clone(), torch.cat([p.grad.reshape(-1) for p in params])

    # ---- quiet reference + the 10-step soak
    ref = {k: step(k) for k in kinds}
    assert all(np.isfinite(r[0]) and bool(torch.isfinite(r[2]).all()) for r in ref.values())
    assert torch.equal(ref["sel0"][1], ref["full"][1]), "the sample-0 step ran the same forward as the full step: same prediction bits"
    for i in range(9):
        l, p, g = step("full")
        assert l == ref["full"][0] and torch.equal(p, ref["full"][1]) and torch.equal(g, ref["full"][2]), f"quiet step {i + 1} of the soak differs from step 0"
        del p, g

    tally = {"equal": 0, "raised": 0}

    def checked(kind, where):
        try:
            l, p, g = step(kind)
        except RuntimeError as e:
            assert "device error" in str(e), f"{where}: {e}"
            torch.cuda.synchronize()
            try:                                        # (a second record of the same starved launch wave, if any)
                _cabi.poll_device_error()
            except RuntimeError:
                pass
            tally["raised"] += 1
            return
        r = ref[kind]
        same = l == r[0] and torch.equal(p, r[1]) and torch.equal(g, r[2])
        if not same:
            off = float((g - r[2]).norm() / r[2].norm())
            pytest.fail(f"{where}: the {kind} step returned a DIFFERENT gradient (rel-L2 {off:.3e}, prediction equal: {torch.equal(p, r[1])}) and raised nothing")
        tally["equal"] += 1

    # ---- (b) CUs held by another stream of this process while the step runs
    # (HIP multiplexes streams onto a few hardware queues; a side stream that shares the training stream's queue would only SERIALISE with the step.  The hog
    # is therefore dealt over three fresh streams: at least two of them run beside the step whatever the mapping is.)
    sides = [torch.cuda.Stream(device=cuda) for _ in range(3)]
    n_cu = torch.cuda.get_device_properties(cuda).multi_processor_count
    for hog in (32, 128, n_cu - 1):
        for kind in ("full", "b1"):
            for si, side in enumerate(sides):
                with torch.cuda.stream(side):
                    _cabi.check(lib.dmx_test_occupy_cus((hog + 2 - si) // 3, 15_000_000, _cabi.current_stream()), "occupy")     # 150 ms
            time.sleep(0.01)
            checked(kind, f"{hog} CUs held by a side stream")
    torch.cuda.synchronize()

    # ---- (a) a second process looping the denoise loop on the same GPU
    proc = COTENANT["proc"]
    if proc is None:
        pytest.skip("the co-tenant process was not started (session not selected with -m gpu)")
    open(COTENANT["ctl"] + ".go", "w").close()
    t0 = time.time()
    while not os.path.exists(COTENANT["ctl"] + ".ready"):
        assert proc.poll() is None, f"the co-tenant process died (exit code {proc.returncode}): {open(COTENANT['out']).read()[:800] if os.path.exists(COTENANT['out']) else ''}"
        assert time.time() - t0 < 600, "the co-tenant process did not come up"
        time.sleep(0.25)
    for i in range(8):
        checked(("full", "sel0", "b1", "full")[i % 4], "a second process loops the denoise loop on this GPU")
    open(COTENANT["ctl"] + ".stop", "w").close()
    proc.wait(timeout=120)
    with open(COTENANT["out"]) as f:
        co = json.load(f)
    assert "fatal" not in co, co
    assert co["passes"] >= 2, f"the co-tenant hardly ran ({co}): no co-tenancy was exercised"
    assert co["silently_different"] == 0, f"the CO-TENANT's own denoise passes changed silently next to the training step: {co}"
    print(f"co-tenancy: {tally['equal']} steps bit-equal to the quiet run, {tally['raised']} raised DMX_ERR_DEVICE; co-tenant: {co['passes']} passes, "
          f"{co['equal']} equal, {co['raised']} raised, 0 silently different")
