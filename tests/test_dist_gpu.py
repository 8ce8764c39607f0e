"""D1 composed (train_diffute_v1.py:780,925): world_size = 2 through `set_gradient_sync` + the HIP backward.
The two ranks were started by conftest.py at session start (tests/d1_world2_worker.py says what each computes)."""
import json
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_gradient_sync_world2_through_the_hip_backward(cuda):
    from conftest import D1_WORLD2
    procs, prefix = D1_WORLD2["procs"], D1_WORLD2["prefix"]
    if not procs:
        pytest.skip("world-2 workers were not started (session not selected with -m gpu)")
    for p in procs:
        p.wait(timeout=600)
    res = []
    for r in range(2):
        path = f"{prefix}.rank{r}.json"
        assert os.path.exists(path), f"rank {r} wrote no result (exit code {procs[r].returncode})"
        with open(path) as f:
            res.append(json.load(f))
    for r in res:
        assert r["ok"], r.get("error")
    g0 = torch.load(f"{prefix}.rank0.pt"); g1 = torch.load(f"{prefix}.rank1.pt")
    assert torch.equal(g0, g1), "the two ranks hold different synchronised gradients"
    for r in res:
        # mean of the two single-rank gradients to fp32 rounding: 1/world = 1/2 is folded into dLoss/dpred before the backward,
        # a power of two commutes with every rounding of the backward, so only the final fp32 sum differs from the fp64 mean
        assert r["rel_err"] < 1e-5, r
        assert r["differs_from_own"] > 1e-2, "the synchronised gradient equals the rank's own: nothing was exchanged"
        assert r["exposed_ms"] is not None and r["exposed_ms"] >= 0.0
        assert abs(r["loss_mean"] - r["loss_expected"]) < 1e-5 * max(1.0, abs(r["loss_expected"]))
    print(f"D1 world 2: rel err vs mean of single-rank gradients {res[0]['rel_err']:.2e} / {res[1]['rel_err']:.2e}, "
          f"exposed exchange {res[0]['exposed_ms']:.3f} / {res[1]['exposed_ms']:.3f} ms, ranks bit-equal")
