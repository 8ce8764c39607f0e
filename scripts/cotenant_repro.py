"""Deliberate reproduction of the one anomalous training gradient of round 5 (EXPERIMENTS.md round 5 item 6; VERDICT r5 "next" item 1): the cfg4 training
step (train_diffute_v1.py:913-925 at 8 x 512 px) in a quiet process, then again while something else holds the GPU:

    quiet        reference: N full steps (must be bit-equal to each other), the sample-0 step, the B = 1 step
    serialize    AMD_SERIALIZE_KERNEL=3: every kernel starts on an idle GPU (what a pathologically slow host looks like to the device)
    occupy32/128 dmx_test_occupy_cus holds 32 / 128 CUs on a side stream of the SAME process during every step
    cotenant-denoise   a second PROCESS loops the headline denoise loop on the same GPU (and classifies its own passes: equal / raised / silently different)
    cotenant-world2    two more processes loop the tiny-UNet training step with the gloo exchange (tests/d1_world2_worker.py's work): the original condition
    cotenant-idle      two more processes that only hold a GPU context
    cotenant-churn     a process that keeps allocating and freeing 2 GB blocks (page-table traffic: a process starting up next to the step)
    cotenant-all       denoise loop + world-2 workers + churner at once

This orchestrator never touches the GPU (it only starts children), so it may start processes at any time.  Every step of every phase is compared with the
quiet run checksum by checksum: the verdict per step is EQUAL, RAISED (DMX_ERR_DEVICE surfaced) or SILENT-DIFF (the bug).

    python scripts/cotenant_repro.py [--out gpurun_out/cotenant] [--steps 10] [--phases quiet,serialize,...]
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = os.path.join(ROOT, "scripts", "cotenant_child.py")


def run_train(out, tag, steps, occupy=0, env=None, timeout=900):
    e = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    e.update(env or {})
    cmd = [sys.executable, CHILD, "train", "--out", out, "--tag", tag, "--steps", str(steps), "--occupy", str(occupy)]
    t0 = time.time()
    try:
        r = subprocess.run(cmd, env=e, timeout=timeout, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        tail = r.stdout[-1500:]
        rc = r.returncode
    except subprocess.TimeoutExpired:
        tail, rc = "TIMEOUT", -9
    if not os.path.exists(out):
        return {"fatal": f"no result file (rc {rc}): {tail}", "wall_s": time.time() - t0}
    with open(out) as f:
        d = json.load(f)
    d["wall_s"] = time.time() - t0
    return d


def start_cotenants(kind, ctl, outdir):
    e = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs, ready = [], []
    if kind == "denoise":
        procs.append(subprocess.Popen([sys.executable, CHILD, "denoise-loop", "--out", os.path.join(outdir, "cotenant_denoise.json"), "--ctl", ctl],
                                      env=e, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL))
        ready = [ctl + ".ready"]
    elif kind == "world2":
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
        for r in range(2):
            procs.append(subprocess.Popen([sys.executable, CHILD, "tiny-train-loop", "--out", os.path.join(outdir, f"cotenant_world2_r{r}.json"), "--ctl", ctl,
                                           "--rank", str(r), "--port", str(port)], env=e, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL))
        ready = [ctl + ".ready0", ctl + ".ready1"]
    elif kind == "all":                                 # the denoise loop AND the two world-2 workers AND a churner at once
        return start_cotenants("denoise", ctl, outdir) + start_cotenants("world2", ctl, outdir) + start_cotenants("churn", ctl, outdir)
    elif kind == "churn":
        procs.append(subprocess.Popen([sys.executable, CHILD, "churn", "--out", os.path.join(outdir, "cotenant_churn.json"), "--ctl", ctl, "--rank", "7"],
                                      env=e, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL))
        ready = [ctl + ".ready7"]
    elif kind == "idle":
        for r in range(2):
            procs.append(subprocess.Popen([sys.executable, CHILD, "idle", "--out", os.path.join(outdir, f"cotenant_idle_r{r}.json"), "--ctl", ctl, "--rank", str(r)],
                                          env=e, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL))
        ready = [ctl + ".ready0", ctl + ".ready1"]
    t0 = time.time()
    while not all(os.path.exists(p) for p in ready):
        if time.time() - t0 > 600 or any(p.poll() is not None for p in procs):
            break
        time.sleep(0.5)
    return procs


def stop_cotenants(procs, ctl):
    open(ctl + ".stop", "w").close()
    for p in procs:
        try:
            p.wait(timeout=120)
        except subprocess.TimeoutExpired:
            p.kill()


def compare(ref, got):
    """per step kind: EQUAL / RAISED / SILENT-DIFF (+ which tensors, and how far off their norms are)"""
    out = []
    rs = {s["kind"]: s for s in ref.get("steps", [])}
    for s in got.get("steps", []):
        k = s["kind"] if s["kind"] in rs else ("full0" if s["kind"].startswith("full") else s["kind"])
        r = rs.get(k)
        if s.get("device_error"):
            out.append({"kind": s["kind"], "verdict": "RAISED", "error": s["device_error"][:200]})
            continue
        if r is None or r.get("device_error"):
            out.append({"kind": s["kind"], "verdict": "NO-REFERENCE"})
            continue
        bad = [n for n in r["grads"] if s["grads"].get(n) != r["grads"][n]]
        pred_same = s["pred"] == r["pred"]
        if not bad and pred_same and s["loss"] == r["loss"]:
            out.append({"kind": s["kind"], "verdict": "EQUAL", "wall_ms": round(s.get("wall_ms", 0), 1)})
        else:
            worst = sorted(((abs((s["grads"][n][1] / max(r["grads"][n][1], 1e-300)) ** 0.5 - 1), n) for n in bad), reverse=True)[:5]
            out.append({"kind": s["kind"], "verdict": "SILENT-DIFF", "pred_equal": pred_same, "loss": [r["loss"], s["loss"]], "n_tensors_differ": len(bad),
                        "of": len(r["grads"]), "worst_norm_ratio_off": [[round(w, 6), n] for w, n in worst]})
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "cotenant"))
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--costeps", type=int, default=3, help="full steps per non-quiet phase")
    ap.add_argument("--phases", default="quiet,serialize,occupy32,occupy128,cotenant-denoise,cotenant-world2,cotenant-idle")
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    phases = args.phases.split(",")
    summary = {"phases": {}}
    quiet = run_train(os.path.join(args.out, "quiet.json"), "quiet", args.steps)
    if "fatal" in quiet:
        print(json.dumps({"fatal": quiet["fatal"]})); sys.exit(2)
    # the quiet run against itself: all full steps bit-equal (the 10-step training soak of the verdict)
    f0 = quiet["steps"][0]
    soak_equal = all(s["grads"] == f0["grads"] and s["pred"] == f0["pred"] and s["loss"] == f0["loss"] for s in quiet["steps"] if s["kind"].startswith("full"))
    summary["phases"]["quiet"] = {"full_steps": args.steps, "all_full_steps_bit_equal": soak_equal, "device_errors": [s["device_error"] for s in quiet["steps"] if s["device_error"]],
                                  "wall_s": round(quiet["wall_s"], 1), "step_ms": [round(s.get("wall_ms", 0), 1) for s in quiet["steps"]]}
    # is the sample-0 step's FORWARD the full step's forward?  (same inputs: the predictions must be bit-equal)
    sel = [s for s in quiet["steps"] if s["kind"] == "sel0"]
    if sel and not sel[0]["device_error"]:
        summary["phases"]["quiet"]["sel0_forward_equals_full_forward"] = sel[0]["pred"] == f0["pred"]
    for ph in phases:
        if ph == "quiet":
            continue
        ctl = os.path.join(args.out, f"ctl_{ph}")
        for suf in (".stop", ".ready", ".ready0", ".ready1", ".ready7"):
            if os.path.exists(ctl + suf):
                os.remove(ctl + suf)
        procs = []
        if ph.startswith("cotenant-"):
            procs = start_cotenants(ph.split("-", 1)[1], ctl, args.out)
        env = {"AMD_SERIALIZE_KERNEL": "3"} if ph == "serialize" else None
        occ = int(ph[len("occupy"):]) if ph.startswith("occupy") else 0
        got = run_train(os.path.join(args.out, f"{ph}.json"), ph, args.costeps, occupy=occ, env=env)
        if procs:
            stop_cotenants(procs, ctl)
        rec = {"wall_s": round(got.get("wall_s", 0), 1)}
        if "fatal" in got:
            rec["fatal"] = got["fatal"][-600:]
        else:
            rec["steps"] = compare(quiet, got)
        for fn in os.listdir(args.out):
            if fn.startswith("cotenant_") and fn.endswith(".json") and (ph.split("-", 1)[-1] in fn or ph == "cotenant-all"):
                with open(os.path.join(args.out, fn)) as f:
                    rec.setdefault("cotenants", []).append(json.load(f))
        summary["phases"][ph] = rec
    verdicts = [s["verdict"] for p in summary["phases"].values() for s in p.get("steps", [])]
    summary["silent_diffs"] = verdicts.count("SILENT-DIFF")
    summary["raised"] = verdicts.count("RAISED")
    summary["equal"] = verdicts.count("EQUAL")
    summary["cotenant_silent_diffs"] = sum(c.get("silently_different", 0) for p in summary["phases"].values() for c in p.get("cotenants", []))
    with open(os.path.join(args.out, "summary.json"), "w") as f:
        json.dump(summary, f, indent=1)
    print(json.dumps(summary, indent=1)[:6000])
    sys.exit(1 if summary["silent_diffs"] or summary["cotenant_silent_diffs"] or not soak_equal else 0)


if __name__ == "__main__":
    main()
