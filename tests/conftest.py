import os
import subprocess
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

D1_WORLD2 = {"procs": None, "prefix": None}
COTENANT = {"proc": None, "ctl": None, "out": None}      # tests/test_cotenant_gpu.py: a second process that loops the denoise loop on the same GPU


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """The world-2 gradient-exchange check (tests/test_dist_gpu.py) needs two more processes on the GPU.  They are started
    HERE, before anything in this process has initialised HIP: a process that has touched the GPU must not fork + exec on
    the GPU pool.  (`device_count()` does not initialise the device; `is_available()` would.)"""
    expr = session.config.getoption("markexpr", "") or ""
    if "gpu" not in expr or "not gpu" in expr:
        return
    import torch
    if torch.cuda.device_count() < 1:
        return
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
    prefix = os.path.join(tempfile.mkdtemp(prefix="d1_world2_"), "res")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    worker = os.path.join(ROOT, "tests", "d1_world2_worker.py")
    D1_WORLD2["procs"] = [subprocess.Popen([sys.executable, worker, str(r), "2", str(port), prefix], env=env,
                                           stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL) for r in range(2)]
    D1_WORLD2["prefix"] = prefix
    # ... and they are given the GPU to themselves: the library's plans assume that nothing else holds CUs while its launches run
    # (include/diffute_hip.h dmx_set_exclusive_device), and every other GPU test checks results produced under that assumption.  On a slow
    # host the workers used to be still running - two more processes on the same GPU - when the first model tests started (round 5 saw one
    # lease with an 8 x slower host produce a different training gradient in test_cfg4; EXPERIMENTS.md round 5 item 6).
    for p in D1_WORLD2["procs"]:
        try:
            p.wait(timeout=900)
        except subprocess.TimeoutExpired:
            pass                                   # test_dist_gpu.py reports it
    # the deliberate co-tenant of tests/test_cotenant_gpu.py: started here for the same reason (no fork + exec once HIP is initialised), but it does
    # not touch the GPU - it does not even import torch - until that test writes <ctl>.go, and it leaves when the test writes <ctl>.stop
    d = tempfile.mkdtemp(prefix="cotenant_")
    COTENANT["ctl"], COTENANT["out"] = os.path.join(d, "co"), os.path.join(d, "co.json")
    COTENANT["proc"] = subprocess.Popen([sys.executable, os.path.join(ROOT, "scripts", "cotenant_child.py"), "denoise-loop", "--wait-go",
                                         "--ctl", COTENANT["ctl"], "--out", COTENANT["out"], "--dsteps", "10"], env=env,
                                        stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)


def pytest_sessionfinish(session, exitstatus):
    for p in D1_WORLD2["procs"] or []:
        if p.poll() is None:
            p.kill()
    p = COTENANT["proc"]
    if p is not None and p.poll() is None:
        open(COTENANT["ctl"] + ".stop", "w").close()
        try:
            p.wait(timeout=30)
        except subprocess.TimeoutExpired:
            p.kill()


@pytest.fixture(scope="session")
def cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    return torch.device("cuda:0")
