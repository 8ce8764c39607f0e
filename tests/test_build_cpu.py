"""Build-time guard for conv_halo.hip: its MFMAs are inline asm (accumulators tied in place), so the compiler does not know their
latency.  A spill of an accumulator right behind the MFMA that wrote it would store stale registers (it happened once: EXPERIMENTS.md
round 4, item 1b).  The device ISA of every dmx_conv_halo_kernel instance must not contain a scratch store of registers an MFMA wrote
within the preceding instructions."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


def _regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
@pytest.mark.parametrize("build", ["bf16", "fp16"])
def test_no_accumulator_spill_behind_an_asm_mfma(tmp_path, build):
    """both builds of the library: the register allocation differs between them (the fp16 build of the 160-column warp-specialised
    instance spilled an accumulator inside the tap where the bf16 build did not: non-deterministic results at 768 px)"""
    src = os.path.join(ROOT, "diffute_amd", "csrc", "conv_halo.hip")
    out = tmp_path / "conv_halo.s"
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off"] + (["-DDMX_F16"] if build == "fp16" else []) +
                   ["-S", "--cuda-device-only", src, "-o", str(out)], check=True, capture_output=True, cwd=os.path.dirname(src))
    lines = [l.strip() for l in open(out)]
    WINDOW = 12                                        # instructions; an MFMA's result is due well within that many issue slots
    recent = []                                        # (instruction index, registers written by an MFMA)
    n_inst = n_mfma = 0
    bad = []
    in_halo = False
    for l in lines:
        if l.startswith("_Z") and "dmx_conv_halo_kernel" in l and ":" in l:
            in_halo = True
        elif l.startswith(".Lfunc_end"):
            in_halo = False
        if not in_halo or not l or l.startswith((";", ".", "//", "_Z")):
            continue
        n_inst += 1
        if l.startswith("v_mfma"):
            n_mfma += 1
            recent.append((n_inst, _regs(l.split()[1].rstrip(","))))
        elif l.startswith("scratch_store"):
            ops = l.split(",")
            regs = _regs(ops[1].strip()) if len(ops) > 1 else set()
            for (i, wr) in recent[-16:]:
                if n_inst - i <= WINDOW and regs & wr:
                    bad.append(l)
    assert n_mfma > 1000, "the halo kernels were not found in the ISA"
    assert not bad, f"accumulator registers spilled right behind the MFMA that writes them: {bad[:3]}"
