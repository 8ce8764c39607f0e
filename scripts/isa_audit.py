"""ISA audit of the SHIPPED device code (diffute_amd/lib/*.so): what the compiler cannot check because the instruction sits in inline asm.

    python scripts/isa_audit.py [lib.so ...]        (default: both builds)      exit code 1 when a rule is violated

The library's hand-scheduled kernels issue MFMAs, LDS-DMA loads and counted waits from `asm volatile` statements.  hipcc treats such a
statement as one opaque instruction: LLVM's hazard recogniser neither sees the MFMA inside (no software wait states behind it) nor
counts an asm load (no drain before the wave ends).  This tool disassembles every gfx950 code object of the built library
(.hip_fatbin -> clang offload bundles -> llvm-objdump) and checks, instruction by instruction and along both arms of every branch:

  M  MFMA result hazards (CDNA3/4 ISA, "required software wait states"; the numbers LLVM's GCNHazardRecognizer uses for gfx940+):
     after an XDL MFMA of P passes that writes D, an instruction that reads or writes any register of D needs P + 3 issue states in
     between (s_nop N counts N + 1); exempt: an MFMA taking D WHOLE as its C (the accumulate chain, 0 states);
     an MFMA overlapping D only through a partially overlapping C needs P + 1.  Passes: 32x32x16 16-bit / 32x32x8 = 8, 16x16x32 / 16x16x16 = 4,
     4x4 = 2; anything unknown is treated as 16 (conservative).
     The rule is applied to EVERY MFMA of the library, compiler-issued ones included: those must all pass (they do - that calibrates the
     rule), so a violation can only come from an asm statement.
  L  LDS-DMA drain: a kernel that issues `global_load_lds_*` / `buffer_load_* ... lds` must not reach `s_endpgm` with such a request possibly
     in flight - the DMA would land in LDS that belongs to the NEXT block on that CU (it happened once: EXPERIMENTS.md round 3).  Walking
     backwards from every `s_endpgm` over all predecessors, an `s_waitcnt vmcnt(0)` must come before any LDS-DMA instruction.
"""
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
OBJCOPY = "/opt/rocm/lib/llvm/bin/llvm-objcopy"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(so_path, workdir):
    """-> paths of the gfx950 ELF code objects bundled in the library's .hip_fatbin section"""
    fat = os.path.join(workdir, "fat.bin")
    subprocess.run([OBJCOPY, f"--dump-section=.hip_fatbin={fat}", so_path, os.path.join(workdir, "discard.so")], check=True, capture_output=True)
    d = open(fat, "rb").read()
    out = []
    p = d.find(MAGIC)
    k = 0
    while p >= 0:
        q = p + len(MAGIC)
        num, = struct.unpack_from("<Q", d, q); q += 8
        for _ in range(num):
            off, size, ts = struct.unpack_from("<QQQ", d, q); q += 24
            trip = d[q:q + ts].decode(); q += ts
            if "gfx950" in trip and size:
                path = os.path.join(workdir, f"co_{k}.elf")
                open(path, "wb").write(d[p + off:p + off + size])
                out.append(path)
                k += 1
        p = d.find(MAGIC, p + 1)
    return out


REG = re.compile(r"\b([va])\[(\d+):(\d+)\]|\b([va])(\d+)\b")


def regs_of(tok):
    s = set()
    for m in REG.finditer(tok):
        if m.group(1):
            s.update((m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1))
        else:
            s.add((m.group(4), int(m.group(5))))
    return s


def split_ops(text):
    """operands of an instruction, brackets kept together"""
    ops, cur, depth = [], "", 0
    for ch in text:
        if ch == "[":
            depth += 1
        elif ch == "]":
            depth -= 1
        if ch == "," and depth == 0:
            ops.append(cur.strip()); cur = ""
        else:
            cur += ch
    if cur.strip():
        ops.append(cur.strip())
    return ops


class Inst:
    __slots__ = ("addr", "mn", "ops", "raw", "target", "all_regs")

    def __init__(self, addr, mn, ops, raw, target):
        self.addr, self.mn, self.ops, self.raw, self.target = addr, mn, ops, raw, target
        self.all_regs = set()
        for o in ops:
            self.all_regs |= regs_of(o)


FUNC = re.compile(r"^([0-9a-f]+) <(.+)>:$")
LINE = re.compile(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-F]+):")
TGT = re.compile(r"<([^>+]+)(?:\+0x([0-9a-fA-F]+))?>\s*$")


def disassemble(elf):
    """-> {function name: [Inst]}"""
    txt = subprocess.run([OBJDUMP, "-d", "--mcpu=gfx950", elf], check=True, capture_output=True, text=True).stdout
    funcs, cur, start = {}, None, {}
    for l in txt.splitlines():
        m = FUNC.match(l)
        if m:
            cur = m.group(2); funcs[cur] = []; start[cur] = int(m.group(1), 16)
            continue
        if cur is None:
            continue
        m = LINE.match(l)
        if not m:
            continue
        mn, rest, addr = m.group(1), m.group(2), int(m.group(3), 16)
        target = None
        if mn.startswith(("s_cbranch", "s_branch")):
            t = TGT.search(l)
            if t and t.group(1) in start:
                target = start[t.group(1)] + (int(t.group(2), 16) if t.group(2) else 0)
        funcs[cur].append(Inst(addr, mn, split_ops(rest), l.strip(), target))
    return funcs


def mfma_passes(mn):
    m = re.match(r"v_s?mfma[c]?_\w+?_(\d+)x(\d+)x(\d+)", mn)
    if not m:
        return 16
    a, _, k = int(m.group(1)), int(m.group(2)), int(m.group(3))
    if a == 32:
        return 8 if k >= 8 else 16
    if a == 16:
        return 4 if k >= 16 else 8
    if a == 4:
        return 2
    return 16


def states(i):
    if i.mn == "s_nop":
        try:
            return int(i.ops[0], 0) + 1
        except (ValueError, IndexError):
            return 1
    return 1


def audit_mfma(name, insts, report):
    index = {i.addr: k for k, i in enumerate(insts)}
    n_mfma = 0
    for k, mi in enumerate(insts):
        if not mi.mn.startswith(("v_mfma", "v_smfmac")):
            continue
        n_mfma += 1
        D = regs_of(mi.ops[0])
        need = mfma_passes(mi.mn) + 3
        # walk forward along every path until `need` states have gone by
        work, seen = [(k + 1, 0)], set()
        while work:
            j, st = work.pop()
            while j < len(insts) and st < need:
                if (j, st) in seen:
                    break
                seen.add((j, st))
                x = insts[j]
                if x.mn == "s_endpgm":
                    break
                if x.all_regs & D:
                    ok = False
                    if x.mn.startswith(("v_mfma", "v_smfmac")) and len(x.ops) >= 4:
                        xd, xa, xb, xc = (regs_of(o) for o in x.ops[:4])
                        if not ((xa | xb) & D) and (xd == D or not (xd & D)):
                            if xc == D:
                                ok = True                       # D taken whole as C: the accumulate chain (0 states)
                            elif st >= need - 2:
                                ok = True                       # partially overlapping C: passes + 1
                    if not ok:
                        report.append(f"M {name}: {mi.raw.split('//')[0].strip()}  ->  after {st} state(s) (need {need}):  {x.raw.split('//')[0].strip()}  @{x.addr:x}")
                        break
                st += states(x)
                if x.target is not None and x.target in index:
                    if x.mn == "s_branch":
                        j = index[x.target]
                        continue
                    work.append((index[x.target], st))
                j += 1
    return n_mfma


def is_lds_dma(i):
    return i.mn.startswith("global_load_lds") or (i.mn.startswith("buffer_load") and "lds" in i.ops[-1:][0].split() if i.ops else False) or \
        (i.mn.startswith("buffer_load") and " lds" in i.raw.split("//")[0])


def is_vm_drain(i):
    if i.mn != "s_waitcnt":
        return False
    txt = " ".join(i.ops)
    return "vmcnt(0)" in txt


def audit_lds_dma(name, insts, report):
    if not any(is_lds_dma(i) for i in insts):
        return 0
    index = {i.addr: k for k, i in enumerate(insts)}
    preds = {k: [] for k in range(len(insts))}
    for k, i in enumerate(insts):
        if i.mn not in ("s_branch", "s_endpgm") and k + 1 < len(insts):
            preds[k + 1].append(k)
        if i.target is not None and i.target in index:
            preds[index[i.target]].append(k)
    n = 0
    for k, i in enumerate(insts):
        if i.mn != "s_endpgm":
            continue
        n += 1
        work, seen = list(preds[k]), set()
        while work:
            j = work.pop()
            if j in seen:
                continue
            seen.add(j)
            x = insts[j]
            if is_vm_drain(x):
                continue                                        # this path is drained
            if is_lds_dma(x):
                report.append(f"L {name}: s_endpgm @{i.addr:x} reachable from {x.raw.split('//')[0].strip()} @{x.addr:x} without s_waitcnt vmcnt(0)")
                break
            work.extend(preds[j])
    return n


def audit_library(so_path):
    report, stats = [], {"kernels": 0, "mfma": 0, "lds_dma_kernels": 0}
    with tempfile.TemporaryDirectory() as wd:
        for elf in code_objects(so_path, wd):
            for name, insts in disassemble(elf).items():
                if not insts:
                    continue
                stats["kernels"] += 1
                stats["mfma"] += audit_mfma(name, insts, report)
                stats["lds_dma_kernels"] += 1 if audit_lds_dma(name, insts, report) else 0
    return report, stats


def main():
    libs = sys.argv[1:] or [os.path.join(ROOT, "diffute_amd", "lib", n) for n in ("libdiffute_hip.so", "libdiffute_hip_f16.so")]
    bad = 0
    for so in libs:
        report, stats = audit_library(so)
        print(f"{os.path.basename(so)}: {stats['kernels']} kernels, {stats['mfma']} MFMAs checked, {stats['lds_dma_kernels']} kernels with LDS-DMA checked, "
              f"{len(report)} violation(s)")
        for r in report[:40]:
            print("  " + r)
        bad += len(report)
    raise SystemExit(1 if bad else 0)


if __name__ == "__main__":
    main()
