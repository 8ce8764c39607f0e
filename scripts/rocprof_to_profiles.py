"""rocprofv3 output directories -> the summaries committed under profiles/.

    python scripts/rocprof_to_profiles.py <stats_dir> <fetch_dir> <write_dir> <out_prefix> [<mfma_dir>]     e.g. profiles/r01

  <stats_dir>  rocprofv3 --kernel-trace --stats            -> <out_prefix>_kernel_stats.csv (copied as is)
  <fetch_dir>  rocprofv3 --pmc FETCH_SIZE  (own pass)      \\
  <write_dir>  rocprofv3 --pmc WRITE_SIZE  (own pass)      /-> <out_prefix>_pmc_traffic.csv: per kernel, per launch
  <mfma_dir>   rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE (own pass) -> <out_prefix>_pmc_mfma.csv: per kernel,
               MFMA-busy cycles summed over the chip's 1024 SIMDs / (1024 x GPU-active cycles) = matrix-pipe utilisation
               (GRBM_GUI_ACTIVE is reported as the sum of 16 instances: divided by 16)
  l2 <tcc_dir> <sq_dir> <out_prefix>   (second form: `rocprof_to_profiles.py l2 ...`)
               rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum (own pass) and
               rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT (own pass)
               -> <out_prefix>_pmc_l2.csv: per kernel and launch, the L2-side picture (requests, hit rate, request bytes at 128 B) and where the waves' cycles go
               (parked on s_waitcnt / s_barrier, issue-stalled, issuing; the LDS share) - what bounds a kernel that is at neither the HBM nor the MFMA roof
Counter units and the gfx950 correction follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE and
WRITE_SIZE are in KiB; FETCH_SIZE reports half of the bytes of wide coalesced reads -> doubled; WRITE_SIZE is taken as is."""
import collections
import csv
import glob
import os
import shutil
import sys


def per_kernel(dirname, counter):
    f = glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True)[0]
    tot = collections.defaultdict(float); n = collections.defaultdict(int)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"]
        tot[name] += float(r["Counter_Value"]); n[name] += 1
    return tot, n


def main(stats_dir, fetch_dir, write_dir, out):
    src = glob.glob(os.path.join(stats_dir, "**", "*kernel_stats.csv"), recursive=True)[0]
    shutil.copy(src, out + "_kernel_stats.csv")
    ft, fn = per_kernel(fetch_dir, "FETCH_SIZE")
    wt, wn = per_kernel(write_dir, "WRITE_SIZE")
    rows = []
    for k in ft:
        if "dmx_" not in k:
            continue
        f_kb = ft[k] / fn[k]
        w_kb = wt.get(k, 0.0) / max(wn.get(k, 0), 1)
        rows.append((k, fn[k], f_kb, 2 * f_kb * 1024, w_kb, 2 * f_kb * 1024 + w_kb * 1024))
    rows.sort(key=lambda r: -r[1] * r[5])
    with open(out + "_pmc_traffic.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "launches", "FETCH_SIZE_KB_per_launch_raw", "FETCH_bytes_per_launch_corrected_x2", "WRITE_SIZE_KB_per_launch", "traffic_bytes_per_launch"])
        for r in rows:
            w.writerow([r[0], r[1], f"{r[2]:.1f}", f"{r[3]:.0f}", f"{r[4]:.1f}", f"{r[5]:.0f}"])
    print(f"{len(rows)} kernels -> {out}_pmc_traffic.csv, {out}_kernel_stats.csv")


# rocprofv3 sums GRBM_GUI_ACTIVE over 16 counter instances on this part (cross-checked: value / 16 = kernel duration x ~2 GHz)
GRBM_INSTANCES = 16.0


def mfma(mfma_dir, out):
    bt, bn = per_kernel(mfma_dir, "SQ_VALU_MFMA_BUSY_CYCLES")
    at, an = per_kernel(mfma_dir, "GRBM_GUI_ACTIVE")
    rows = []
    for k in bt:
        if "dmx_" not in k or at.get(k, 0) <= 0:
            continue
        rows.append((k, bn[k], bt[k] / bn[k], at[k] / an[k], bt[k] / (1024.0 * at[k] / GRBM_INSTANCES)))
    rows.sort(key=lambda r: -r[1] * r[3])
    with open(out + "_pmc_mfma.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "launches", "SQ_VALU_MFMA_BUSY_CYCLES_per_launch", "GRBM_GUI_ACTIVE_per_launch_sum_of_16_instances", "mfma_busy_fraction_of_1024_SIMDs"])
        for r in rows:
            w.writerow([r[0], r[1], f"{r[2]:.0f}", f"{r[3]:.0f}", f"{r[4]:.4f}"])
    print(f"{len(rows)} kernels -> {out}_pmc_mfma.csv")


def l2(tcc_dir, sq_dir, out):
    """L2-side and wave-state counters per kernel.  TCC requests are 128-byte lines (guide, L2 section); SQ_* are summed over all waves
    (quad-cycles: only their RATIOS to SQ_WAVE_CYCLES are used)."""
    names_t = ["TCC_REQ_sum", "TCC_HIT_sum", "TCC_MISS_sum", "TCP_TCC_READ_REQ_sum"]
    names_s = ["SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_LDS", "SQ_INSTS_LDS", "SQ_WAIT_INST_LDS", "SQ_LDS_BANK_CONFLICT"]
    T = {c: per_kernel(tcc_dir, c) for c in names_t}
    S = {c: per_kernel(sq_dir, c) for c in names_s}
    kernels = [k for k in T["TCC_REQ_sum"][0] if "dmx_" in k]
    rows = []
    for k in kernels:
        n = T["TCC_REQ_sum"][1][k]
        t = {c: (T[c][0].get(k, 0.0) / max(T[c][1].get(k, 0), 1)) for c in names_t}
        q = {c: (S[c][0].get(k, 0.0) / max(S[c][1].get(k, 0), 1)) for c in names_s}
        wc = max(q["SQ_WAVE_CYCLES"], 1.0)
        hm = t["TCC_HIT_sum"] + t["TCC_MISS_sum"]
        rows.append([k, n, f"{t['TCC_REQ_sum']:.0f}", f"{t['TCC_HIT_sum']:.0f}", f"{t['TCC_MISS_sum']:.0f}", f"{(t['TCC_HIT_sum'] / hm if hm else 0):.4f}",
                     f"{t['TCC_REQ_sum'] * 128:.0f}", f"{t['TCP_TCC_READ_REQ_sum']:.0f}", f"{q['SQ_WAVE_CYCLES']:.0f}",
                     f"{q['SQ_WAIT_ANY'] / wc:.4f}", f"{q['SQ_WAIT_INST_ANY'] / wc:.4f}", f"{q['SQ_ACTIVE_INST_ANY'] / wc:.4f}", f"{q['SQ_ACTIVE_INST_LDS'] / wc:.4f}",
                     f"{q['SQ_WAIT_INST_LDS'] / wc:.4f}", f"{q['SQ_INSTS_LDS']:.0f}", f"{(q['SQ_LDS_BANK_CONFLICT'] / max(q['SQ_ACTIVE_INST_LDS'], 1.0)):.4f}"])
    rows.sort(key=lambda r: -r[1] * float(r[8]))
    with open(out + "_pmc_l2.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "launches", "TCC_REQ_per_launch", "TCC_HIT_per_launch", "TCC_MISS_per_launch", "l2_hit_rate", "l2_request_bytes_per_launch_at_128B", "TCP_TCC_READ_REQ_per_launch",
                    "SQ_WAVE_CYCLES_per_launch", "wave_parked_frac_SQ_WAIT_ANY", "issue_stall_frac_SQ_WAIT_INST_ANY", "issuing_frac_SQ_ACTIVE_INST_ANY", "lds_issuing_frac_SQ_ACTIVE_INST_LDS",
                    "lds_issue_stall_frac_SQ_WAIT_INST_LDS", "SQ_INSTS_LDS_per_launch", "lds_bank_conflict_cycles_per_lds_active_cycle"])
        w.writerows(rows)
    print(f"{len(rows)} kernels -> {out}_pmc_l2.csv")


if __name__ == "__main__":
    if sys.argv[1] == "l2":
        l2(*sys.argv[2:5])
        sys.exit(0)
    main(*sys.argv[1:5])
    if len(sys.argv) > 5:
        mfma(sys.argv[5], sys.argv[4])
