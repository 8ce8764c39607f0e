R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pm1 /tmp/pm2 /tmp/pm3
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES --output-format csv -d /tmp/pm1 -o p -- python3 $R/scripts/pmc_two_kernels.py > /dev/null 2>&1
timeout 200 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d /tmp/pm2 -o p -- python3 $R/scripts/pmc_two_kernels.py > /dev/null 2>&1
timeout 200 rocprofv3 --pmc SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_MFMA --output-format csv -d /tmp/pm3 -o p -- python3 $R/scripts/pmc_two_kernels.py > /dev/null 2>&1
cd $R
python3 - <<'PY'
import csv, glob, collections
for d in ("/tmp/pm1","/tmp/pm2","/tmp/pm3"):
    fs = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not fs: print(d, "no output"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set)
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"]
        if "xf_chain" not in k and "attn" not in k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k].add(r["Dispatch_Id"])
    for k in sorted(acc):
        n = len(cnt[k])
        print(k[:48], " ".join(f"{c}={v / n:.3g}" for c, v in sorted(acc[k].items())))
PY
