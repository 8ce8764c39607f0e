# PMC counters of the halo conv kernel on one shape (separate --pmc passes, never combined with trace domains).  bash scripts/pmc_halo.sh "4 64 64 320 0 320 0"
R=$PWD; SH=${1:-"4 64 64 320 0 320 0"}; KN=${2:-dmx_conv_halo}
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1)); rm -rf /tmp/pm$i
  timeout 200 rocprofv3 --pmc $C --output-format csv -d /tmp/pm$i -o p -- python3 $R/scripts/halo_one.py $SH 6 > /tmp/pm$i.log 2>&1
  python3 - "$KN" /tmp/pm$i <<'PY'
import csv, glob, sys, collections
kn, d = sys.argv[1], sys.argv[2]
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
if not f: print("no counters in", d); sys.exit()
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    if kn in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items(): print(f"  {k:28s} per launch: {sum(v[1:]) / max(len(v) - 1, 1):.4g}   ({len(v)} launches)")
PY
done
