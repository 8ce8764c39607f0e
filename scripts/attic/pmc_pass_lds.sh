# LDS bank-conflict / wait counters per kernel over a short denoise pass (separate --pmc run, no trace domains)
R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pml
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --output-format csv -d /tmp/pml -o p -- python3 $R/bench.py --steps 1 --warmup 0 --denoise-steps 2 --no-cpu-baseline --no-profile --no-secondary > /dev/null 2>&1
cd $R
python3 - <<'PY'
import csv, glob, collections
fs = glob.glob("/tmp/pml/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set)
for r in csv.DictReader(open(fs[0])):
    k = r["Kernel_Name"]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k].add(r["Dispatch_Id"])
rows = []
for k in acc:
    a = acc[k]
    rows.append((a["SQ_WAVE_CYCLES"], k, len(cnt[k]), a))
print("kernel | launches | wave cycles (total) | LDS conflict / LDS active | conflict / wave cycles | wait LDS / wave cycles | wait any / wave cycles")
for wc, k, n, a in sorted(rows, reverse=True)[:22]:
    la = a["SQ_ACTIVE_INST_LDS"] or 1
    print(f"{k[:70]:70s} {n:5d} {wc:10.3g} {a['SQ_LDS_BANK_CONFLICT'] / la:6.2f} {a['SQ_LDS_BANK_CONFLICT'] / wc:7.3f} {a['SQ_WAIT_INST_LDS'] / wc:7.3f} {a['SQ_WAIT_INST_ANY'] / wc:7.3f}")
PY
