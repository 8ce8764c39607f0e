# usage: bash scripts/pmc_kernel.sh <kernel-name-substring> <counters...> -- <python script + args>
# Runs rocprofv3 --pmc for the counters and prints their per-launch mean for kernels matching the substring.
# (counters only: never combine --pmc with the trace domains on this pool)
PAT=$1; shift
CTRS=()
while [ "$1" != "--" ]; do CTRS+=("$1"); shift; done
shift
R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_k
rocprofv3 --pmc "${CTRS[@]}" --output-format csv -d /tmp/pmc_k -o k -- python3 $R/"$@" > /tmp/pmc_k.log 2>&1
cd $R
python3 - "$PAT" <<'PY'
import csv, glob, collections, sys
pat = sys.argv[1]
fs = glob.glob("/tmp/pmc_k/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.Counter())
for f in fs:
    for r in csv.DictReader(open(f)):
        if pat not in r["Kernel_Name"]:
            continue
        k = r["Kernel_Name"][:60]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
for k, v in agg.items():
    print(k)
    for c, x in v.items():
        print(f"    {c:32s} {x / cnt[k][c]:16.0f}   (x{cnt[k][c]})")
PY
