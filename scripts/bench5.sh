#!/bin/bash
# VERDICT r4 item 1 "Done": the driver's bench command five times in a row on one lease (secondary / CPU legs skipped after the first: the failure of
# round 4 sat in the 25 timed passes); one line per invocation: rc, value, ms per pass, passes checked
mkdir -p gpurun_out
: > gpurun_out/bench5.txt
for i in 1 2 3 4 5; do
  extra="--no-secondary --no-cpu-baseline"; [ $i = 1 ] && extra=""
  python3 bench.py --gpus 1 --steps 20 --warmup 5 $extra > gpurun_out/bench5_$i.json 2> gpurun_out/bench5_$i.err; rc=$?
  python3 - <<PY >> gpurun_out/bench5.txt
import json
try:
    d = json.loads(open("gpurun_out/bench5_$i.json").read().strip().splitlines()[-1])
    print("run $i rc=$rc value", d.get("value"), "ms_per_step", d.get("ms_per_step"), "passes", d.get("passes_checked"), "error", d.get("error"))
except Exception as e:
    print("run $i rc=$rc no JSON line:", e)
PY
done
cat gpurun_out/bench5.txt
