#!/bin/bash
mkdir -p gpurun_out
L=gpurun_out/soak_bisect4.jsonl
: > $L
run() { tag=$1; shift; timeout 300 python3 scripts/soak.py --passes 6 --tag "$tag" "$@" >> $L 2>gpurun_out/soak_$tag.err; echo "$tag rc=$?" >> $L; }
run sync_counts --sync --pool-counts
run sync_nohold_counts --sync --no-hold --pool-counts
cat $L; tail -3 gpurun_out/soak_sync_counts.err
