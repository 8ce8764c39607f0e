#!/bin/bash
# The bisection of the non-finite latents of BENCH_r04 (EXPERIMENTS.md round 5 item 0): one configuration of scripts/soak.py per process, one JSON
# line each.  The last three need a probe build (make -C diffute_amd/csrc EXTRA=-DDMX_PROBES): they switch the pools back to memset nodes.
mkdir -p gpurun_out
L=gpurun_out/soak_bisect.jsonl
: > $L
run() { tag=$1; shift; timeout 300 python3 scripts/soak.py --passes 8 --tag "$tag" "$@" >> $L 2>gpurun_out/soak_$tag.err; echo "$tag rc=$?" >> $L; }
run hold
run nohold --no-hold
run sync --sync
run sync_nohold --sync --no-hold
run sync_nograph --sync --no-graph
run sync_prefetch0 --sync --prefetch 0
run sync_halows0 --sync --halo-ws 0
run sync_halo0 --sync --halo 0
run sync_xf0 --sync --xf-chain 0
run sync_gnstats0 --sync --gn-stats 0
run sync_steps --sync --step-report
run probe_memset_nodes --sync --memset-nodes
run probe_pool_counts --sync --pool-counts
run probe_pool_counts_nohold --sync --no-hold --pool-counts
cut -c1-400 $L
