#!/bin/bash
# long soak of the headline loop and its variants (one JSON line each): evidence for the round-5 fix beyond the test suite's pass counts
mkdir -p gpurun_out; L=gpurun_out/soak_long.jsonl; : > $L
run() { tag=$1; shift; timeout 600 python3 scripts/soak.py --tag "$tag" "$@" >> $L 2>gpurun_out/soak_$tag.err; echo "$tag rc=$?" >> $L; }
run hold100 --passes 100
run sync40 --passes 40 --sync
run ddpm30 --passes 30 --ddpm --sync
run fp16_768 --passes 12 --fp16 --batch 2 --latent 96 --sync
run b1_sync30 --passes 30 --batch 1 --sync
run b16_8 --passes 8 --batch 16 --sync
cat $L | cut -c1-330
