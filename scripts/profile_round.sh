# Regenerates profiles/rNN_* (default r06): rocprofv3 kernel stats of one bench pass + separate PMC passes (FETCH_SIZE, WRITE_SIZE, MFMA busy),
# plus kernel-stat summaries of the secondary configurations (cfg3 VAE 32 x 512 px, cfg5 768 px loop, cfg4 per-GPU training step).
# Run on the GPU box from the repo root: bash scripts/profile_round.sh ; outputs land in gpurun_out/ (copy the CSVs to profiles/).
# Counter passes are separate runs with --pmc only (never combined with trace domains).
R=$PWD
P=${1:-r06}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/rp_stats /tmp/rp_fetch /tmp/rp_write /tmp/rp_mfma /tmp/rp_extra /tmp/rp_train
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_stats -o $P -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary > $R/gpurun_out/rp_stats.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/rp_fetch -o $P -- python3 $R/bench.py --steps 1 --warmup 0 --denoise-steps 4 --no-cpu-baseline --no-profile --no-secondary > $R/gpurun_out/rp_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/rp_write -o $P -- python3 $R/bench.py --steps 1 --warmup 0 --denoise-steps 4 --no-cpu-baseline --no-profile --no-secondary > $R/gpurun_out/rp_write.log 2>&1
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/rp_mfma -o $P -- python3 $R/bench.py --steps 1 --warmup 0 --denoise-steps 4 --no-cpu-baseline --no-profile --no-secondary > $R/gpurun_out/rp_mfma.log 2>&1
# L2-side and wave-state counters (their own passes; TCC has 4 slots, SQ 8): what bounds the kernels that sit at neither the HBM nor the MFMA roof
rm -rf /tmp/rp_tcc /tmp/rp_sq
timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum --output-format csv -d /tmp/rp_tcc -o $P -- python3 $R/bench.py --steps 1 --warmup 0 --denoise-steps 4 --no-cpu-baseline --no-profile --no-secondary > $R/gpurun_out/rp_tcc.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d /tmp/rp_sq -o $P -- python3 $R/bench.py --steps 1 --warmup 0 --denoise-steps 4 --no-cpu-baseline --no-profile --no-secondary > $R/gpurun_out/rp_sq.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_extra -o $P -- python3 $R/scripts/bench_extra.py --skip-vit > $R/gpurun_out/${P}_cfg3_cfg5.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_train -o $P -- python3 $R/scripts/bench_train.py --steps 2 --warmup 1 > $R/gpurun_out/${P}_train.log 2>&1
python3 $R/scripts/bench_train.py --steps 3 --warmup 2 --mixed-precision fp16 > $R/gpurun_out/${P}_train_fp16.log 2>&1
cd $R
# the L2 -> LDS ceiling roofline.secondary_bound is priced against (LDS-DMA streams, no compute; "shared" = an L2-resident stream, "private" = HBM)
hipcc --offload-arch=gfx950 -O3 -o /tmp/l2_lds_probe scripts/l2_lds_probe.hip > gpurun_out/l2_lds_probe_build.log 2>&1 && timeout 120 /tmp/l2_lds_probe > gpurun_out/${P}_l2_lds_probe.txt 2>&1
python3 scripts/rocprof_to_profiles.py /tmp/rp_stats /tmp/rp_fetch /tmp/rp_write gpurun_out/$P /tmp/rp_mfma
python3 scripts/rocprof_to_profiles.py l2 /tmp/rp_tcc /tmp/rp_sq gpurun_out/$P || tail -5 gpurun_out/rp_tcc.log gpurun_out/rp_sq.log
cp $(find /tmp/rp_extra -name "*kernel_stats.csv" | head -1) gpurun_out/${P}_cfg3_cfg5_kernel_stats.csv
cp $(find /tmp/rp_train -name "*kernel_stats.csv" | head -1) gpurun_out/${P}_train_kernel_stats.csv
grep -h '^{"metric"' gpurun_out/rp_stats.log > gpurun_out/${P}_bench_line_profiled.json
grep -h '^{' gpurun_out/${P}_cfg3_cfg5.log gpurun_out/${P}_train.log gpurun_out/${P}_train_fp16.log > gpurun_out/${P}_secondary_configs.jsonl
head -8 gpurun_out/${P}_kernel_stats.csv | cut -c1-160
head -6 gpurun_out/${P}_pmc_traffic.csv
head -8 gpurun_out/${P}_pmc_mfma.csv
head -8 gpurun_out/${P}_pmc_l2.csv | cut -c1-400
cat gpurun_out/${P}_secondary_configs.jsonl
