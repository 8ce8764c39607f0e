# Regenerates profiles/r01_*: rocprofv3 kernel stats of one bench pass + separate PMC passes (FETCH_SIZE, WRITE_SIZE, MFMA busy).
# Run on the GPU box from the repo root: bash scripts/profile_round.sh ; outputs land in gpurun_out/ (copy the CSVs to profiles/).
R=$PWD
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/rp_stats /tmp/rp_fetch /tmp/rp_write /tmp/rp_mfma
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_stats -o r01 -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/rp_stats.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/rp_fetch -o r01 -- python3 $R/bench.py --steps 1 --warmup 0 --denoise-steps 4 --no-cpu-baseline --no-profile > $R/gpurun_out/rp_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/rp_write -o r01 -- python3 $R/bench.py --steps 1 --warmup 0 --denoise-steps 4 --no-cpu-baseline --no-profile > $R/gpurun_out/rp_write.log 2>&1
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/rp_mfma -o r01 -- python3 $R/bench.py --steps 1 --warmup 0 --denoise-steps 4 --no-cpu-baseline --no-profile > $R/gpurun_out/rp_mfma.log 2>&1
cd $R
python3 scripts/rocprof_to_profiles.py /tmp/rp_stats /tmp/rp_fetch /tmp/rp_write gpurun_out/r01 /tmp/rp_mfma
head -8 gpurun_out/r01_kernel_stats.csv | cut -c1-160
head -6 gpurun_out/r01_pmc_traffic.csv
head -8 gpurun_out/r01_pmc_mfma.csv
