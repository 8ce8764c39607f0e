# one-off: validate the L2 / SQ counter passes of profile_round.sh on their own
R=$PWD; P=r06
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/rp_tcc /tmp/rp_sq
timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum --output-format csv -d /tmp/rp_tcc -o $P -- python3 $R/bench.py --steps 1 --warmup 0 --denoise-steps 4 --no-cpu-baseline --no-profile --no-secondary > $R/gpurun_out/rp_tcc.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d /tmp/rp_sq -o $P -- python3 $R/bench.py --steps 1 --warmup 0 --denoise-steps 4 --no-cpu-baseline --no-profile --no-secondary > $R/gpurun_out/rp_sq.log 2>&1
cd $R
python3 scripts/rocprof_to_profiles.py l2 /tmp/rp_tcc /tmp/rp_sq gpurun_out/$P || { tail -8 gpurun_out/rp_tcc.log gpurun_out/rp_sq.log; rocprofv3 -L 2>/dev/null | grep -o "TCC_[A-Za-z_0-9]*\|TCP_TCC[A-Za-z_0-9]*" | sort -u | head -60; }
head -12 gpurun_out/${P}_pmc_l2.csv | cut -c1-330
