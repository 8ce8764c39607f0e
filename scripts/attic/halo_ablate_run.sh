# kernel time of each ablation library on one shape (rocprofv3 kernel trace).  bash scripts/halo_ablate_run.sh "4 64 64 320 0 320 0" "0 1 2 ..."
R=$PWD; SH=${1:-"4 64 64 320 0 320 0"}
cd /tmp && export TMPDIR=/tmp
LIST=${2:-0 1 2 4 8 6 14 15 31}
for n in $LIST; do
  rm -rf /tmp/ha$n
  if [ $n = 0 ]; then unset DIFFUTE_HIP_LIB; else export DIFFUTE_HIP_LIB=$R/ab/libhalo_$n.so; fi
  timeout 120 rocprofv3 --kernel-trace --output-format csv -d /tmp/ha$n -o h -- python3 $R/scripts/halo_one.py $SH 8 > /tmp/ha$n.log 2>&1
  echo -n "dbg $n: "; python3 $R/scripts/trace_runs.py /tmp/ha$n 4 dmx_conv_halo | cut -c1-60
done
