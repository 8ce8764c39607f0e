for i in 1 2 3; do
python3 scripts/ab_libs.py - 2>&1 | tail -1 | sed 's/^/default            /'
HIP_FORCE_DEV_KERNARG=1 python3 scripts/ab_libs.py - 2>&1 | tail -1 | sed 's/^/DEV_KERNARG=1      /'
HIP_FORCE_DEV_KERNARG=0 python3 scripts/ab_libs.py - 2>&1 | tail -1 | sed 's/^/DEV_KERNARG=0      /'
done
