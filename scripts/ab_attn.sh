for i in 1 2 3; do
python3 scripts/ab_libs.py ab/lib_gm8.so 2>&1 | tail -1
python3 scripts/ab_libs.py - 2>&1 | tail -1
done
