python3 -m pytest tests/test_ops_gpu.py -q -x -k "attention" 2>&1 | tail -2
python3 -m pytest tests/test_train_ops_gpu.py -q -x -k "attention" 2>&1 | tail -2
DIFFUTE_HIP_LIB=ab/lib_base.so python3 scripts/attn_ab.py dump /tmp/attn_base.pt 2>/dev/null
python3 scripts/attn_ab.py cmp /tmp/attn_base.pt 2>/dev/null | head -3
DIFFUTE_HIP_LIB=ab/lib_base.so python3 scripts/ab_libs.py ab/lib_base.so 2>/dev/null | tail -1
python3 scripts/ab_libs.py - 2>/dev/null | tail -1
DIFFUTE_HIP_LIB=ab/lib_base.so python3 scripts/ab_libs.py ab/lib_base.so 2>/dev/null | tail -1
python3 scripts/ab_libs.py - 2>/dev/null | tail -1
