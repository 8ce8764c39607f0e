DIFFUTE_HIP_LIB=ab/lib_cur.so python3 scripts/attn_ab.py dump /tmp/attn_base.pt 2>/dev/null
DIFFUTE_HIP_LIB=ab/lib_split.so python3 scripts/attn_ab.py cmp /tmp/attn_base.pt 2>/dev/null | head -3
DIFFUTE_HIP_LIB=ab/lib_cur.so python3 scripts/attn_ab.py dump /tmp/attn_base.pt 2>/dev/null
DIFFUTE_HIP_LIB=ab/lib_split.so python3 scripts/attn_ab.py cmp /tmp/attn_base.pt 2>/dev/null | head -1
