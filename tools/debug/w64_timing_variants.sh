for t in ${VARIANTS:-timing}; do echo "== $t"; DUO_ATTN_HIP_LIB=$PWD/duo-attention_amd/lib/ab/lib_$t.so python tools/debug/w64_timing.py 2>/dev/null | python -c "
import sys,json
txt=sys.stdin.read(); i=txt.index('{\n'); d=json.loads(txt[i:]); first=json.loads(txt[:i].strip().splitlines()[-1])
print(round(first['avg_ms'],3),'ms', [round(v) for v in d['cycles_per_tile'].values()], d['sum'])"; done
