#!/bin/bash
# Audit of the 4-wave prefill kernel's register ownership (duo_prefill_w64.h): the accumulator half of the register
# file is addressed by literal register numbers inside asm statements, so the compiler must not use it at all.
set -e
d=$(mktemp -d)
cd $d
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -save-temps -c ${1:-/root/repo}/duo-attention_amd/csrc/duo_prefill.hip -o pf.o 2>/dev/null
python3 - <<'PY'
import re, collections, sys
s = open('duo_prefill-hip-amdgcn-amd-amdhsa-gfx950.s').read()
rc = 0
for name in ('duo_prefill_w64_kernelENS_13PrefillParamsE:', 'duo_prefill_w64_f16_kernelENS_13PrefillParamsE:'):
    a = s.index(name)
    k = s[a:]; k = k[:k.index('.Lfunc_end')]
    tail = s[a:]; tail = tail[tail.index('.Lfunc_end'):][:3000]
    stats = dict(re.findall(r'; (NumVgprs|NumAgprs|ScratchSize): (\d+)', tail))
    inasm, bad = False, []
    for l in k.splitlines():
        if 'ASMSTART' in l: inasm = True; continue
        if 'ASMEND' in l: inasm = False; continue
        if not inasm and ('accvgpr' in l or re.search(r'\ba\[?\d+', l.split(';')[0])): bad.append(l.strip())
    print(name.split('ENS')[0], stats, 'compiler instructions touching AGPRs:', len(bad))
    if bad or stats.get('ScratchSize') != '0': rc = 1
sys.exit(rc)
PY
