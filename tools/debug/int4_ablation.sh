#!/bin/bash
# Where do the issue slots of duo_int4_decode_mfma_kernel go?  Ablation builds of csrc/duo_int4.hip (-DDUO_I4_PROBE=<bits>:
# 1 = no transcendental, 2 = K words to the MFMA without dequantisation, 4 = V words to LDS without dequantisation, 8 = no LDS round trip for V at all; results
# are wrong by design) timed on the one-layer 1 M-context leg of bench.py, alternating with the shipped library.
#   (here)      tools/debug/int4_ablation.sh build          -> duo-attention_amd/lib/probe_int4_<bits>.so
#   (GPU box)   tools/debug/int4_ablation.sh run > gpurun_out/int4_ablation.txt
set -e
cd "$(dirname "$0")/../.."
LIBDIR=duo-attention_amd/lib
CS=duo-attention_amd/csrc
BITS="1 2 4 6 7 8 14 15"
if [ "$1" = build ]; then
  make -s -C $CS
  for b in $BITS; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DDUO_I4_PROBE=$b -c $CS/duo_int4.hip -o /tmp/duo_int4_probe_$b.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $LIBDIR/probe_int4_$b.so /tmp/duo_int4_probe_$b.o \
        $(ls $CS/*.o | grep -v duo_int4.o)
  done
  ls -la $LIBDIR
else
  for rep in 1 2; do
    for b in 0 $BITS; do
      lib=$LIBDIR/libduoattn_hip.so; [ $b != 0 ] && lib=$LIBDIR/probe_int4_$b.so
      echo -n "probe=$b rep=$rep "
      DUO_ATTN_HIP_LIB=$PWD/$lib python tools/debug/int4_legs.py kernel 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin)['kernel_1M']
print('kernel_ms %.5f  %.0f GB/s' % (d['kernel_ms'], d['kernel_GBps']))"
    done
    echo -n "loads only (DUO_DEBUG_FLAGS=32) rep=$rep "
    DUO_DEBUG_FLAGS=32 python tools/debug/int4_legs.py kernel 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin)['kernel_1M']
print('kernel_ms %.5f  %.0f GB/s' % (d['kernel_ms'], d['kernel_GBps']))"
  done
fi
