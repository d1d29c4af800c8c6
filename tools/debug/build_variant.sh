#!/bin/bash
# Builds a measurement variant of the HIP library: [SRC=duo_decode] tools/debug/build_variant.sh <tag> [-DSWITCH ...]
#   -> duo-attention_amd/lib/ab/lib_<tag>.so   (only $SRC.hip — default duo_prefill — is rebuilt with the switches)
set -e
tag=$1; shift
SRC=${SRC:-duo_prefill}
root=$(cd "$(dirname "$0")/../.." && pwd)
src=$root/duo-attention_amd/csrc
make -s -C $src
mkdir -p $root/duo-attention_amd/lib/ab /tmp/ab_$tag
extra=""   # e.g. EXTRA="-mllvm -amdgpu-kernarg-preload-count=14" (measured: no effect on the decode scan, profiles/r3_decode.md)
extra="$EXTRA"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $extra "$@" -c $src/$SRC.hip -o /tmp/ab_$tag/$SRC.o
objs=""
for f in duo_decode duo_prefill duo_prefill_w32_debug duo_rope_kv duo_int4 duo_linear duo_tuple; do
  if [ $f = $SRC ]; then objs="$objs /tmp/ab_$tag/$f.o"; else objs="$objs $src/$f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/duo-attention_amd/lib/ab/lib_$tag.so $objs
echo built lib_$tag.so
