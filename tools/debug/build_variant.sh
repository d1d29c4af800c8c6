#!/bin/bash
# Builds a measurement variant of the HIP library: tools/debug/build_variant.sh <tag> [-DSWITCH ...]
#   -> duo-attention_amd/lib/ab/lib_<tag>.so   (only duo_prefill.hip is rebuilt with the switches; other objects reused)
set -e
tag=$1; shift
root=$(cd "$(dirname "$0")/../.." && pwd)
src=$root/duo-attention_amd/csrc
make -s -C $src
mkdir -p $root/duo-attention_amd/lib/ab /tmp/ab_$tag
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function "$@" -c $src/duo_prefill.hip -o /tmp/ab_$tag/duo_prefill.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/duo-attention_amd/lib/ab/lib_$tag.so /tmp/ab_$tag/duo_prefill.o $src/duo_decode.o $src/duo_rope_kv.o $src/duo_int4.o
echo built lib_$tag.so
