#!/bin/bash
# usage: tools/build_variant.sh <tag> [-DFLAG=..]...  ->  tools/diag/_lib_<tag>.so (git-ignored; travels with gpurun) built from the current
# sources with extra compiler flags: diagnostic / A-B builds selected at run time with MTL_LIB_PATH.
set -e
TAG=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd)
OBJ=$R/build/obj_$TAG
mkdir -p $OBJ $R/tools/diag
for f in mtl_gemm mtl_attention mtl_norm mtl_elementwise mtl_tokenizer mtl_backbone mtl_optim mtl_stats; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function "$@" -c $R/med-ts-llm_amd/csrc/$f.hip -o $OBJ/$f.o &
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/diag/_lib_$TAG.so $OBJ/*.o
echo built $R/tools/diag/_lib_$TAG.so
