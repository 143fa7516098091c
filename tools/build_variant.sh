#!/bin/bash
# usage: tools/build_variant.sh <tag> [only=<file>[,<file>...]] [-DFLAG=..]...  ->  tools/diag/_lib_<tag>.so (git-ignored; travels with gpurun) built from
# the current sources with extra compiler flags: diagnostic / A-B builds selected at run time with MTL_ALLOW_DIAG_LIB=1 MTL_LIB_PATH=... (pass -DMTL_DIAG for the environment switches). `only=mtl_gemm,mtl_norm`: only those
# sources are recompiled with the flags, the other objects are the in-tree build's (build/obj — run `make -C med-ts-llm_amd/csrc` first).
set -e
TAG=$1; shift
ONLY=""
case "$1" in only=*) ONLY=${1#only=}; shift;; esac
R=$(cd "$(dirname "$0")/.." && pwd)
OBJ=$R/build/obj_$TAG
mkdir -p $OBJ $R/tools/diag
ALL="mtl_gemm mtl_attention mtl_norm mtl_elementwise mtl_tokenizer mtl_backbone mtl_optim mtl_stats"
for f in $ALL; do
  # mtl_elementwise holds mtl_build_flags(): it is always rebuilt with the variant's flags, so that the variant reports what it is
  if [ -n "$ONLY" ] && [ "$f" != mtl_elementwise ] && ! echo ",$ONLY," | grep -q ",$f,"; then
    cp $R/build/obj/$f.o $OBJ/$f.o
  else
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function "$@" -c $R/med-ts-llm_amd/csrc/$f.hip -o $OBJ/$f.o &
  fi
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/diag/_lib_$TAG.so $OBJ/*.o
echo built $R/tools/diag/_lib_$TAG.so
