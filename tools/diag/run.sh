#!/bin/bash
# diagnostic: where does a k-step's time go?  (GPU box) — same shapes through the real kernel and the crippled builds
cd $GRAFT_REPO_ROOT
S="${DIAG_SHAPES:-qkv mproj b_dxfc b_dO dx_fc llama_down}"
echo "== real";      timeout 200 python tools/bench_gemm.py $S 2>&1 | grep "M="
for v in ${DIAG_VARIANTS:-NOLOAD NOCOMPUTE NOMFMA}; do echo "== $v"; MTL_ALLOW_DIAG_LIB=1 MTL_LIB_PATH=$GRAFT_REPO_ROOT/tools/diag/libdiag_$v.so timeout 200 python tools/bench_gemm.py $S 2>&1 | grep "M="; done
