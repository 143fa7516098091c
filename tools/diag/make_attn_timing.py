#!/usr/bin/env python3
"""Diagnostic build of the attention library: attn_fwd_res_kernel stamps s_memtime at its phase boundaries and writes, per workgroup,
the cycle counts of wave 0 into the `lse` output (whose values are then meaningless): [start->loads issued+landed+barrier, tile 1,
tile 2 + epilogue, total] — tools/diag/attn_timing.py prints their distribution. Not shipped."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
HERE = os.path.join(ROOT, "tools", "diag")
s = open(os.path.join(ROOT, "med-ts-llm_amd", "csrc", "mtl_attention.hip")).read()
i = s.index("__global__ __launch_bounds__(NW * 64) void attn_fwd_res_kernel")
j = s.index("template <int D, int NW, bool DROP = false>", i)
body = s[i:j]
body = body.replace("    const bf16_t* Q = reinterpret_cast<const bf16_t*>(a.q) + b * a.q_bs + h * a.q_hs;\n    load_rows_pair",
                    "    const uint64_t T0 = __builtin_amdgcn_s_memtime();\n    const bf16_t* Q = reinterpret_cast<const bf16_t*>(a.q) + b * a.q_bs + h * a.q_hs;\n    load_rows_pair", 1)
body = body.replace("    __syncthreads();\n    const uint32_t dbase", "    __syncthreads();\n    const uint64_t T1 = __builtin_amdgcn_s_memtime();\n    uint64_t T2 = T1;\n    const uint32_t dbase", 1)
body = body.replace("        if (half == 1 && tile == pi) break;", "        if (half == 1 && tile == pi) break;\n        if (half == 1) T2 = __builtin_amdgcn_s_memtime();", 1)
# final stamp: after the loop over halves
k = body.rindex("}\n")
body = body[:k] + ("    if (threadIdx.x == 0) {\n        const uint64_t T3 = __builtin_amdgcn_s_memtime();\n"
                   "        float* tb = a.lse + ((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 8;\n"
                   "        tb[0] = (float)(T1 - T0); tb[1] = (float)(T2 - T1); tb[2] = (float)(T3 - T2); tb[3] = (float)(T3 - T0); tb[4] = (float)((T0 >> 12) & 0xffffff); tb[5] = (float)(T0 & 0xfff);\n"
                   "        tb[6] = (float)(__builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11)) & 0xffff); tb[7] = (float)(__builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11)) & 0xf);\n    }\n") + body[k:]
# the regular lse store must not clobber the stamps
body = body.replace("if (g == 0 && a.lse) a.lse[", "if (false) a.lse[")
s = s[:i] + body + s[j:]
src = os.path.join(HERE, "_attn_timing.hip")
open(src, "w").write(s)
objs = [os.path.join(ROOT, "build", "obj", f"mtl_{n}.o") for n in ("gemm", "norm", "elementwise", "tokenizer", "backbone", "optim")]
o = os.path.join(HERE, "_attn_timing.o")
subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", os.path.join(ROOT, "med-ts-llm_amd", "csrc"), "-c", src, "-o", o], check=True, stderr=subprocess.DEVNULL)
subprocess.run(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(HERE, "libdiag_attn_timing.so"), o, *objs], check=True)
os.remove(o); os.remove(src)
print("built libdiag_attn_timing.so")
