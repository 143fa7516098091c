#!/usr/bin/env python3
"""Build crippled copies of the GEMM library to see where a k-step's time goes (diagnostic, not shipped):
  NOLOAD     the in-loop global->LDS DMA is skipped (ds_read + MFMA only)
  NOCOMPUTE  ds_read + MFMA are skipped (the DMA pipeline alone)
  NOMFMA     DMA + ds_read, MFMA replaced by one add
  NOCOMPUTE_NOSWZ  as NOCOMPUTE without the source-side XOR swizzle
  NOEPI      the epilogue (math + stores) of the persistent kernel is skipped: what the main loops alone cost
  NOSTORE / NOAUX / NOSTORE_NOAUX  the wave-level epilogue without its stores / its residual-type loads / both
Run after `make -C med-ts-llm_amd/csrc` (reuses build/obj/*.o); then on the GPU box `bash tools/diag/run.sh`.
Results of 2026-09-28 are in profiles/r01_gemm_diag.txt."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
HERE = os.path.join(ROOT, "tools", "diag")
s = open(os.path.join(ROOT, "med-ts-llm_amd", "csrc", "mtl_gemm.hip")).read()


def sub(a, b, count=1):
    global s
    assert s.count(a) == count, (s.count(a), a[:60])
    s = s.replace(a, b)


sub("            stage(s_buf, s_kt);\n            s_buf = (s_buf + 1 == STAGES) ? 0 : s_buf + 1;\n            if (++s_kt == nkt) { s_kt = 0; ++s_i; }\n        }\n        const char* la",
    "#ifndef DIAG_NOLOAD\n            stage(s_buf, s_kt);\n#endif\n            s_buf = (s_buf + 1 == STAGES) ? 0 : s_buf + 1;\n            if (++s_kt == nkt) { s_kt = 0; ++s_i; }\n        }\n        const char* la")
sub("#pragma unroll\n        for (int ks = 0; ks < 2; ++ks) {\n            const int pc16 = ((ks * 4 + g) ^ sw) * 16;\n            bf16x8 af[4], bfr[NI]",
    "#ifdef DIAG_NOCOMPUTE\n        if (p.alpha == 12345.f)\n#endif\n#pragma unroll\n        for (int ks = 0; ks < 2; ++ks) {\n            const int pc16 = ((ks * 4 + g) ^ sw) * 16;\n            bf16x8 af[4], bfr[NI]")
sub("                    acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[ni], af[mi], acc[ni][mi], 0, 0, 0);\n        }\n        c_buf =",
    "#ifdef DIAG_NOMFMA\n                    { acc[ni][mi][0] += (float)af[mi][0] + (float)bfr[ni][0]; }\n#else\n                    acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[ni], af[mi], acc[ni][mi], 0, 0, 0);\n#endif\n        }\n        c_buf =")
sub("__device__ __forceinline__ int swz64(int row) { return (row >> 1) & 7; }",
    "#ifdef DIAG_NOSWZ\n__device__ __forceinline__ int swz64(int row) { (void)row; return 0; }\n#else\n__device__ __forceinline__ int swz64(int row) { return (row >> 1) & 7; }\n#endif")
sub("    auto finish = [&](int item) {\n        int tm, tn;",
    "    auto finish = [&](int item) {\n#ifdef DIAG_NOEPI\n        if (p.alpha != 12345.f) return;\n#endif\n        int tm, tn;")
sub("            const bool ok = mok[mi] && nok[ni];", "            const bool ok = mok[mi] && nok[ni] DIAG_OK;")
sub("    if constexpr (EPI == MTL_EPI_RESID || EPI == MTL_EPI_ACCUM || EPI == MTL_EPI_DGELU || EPI == MTL_EPI_DSWIGLU) {\n#pragma unroll\n        for (int mi = 0; mi < 4; ++mi)",
    "    if constexpr (EPI == MTL_EPI_RESID || EPI == MTL_EPI_ACCUM || EPI == MTL_EPI_DGELU || EPI == MTL_EPI_DSWIGLU) {\n        if (DIAG_AUX)\n#pragma unroll\n        for (int mi = 0; mi < 4; ++mi)")
s = ("#ifdef DIAG_NOSTORE\n#define DIAG_OK && p.alpha == 12345.f\n#else\n#define DIAG_OK\n#endif\n"
     "#ifdef DIAG_NOAUX\n#define DIAG_AUX p.alpha == 12345.f\n#else\n#define DIAG_AUX true\n#endif\n") + s
src = os.path.join(HERE, "_gemm_diag.hip")
open(src, "w").write(s)
objs = [os.path.join(ROOT, "build", "obj", f"mtl_{n}.o") for n in ("attention", "norm", "elementwise", "tokenizer", "backbone", "optim")]
for name, defs in (("NOLOAD", ["NOLOAD"]), ("NOCOMPUTE", ["NOCOMPUTE"]), ("NOMFMA", ["NOMFMA"]), ("NOEPI", ["NOEPI"]), ("NOSTORE", ["NOSTORE"]), ("NOAUX", ["NOAUX"]), ("NOSTORE_NOAUX", ["NOSTORE", "NOAUX"])):
    o = os.path.join(HERE, f"_g_{name}.o")
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", *[f"-DDIAG_{d}" for d in defs],
                    "-I", os.path.join(ROOT, "med-ts-llm_amd", "csrc"), "-c", src, "-o", o], check=True, stderr=subprocess.DEVNULL)
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(HERE, f"libdiag_{name}.so"), o, *objs], check=True)
    os.remove(o)
os.remove(src)
print("built", sorted(f for f in os.listdir(HERE) if f.endswith(".so")))
