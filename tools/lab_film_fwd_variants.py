"""Ablation / stamp builds of the PAIR-SPECIALISED forward kernel experiment of round 4 (profiles/r04_film_fwd_pair_experiment.txt).
The text substitutions below match film_chain.hip with profiles/r04_film_fwd_pair_kernel.patch applied (git apply it on commit aa8f1f1 first).
python tools/lab_film_fwd_variants.py <names...> -> scratch/r4/lab/libfwd_<name>.so ; timed by tools/lab_film_fwd_bench.py <names...>"""
import os, subprocess, sys
R = "/root/repo"
SRC = open(f"{R}/neusky_amd/csrc/film_chain.hip").read()
CH = open(f"{R}/neusky_amd/csrc/chain.h").read()
LAB = f"{R}/scratch/r4/lab"
os.makedirs(LAB, exist_ok=True)
def build(name, src_edits=(), ch_edits=()):
    s, ch = SRC, CH
    for old, new in src_edits:
        if old == "SHAPE16": s = s.replace("// ------------------------------------------------------------------ role A", "__SPLIT__// ---- role A"); a, b = s.split("__SPLIT__"); i0 = a.index("constexpr int RF = 7;"); s = a[:i0] + shape16(a[i0:]) + b; continue
        assert s.count(old) >= 1, (name, old[:70]); s = s.replace(old, new)
    for old, new in ch_edits:
        assert ch.count(old) >= 1, (name, old[:70]); ch = ch.replace(old, new)
    open(f"{LAB}/chain_{name}.h", "w").write(ch)
    s = s.replace('#include "chain.h"', f'#include "{LAB}/chain_{name}.h"')
    p = f"{LAB}/film_{name}.hip"
    open(p, "w").write(s)
    o = f"{LAB}/film_{name}.o"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", f"-I{R}/neusky_amd/csrc", f"-I{R}/include", "-c", p, "-o", o])
    objs = [f"{R}/build/{f}" for f in os.listdir(f"{R}/build") if f.endswith(".o") and not f.startswith("film_chain")]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", f"{LAB}/libfwd_{name}.so", o] + objs)
    os.remove(o)
    print("built", name, flush=True)

def X(n):  # outstanding DMA groups allowed at a transition
    return [('asm volatile("s_waitcnt vmcnt(8)\\n\\ts_barrier" ::: "memory");\n  fr_issue(r, r.slot);', f'asm volatile("s_waitcnt vmcnt({4*n})\\n\\ts_barrier" ::: "memory");\n  fr_issue(r, r.slot);')]
GLDS = 'asm volatile("s_mov_b32 %0, m0\\n\\ts_mov_b32 m0, %2\\n\\ts_nop 0\\n\\tglobal_load_lds_dwordx4 %1, off\\n\\ts_mov_b32 m0, %0"\n               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");'
NODMA = [(GLDS, 'asm volatile("" : "=s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");')]
NOEPI = [("      yy[q] = sin_cw(fmaf(fmaf(15.0f, F, 30.0f), z, P));", "      yy[q] = F + P;")]
NOBIASG = [("    const float4 b4F = ldg4(bF + fo), b4P = ldg4(bP + fo), b4Z = ldg4(bZ + fo);", "    const float4 b4F = make_float4(iF, iP, iZ, 1.f), b4P = b4F, b4Z = b4F;")]
# stamps: s_memtime of wave 0 (A) and wave 4 (B) of one workgroup at the phase boundaries
STAMP_DEF = """
__device__ unsigned long long g_stamps[64];
#define STAMPR(k) do { if (blockIdx.x == 700 && (threadIdx.x & 255) == 0) g_stamps[(threadIdx.x >> 8) * 16 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define STAMP(k) do { if (blockIdx.x == 700 && (threadIdx.x & 255) == 0) g_stamps[(threadIdx.x >> 8) * 16 + (k)] = __builtin_readcyclecounter(); } while (0)
"""
STAMPS = [("constexpr int RF = 7;", STAMP_DEF + "constexpr int RF = 7;"),
          ("    // ---- mapping layer 0\n    float m = 0.0f;", "    STAMP(1);\n    // ---- mapping layer 0\n    float m = 0.0f;"),
          ("  hinv_s[lane] = h_inv;  // B's", "  STAMP(2);\n  hinv_s[lane] = h_inv;  // B's"),
          ("  asm volatile(\"s_waitcnt lgkmcnt(0)\" ::: \"memory\");\n  // ---- drain:", "  STAMP(3);\n  asm volatile(\"s_waitcnt lgkmcnt(0)\" ::: \"memory\");\n  // ---- drain:"),
          ("  for (int k = 0; k < nd; ++k) trans_a<1>(r);\n}", "  for (int k = 0; k < nd; ++k) trans_a<1>(r);\n  STAMP(4);\n}"),
          ("  for (int k = 0; k < nm; ++k) trans_b<1>(r);", "  STAMP(1);\n  for (int k = 0; k < nm; ++k) trans_b<1>(r);\n  STAMP(2);"),
          ("  // ---- head\n  {\n    f32x16 acc;", "  STAMP(3);\n  // ---- head\n  {\n    f32x16 acc;"),
          ("  else fwd_role_b<H>(a, r, xch, sl, mb, hinv_s + pair * 64, rt, lane, wave_live, tid - 256);", "  else fwd_role_b<H>(a, r, xch, sl, mb, hinv_s + pair * 64, rt, lane, wave_live, tid - 256);\n  STAMP(5);"),
          ("  FRing r;\n  r.lds_lane", "  STAMP(0);\n  FRing r;\n  r.lds_lane"),
          ("  STAMP(0);\n", "  STAMP(0); STAMPR(8);\n"),
          ("tid - 256);\n  STAMP(5);", "tid - 256);\n  STAMP(5); STAMPR(9);"),
          # A, slot 20: before the frequency product, between, after the phase product
          ("    if (s == 0) product_a<KS, 1, 1>(r, cy, hh, hl, aF);", "    if (s == 20) STAMP(10);\n    if (s == 0) product_a<KS, 1, 1>(r, cy, hh, hl, aF);"),
          ("    product_a<KS, 1, 1>(r, cy, hh, hl, aP);\n  }", "    if (s == 20) STAMP(11);\n    product_a<KS, 1, 1>(r, cy, hh, hl, aP);\n    if (s == 20) STAMP(12);\n  }"),
          # B, tile 19 (slot 20): before z product, after, after part 0, after part 1
          ("    if (i == 0) {\n      if (!drain) {\n        product_b<1, 2, 1>", "    if (u == 19) STAMP(10);\n    if (i == 0) {\n      if (!drain) {\n        product_b<1, 2, 1>"),
          ("    const float z_unscale = i == 0 ? x_inv : 1.0f / Y_SCALE;", "    if (u == 19) STAMP(11);\n    const float z_unscale = i == 0 ? x_inv : 1.0f / Y_SCALE;"),
          ("      if (GH == 2) {\n        trans_b<1>(r);", "      if (u == 19) STAMP(12);\n      if (GH == 2) {\n        trans_b<1>(r);\n        if (u == 19) STAMP(13);"),
          ("      if (t == NT - 1) rebuild_y<KS>(a.y_save[i] + blk, lane, yh, yl);\n      trans_b<1>(r);", "      if (u == 19) STAMP(14);\n      if (t == NT - 1) rebuild_y<KS>(a.y_save[i] + blk, lane, yh, yl);\n      trans_b<1>(r);\n      if (u == 19) STAMP(15);"),
          ('extern "C" int nsky_film_stream_layout(', 'extern "C" int nsky_lab_stamps(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 64); }\nextern "C" int nsky_film_stream_layout(')]
MFB = ["        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh[s % 3], bl[ks], acc, 0, 0, 0);\n", "        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fl[s % 3], bh[ks], acc, 0, 0, 0);\n", "        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh[s % 3], bh[ks], acc, 0, 0, 0);\n"]
NOMFMA_B = [(MFB[0], '        asm volatile("" :: "v"(fh[s % 3]), "v"(bl[ks]));\n'), (MFB[1], '        asm volatile("" :: "v"(fl[s % 3]), "v"(bh[ks]));\n'), (MFB[2], "")]
MFA = ["    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh[ks % 3], bl[ks], acc, 0, 0, 0);\n", "    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fl[ks % 3], bh[ks], acc, 0, 0, 0);\n", "    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh[ks % 3], bh[ks], acc, 0, 0, 0);\n"]
NOMFMA_A = [(MFA[0], '    asm volatile("" :: "v"(fh[ks % 3]), "v"(bl[ks]));\n'), (MFA[1], '    asm volatile("" :: "v"(fl[ks % 3]), "v"(bh[ks]));\n'), (MFA[2], "")]
NOSTORE = [("    if (wave_live) {\n      if (zblk)", "    if (wave_live && zz[0] == 1.2345e30f) {\n      if (zblk)")]
HALFDMA = [("  for (int p = 0; p < 4; ++p) glds16(r.src + p * 1024, d + p * 1024);", "  for (int p = 0; p < 4; p += 2) glds16(r.src + p * 1024, d + p * 1024);"),
           ('asm volatile("s_waitcnt vmcnt(8)\\n\\ts_barrier" ::: "memory");\n  fr_issue(r, r.slot);', 'asm volatile("s_waitcnt vmcnt(4)\\n\\ts_barrier" ::: "memory");\n  fr_issue(r, r.slot);')]
PREFILL = [("  __syncthreads();\n  asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");\n  STAMP(0);", """  { unsigned sd = tid * 2654435761u + 12345u;
    for (int i = tid; i < RF * GROUP / 4; i += 512) { sd = sd * 1664525u + 1013904223u; const unsigned e0 = 0x3000u + ((sd >> 8) & 0x0fffu) + ((sd >> 30) << 15), e1 = 0x3000u + ((sd >> 20) & 0x0fffu) + (((sd >> 29) & 1u) << 15); reinterpret_cast<unsigned*>(smem)[i] = e0 | (e1 << 16); } }
  __syncthreads();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  STAMP(0);""")]
def shape16(txt):
    import re
    # acc = mfma_32x32x16(A, B, acc) -> two 16x16x32 on 4-register slices of acc
    pat = re.compile(r"(\w+) = __builtin_amdgcn_mfma_f32_32x32x16_f16\(([^;]*?), \1, 0, 0, 0\);")
    def rep(m):
        acc, ops = m.group(1), m.group(2)
        a_, b_ = [x.strip() for x in ops.split(",")]
        ops2 = b_ + ", " + a_  # (swapped operands: the second chain must not be a common subexpression of the first)
        return ("{ f32x4 t0 = {%s[0], %s[1], %s[2], %s[3]}, t1 = {%s[4], %s[5], %s[6], %s[7]}; t0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(%s, t0, 0, 0, 0); t1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(%s, t1, 0, 0, 0); "
                "%s[0] = t0[0]; %s[1] = t0[1]; %s[2] = t0[2]; %s[3] = t0[3]; %s[4] = t1[0]; %s[5] = t1[1]; %s[6] = t1[2]; %s[7] = t1[3]; }") % ((acc,) * 8 + (ops, ops2) + (acc,) * 8)
    return pat.sub(rep, txt)
V = {"base": ((), ()), "x3": (X(3), ()), "x4": (X(4), ()), "nodma": ((), NODMA), "noepi": (NOEPI, ()), "nobiasg": (NOBIASG, ()), "stamps": (STAMPS, ()), "nomfma_b": (NOMFMA_B, ()), "nomfma_a": (NOMFMA_A, ()), "nomfma_ab": (NOMFMA_A + NOMFMA_B, ()), "nostore": (NOSTORE, ()), "stamps_nodma": (STAMPS, NODMA), "st_shape16": (STAMPS + [("SHAPE16", "")], ()), "st_nodma_prefill": (STAMPS + PREFILL, NODMA), "st_mfma_only": (STAMPS + PREFILL + NOSTORE + NOEPI + NOBIASG, NODMA), "st_nostore_noepi": (STAMPS + NOSTORE + NOEPI + NOBIASG, ()), "st_halfdma": (STAMPS + HALFDMA, ()), "st_nomfma": (STAMPS + NOMFMA_A + NOMFMA_B, ()), "st_nostore": (STAMPS + NOSTORE, ()), "st_noepi": (STAMPS + NOEPI + NOBIASG, ()), "st_nodma_nostore": (STAMPS + NOSTORE, NODMA), "st_nomfma_nodma": (STAMPS + NOMFMA_A + NOMFMA_B, NODMA),
     "noepi_nobiasg": (NOEPI + NOBIASG, ()), "nodma_noepi_nobiasg": (NOEPI + NOBIASG, NODMA)}
if __name__ == "__main__":
    for n in sys.argv[1:]:
        build(n, *V[n])
