#!/usr/bin/env python3
"""python tools/kernel_resources.py [objects ...] : registers, scratch (spills), LDS and the resulting waves per SIMD of every kernel, read
from the code objects' own metadata (llvm-readelf --notes of the gfx950 code object unbundled from build/*.o).  No GPU involved.
gfx950: 512 VGPRs per SIMD lane shared by ArchVGPRs and AGPRs (unified file, allocation granule 8), 160 KB LDS per CU."""
import glob
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"


def metadata(fat, tmp):
    co = os.path.join(tmp, os.path.basename(fat) + ".co")
    sections = subprocess.run([f"{LLVM}/llvm-readelf", "-S", fat], capture_output=True, text=True).stdout
    if ".hip_fatbin" not in sections:
        return []  # host-only object (api.cpp)
    blob = os.path.join(tmp, os.path.basename(fat) + ".fatbin")
    subprocess.run([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={blob}", fat], check=True)
    subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--targets={TARGET}", f"--input={blob}", f"--output={co}"], check=True)
    if not os.path.getsize(co):
        return []
    notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], check=True, capture_output=True, text=True).stdout
    kernels, cur = [], None
    for line in notes.splitlines():
        m = re.match(r"\s*-?\s*\.(\w+):\s*(.*)$", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2).strip().strip("'")
        if k == "agpr_count":  # first key of a kernel's map in the emitted order
            cur = {"agpr_count": int(v)}
            kernels.append(cur)
        elif cur is not None and k in ("vgpr_count", "sgpr_count", "private_segment_fixed_size", "group_segment_fixed_size", "vgpr_spill_count",
                                       "sgpr_spill_count", "max_flat_workgroup_size"):
            cur[k] = int(v)
        elif cur is not None and k == "symbol":  # (the .name keys in front of it are the arguments')
            cur["name"] = v[:-3] if v.endswith(".kd") else v
    return [k for k in kernels if "name" in k and "vgpr_count" in k]


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    return [re.sub(r"\(.*", "", o.replace("void ", "").replace("(anonymous namespace)::", "")) for o in out]


def main(argv):
    objs = argv or sorted(glob.glob("build/*.o"))
    rows = []
    with tempfile.TemporaryDirectory() as tmp:
        for o in objs:
            ks = metadata(o, tmp)
            for k, short in zip(ks, demangle([k["name"] for k in ks])):
                total = (k["vgpr_count"] + 7) // 8 * 8  # vgpr_count already holds ArchVGPRs + AGPRs of the unified file
                waves = max(1, min(8, 512 // max(total, 1)))
                wg = k.get("max_flat_workgroup_size", 256)
                lds = k.get("group_segment_fixed_size", 0)
                rows.append((os.path.basename(o).replace(".hip.o", ""), short, k["vgpr_count"], k["agpr_count"], k["sgpr_count"],
                             k.get("private_segment_fixed_size", 0), k.get("vgpr_spill_count", 0), lds, wg, waves))
    print(f"{'file':<13} {'kernel':<44} {'vgpr(+agpr)':>11} {'agpr':>5} {'sgpr':>5} {'scratch B':>9} {'spills':>6} {'static LDS B':>12} {'max wg':>6} {'waves/SIMD by regs':>18}")
    for r in sorted(rows):
        print(f"{r[0]:<13} {r[1][:44]:<44} {r[2]:>11} {r[3]:>5} {r[4]:>5} {r[5]:>9} {r[6]:>6} {r[7]:>12} {r[8]:>6} {r[9]:>18}")
    spilled = [r for r in rows if r[5] or r[6]]
    print(f"{len(rows)} kernels; with scratch or spills: {len(spilled)}" + ("" if not spilled else " -> " + ", ".join(f"{r[1][:30]} ({r[5]} B)" for r in spilled)))
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
