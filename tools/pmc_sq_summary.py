"""per-kernel averages of SQ counters from rocprofv3 --pmc passes: python tools/pmc_sq_summary.py <dir> [<dir> ...]"""
import csv, glob, os, re, sys
from collections import defaultdict


def short(name):
    m = re.search(r"(\w+_kernel(<[^>]*>)?)", name)
    return m.group(1) if m else name[:60]


acc = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(lambda: defaultdict(int))
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[k][r["Counter_Name"]] += 1
dur = defaultdict(list)
for d in sys.argv[1:2]:
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            dur[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
names = sorted({c for k in acc for c in acc[k]})
rows = sorted(acc, key=lambda k: -acc[k].get("SQ_WAVE_CYCLES", 0.0))[:22]
for k in rows:
    print(k)
    a = {c: acc[k][c] / max(cnt[k][c], 1) for c in acc[k]}
    for c in names:
        if c in a:
            print(f"    {c:32s} {a[c]:16.0f}   ({cnt[k][c]} dispatches)")
    if dur.get(k):
        us = sum(dur[k]) / len(dur[k])
        print(f"    {'duration under the counter pass (us)':44s} {us:10.1f}")
        if "GRBM_GUI_ACTIVE" in a:
            # GRBM_GUI_ACTIVE is summed over the 8 XCDs (MI355X_MICROARCH.md): a dispatch lasts GUI_ACTIVE / 8 cycles.  (Round 3's file divided
            # by GUI_ACTIVE itself: "clock 14.9 GHz", "matrix pipe busy 0.046" -- both off by the 8 XCDs)
            cyc = a["GRBM_GUI_ACTIVE"] / 8.0
            print(f"    {'clock = GRBM_GUI_ACTIVE / 8 / duration (GHz)':44s} {cyc / us * 1e-3:8.3f}")
            if "SQ_VALU_MFMA_BUSY_CYCLES" in a:
                print(f"    {'matrix pipe busy = MFMA_BUSY / (1024 SIMDs x GUI_ACTIVE / 8)':44s} {a['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024 / cyc:8.3f}")
    w = a.get("SQ_WAVE_CYCLES")
    if w:
        # WAVE_CYCLES, WAIT_*, ACTIVE_INST_* count quad-cycles summed over waves; VALU_MFMA_BUSY_CYCLES counts cycles per SIMD (guide, constants table)
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_MISC", "SQ_ACTIVE_INST_SCA"):
            if c in a:
                print(f"    {c + ' / WAVE_CYCLES':44s} {a[c] / w:8.3f}")
    b = a.get("SQ_BUSY_CYCLES")
    if b and "SQ_VALU_MFMA_BUSY_CYCLES" in a:
        print(f"    {'MFMA_BUSY / BUSY_CYCLES':44s} {a['SQ_VALU_MFMA_BUSY_CYCLES'] / b:8.3f}")
        if "SQ_VALU_MFMA_COEXEC_CYCLES" in a:
            print(f"    {'COEXEC / MFMA_BUSY':44s} {a['SQ_VALU_MFMA_COEXEC_CYCLES'] / max(a['SQ_VALU_MFMA_BUSY_CYCLES'], 1):8.3f}")
