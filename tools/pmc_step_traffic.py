"""whole-step HBM traffic by kernel from the two PMC passes of tools/pmc_bench.sh (eager, 1 warm-up + 2 timed + 1 roofline
iteration = 4 iterations): 2 x FETCH_SIZE + WRITE_SIZE (KiB counters; FETCH_SIZE doubled per MI355X_MICROARCH.md)"""
import csv, glob, os, re, sys, collections, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
iters = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
out_name = sys.argv[2] if len(sys.argv) > 2 else "r05_pmc_traffic.json"
tot = collections.defaultdict(lambda: [0.0, 0.0, 0])
for c, idx in (("FETCH_SIZE", 0), ("WRITE_SIZE", 1)):
    files = glob.glob(os.path.join(ROOT, "gpurun_out", f"pmc_bench_{c}", "**", "*counter_collection.csv"), recursive=True)
    for f in sorted(files, key=os.path.getmtime)[-1:]:  # the newest pass only (gpurun_out keeps older rounds' files)
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == c:
                k = re.sub(r"\(anonymous namespace\)::|^void ", "", row["Kernel_Name"])
                k = re.split(r"\((?![a-z])", k)[0][:90]
                tot[k][idx] += float(row["Counter_Value"]) * 1024.0
                if idx == 0:
                    tot[k][2] += 1
if iters <= 0:  # one hemisphere-composite launch per training iteration
    iters = float(max(v[2] for k, v in tot.items() if k.startswith("hemi_fwd_kernel")))
rows = sorted(((2 * v[0] + v[1], k, v) for k, v in tot.items()), reverse=True)
total = sum(r[0] for r in rows) / iters
print(f"whole step: {total/1e9:.2f} GB/iteration (fetch x2 {sum(2*r[2][0] for r in rows)/iters/1e9:.2f} + write {sum(r[2][1] for r in rows)/iters/1e9:.2f})")
for b, k, v in rows[:18]:
    print(f"{k:72s} {v[2]/iters:6.1f} launches/it {b/iters/1e9:8.2f} GB/it")
# per launch, by kernel family (template arguments of the dense-layer kernels folded together; the chain kernels by hidden width)
fam = collections.defaultdict(lambda: [0.0, 0])
for b, k, v in rows:
    name = k.split("(")[0].strip()
    key = re.sub(r"<(\d+)[^>]*>", r"<\1>", name) if name.startswith("film_") else name.split("<")[0]
    fam[key][0] += b
    fam[key][1] += v[2]
out = {"command": "tools/pmc_bench.sh: rocprofv3 --kernel-trace --pmc <FETCH_SIZE|WRITE_SIZE> -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-graph --no-exact-f32 (two passes)",
       "units": "bytes = 1024 * (2 * FETCH_SIZE + WRITE_SIZE): counters are KiB; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half the bytes of wide coalesced reads)",
       "whole_step_GB": total / 1e9, "iterations": iters,
       "bytes_per_launch": {k: v[0] / max(v[1], 1) for k, v in fam.items() if v[1] > 0 and v[0] / iters > 5e7},
       "GB_per_iteration": {k: v[0] / iters / 1e9 for k, v in sorted(fam.items(), key=lambda kv: -kv[1][0])[:20]}}
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (for the source hash bench.py compares against: the counters describe THESE kernels)
csrc = os.path.join(ROOT, "neusky_amd", "csrc")
out["kernel_sources_sha"] = bench._sources_sha(bench.step_kernel_sources())
json.dump(out, open(os.path.join(ROOT, "profiles", out_name), "w"), indent=1)
