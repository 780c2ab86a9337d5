"""whole-step HBM traffic by kernel from the two PMC passes of tools/pmc_bench.sh (eager, 1 warm-up + 2 timed + 1 roofline
iteration = 4 iterations): 2 x FETCH_SIZE + WRITE_SIZE (KiB counters; FETCH_SIZE doubled per MI355X_MICROARCH.md)"""
import csv, glob, os, re, sys, collections, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
iters = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
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
json.dump({"GB_per_iteration": total / 1e9, "iterations": iters, "top": [(k, b / iters / 1e9) for b, k, v in rows[:25]]},
          open(os.path.join(ROOT, "gpurun_out", "pmc_step_traffic.json"), "w"), indent=1)
