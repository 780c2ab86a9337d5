"""Summarise the FETCH_SIZE / WRITE_SIZE passes of tools/pmc_bench.sh into profiles/r01_pmc_traffic.json
(HBM bytes per launch of the dominant kernel; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950).

    python tools/pmc_traffic_summary.py <kernel-name-substring> <precision-policy>
"""
import csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
needle, policy = sys.argv[1], sys.argv[2]
out = {"command": "rocprofv3 --kernel-trace --pmc <FETCH_SIZE|WRITE_SIZE> -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-graph (two separate passes)",
       "kernel": needle, "precision_policy": policy,
       "units": "counter values are KiB; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half the bytes of wide coalesced reads)"}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    files = glob.glob(os.path.join(ROOT, "gpurun_out", f"pmc_bench_{c}", "**", "*counter_collection.csv"), recursive=True)
    tot, n = 0.0, 0
    for f in files:
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == c and needle in row["Kernel_Name"]:
                tot += float(row["Counter_Value"]); n += 1
    out[f"{c}_KiB_avg_per_launch"] = tot / max(n, 1)
    out[f"{c}_launches"] = n
out["traffic_bytes_per_launch"] = 1024.0 * (2.0 * out["FETCH_SIZE_KiB_avg_per_launch"] + out["WRITE_SIZE_KiB_avg_per_launch"])
json.dump(out, open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
