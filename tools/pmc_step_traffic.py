"""whole-step HBM traffic by kernel from the two PMC passes of tools/pmc_bench.sh (gpurun_out/pmc_bench_{FETCH,WRITE}_SIZE) ->
profiles/<name>.json (bench.pmc_traffic does the arithmetic: 2 x FETCH_SIZE + WRITE_SIZE, KiB counters)"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
iters = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
out_name = sys.argv[2] if len(sys.argv) > 2 else bench.PMC_TRAFFIC_FILE
out = bench.pmc_traffic(os.path.join(ROOT, "gpurun_out", "pmc_bench_FETCH_SIZE"), os.path.join(ROOT, "gpurun_out", "pmc_bench_WRITE_SIZE"), iters)
print(f"whole step: {out['whole_step_GB']:.2f} GB/iteration (fetch x2 {out['fetch_GB']:.2f} + write {out['write_GB']:.2f})")
for k, v in out["GB_per_iteration"].items():
    print(f"{k:72s} {out['launches_per_iteration'].get(k, 0):6.1f} launches/it {v:8.2f} GB/it")
out["command"] = "tools/pmc_bench.sh: rocprofv3 --kernel-trace --pmc <FETCH_SIZE|WRITE_SIZE> -- python3 bench.py --no-spawn --steps 2 --warmup 1 --no-cpu-baseline --no-graph --no-exact-f32 --no-extra-configs (two passes)"
out["units"] = "bytes = 1024 * (2 * FETCH_SIZE + WRITE_SIZE): counters are KiB; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half the bytes of wide coalesced reads)"
out["kernel_sources_sha"] = bench._sources_sha(bench.step_kernel_sources())
json.dump(out, open(os.path.join(ROOT, "profiles", out_name), "w"), indent=1)
