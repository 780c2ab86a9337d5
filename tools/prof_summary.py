"""print a per-iteration summary of a rocprofv3 kernel-stats CSV: python tools/prof_summary.py <kernel_stats.csv> <iterations>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
iters = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
tot = sum(float(r['TotalDurationNs']) for r in rows)
native = sum(int(r['Calls']) for r in rows if 'at::native' in r['Name'] or 'rocclr' in r['Name'])
print(f"total {tot/1e6/iters:.3f} ms/iter, launches/iter {sum(int(r['Calls']) for r in rows)/iters:.0f}, torch-native+copy launches/iter {native/iters:.0f}")
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 24]:
    print(f"{r['Name'][:120]:120s} {int(r['Calls'])/iters:7.1f}/it {float(r['TotalDurationNs'])/1e6/iters:8.3f} ms/it {float(r['AverageNs'])/1e3:9.1f} us")
