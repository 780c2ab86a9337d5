"""Condensed timeline of one graph-replayed train step from a rocprofv3 --kernel-trace csv: runs of small kernels (< 20 us) are
folded into one line (count, wall span, summed kernel time), big kernels are listed.  Run on the GPU box via tools/trace_gaps.sh."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f)))
adam = [i for i, r in enumerate(rows) if "adam_kernel" in r[2]]
groups = []
for i in adam:
    if groups and i - groups[-1][-1] <= 3: groups[-1].append(i)
    else: groups.append([i])
best = None
for g0, g1 in zip(groups[:-1], groups[1:]):
    a, b = g0[-1] + 1, g1[0]
    sp = rows[b - 1][1] - rows[a][0]
    if best is None or sp < best[0]: best = (sp, a, b)
_, a, b = best
step = rows[a:b]
t0 = step[0][0]
run = []
import collections
def flush():
    if run:
        span = run[-1][1] - run[0][0]
        if len(run) >= 40:
            c = collections.Counter(n.replace("void at::native::", "").replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")[:70] for _, _, n in run)
            for k, v in c.most_common(14):
                print(f"                 {v:4d} x {k}")
        print(f"{(run[0][0]-t0)/1e3:9.1f} us  [{len(run):4d} small kernels]  wall {span/1e3:8.1f} us  busy {sum(e-s for s,e,_ in run)/1e3:8.1f} us")
        run.clear()
tot_small_wall = 0
for s, e, n in step:
    if e - s < 20000:
        run.append((s, e, n))
    else:
        if run: tot_small_wall += run[-1][1] - run[0][0]
        flush()
        short = n.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")[:60]
        print(f"{(s-t0)/1e3:9.1f} us  {(e-s)/1e3:8.1f} us  {short}")
if run: tot_small_wall += run[-1][1] - run[0][0]
flush()
if len(sys.argv) >= 4:  # every kernel whose start lies in [argv[2], argv[3]] us of the step
    lo, hi = float(sys.argv[2]) * 1e3, float(sys.argv[3]) * 1e3
    for s, e, n in step:
        if lo <= s - t0 <= hi:
            print(f"  {(s-t0)/1e3:9.1f} {(e-s)/1e3:7.1f}  " + n.replace("void at::native::", "").replace("(anonymous namespace)::", "")[:150])
print("total wall inside small-kernel runs: %.1f us of %.1f" % (tot_small_wall / 1e3, (step[-1][1] - t0) / 1e3))
