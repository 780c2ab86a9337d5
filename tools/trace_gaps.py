"""Idle time between kernels of the graph-replayed train steps, from a rocprofv3 --kernel-trace csv (run on the GPU box).
Prints, for the last replayed step: wall span, summed kernel time, union busy time (kernels overlap across streams), idle time,
and the idle time attributed to the kernel that FOLLOWS each gap (grouped by kernel name)."""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
# steps are delimited by the adam kernels (5 per step, outside the graph)
adam_idx = [i for i, r in enumerate(rows) if "adam_kernel" in r[2]]
# group consecutive adam launches
groups = []
for i in adam_idx:
    if groups and i - groups[-1][-1] <= 3:
        groups[-1].append(i)
    else:
        groups.append([i])
print("adam groups:", len(groups))
best = None
for g0, g1 in zip(groups[:-1], groups[1:]):
    a, b = g0[-1] + 1, g1[0]
    sp = rows[b - 1][1] - rows[a][0]
    print(f"  between adam groups: {b - a} kernels, span {sp / 1e6:.3f} ms")
    if best is None or sp < best[0]:
        best = (sp, a, b)
_, a, b = best
step = rows[a:b]
span = step[-1][1] - step[0][0]
ksum = sum(e - s for s, e, _ in step)
busy, cur_s, cur_e = 0, step[0][0], step[0][1]
gaps = collections.Counter(); gapn = collections.Counter()
for s, e, n in step[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps[n[:70]] += s - cur_e; gapn[n[:70]] += 1
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print(f"kernels {len(step)}  span {span/1e6:.3f} ms  kernel-sum {ksum/1e6:.3f} ms  busy(union) {busy/1e6:.3f} ms  idle {(span-busy)/1e6:.3f} ms")
short = [(e - s) for s, e, n in step if e - s < 10000]
print(f"kernels < 10 us: {len(short)}, {sum(short)/1e6:.3f} ms")
print("idle attributed to the following kernel:")
for n, t in gaps.most_common(25):
    print(f"  {t/1e3:9.1f} us  {gapn[n]:5d} gaps  avg {t/gapn[n]/1e3:6.2f} us  {n}")
native = collections.Counter()
for s, e, n in step:
    if "at::native" in n or "rocclr" in n or "at::cuda" in n:
        key = "copyBuffer" if "rocclr" in n else n.split("at::native::")[-1][:90]
        native[key] += 1
print(f"torch-native / copy launches in the step: {sum(native.values())} of {len(step)}")
for n, c in native.most_common(12):
    print(f"  {c:4d}  {n}")
