"""Post-process a rocprofv3 --kernel-trace CSV of scripts/train_bench.py: the LAST training step (between the last two k_adam
launches) as a timeline -- per kernel name: calls, total, and the idle gap in front of it; then the launches in order."""
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"\(.*", "", n)
    return n.replace("mdg::", "")[:44]
adam = [i for i, r in enumerate(rows) if "k_adam" in r["Kernel_Name"]]
lo, hi = adam[-2] + 1, adam[-1] + 1
step = rows[lo:hi]
t0 = int(step[0]["Start_Timestamp"]); t1 = int(step[-1]["End_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in step)
print(f"last step: {len(step)} launches, span {(t1 - t0) / 1e6:.2f} ms, kernel time {busy / 1e6:.2f} ms")
qkey = "Queue_Id" if "Queue_Id" in step[0] else None
if qkey:
    per_q = collections.Counter()
    for r in step:
        per_q[r[qkey]] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    print("kernel time per queue (ms):", {k: round(v / 1e6, 2) for k, v in per_q.items()})
agg = collections.OrderedDict()
prev_end = t0
for r in step:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    a = agg.setdefault(short(r["Kernel_Name"]), [0, 0, 0])
    a[0] += 1; a[1] += e - s; a[2] += max(0, s - prev_end)
    prev_end = max(prev_end, e)
print(f"{'kernel':46s} {'calls':>5s} {'ms':>8s} {'avg us':>8s} {'gap ms':>8s}")
for k, a in sorted(agg.items(), key=lambda kv: -(kv[1][1] + kv[1][2])):
    print(f"{k:46s} {a[0]:5d} {a[1] / 1e6:8.3f} {a[1] / a[0] / 1e3:8.1f} {a[2] / 1e6:8.3f}")
print("total gap ms", sum(a[2] for a in agg.values()) / 1e6)
if len(sys.argv) > 2:
    prev_end = t0
    for i, r in enumerate(step):
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        g = r.get("Grid_Size", r.get("Grid_Size_X", "?"))
        print(f"{i:4d} q{r.get('Queue_Id', '?'):>2s} +{(s - t0) / 1e3:9.1f} us  gap {max(0, s - prev_end) / 1e3:6.1f}  dur {(e - s) / 1e3:7.1f}  grid {g:>9s}  {short(r['Kernel_Name'])}")
        prev_end = max(prev_end, e)
