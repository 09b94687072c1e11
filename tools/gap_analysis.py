"""Where is the GPU idle inside a clip fit?  (analysis tool)
    python tools/gap_analysis.py <rocprofv3 kernel_trace.csv> [tail_fraction]
Takes the last ``tail_fraction`` (default: everything after the longest pause = the timed fit after the warm-up fit) of
the kernel trace, and reports the busy time (union of the dispatch intervals), the idle time, and the idle gaps grouped
by the kernels on either side."""
import csv
import sys
from collections import defaultdict


def short(n):
    n = n.split("(")[0]
    n = n.replace("void gfl::", "").replace("gfl::", "")
    return n[-44:]


rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
rows.sort()
# the timed fit starts after the longest pause of the second half of the trace (clip synthesis / warm-up are before it)
if len(sys.argv) > 2:
    k0 = int(len(rows) * (1.0 - float(sys.argv[2])))
else:
    best, k0 = 0, 0
    for i in range(len(rows) // 4, len(rows) - 1):
        g = rows[i + 1][0] - rows[i][1]
        if g > best:
            best, k0 = g, i + 1
rows = rows[k0:]
t0, t1 = rows[0][0], max(r[1] for r in rows)
busy, cur_end = 0, rows[0][0]
gaps = []
for i, (a, b, n) in enumerate(rows):
    if a > cur_end:
        gaps.append((a - cur_end, rows[i - 1][2] if i else "-", n))
        busy += 0
        cur_start = a
    busy += max(0, b - max(a, cur_end))
    cur_end = max(cur_end, b)
span = t1 - t0
print(f"kernels {len(rows)}  span {span/1e6:.2f} ms  busy {busy/1e6:.2f} ms ({100*busy/span:.1f} %)  idle {(span-busy)/1e6:.2f} ms")
ktime = defaultdict(lambda: [0, 0])
for a, b, n in rows:
    ktime[n][0] += 1
    ktime[n][1] += b - a
print("-- kernel time")
for n, (c, t) in sorted(ktime.items(), key=lambda kv: -kv[1][1])[:24]:
    print(f"{n:46s} calls {c:6d} total {t/1e6:8.2f} ms avg {t/c/1e3:7.1f} us")
hist = defaultdict(lambda: [0, 0])
for g, p, n in gaps:
    b = "<2us" if g < 2000 else "<5us" if g < 5000 else "<20us" if g < 20000 else "<100us" if g < 100000 else "<1ms" if g < 1e6 else ">=1ms"
    hist[b][0] += 1
    hist[b][1] += g
print("-- idle gaps by size")
for b in ("<2us", "<5us", "<20us", "<100us", "<1ms", ">=1ms"):
    print(f"{b:8s} count {hist[b][0]:6d} total {hist[b][1]/1e6:8.2f} ms")
pair = defaultdict(lambda: [0, 0])
for g, p, n in gaps:
    pair[(p, n)][0] += 1
    pair[(p, n)][1] += g
print("-- idle gaps by (kernel before -> kernel after), top 25 by total")
for (p, n), (c, t) in sorted(pair.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"{p:40s} -> {n:40s} count {c:5d} total {t/1e6:7.2f} ms avg {t/c/1e3:7.1f} us")
