"""Concurrency of a context's lanes from a rocprofv3 kernel_trace.csv: over the last <ms> milliseconds of the trace, how long
the chip ran no / one / several kernels, the sum of the kernels' durations against the wall time, and a time line of the
first <rows> kernels of that window with the queue (= lane stream) each ran on.
usage: python tools/overlap_timeline.py <kernel_trace.csv> [ms=20] [rows=140]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ms = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
n_rows = int(sys.argv[3]) if len(sys.argv) > 3 else 140
rows = [r for r in rows if r["Kernel_Name"].startswith(("pd::", "void pd::"))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
end = max(int(r["End_Timestamp"]) for r in rows)
t0 = end - int(ms * 1e6)
win = [r for r in rows if int(r["Start_Timestamp"]) >= t0]
events = []
for r in win:
    events.append((int(r["Start_Timestamp"]), 1))
    events.append((int(r["End_Timestamp"]), -1))
events.sort()
level, last, at = 0, t0, {0: 0, 1: 0, 2: 0}
for t, d in events:
    at[min(level, 2)] += t - last
    last = t
    level += d
total = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in win)
wall = end - t0
queues = sorted({r.get("Queue_Id", "?") for r in win})
scans = [r for r in win if ("screenKernel" in r["Kernel_Name"] or "stage1Kernel" in r["Kernel_Name"])]
print("window: last %.1f ms of the trace, %d kernels on %d queues (%s), %d scans -> %.3f ms per scan" % (
    ms, len(win), len(queues), ", ".join(queues), len(scans), wall / 1e6 / max(1, len(scans))))
print("chip idle %.1f %%, one kernel %.1f %%, two or more kernels %.1f %% of the window; kernel durations add up to %.2f x the wall time" % (
    100.0 * at[0] / wall, 100.0 * at[1] / wall, 100.0 * at[2] / wall, total / wall))
print("\n   start us   dur us  queue  kernel")
for r in win[:n_rows]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("void ", "").split("(")[0][:34]
    print("%10.1f %8.1f  %5s  %s" % ((s - t0) / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?"), name))
