"""One warmed round of `popdel call` from rocprofv3's kernel_trace.csv + memory_copy_trace.csv: every copy and kernel between two
submissions of the window walk (pd::walk* kernels mark a round), with start (ms from the round's begin), duration, bytes / rate.
usage: loader_timeline.py <kernel_trace.csv> <memory_copy_trace.csv> [rounds back from the end = 4]"""
import csv
import sys

k = list(csv.DictReader(open(sys.argv[1])))
m = list(csv.DictReader(open(sys.argv[2])))
back = int(sys.argv[3]) if len(sys.argv) > 3 else 4
ev = []
for r in k:
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K", r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0][:44], 0,
               r.get("Stream_Id", r.get("Queue_Id", ""))))
for r in m:
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C", r.get("Direction", r.get("Name", "?"))[:44], int(r.get("Size", r.get("Bytes", 0)) or 0),
               r.get("Stream_Id", "")))
ev.sort()
marks = [e[0] for e in ev if e[2] == "K" and ("stage1Kernel" in e[3] or "screenKernel" in e[3])]
if not marks:
    print(sorted(set(e[3] for e in ev))[:60])
    sys.exit(0)
# rounds: gaps of more than 5 ms between scan kernels
rounds = [marks[0]]
for a, b in zip(marks, marks[1:]):
    if b - a > 5e6:
        rounds.append(b)
if len(rounds) < back + 1:
    back = len(rounds) - 1
t0, t1 = rounds[-back - 1], rounds[-back]
print("round of %.2f ms (%d rounds seen)" % ((t1 - t0) / 1e6, len(rounds)))
tot = {}
for s, e, kind, name, size, stream in ev:
    if s < t0 or s >= t1:
        continue
    key = (kind, name)
    d = tot.setdefault(key, [0, 0.0, 0])
    d[0] += 1
    d[1] += (e - s) / 1e6
    d[2] += size
print("%-46s %6s %10s %12s" % ("what", "n", "ms", "MB"))
for (kind, name), (n, ms, size) in sorted(tot.items(), key=lambda x: -x[1][1]):
    print("%s %-44s %6d %10.3f %12.2f%s" % (kind, name, n, ms, size / 1e6, ("  %.1f GB/s" % (size / ms / 1e6)) if size and ms else ""))
print()
print("copies and kernels longer than 0.15 ms, in order:")
for s, e, kind, name, size, stream in ev:
    if s < t0 or s >= t1 or e - s < 150000:
        continue
    print("%9.3f %8.3f %s %-44s %10.2f MB %s" % ((s - t0) / 1e6, (e - s) / 1e6, kind, name, size / 1e6, stream))
