"""Per-pass summary of a rocprofv3 kernel_stats.csv: python tools/kstats.py <csv> <passes>"""
import csv, sys
passes = float(sys.argv[2])
rows = list(csv.DictReader(open(sys.argv[1])))
tot = 0
for r in rows:
    name = r["Name"].replace("(anonymous namespace)::", "").split("(")[0]
    if name.startswith("void "):                 # template kernels are printed with their return type
        name = name[5:]
    if not (name.startswith("pd::") or "copyBuffer" in name or "fillBuffer" in name or "inflateMembersKernel" in name or name.startswith("parse")):
        continue
    ms = float(r["TotalDurationNs"]) / 1e6 / passes
    tot += ms
    print("%-34s calls/pass %6.1f  avg %8.1f us  per pass %7.3f ms" % (name[:34], int(r["Calls"]) / passes, float(r["AverageNs"]) / 1e3, ms))
print("sum per pass %.3f ms" % tot)
