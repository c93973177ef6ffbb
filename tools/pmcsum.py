"""Per-kernel means of the rocprofv3 PMC passes of tools/profile_round.sh: python tools/pmcsum.py gpurun_out/prof_<tag> [json-out]
Prints one line per (kernel, counter): launches and mean value per launch; with a second argument also writes them as JSON."""
import collections, csv, glob, json, os, sys
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(os.path.join(root, "pmc_*", "**", "*counter_collection.csv"), recursive=True)):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0]
        if name.startswith("void "):
            name = name[5:]
        if name.startswith("pd::"):
            acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k in sorted(acc):
    out[k] = {c: {"launches": len(v), "mean": sum(v) / len(v), "max": max(v)} for c, v in sorted(acc[k].items())}
    print(k)
    for c, d in out[k].items():
        print("    %-34s launches %5d  mean %.6g" % (c, d["launches"], d["mean"]))
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=1)
