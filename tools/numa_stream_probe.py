#!/usr/bin/env python3
"""Is the streaming leg's run-to-run spread (0.55 vs 1.8e9 w x s/s on the same box) a NUMA effect? Prints the box's nodes, the
GPU's node and this job's CPUs, then runs bench.py's streaming leg with the process bound to each node's CPUs in turn
(affinity set in the child before anything touches the GPU; memory follows first touch)."""
import glob, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def cpulist(s):
    out = []
    for part in s.strip().split(","):
        if "-" in part:
            a, b = part.split("-")
            out += list(range(int(a), int(b) + 1))
        elif part:
            out.append(int(part))
    return out


nodes = {}
for d in sorted(glob.glob("/sys/devices/system/node/node[0-9]*")):
    nodes[int(d.rsplit("node", 1)[1])] = cpulist(open(d + "/cpulist").read())
print("nodes:", {k: "%d cpus (%d..%d)" % (len(v), v[0], v[-1]) for k, v in nodes.items()})
for f in sorted(glob.glob("/sys/class/drm/card*/device/numa_node")) + sorted(glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties")):
    try:
        txt = open(f).read()
        if f.endswith("properties"):
            txt = " ".join(l for l in txt.splitlines() if l.split()[0] in ("simd_count", "domain", "location_id", "cpu_cores_count"))
        print(f, txt.strip())
    except OSError as e:
        print(f, e)
allowed = sorted(os.sched_getaffinity(0))
print("allowed cpus: %d (%s..%s)" % (len(allowed), allowed[0], allowed[-1]))
try:
    print("cpu.max:", open("/sys/fs/cgroup/cpu.max").read().strip())
except OSError:
    pass
cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu", "--e2e-samples", "0", "--e2e-cohort-samples", "0", "--inflate-members", "0",
       "--dense-deletions", "0", "--small-cohort", "0", "--sharded-world1", "", "--concurrent", "1", "--steps", "1", "--warmup", "0", "--passes", "40"]
for name, cpus in [("unbound", allowed)] + [("node %d" % k, sorted(set(v) & set(allowed))) for k, v in nodes.items()] + [("unbound", allowed)]:
    if not cpus:
        continue
    r = subprocess.run(cmd, capture_output=True, text=True, preexec_fn=lambda c=cpus: os.sched_setaffinity(0, c))
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if not line:
        print(name, "failed", r.stderr[-300:])
        continue
    d = json.loads(line[0])
    s = d["streaming"]
    print(name, "streaming %.3g w*s/s, stage %.3f s/segment, first stage %.3f s; resident staging %.3f s/segment; headline %.4g" % (
        s["value"], s["stage_seconds_per_segment"], s["first_stage_seconds"], d["staging_seconds_per_segment"], d["value"]), flush=True)
