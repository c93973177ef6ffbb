#!/usr/bin/env python3
"""`popdel call` at cohort scale under several settings of the environment, on ONE generated cohort: steady-state seconds per
segment and the host phases behind it (POPDEL_TIMING's "steady state" line), output compared between the variants.
usage: e2e_cohort_probe.py [samples=4000] [mb=2] [VAR=value,VAR=value ...]   (each argument after mb: one variant; "-" = none)
-> gpurun_out/e2e_cohort_probe.json"""
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import bench
    from popdel_amd import synth
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
    mb = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    variants = sys.argv[3:] or ["-"]
    region = (1000000, 1000000 + mb * 1000000)
    d = tempfile.mkdtemp(prefix="popdel_probe_")
    rows, first = [], None
    try:
        t0 = time.perf_counter()
        lst = synth.write_cohort(d, n, region=region, deletions=synth.single_deletion, workers=bench.usable_cores())
        t_write = time.perf_counter() - t0
        for v in variants:
            env = dict(os.environ, POPDEL_TIMING="1")
            env.pop("POPDEL_HIP_LANES", None)
            binary = "popdel"
            if v != "-":
                for kv in v.split(","):
                    k, val = kv.split("=", 1)
                    if k == "BINARY":           # (another build of the host side in popdel_amd/lib: measurements)
                        binary = val
                    else:
                        env[k] = val
            product = os.path.join(ROOT, "popdel_amd", "lib", binary)
            out = os.path.join(d, "out.vcf")
            q = time.perf_counter()
            r = subprocess.run([product, "call", lst, "-o", out, "-r", "chr20:%d-%d" % region], env=env, capture_output=True, text=True, timeout=900)
            wall = time.perf_counter() - q
            if r.returncode != 0:
                rows.append(dict(variant=v, error=(r.stderr or r.stdout)[-400:]))
                print(json.dumps(rows[-1]), flush=True)
                continue
            text = "".join(l for l in open(out) if not l.startswith("##fileDate"))
            if first is None:
                first = text
            m = re.search(r"steady state: segments (\d+) of (\d+); windows (\d+); samples (\d+); wall ([0-9.e+-]+) s; decode ([0-9.e+-]+) s; "
                          r"replay ([0-9.e+-]+) s; device kernels ([0-9.e+-]+) s", r.stderr)
            cpu = re.search(r"CPU time ([0-9.e+-]+) s user, ([0-9.e+-]+) s system", r.stderr)
            row = dict(variant=v, wall_seconds=round(wall, 3), same_vcf_as_first=text == first,
                       cpu_seconds=(float(cpu.group(1)) + float(cpu.group(2))) if cpu else None,
                       cpu_system_seconds=float(cpu.group(2)) if cpu else None)
            if m:
                segs, _, windows, samples = (int(x) for x in m.groups()[:4])
                t_wall, t_dec, t_rep, t_dev = (float(x) for x in m.groups()[4:])
                row.update(steady_seconds_per_segment=round(t_wall / segs, 4), value=windows * samples / t_wall,
                           decode_per_segment=round(t_dec / segs, 4), replay_per_segment=round(t_rep / segs, 4),
                           device_kernels_per_segment=round(t_dev / segs, 5), steady_segments=segs)
            mw = re.search(r"warmed rounds: segments (\d+); windows (\d+); samples (\d+); wall ([0-9.e+-]+) s; decode ([0-9.e+-]+) s; replay ([0-9.e+-]+) s", r.stderr)
            if mw:
                row.update(warmed_segments=int(mw.group(1)), warmed_seconds_per_segment=round(float(mw.group(4)) / int(mw.group(1)), 4),
                           warmed_value=int(mw.group(2)) * int(mw.group(3)) / float(mw.group(4)))
            ph = [l for l in r.stderr.splitlines() if "[popdel timing] threads" in l]
            if ph:
                row["phases"] = {k.strip().replace(" ", "_"): float(v) for k, v in re.findall(r"([a-z][a-z+ ]*?) ([0-9][0-9.e+-]*) s", ph[-1])}
            extra = [l for l in r.stderr.splitlines() if "[popdel timing] loader" in l]
            trace = [l[len("[popdel loader] "):] for l in r.stderr.splitlines() if l.startswith("[popdel loader]")]
            if trace:
                row["loader_trace"] = trace
            timeline = [l for l in r.stderr.splitlines() if l.startswith("[popdel timeline]")]
            if timeline:            # (POPDEL_TIMING=2 among the variant's settings)
                os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
                with open(os.path.join(ROOT, "gpurun_out", "e2e_timeline_%d.txt" % len(rows)), "w") as f:
                    f.write("\n".join(timeline) + "\n")
            if extra:
                row["loader"] = extra[-1][len("[popdel timing] "):]
            filing = [l for l in r.stderr.splitlines() if "[popdel timing] filing" in l]
            if filing:
                row["filing"] = filing[-1][len("[popdel timing] "):]
            mc = re.search(r"warmed rounds: .*CPU ([0-9.e+-]+) s", r.stderr)
            if mc and mw:
                row["warmed_cpu_seconds_per_segment"] = round(float(mc.group(1)) / int(mw.group(1)), 4)
            rows.append(row)
            print(json.dumps(row), flush=True)
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "e2e_cohort_probe.json"), "w") as f:
            json.dump(dict(samples=n, region=region, cohort_written_in_seconds=t_write, usable_cores=bench.usable_cores(), rows=rows), f, indent=1)
    finally:
        shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()
