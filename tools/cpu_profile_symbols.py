#!/usr/bin/env python3
"""Function names for a POPDEL_CPU_PROFILE file (popdel_amd/csrc/host/cpu_sampler.h): per module the symbols of `nm -C -n`
(this repo's binaries by file name, system libraries by path if they exist here), samples summed per function.
usage: cpu_profile_symbols.py <profile> [top=40]"""
import bisect, collections, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def symbols(path):
    here = os.path.join(ROOT, "popdel_amd", "lib", os.path.basename(path))
    use = here if os.path.exists(here) else path
    if not os.path.exists(use):
        return None
    out = subprocess.run(["nm", "-C", "-n", "--defined-only", use], capture_output=True, text=True).stdout
    if not out.strip():
        out = subprocess.run(["nm", "-C", "-n", "-D", "--defined-only", use], capture_output=True, text=True).stdout
    table = []
    for line in out.splitlines():
        parts = line.split(None, 2)
        if len(parts) == 3 and parts[1].lower() in "tw":
            table.append((int(parts[0], 16), parts[2]))
    table.sort()
    return table


def main():
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    per_fn, per_mod, total, cache = collections.Counter(), collections.Counter(), 0, {}
    for line in open(sys.argv[1]):
        if line.startswith("#"):
            continue
        n, module, off = line.split()
        n, off = int(n), int(off, 16)
        total += n
        per_mod[os.path.basename(module)] += n
        if module not in cache:
            cache[module] = symbols(module)
        table = cache[module]
        name = "?"
        if table:
            i = bisect.bisect_right(table, (off, "\xff")) - 1
            if i >= 0:
                name = table[i][1]
        per_fn[(os.path.basename(module), name[:110])] += n
    print("%d samples (ms of CPU time)" % total)
    for m, n in per_mod.most_common(12):
        print("  %6.1f %%  %s" % (100.0 * n / total, m))
    print()
    for (m, f), n in per_fn.most_common(top):
        print("%6.1f %%  %-24s %s" % (100.0 * n / total, m, f))


if __name__ == "__main__":
    main()
