"""Writes tests/golden/residual_<seed>.json (GPU box): python tools/make_residual_goldens.py [--out DIR] <seed> ...

For every seed the random cohort of tests/test_gpu_fuzz.py::random_case goes through the SHIPPED HIP library and through the
oracle (C ABI both); the record of every call that differs in any way (tests/residual.py) is the expectation that
tests/test_gpu_fuzz.py::test_pinned_residual asserts. Run it again -- and read the diff -- whenever a kernel change is MEANT
to move the residual. On the GPU box write under gpurun_out/ (--out gpurun_out/residual) and copy the files into tests/golden/."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main(argv):
    from residual import residual_record, summary
    from test_gpu_fuzz import random_case
    from test_gpu_parity import both
    out = os.path.join(ROOT, "tests", "golden")
    if argv and argv[0] == "--out":
        out, argv = argv[1], argv[2:]
    os.makedirs(out, exist_ok=True)
    for seed in [int(a) for a in argv]:
        seg, rgs, offs, opts = random_case(seed)
        gpu, cpu = both(seg, rgs, offs, **opts)
        rec = residual_record(seed, gpu, cpu)
        rec["options"] = {k: (bool(v) if isinstance(v, bool) else v) for k, v in opts.items()}
        with open(os.path.join(out, "residual_%d.json" % seed), "w") as fh:
            json.dump(rec, fh, indent=1, sort_keys=True)
            fh.write("\n")
        print(json.dumps(summary(rec)), flush=True)


if __name__ == "__main__":
    main(sys.argv[1:])
