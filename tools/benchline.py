"""Pick the interesting numbers out of a bench.py JSON line on stdin: python bench.py ... | python tools/benchline.py [passes]"""
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
p = d["config"]["segment_passes_per_step"]
print("value %.4g  ms/pass %.3f  stage1 %.3f  stage2 %.3f  frac %.3f  pairs/cell %.1f  K %d  calls/pass %.0f" % (
    d["value"], d["ms_per_step"] / p, d["stage1_ms_per_step"] / p, d["stage2_ms_per_step"] / p, d["roofline"]["frac"],
    d["roofline"]["mean_active_pairs_per_cell"], d["max_life"], d["calls_per_step"] / p))
