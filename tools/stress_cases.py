"""Random cohorts for the self-consistency stress (tools/stress_consistency.py) and the sharded stress cases of
tests/shard_worker.py."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from popdel_amd import capi, synth


def case(seed):
    rng = np.random.default_rng(4242 + seed)
    lib = synth.Library(median=int(rng.integers(330, 600)), stddev=float(rng.uniform(20, 70)), read_len=int(rng.choice([100, 150])))
    base = synth.conditioned_readgroup(lib)
    n = int(rng.integers(300, 1500))
    W = int(rng.integers(800, 2500))
    first = 1000020
    region = (first - 800, first + 30 * W)
    dels = [(int(rng.integers(region[0] + 3000, region[1] - 9000)), int(np.exp(rng.uniform(np.log(300), np.log(5000)))),
             rng.binomial(2, float(rng.beta(0.7, 2.0)), size=n)) for _ in range(int(rng.integers(1, 4)))]
    parts, offsets, rgs, sample_offsets = [], [0], [], [0]
    for k in range(n):
        cov = float(rng.choice([30.0, 30.0, 15.0, 50.0, 70.0]))
        pos, dev = synth.sample_pairs(70000 * seed + k, region,
                                      synth.Library(median=lib.median, stddev=lib.stddev, read_len=lib.read_len, coverage=cov),
                                      [synth.Deletion(s, l, int(g[k])) for s, l, g in dels])
        d, s_, a, b = synth.instances_by_interval_rule(pos, dev, lib, first, W)
        R = int(rng.choice([1, 1, 1, 2, 3]))
        which = rng.integers(0, R, size=len(d))
        for r in range(R):
            sel = which == r
            parts.append((d[sel], s_[sel], a[sel], b[sel]))
            offsets.append(offsets[-1] + int(sel.sum()))
            rgs.append(base)
        sample_offsets.append(len(rgs))
    seg = capi.Segment(first, W, np.array(offsets, np.uint64), np.concatenate([p[0] for p in parts]).astype(np.int32),
                       np.concatenate([p[1] for p in parts]).astype(np.uint32), np.concatenate([p[2] for p in parts]).astype(np.uint32),
                       np.concatenate([p[3] for p in parts]).astype(np.uint32))
    return seg, rgs, sample_offsets, dict(window_wise_output=bool(rng.integers(0, 2)))
