import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import test_gpu_parity as T
import conftest
# re-create the ragged segment via the test body pieces
from popdel_amd import capi, synth
lib = synth.Library(); lowcov = synth.Library(coverage=1.0)
rg = synth.conditioned_readgroup(lib); rg_tight = synth.conditioned_readgroup(lib, max_load=20)
region, first, W = (1000000, 1090000), 1000020, 2900
dels = lambda k, r: [synth.Deletion(1040000, 2500, [2, 1, 0][k % 3]), synth.Deletion(1060000, 400, [1, 2, 2][k % 3])]
parts, offsets, rgs, sample_offsets = [], [0], [], [0]
specs = [("normal", lib, rg)] * 5 + [("empty", None, rg), ("low", lowcov, rg), ("tight", lib, rg_tight), ("multi", lib, rg), ("multi", lib, rg)]
seed = 4000
for kind, l, r in specs:
    n_rg = 3 if kind == "multi" else 1
    for j in range(n_rg):
        seed += 1
        if l is None:
            d = s = a = b = np.zeros(0, dtype=np.int64)
        else:
            sub = synth.Library(coverage=l.coverage / n_rg)
            pos, dev = synth.sample_pairs(seed, region, sub, dels(seed, region))
            d, s, a, b = synth.instances_by_interval_rule(pos, dev, lib, first, W)
        parts.append((d, s, a, b)); offsets.append(offsets[-1] + len(d)); rgs.append(r)
    sample_offsets.append(len(rgs))
seg = capi.Segment(first, W, np.array(offsets, np.uint64), np.concatenate([p[0] for p in parts]).astype(np.int32),
                   np.concatenate([p[1] for p in parts]).astype(np.uint32), np.concatenate([p[2] for p in parts]).astype(np.uint32),
                   np.concatenate([p[3] for p in parts]).astype(np.uint32))
gpu, cpu = T.both(seg, rgs, sample_offsets)
(gc, gq, gf), (oc, oq, of) = gpu, cpu
print("q3 equal", np.array_equal(gq, oq), "flags", np.array_equal(gf, of), "ncalls", gc["n_calls"], oc["n_calls"])
if gc["n_calls"] == oc["n_calls"]:
    for k in T.INT_KEYS:
        if not np.array_equal(gc[k], oc[k]):
            idx = np.argwhere(gc[k] != oc[k])
            print("MISMATCH", k, len(idx), idx[:8].tolist())
            for i in idx[:4]:
                print("  gpu", gc[k][tuple(i)], "cpu", oc[k][tuple(i)], "call win", oc["window_position"][i[0]], "len", oc["deletion_length"][i[0]])
    print("maxLR", np.abs(gc["likelihood_ratio"] - oc["likelihood_ratio"]).max())
else:
    print(gc["window_position"][:20], oc["window_position"][:20])
d = np.abs(gc["frequency"] - oc["frequency"]); i = int(d.argmax())
print("maxAF diff", d.max(), "at", i, gc["frequency"][i], oc["frequency"][i], "it", oc["iterations"][i], "LRs", gc["likelihood_ratio"][i], oc["likelihood_ratio"][i])
