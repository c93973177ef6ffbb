import os, sys
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import ORACLE_LIB
from test_gpu_fuzz import random_case
from popdel_amd import capi
seed = int(sys.argv[1])
seg, rgs, offs, opts = random_case(seed)
print("opts", opts, "n", len(offs) - 1, "nrg", len(rgs), "W", seg.n_windows)
res = []
for lib in (capi.load_hip(), capi.load_library(ORACLE_LIB)):
    ctx = capi.Context(lib, rgs, offs, **opts)
    res.append(ctx.collect(ctx.submit(seg)))
g, o = res
lr_g, lr_o = np.asarray(g["likelihood_ratio"]), np.asarray(o["likelihood_ratio"])
rel = np.abs(lr_g - lr_o) / np.maximum(1, np.abs(lr_o))
for c in np.argsort(rel)[-3:]:
    print("call", c, "win", g["window_position"][c], "LR gpu %.17g cpu %.17g rel %.3g" % (lr_g[c], lr_o[c], rel[c]), "AF", g["frequency"][c], o["frequency"][c],
          "len", g["deletion_length"][c], o["deletion_length"][c], "it", g["iterations"][c], o["iterations"][c], "init", g["initial_length"][c], "filter", g["filter"][c], o["filter"][c])
