"""Debug helper (GPU box): does torch still find the GPU after libpopdel_hip.so has used it in this process?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from popdel_amd import capi, synth
lib = synth.Library(); rg = synth.conditioned_readgroup(lib)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1
for i in range(n):
    seg = synth.cohort_segment(6, (1000000, 1030000), 1000020, 900, lib)
    ctx = capi.Context(capi.load_hip(), [rg] * 6, list(range(7)))
    ctx.collect(ctx.submit(seg)); ctx.close()
print("contexts done:", n, flush=True)
import torch
print("torch sees", torch.cuda.device_count(), torch.cuda.is_available(), flush=True)
x = torch.ones(4, device="cuda"); print(float(x.sum()))
