"""GPU box (one GPU): the RCCL leg of the sample-sharded exchange with a world of ONE rank -- the collective calls, the
ExternalStream wrapping of a foreign HIP stream and the in-place all_gather_into_tensor exactly as popdel_amd.distributed
issues them with N ranks (RCCL refuses two ranks on one device, so the multi-rank data path is covered over gloo in
tests/test_gpu_sample_shard.py and this covers the backend-specific calls)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch
import torch.distributed as dist
from popdel_amd.distributed import SampleShardExchange

torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
x = SampleShardExchange(0, 1, torch.device("cuda", 0))
assert x.direct, "expected the direct RCCL path"
ptr = x.reserve(1 << 20)
assert ptr
foreign = torch.cuda.Stream()                      # stands in for the library's own hipStream_t
with torch.cuda.stream(foreign):
    x.arena[:4096] = torch.arange(4096, device="cuda", dtype=torch.int32).to(torch.uint8)[:4096]
for k in range(50):
    rc = x.all_gather(256 * k, 256, foreign.cuda_stream)
    assert rc == 0, x.last_error
foreign.synchronize()
t = torch.zeros(1, device="cuda")
dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier()
print("rccl smoke ok: %d all_gather_into_tensor calls on an external stream, %d bytes; backend %s" % (x.calls, x.bytes, dist.get_backend()))
dist.destroy_process_group()
