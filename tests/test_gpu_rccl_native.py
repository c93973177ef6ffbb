"""GPU (one device): the library's own RCCL exchange (pd_ctx_set_shard_rccl) with a communicator of ONE rank -- RCCL is loaded,
the communicator created, and every all-gather of the sample-sharded path issued on the context's stream (in place, a
no-op in data terms) -- against the plain context. RCCL refuses two ranks on one device; the multi-rank data path is
covered over gloo in test_gpu_sample_shard.py."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_native_rccl_exchange_with_one_rank(built):
    from popdel_amd import capi, synth
    from test_gpu_fuzz import random_case
    seg, rgs, offs, opts = random_case(7)
    hip = capi.load_hip()
    plain = capi.Context(hip, rgs, offs, **opts)
    a = plain.collect(plain.submit(seg))
    plain.close()
    ctx = capi.Context(hip, rgs, offs, **opts)
    ctx.set_shard_rccl(0, 1, 0, len(offs) - 1, [r.min_init_del_len for r in rgs], capi.rccl_unique_id(hip))
    b = ctx.collect(ctx.submit(seg))
    c = ctx.collect(ctx.submit(seg))
    ctx.close()
    assert a["n_calls"] > 0
    for k, v in a.items():
        if isinstance(v, np.ndarray):
            assert np.array_equal(v, b[k]) and np.array_equal(v, c[k]), k
