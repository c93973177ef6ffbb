"""GPU: bench.py prints exactly one JSON line with the fields the round driver reads."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_bench_line_has_the_contract_fields(built):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--samples", "512", "--passes", "3", "--steps", "2",
                          "--warmup", "1", "--cpu-samples", "40", "--cpu-threads", "2", "--stream-segments", "2", "--small-cohort", "128",
                          "--dense-deletions", "3", "--e2e-samples", "24", "--sharded-world1", "256,0", "--distinct-segments", "3",
                          "--resident-segments", "3", "--e2e-cohort-samples", "32", "--e2e-cohort-mb", "1", "--inflate-members", "600",
                          "--stream50k-samples", "768", "--stream50k-segments", "2"],
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["vs_baseline"] is None and d["data"] == "synthetic" and "workload" in d["config"]
    assert d["value"] > 0 and d["ms_per_step"] > 0
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    c = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in c, key
    assert c["kind"] == "port" and c["cores"] == 1 and c["value"] > 0
    # value is cells per second of the timed region
    cells = d["config"]["windows_per_step"] * d["config"]["samples"]
    assert abs(d["value"] - cells * d["steps"] / (d["ms_per_step"] * d["steps"] / 1e3)) / d["value"] < 1e-6
    # secondary legs are reported beside the headline, never instead of it
    assert d["contexts_per_gpu"] == 1
    cs = d["concurrent_segments"]
    assert cs["contexts_per_gpu"] == 3 and cs["value"] > 0 and cs["unit"] == d["unit"]
    assert d["exchange"]["all_gathers_per_pass"] == 0
    assert d["small_cohort"]["value"] > 0 and d["candidate_dense"]["candidate_cells_per_pass"] > 0
    assert c["all_cores"]["cores"] == 2 and c["cpu_model"]
    assert r["limiter"] == "hbm" and r["scan_source_stamp"] and r["kernel"] == "screenKernel" and 0 < r["frac"] < 1
    assert r["contract_frac"] > r["frac"] and r["algorithmic_bytes_per_launch"] < r["contract_bytes_per_launch"]
    # the whole path against its own roofline, next to the scan's
    assert 0 < d["whole_path_frac"] < 1 and abs(d["whole_path_frac"] - d["value"] * d["algorithmic_bytes_per_unit"] / 8e12) < 1e-9
    # the sample-sharded code path with one rank, costed per pass next to the unsharded context
    rows = d["sharded_world1"]["rows"]
    assert [x["samples"] for x in rows] == [256, 512]
    for x in rows:
        assert x["all_gathers_per_pass"] >= 9 and x["exchange_ms_per_pass"] > 0 and x["sharded_ms_per_pass"] > 0 and x["unsharded_ms_per_pass"] > 0
        # ... and with the wire time of an exchange injected behind every one of them (the latency model of several GPUs)
        assert x["modelled_40us_ms_per_pass"] > x["sharded_ms_per_pass"] and 0 < x["modelled_efficiency"] == x["modelled_40us_efficiency"]
        assert x["modelled_efficiency_vs_plain_context"] > 0 and x["unsharded_plain_ms_per_pass"] > 0
        assert x["modelled_20us_ms_per_pass"] > 0 and x["modelled_10us_ms_per_pass"] > 0 and x["lanes"] == d["lanes_per_context"]
    # the headline's passes take several different resident segments in turn (the third of five has no deletion); the
    # single re-scanned segment of rounds 1-3 and the context without lanes are legs
    assert d["lanes_per_context"] == 3 and d["tickets_in_flight"] == 6
    assert d["config"]["candidate_cells_per_segment"][2] == 0 and len(d["config"]["planted_deletion"]) == 3
    assert d["one_segment"]["value"] > 0 and 0 < d["one_segment"]["whole_path_frac"] < 1
    assert d["plain_context"]["value"] > 0
    assert d["roofline"]["mean_launch_ms"] > 0 and d["roofline"]["mean_launch_ms_in_timed_region"] > 0
    ds = d["distinct_segments"]
    assert ds["value"] > 0 and len(ds["segments"]) == 3 and ds["segments"][-1]["del_len"] is None
    # segments streamed from host memory: the leg says what the staging cost on the host and on the link, and what the host's
    # memory system gave in the same run
    st = d["streaming"]
    assert st["value"] > 0 and st["pack_seconds_per_segment"] > 0 and st["uploaded_seconds_per_segment"] >= st["pack_seconds_per_segment"] * 0.5
    assert st["bytes_per_instance_on_the_link"] == 5 and st["h2d_bytes_per_segment"] > 0 and len(st["host_copy_probe_GBs"]) == 2
    assert st["cold_stage_seconds_per_segment"] > 0 and st["stage_call_seconds_per_segment"] >= st["counted_seconds_per_segment"] > 0
    assert "hot path only" in c["scope"]
    assert min(st["host_copy_probe_GBs"]) > 0.1
    s5 = d["streaming_50000"]
    assert s5["samples"] == 768 and s5["segments"] == 2 and s5["value"] > 0 and s5["pair_instances_per_segment"] > 0
    # gunzip of profile blocks on the device against the host's libdeflate path: same bytes
    g = d["device_gunzip"]
    assert g["members"] == 600 and g["all_accepted"] and g["identical_to_host"] and g["kernel_ms"] > 0 and g["host_seconds"] > 0
    # the cohort-scale end-to-end leg (here a toy cohort: over before the steady state starts, or with its figures)
    ec = d["e2e_cohort"]
    assert "error" not in ec and ec["whole_run_wall_seconds"] > 0 and (ec["value"] is None or ec["steady_seconds_per_segment"] > 0)
    # end to end: the product binary on profile files, the oracle-backed driver beside it, same VCF
    e = d["e2e"]
    assert e["value"] > 0 and e["wall_seconds"] > 0 and "decode" in e["phases_seconds"]
    assert e["cpu_baseline"]["identical_vcf"] is True and e["cpu_baseline"]["one_thread"]["cores"] == 1
    assert e["cpu_baseline"]["all_cores"]["value"] > 0


def test_gpus_2_spawns_two_ranks_and_reports_the_sample_split(built):
    """`bench.py --gpus 2` with no launcher around it starts two ranks itself; here both share the box's one GPU and
    exchange over gloo (RCCL refuses two ranks on one device). The headline is the cohort sharded by samples."""
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-gpu", "--backend", "gloo",
                          "--samples", "512", "--passes", "3", "--steps", "2", "--warmup", "1", "--configs3-samples", "768",
                          "--configs4-samples", "1280", "--configs-passes", "3", "--configs4-segments", "3"],
                         capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "weak" and "headline_fallback" not in d
    assert d["config"]["samples"] == 1024 and d["config"]["samples_per_gpu"] == 512
    assert d["exchange"]["all_gathers_per_pass"] >= 10 and d["exchange"]["backend"] == "gloo"
    cells = d["config"]["windows_per_step"] * d["config"]["samples"]
    assert abs(d["value"] - cells / (d["ms_per_step"] / 1e3)) / d["value"] < 1e-6
    assert d["segment_sharded"]["value"] > 0 and d["segment_sharded"]["all_gathers_per_pass"] == 0
    assert d["calls_per_step"] > 0
    assert d["exchange"]["exchange_ms_per_pass"] > 0
    # the shapes BASELINE names for several GPUs: a cohort of FIXED total size, resident and streamed
    c3, c4 = d["configs3"], d["configs4"]
    assert c3["samples"] == 768 and c3["samples_per_rank"] == 512 and c3["samples_last_rank"] == 256 and c3["passes"] == 3
    assert c3["value"] > 0 and c3["all_gathers_per_pass"] >= 9 and c3["exchange_ms_per_pass"] > 0 and c3["scaling"] == "strong"
    assert c4["samples"] == 1280 and c4["samples_per_rank"] == 768 and c4["segments"] == 3
    assert c4["value"] > 0 and c4["all_gathers_per_pass"] >= 9 and c4["exchange_ms_per_pass"] > 0


def test_gpus_mismatch_fails_loudly(built):
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True,
                         timeout=300, env=env)
    assert out.returncode != 0 and "refusing" in out.stderr
