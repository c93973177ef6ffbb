"""Turns the output of tools/profile_round.sh (gpurun_out/prof_<tag>/) into the summaries committed under profiles/:
python tools/make_profile_summary.py <tag> <name> <passes in the trace>   ->  profiles/<name>_kernel_stats.{csv,md},
profiles/<name>_pmc_stage1.json (what bench.py quotes as roofline.traffic), profiles/<name>_pmc_kernels.md"""
import collections, csv, glob, json, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
tag, name, passes = sys.argv[1], sys.argv[2], float(sys.argv[3])
src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
dst = os.path.join(ROOT, "profiles")
stats = max(glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)   # (the newest run's)
shutil.copy(stats, os.path.join(dst, name + "_kernel_stats.csv"))
table = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kstats.py"), stats, str(passes)], capture_output=True, text=True).stdout
line = json.loads([l for l in open(os.path.join(src, "trace_bench.json")) if l.startswith("{")][0])
with open(os.path.join(dst, name + "_kernel_stats.md"), "w") as f:
    f.write("# %s: rocprofv3 --kernel-trace --stats of `bench.py --serial --lanes 1 --no-legs --no-cpu --steps 1 --warmup 1`\n\n" % name)
    f.write("(%d passes over the resident 10,000-sample segments in the trace -- warm-up, the timed step and the passes that size the segments and time the scan alone --, one ticket at a time on a context without lanes: see bench.py --serial.\n" % passes)
    f.write("Bench line of the traced run: value %.4g, stage 1 %.3f ms and stage 2 %.3f ms per pass by HIP events, frac %.3f.)\n\n```\n" % (
        line["value"], line["stage1_ms_per_step"] / line["config"]["segment_passes_per_step"],
        line["stage2_ms_per_step"] / line["config"]["segment_passes_per_step"], line["roofline"]["frac"]))
    f.write(table + "```\n")
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in sorted(glob.glob(os.path.join(src, "pmc_*", "**", "*counter_collection.csv"), recursive=True)):
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"].split("(")[0]
        k = k[5:] if k.startswith("void ") else k
        if k.startswith("pd::"):
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
mean = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}
SCAN = "pd::screenKernel" if "pd::screenKernel" in mean else "pd::stage1Kernel"
s1 = mean[SCAN]
cycles = s1["GRBM_GUI_ACTIVE"] / 8.0                     # summed over the 8 XCDs
fetch_raw, write = s1["FETCH_SIZE"] * 1024.0, s1["WRITE_SIZE"] * 1024.0
out = {
    "kernel": SCAN, "config": {"samples": 10000, "windows": 6667},
    "scan_source_stamp": bench.scan_source_stamp(),
    "command": "tools/profile_round.sh: rocprofv3 --pmc <one block's counters> --kernel-trace -- python3 bench.py --serial --lanes 1 --no-legs --no-cpu --passes 4 --steps 1 --warmup 0 (one pass per counter group, no other trace domains)",
    "raw_mean_per_launch": s1,
    "fetch_bytes_raw": fetch_raw, "write_bytes": write,
    "gfx950_correction": "FETCH_SIZE x2 (MI355X_MICROARCH.md, HBM: 128-B requests tallied at 64 B for coalesced streams); calibrated on this access pattern in round 1: 2.159 GB to read by construction, 2 x FETCH_SIZE = 2.28 GB. WRITE_SIZE is exact.",
    "hbm_bytes_per_launch": 2.0 * fetch_raw + write,
    "kernel_cycles": cycles,
    "valu_issue_utilisation": s1["SQ_INSTS_VALU"] * 4.0 / (cycles * 1024.0),
    "lds_bank_conflict_share_of_lds_active": s1["SQ_LDS_BANK_CONFLICT"] / max(1.0, s1["SQ_LDS_IDX_ACTIVE"]),
    "wave_cycles_waiting_on_instruction_operands": s1["SQ_WAIT_INST_ANY"] / s1["SQ_WAVE_CYCLES"],
    "wave_cycles_parked": s1["SQ_WAIT_ANY"] / s1["SQ_WAVE_CYCLES"],
}
json.dump(out, open(os.path.join(dst, name + "_pmc_stage1.json"), "w"), indent=1)
with open(os.path.join(dst, name + "_pmc_kernels.md"), "w") as f:
    f.write("# %s: PMC passes (tools/profile_round.sh), means per launch on the bench segment (10,000 samples x 6,667 windows, 74 cells)\n\n" % name)
    f.write("cycles = GRBM_GUI_ACTIVE / 8 XCDs; VALU = SQ_INSTS_VALU x 4 / (cycles x 1024 SIMDs); parked = SQ_WAIT_ANY / SQ_WAVE_CYCLES; "
            "operand stall = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES; L2 hit = TCC_HIT / (TCC_HIT + TCC_MISS); L1->L2 = TCP_TCC_READ_REQ per cycle and CU; "
            "LDS conflicts = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE; HBM = 2 x FETCH_SIZE + WRITE_SIZE (gfx950 correction).\n\n")
    f.write("| kernel | launches | cycles | VALU issue | parked | operand stall | L2 hit | L1->L2 req/clk/CU | LDS conflicts | HBM MB |\n|---|---|---|---|---|---|---|---|---|---|\n")
    for k in ("pd::screenKernel", "pd::exactFlaggedKernel", "pd::stage1Kernel", "pd::stage1BigKernel", "pd::cellGatherKernel", "pd::dataLikDenseKernel", "pd::updateLenDenseKernel",
              "pd::finalDenseKernel", "pd::clusterKernel", "pd::suppSelectLocalKernel"):
        if k not in mean:
            continue
        v = mean[k]
        cyc = v["GRBM_GUI_ACTIVE"] / 8.0
        lds = (v["SQ_LDS_BANK_CONFLICT"] / v["SQ_LDS_IDX_ACTIVE"]) if v.get("SQ_LDS_IDX_ACTIVE") else 0.0
        f.write("| %s | %d | %.3g | %.2f | %.2f | %.2f | %.2f | %.3f | %.2f | %.1f |\n" % (
            k[4:], len(acc[k]["SQ_WAVES"]), cyc, v["SQ_INSTS_VALU"] * 4 / (cyc * 1024), v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"],
            v["SQ_WAIT_INST_ANY"] / v["SQ_WAVE_CYCLES"], v["TCC_HIT_sum"] / max(1.0, v["TCC_HIT_sum"] + v["TCC_MISS_sum"]),
            v["TCP_TCC_READ_REQ_sum"] / (cyc * 256), lds, (2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024 / 1e6))
print(open(os.path.join(dst, name + "_pmc_kernels.md")).read())
