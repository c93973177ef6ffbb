"""End-to-end timing of the product binary on a synthetic cohort (GPU box): python tools/e2e_timing.py <samples> [threads...]"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from popdel_amd import synth
n = int(sys.argv[1])
threads = sys.argv[2:] or ["64"]
d = "/tmp/e2e_cohort_%d" % n
if not os.path.exists(os.path.join(d, "profiles.txt")):
    t = time.time()
    synth.write_cohort(d, n, deletions=synth.single_deletion, workers=16)
    print("cohort written in %.1f s" % (time.time() - t), flush=True)
if not os.path.exists(os.path.join(d, "profiles.txt")):
    pass
for th, host_walk in [(t, h) for t in threads for h in (False, True)]:
    env = dict(os.environ, POPDEL_TIMING=os.environ.get("POPDEL_TIMING", "1"), POPDEL_THREADS=th)
    if host_walk:
        env["POPDEL_HOST_WALK"] = "1"
    t = time.time()
    r = subprocess.run([os.path.join(ROOT, "popdel_amd", "lib", "popdel"), "call", os.path.join(d, "profiles.txt"), "-o", "/tmp/e2e.vcf",
                        "-r", "chr20:1000000-2000000"], env=env, capture_output=True, text=True)
    print("spawned at %.4f, returned at %.4f" % (t, time.time()))
    print("threads", th, "window loop on the", "host" if host_walk else "device", "wall %.2f s" % (time.time() - t), "rc", r.returncode)
    print("\n".join(l for l in (r.stdout + r.stderr).splitlines() if "timing" in l.lower() or "timeline" in l.lower() or "memory]" in l or "phase" in l.lower() or "s " in l[-12:])[-6000:])
