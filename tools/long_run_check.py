"""GPU box: `popdel call` over many rounds (default: 256 samples x 12 Mb = 60 segments) with the window loop on the device,
with every hand-over carrying a copy, and walked on the host -- the three VCFs must be the same bytes.
python tools/long_run_check.py [samples] [megabases]"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from popdel_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
mb = int(sys.argv[2]) if len(sys.argv) > 2 else 12
region = (1000000, 1000000 + mb * 1000000)
d = "/tmp/long_cohort_%d_%d" % (n, mb)
if not os.path.exists(os.path.join(d, "profiles.txt")):
    t = time.time()
    synth.write_cohort(d, n, region=region, deletions=synth.dense_deletions, workers=16)
    print("cohort written in %.1f s" % (time.time() - t), flush=True)
outs = {}
for name, extra in (("device walk", {}), ("copied hand-overs", {"POPDEL_COPY_HAND_OVER": "1"}), ("host walk", {"POPDEL_HOST_WALK": "1"})):
    out = "/tmp/long_%s.vcf" % name.replace(" ", "_")
    t = time.time()
    r = subprocess.run([os.path.join(ROOT, "popdel_amd", "lib", "popdel"), "call", os.path.join(d, "profiles.txt"), "-o", out,
                        "-r", "chr20:%d-%d" % region], env=dict(os.environ, POPDEL_TIMING="1", **extra), capture_output=True, text=True)
    print("%-18s rc %d wall %.2f s   %s" % (name, r.returncode, time.time() - t,
                                            " | ".join(l.split("]", 1)[1].strip()[:110] for l in r.stderr.splitlines() if "window loop" in l or "timing] run" in l)), flush=True)
    outs[name] = "\n".join(l for l in open(out).read().splitlines() if not l.startswith("##fileDate"))
ok = len(set(outs.values())) == 1
print("records", sum(1 for l in outs["device walk"].splitlines() if not l.startswith("#")), "identical" if ok else "DIFFERENT")
sys.exit(0 if ok else 1)
