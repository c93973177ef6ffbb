"""Wall time of the product binary under its A/B switches (GPU box): python tools/e2e_env_probe.py <samples>
Per run: wall, time before main / inside / after the exit call, CPU seconds, page faults; phase split of the last round."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from popdel_amd import synth
n = int(sys.argv[1])
d = "/tmp/e2e_cohort_%d" % n
if not os.path.exists(os.path.join(d, "profiles.txt")):
    synth.write_cohort(d, n, deletions=synth.single_deletion, workers=16)
variants = [("default", {}, []), ("zlib instead of libdeflate", {"POPDEL_ZLIB_INFLATE": "1"}, []), ("no node pin", {"POPDEL_NO_NODE_PIN": "1"}, []),
            ("every hand-over a copy", {"POPDEL_COPY_HAND_OVER": "1"}, []), ("exit frees alone", {"POPDEL_NO_PARALLEL_DISCARD": "1"}, []),
            ("16 threads", {"POPDEL_THREADS": "16"}, [])]
for rep in range(3):
    for name, extra, prefix in variants:
        env = dict(os.environ, POPDEL_TIMING="2")
        env.update(extra)
        t = time.time()
        r = subprocess.run(prefix + [os.path.join(ROOT, "popdel_amd", "lib", "popdel"), "call", os.path.join(d, "profiles.txt"), "-o", "/tmp/e2e.vcf",
                            "-r", "chr20:1000000-2000000"], env=env, capture_output=True, text=True)
        t1 = time.time()
        lines = (r.stdout + r.stderr).splitlines()
        ep = [l for l in lines if "epoch" in l]
        run = [l for l in lines if "timing] run" in l]
        ph = [l for l in lines if "timing] threads" in l or "teardown:" in l]
        at_exit = float(ep[0].split("at exit")[1]) if ep else 0
        at_main = float(ep[0].split("at main")[1].split(",")[0]) if ep else 0
        print("%-32s wall %.3f s  (before main %.3f, main to exit call %.3f, exit to return %.3f) %s" %
              (name, t1 - t, at_main - t, at_exit - at_main, t1 - at_exit, run[0].split("]")[1] if run else r.returncode), flush=True)
        for l in ph if rep == 2 else []:
            print("   ", l[:400])
