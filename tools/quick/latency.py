"""Single-ciphertext latency (and small batches) of CKKS multiply + relinearize + rescale at the headline parameters, device
resident: ms per step for batches 1, 2, 4, 8 (HIP events around `steps` steps after warm-up)."""
import json, os, subprocess, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
for b in [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "1,2,4,8").split(",")]:
    extra = ["--graph"] if os.environ.get("LATENCY_GRAPH") else []
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--batch", str(b), "--steps", "50", "--warmup", "5", "--no-cpu-baseline"] + extra,
                         capture_output=True, text=True).stdout.strip().splitlines()[-1]
    j = json.loads(out)
    print("batch %d: %.3f ms/step  %.1f ct/s   (SEALHIP_KS_SPLIT=%s, %s)" % (b, j["ms_per_step"], j["value"], os.environ.get("SEALHIP_KS_SPLIT", "auto"), j["config"]["launch"]), flush=True)
