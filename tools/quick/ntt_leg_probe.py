"""Why the N = 2^16 NTT leg reads 2.44 or 2.63 TB/s for the same kernels (round 4): where the intermediate sits relative to the data.
usage: python tools/quick/ntt_leg_probe.py [full]     (SEALHIP_MID_SKEW=<bytes> displaces the intermediate inside its scratch block)"""
import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch
import seal_amd as S
from seal_amd import shard
from benchlib import workloads, launcher
full = "full" in sys.argv[1:]
args = launcher.parse(["--no-cpu-baseline", "--no-pmc", "--no-children"])
r = launcher.init_ranks(args)
w = workloads.build(args, S, shard, torch, None, r.device, r.dev_sync, 1, 0)
def blocks(tag, n=4, reps=20):
    class _Buf: ptr = w.xs.data_ptr()
    t = S.HipTimer(); out = []
    for _ in range(n):
        t.start()
        for _ in range(reps): S.ntt_forward(w.ctx, _Buf, 2 * w.B, w.K)
        out.append(round(t.stop() / reps, 4))
    alg = 16.0 * w.n * w.K * 2 * w.B
    print("skew %s: %s ms per launch per block:" % (os.environ.get("SEALHIP_MID_SKEW", "0"), tag), out, "GB/s:", [round(alg / (m * 1e-3) / 1e9) for m in out], "data at", hex(w.xs.data_ptr()), flush=True)
blocks("fresh process, before any step")
if full:
    el = shard.timed_steps(w.step, 10, 2, None, r.dev_sync, torch, r.device)
    print("step ms", 1e3 * el / 10, "pool bytes", S.pool_stats()[0], flush=True)
    blocks("right after the timed steps")
    time.sleep(12)
    blocks("after 12 s of an idle GPU")
    S.release_pool()
    blocks("after the pool was released (fresh scratch block)")
os._exit(0)
