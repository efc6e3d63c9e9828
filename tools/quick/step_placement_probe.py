"""Does the headline STEP move with where the pool's scratch blocks lie (as the NTT leg does, profiles/r04_ntt_leg_placement.txt)?
One process, the same inputs and keys: time the step, release the pool (every scratch block is freed and newly allocated by the next
step), time it again - ROUNDS times.  usage: python tools/quick/step_placement_probe.py [rounds]"""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch
import seal_amd as S
from seal_amd import shard
from benchlib import workloads, launcher
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
args = launcher.parse(["--no-cpu-baseline", "--no-pmc", "--no-children"])
r = launcher.init_ranks(args)
w = workloads.build(args, S, shard, torch, None, r.device, r.dev_sync, 1, 0)
class _Buf: ptr = w.xs.data_ptr()
def leg(reps=10):
    t = S.HipTimer()
    for _ in range(3): S.ntt_forward(w.ctx, _Buf, 2 * w.B, w.K)
    t.start()
    for _ in range(reps): S.ntt_forward(w.ctx, _Buf, 2 * w.B, w.K)
    ms = t.stop() / reps
    return round(16.0 * w.n * w.K * 2 * w.B / (ms * 1e-3) / 1e9 / 8000.0, 4)
for i in range(rounds):
    a = 1e3 * shard.timed_steps(w.step, 8, 3, None, r.dev_sync, torch, r.device) / 8
    b = 1e3 * shard.timed_steps(w.step, 8, 0, None, r.dev_sync, torch, r.device) / 8
    print("round %d: step %.3f / %.3f ms (%.0f ct/s), NTT leg with these blocks %.4f, pool bytes %d" % (i, a, b, 1e3 * w.B / min(a, b), leg(), S.pool_stats()[0]), flush=True)
    r.dev_sync()
    S.release_pool()
os._exit(0)
