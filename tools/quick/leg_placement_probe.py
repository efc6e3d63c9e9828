"""The 2^16 NTT leg against the VIRTUAL address of its scratch block (a development library built with SEALHIP_AB_SWITCHES prints the
pool's hipMalloc calls under SEALHIP_POOL_TRACE=1): ROUNDS times release the pool, run the leg, print rate + data address.
usage: python tools/quick/leg_placement_probe.py [rounds] [hold_mib]   (addresses: a variant library built with
tools/quick/build_variant.sh trace "" pool.cpp copied over seal_amd/lib/libsealhip.so, SEALHIP_POOL_TRACE=1)
hold_mib: allocate (and keep) that many MiB with hipMalloc before each round, so that the next block lands somewhere else"""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch
import seal_amd as S
from seal_amd import shard
from benchlib import workloads, launcher
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 10
hold_mib = int(sys.argv[2]) if len(sys.argv) > 2 else 0
args = launcher.parse(["--no-cpu-baseline", "--no-pmc", "--no-children"])
r = launcher.init_ranks(args)
w = workloads.build(args, S, shard, torch, None, r.device, r.dev_sync, 1, 0)
class _Buf: ptr = w.xs.data_ptr()
held = []
for i in range(rounds):
    r.dev_sync(); S.release_pool()
    if hold_mib:
        held.append(torch.empty(hold_mib << 20, dtype=torch.uint8, device=r.device))
    t = S.HipTimer()
    for _ in range(3): S.ntt_forward(w.ctx, _Buf, 2 * w.B, w.K)
    t.start()
    for _ in range(10): S.ntt_forward(w.ctx, _Buf, 2 * w.B, w.K)
    ms = t.stop() / 10
    print("round %d: leg %.4f of peak, data at %s" % (i, 16.0 * w.n * w.K * 2 * w.B / (ms * 1e-3) / 1e9 / 8000.0, hex(w.xs.data_ptr())), file=sys.stderr, flush=True)
os._exit(0)
