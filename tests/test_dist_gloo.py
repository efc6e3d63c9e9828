"""CPU, world_size 2, gloo: the N>1 paths - batch sharding as bench.py runs it, and digit-parallel key switching
(seal_amd.shard.DigitParallel: key digits split over the ranks, one all-reduce per key switch).
The N>1 path of bench.py (seal_amd/shard.py: shard the batch, barrier-bracketed
timing, max over ranks, whole-job rate) on the fiber-emulated library, results checked against the oracle."""
import os
import subprocess
import sys

import pytest

from seal_amd import shard

HERE = os.path.dirname(os.path.abspath(__file__))


def test_split_covers_everything_once():
    for total in (0, 1, 5, 16, 1024, 1025):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                s, c = shard.split(total, world, r)
                seen += list(range(s, s + c))
                assert c in (total // world, total // world + 1)
            assert seen == list(range(total))
    with pytest.raises(ValueError):
        shard.split(4, 2, 2)


def test_world_size_2_gloo(emu):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29531", WORLD_SIZE="2", OMP_NUM_THREADS="1")
    procs = []
    for rank in range(2):
        e = dict(env, RANK=str(rank), LOCAL_RANK=str(rank))
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "dist_worker.py")], env=e,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out)
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    assert "DIST_OK world=2 items=5" in outs[0], outs[0]
    # digit-parallel key switching: ranks 0 and 1 served digits [0,2) and [2,3) of K = 3 and both match the oracle
    assert "DIGIT_PARALLEL_OK rank=0 digits=[0,2)" in outs[0], outs[0]
    assert "DIGIT_PARALLEL_OK rank=1 digits=[2,3)" in outs[1], outs[1]
