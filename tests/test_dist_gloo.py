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
    # ... and at a two-pass size followed by a rescale: both ranks fold the pending mod-down into it
    assert "DIGIT_PARALLEL_FOLDED_OK rank=0" in outs[0] and "DIGIT_PARALLEL_FOLDED_OK rank=1" in outs[1], outs[0] + outs[1]


@pytest.mark.parametrize("workload", ["headline", "bfv_c4", "rotate_c5"])
def test_bench_gpus_2_starts_two_ranks(emu, workload):
    """`python bench.py --gpus 2` with no torchrun environment starts the two ranks itself; here on the
    emulated kernels with gloo (SEALHIP_BENCH_EMU=1).  The line must say n_gpus 2, both ranks' sampled items must have
    been checked against the reference, and a WORLD_SIZE that disagrees with --gpus is refused."""
    import json
    root = os.path.dirname(HERE)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["SEALHIP_BENCH_EMU"] = "1"
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                          "--workload", workload], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["value"] > 0 and line["steps"] == 2
    assert line["scaling"] == ("weak" if workload == "headline" else "strong")
    # the probe collective's rank count and every rank's own rate travel in the line
    assert line["rccl_ranks"] == 2 and line["collective_backend"] == "gloo"
    assert [p["rank"] for p in line["per_rank"]] == [0, 1] and all(p["value"] > 0 for p in line["per_rank"])
    assert line["workloads"] is None   # the appended BASELINE workloads belong to the one-GPU headline line only
    import sealref
    if sealref.available():
        assert line["verified_items"] == 4  # two items per rank
    # a launcher that started a different number of ranks than --gpus says is an error, not a silent n_gpus
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--steps", "1", "--warmup", "0"],
                         env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), capture_output=True, text=True, timeout=300)
    assert bad.returncode != 0 and "WORLD_SIZE" in (bad.stdout + bad.stderr)


@pytest.mark.parametrize("workload", ["headline", "bfv_c4", "rotate_c5"])
def test_bench_gpus_8_emulated(emu, workload):
    """`python bench.py --gpus 8` - the shape the driver's scaling run has - as eight real processes over gloo on the emulated kernels
    (VERDICT r5 #4): the probe collective reaches all eight, every rank reports its own rate, the sampled items of every rank that
    owns any equal the reference's.  bfv_c4 shards a total batch of 4 over 8 ranks (four ranks own NOTHING and still pass every
    barrier), rotate_c5 spreads K = 3 key-switch digits over 8 ranks (five ranks own no digit and contribute zero sums)."""
    import json
    root = os.path.dirname(HERE)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["SEALHIP_BENCH_EMU"] = "1"
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "1",
                          "--workload", workload], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 8 and line["rccl_ranks"] == 8 and line["collective_backend"] == "gloo", line
    assert [p["rank"] for p in line["per_rank"]] == list(range(8))
    assert line["value"] > 0 and line["scaling"] == ("weak" if workload == "headline" else "strong")
    if workload == "bfv_c4":
        assert sum(1 for p in line["per_rank"] if p["value"] > 0) == 4, line["per_rank"]   # 4 items: ranks 4..7 have empty shards
    else:
        assert all(p["value"] > 0 for p in line["per_rank"])
    if workload == "rotate_c5":
        assert line["latency_ms_per_ciphertext"] > 0
    ncpu = len(os.sched_getaffinity(0))
    assert line["config"]["cpus_per_rank"] == (ncpu // 8 if ncpu >= 8 else None)   # every rank pinned to its own share of the host
    import sealref
    if sealref.available():
        assert line["verified_items"] == (4 if workload == "bfv_c4" else 16), line["verified_items"]


@pytest.mark.parametrize("workload", ["headline", "bfv_c4"])
def test_bench_streams_divides_the_batch(emu, workload):
    """`bench.py --streams 2`: two evaluators on two streams, each with half of the rank's batch; the sampled items of the
    joined output are still the reference's (emulated kernels)."""
    import json
    root = os.path.dirname(HERE)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["SEALHIP_BENCH_EMU"] = "1"
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--streams", "2", "--steps", "1", "--warmup", "1", "--batch", "5",
                          "--workload", workload, "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert "2 evaluators" in line["config"]["launch"] and line["value"] > 0
    import sealref
    if sealref.available():
        assert line["verified_items"] >= 2


def test_bench_rotate_c5_pipelined_exchange(emu):
    """`bench.py --workload rotate_c5 --gpus 2 --streams 2`: two sub-batches per rank, each with its own evaluator / stream, sharing
    the exchange; every rank's sampled items equal the reference's rotate + rescale (the pipelined form)"""
    import json
    root = os.path.dirname(HERE)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["SEALHIP_BENCH_EMU"] = "1"
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--workload", "rotate_c5",
                          "--streams", "2", "--batch", "4"], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])
    assert line["n_gpus"] == 2 and line["value"] > 0
    assert "2 sub-batches on 2 streams" in line["config"]["exchange_overlap"] and line["config"]["exchange_bytes_per_ciphertext"] > 0
    import sealref
    if sealref.available():
        assert line["verified_items"] == 8  # four items per rank


def test_bench_default_line_carries_configs3_and_configs4(emu):
    """`python bench.py` (one GPU, headline): BASELINE configs[3] and configs[4] are timed by child runs and attached to the same
    line with their own value / verified_items; --no-children leaves them out"""
    import json
    root = os.path.dirname(HERE)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["SEALHIP_BENCH_EMU"] = "1"
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "2", "--warmup", "1", "--child-steps", "2"],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 1 and line["rccl_ranks"] == 1 and line["value"] > 0
    assert set(line["workloads"]) == {"bfv_c4", "rotate_c5"}
    import sealref
    for name, wl in line["workloads"].items():
        assert "error" not in wl, wl
        assert wl["value"] > 0 and wl["steps"] == 2 and wl["ms_per_step"] > 0, (name, wl)
        assert ("configs[3]" if name == "bfv_c4" else "configs[4]") in wl["config"]["workload"]
        if sealref.available():
            assert wl["verified_items"] >= 2, (name, wl)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "1", "--warmup", "0", "--no-children"],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])["workloads"] is None


def test_device_memory_estimate_tracks_the_workloads():
    """bench.py refuses a run whose estimate exceeds the rank's free HBM (benchlib/launcher.py: check_device_memory): the estimate must be
    of the right order - the headline at batch 256 holds ~50 GB on the device (profiles/r05_bench.json: device_memory) - and follow
    the batch, the shard size and the scratch cap"""
    import argparse
    import importlib
    from benchlib import workloads
    importlib.reload(workloads)

    def est(workload, world=1, rank=0, **kw):
        a = argparse.Namespace(workload=workload, batch=kw.get("batch", 0), total_batch=kw.get("total_batch", 1024))
        return workloads.estimate_device_bytes(a, world, rank)
    gib = 2.0 ** 30
    head = est("headline")
    assert 35 * gib < head < 80 * gib, head / gib
    assert est("headline", batch=512) > 1.5 * head and est("headline", batch=64) < 0.5 * head
    assert est("bfv_c4", world=8, rank=3) < 0.25 * est("bfv_c4")          # 128 of 1024 ciphertexts per rank
    assert est("rotate_c5") < 0.3 * head                                   # batch 32, one operand
    os.environ["SEALHIP_KS_SCRATCH_CAP_MIB"] = "4096"
    try:
        assert est("headline") < head - 7 * gib                            # 12 GB of chunked intermediate -> at most 4 GiB
    finally:
        del os.environ["SEALHIP_KS_SCRATCH_CAP_MIB"]
