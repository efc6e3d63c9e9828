"""Digit-parallel key switching over real GPUs: one process per GPU, the library's own RCCL communicator,
both exchange shapes, keys distributed by the library's broadcast; every rank's result equals the reference's.  Needs at
least two MI355X in the box - skipped on the single-GPU boxes; the ranks-emulated and one-rank-RCCL forms of the same
arithmetic run everywhere (test_gpu_parity.py: test_digit_parallel_*)."""
import os
import subprocess
import sys
import tempfile

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.gpu
@pytest.mark.parametrize("exchange", [0, 1])
def test_digit_parallel_over_real_ranks(gpu, exchange):
    import seal_amd as S
    n_dev = S.device_count()
    if n_dev < 2:
        pytest.skip("one GPU visible: the multi-rank RCCL path needs at least two")
    if not S.Comm.rccl_available():
        pytest.fail("librccl.so.1 did not load on a multi-GPU box")
    nranks = min(n_dev, 8)
    with tempfile.TemporaryDirectory() as tmp:
        id_path = os.path.join(tmp, "rccl_id")
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "multi_gpu_worker.py"), str(r), str(nranks), id_path, str(exchange)],
                                  env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(nranks)]
        outs = []
        for p in procs:
            try:
                out, _ = p.communicate(timeout=900)
            except subprocess.TimeoutExpired:
                for q in procs:
                    q.kill()
                raise
            outs.append(out)
    assert all(p.returncode == 0 for p in procs), "\n".join(o[-1500:] for o in outs)
    for r, out in enumerate(outs):
        assert "MULTI_GPU_OK rank=%d/%d exchange=%d" % (r, nranks, exchange) in out, out[-1500:]
