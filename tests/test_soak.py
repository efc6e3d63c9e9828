"""Soak of the shortest device sequences (tests/soak_cases.py): fresh upload -> one operation -> download against the real
reference, repeated for a fixed time in every host-copy mode.  CPU: a few hundred iterations on the fiber emulator (host
logic of the three product / growth paths and of the copy modes); GPU: about 75 s, tens of thousands of sequences - the
regression guard for the one unexplained wrong result of round 3 (profiles/r03_fuzz_stress.txt)."""
import os

import pytest

import sealref
import soak_cases as K

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_soak_emulated(emu):
    if not sealref.available():
        pytest.skip("needs the real reference (oracle/_ref)")
    stats = K.run_soak(6.0, cases=K.default_cases(small=True), max_iterations=120)
    assert stats["iterations"] >= 60 and all(v > 0 for v in stats["per_op"].values()), stats
    assert len(stats["per_phase"]) == 3 and all(v > 0 for v in stats["per_phase"].values()), stats


@pytest.mark.gpu
def test_soak_gpu(gpu):
    if not sealref.available():
        pytest.skip("needs the real reference (oracle/_ref)")
    seconds = float(os.environ.get("SEALHIP_SOAK_SECONDS", "75"))
    stats = K.run_soak(seconds, dump_dir=os.path.join(ROOT, "gpurun_out"))
    print("soak:", stats)
    assert stats["iterations"] >= 5000, stats   # >= 5 000 iterations inside the time box
    assert all(v > 0 for v in stats["per_phase"].values()), stats
