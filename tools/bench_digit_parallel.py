#!/usr/bin/env python
"""BASELINE configs[4] as written: CKKS N=65536, L=16 - rotate_vector (apply_galois key switch with the decomposition
digits spread over the GPUs of the node) + rescale_to_next.

This is `bench.py --workload rotate_c5` (same contract, same JSON line, `verified_items` against the reference, the exchange
inside libsealhip over RCCL); kept as an entry point of its own for the configuration's name:

  python tools/bench_digit_parallel.py --gpus G [--exchange all_reduce|reduce_scatter] [--batch B] [--steps K] [--warmup W]

With --gpus G > 1 outside torchrun the G ranks are started here (bench.py's launcher); under torchrun it is one rank."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

if __name__ == "__main__":
    os.execv(sys.executable, [sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "rotate_c5"] + sys.argv[1:])
