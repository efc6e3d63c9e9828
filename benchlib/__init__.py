"""Parts of bench.py (repo root): launcher, workloads, counters, CPU baseline."""
