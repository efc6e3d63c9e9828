"""bench.py, CPU baseline: the reference's own Evaluator (oracle/_ref = Microsoft SEAL, HEXL off) timed on this host's cores on a
bounded sample.  Checker / baseline only - never the thing measured."""
import os
import sys


def physical_cores():
    try:
        import psutil
        return psutil.cpu_count(logical=False) or (os.cpu_count() or 1)
    except Exception:
        return os.cpu_count() or 1


def _time_reference(pipeline, scheme, n, primes, t_plain, thread_counts, reps):
    """runs of the loaded flavour of the reference: [{value, cores}], and the single-thread rate"""
    import sealref
    ref = sealref.RefContext(scheme, n, primes, t_plain)
    ref.keygen_relin()
    if pipeline == "rotate":
        ref.keygen_galois_steps([1])
    runs = []
    for threads in thread_counts:
        secs = ref.time_pipeline(pipeline, threads, reps)
        runs.append(dict(value=round(threads * reps / secs, 3), cores=threads))
    one = ref.time_pipeline(pipeline, 1, 2)
    return runs, round(2 / one, 3)


def _time_reference_child(lib_path, pipeline, scheme, n, primes, t_plain, thread_counts, reps):
    """the same in a child process that loads another flavour of the reference (SEALREF_LIB); None when it fails"""
    import json
    import subprocess
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    code = ("import sys, json; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "from benchlib.cpu import _time_reference\n"
            "runs, one = _time_reference(%r, %r, %d, %r, %d, %r, %d)\n"
            "print('RESULT ' + json.dumps(dict(runs=runs, one=one)))\n"
            % (root, os.path.join(root, "tests"), pipeline, scheme, n, list(primes), t_plain, list(thread_counts), reps))
    try:
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, SEALREF_LIB=lib_path), capture_output=True, text=True, timeout=900)
        for line in r.stdout.splitlines():
            if line.startswith("RESULT "):
                j = json.loads(line[7:])
                return j["runs"], j["one"]
        sys.stderr.write("cpu_baseline(%s) failed: %s\n" % (os.path.basename(lib_path), r.stderr[-500:]))
    except Exception as e:
        sys.stderr.write("cpu_baseline(%s) failed: %r\n" % (os.path.basename(lib_path), e))
    return None


def cpu_baseline(workload, scheme, n, primes, t_plain, args):
    pipeline = {"headline": "ckks_mul_relin_rescale", "bfv_c4": "bfv_mul_relin_modswitch", "rotate_c5": "rotate"}[workload]
    try:
        import sealref
        if sealref.available():
            logical = args.cpu_threads or (os.cpu_count() or 1)
            phys = min(physical_cores(), logical)
            reps = args.cpu_reps
            # one thread per logical CPU and, where that differs, one per physical core: the host's BEST rate is the baseline
            # (memory-bound NTTs often lose with SMT siblings: 32 vs 63 ct/s on a 128-core / 256-thread box), all runs are listed.
            # Two builds of the same reference sources: g++ -O3 (the parity checker) and, when oracle/_ref holds it, clang++ -O3 -
            # the compiler the reference's README recommends for speed (/root/reference README.md:232).  The faster one is the baseline.
            counts = sorted({logical, phys}, reverse=True)
            runs, one = _time_reference(pipeline, scheme, n, primes, t_plain, counts, reps)
            for r in runs:
                r["compiler"] = "g++ -O3"
            singles = {"g++ -O3": one}
            if os.path.exists(sealref.CLANG_LIB_PATH) and not os.environ.get("SEALREF_LIB"):
                got = _time_reference_child(sealref.CLANG_LIB_PATH, pipeline, scheme, n, primes, t_plain, counts, reps)
                if got:
                    for r in got[0]:
                        r["compiler"] = "clang++ -O3"
                    runs += got[0]
                    singles["clang++ -O3"] = got[1]
            best = max(runs, key=lambda r: r["value"])
            out = dict(value=best["value"], unit="ciphertexts/s", cores=best["cores"], kind="reference", compiler=best["compiler"],
                       sample="%d threads x %d ciphertexts each; every thread builds its inputs, runs one untimed pass, waits at a "
                              "start barrier; wall time from the barrier to the last thread's finish; per-thread "
                              "MemoryPoolHandle::New(); seal::Evaluator, HEXL off, same parameters; best of the listed runs "
                              "(thread counts x compilers)" % (best["cores"], reps),
                       runs=runs)
            out["single_thread_value"] = max(singles.values())
            out["single_thread_by_compiler"] = singles
            return out
    except Exception as e:  # the baseline must never take the benchmark down
        sys.stderr.write("cpu_baseline(reference) unavailable: %r\n" % (e,))
    if workload != "headline":
        return None
    try:
        import numpy as np
        import sealoracle
        from oracle import rand_ct
        rng = np.random.default_rng(0x5EA1)
        K = len(primes) - 1
        po = sealoracle.PortContext("ckks", n, primes)
        a, b = rand_ct(rng, primes, K, n), rand_ct(rng, primes, K, n)
        rlk = np.stack([np.stack([np.stack([rng.integers(0, q, n, dtype=np.uint64) for q in primes]) for _ in range(2)])
                        for _ in range(K)])
        secs, _ = po.time_ckks_pipeline(a, b, rlk, 1)
        return dict(value=round(1 / secs, 4), unit="ciphertexts/s", cores=1, kind="port",
                    sample="1 ciphertext, plain-C restatement (oracle/seal_oracle.c), 1 thread")
    except Exception as e:
        sys.stderr.write("cpu_baseline(port) unavailable: %r\n" % (e,))
    return None
