"""bench.py, measurement side: the NTT roofline leg, its live PMC traffic, the per-kernel counters of the timed step and the
algorithmic-byte model of SURVEY section 8(d).  Everything here runs OUTSIDE the timed region."""
import json
import os
import subprocess
import sys

from .workloads import WORKLOADS, device_uniform

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
BENCH_PY = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py")


def ntt_leg(S, ctx, xs, B, K, n, reps=40, warm=25):
    """the roofline leg: the batched forward NTT over the resident batch (2*B polynomials x K components), HIP events on the stream
    the transform is launched on; achieved = algorithmic bytes (16*N per RNS-component transform) / time.
    `warm` untimed launches precede the timed ones (the leg follows ten seconds of host-side reference checking).  The figure still
    moves by +-7 % from run to run on one box: it depends on where the scratch block of the intermediate happens to lie relative to
    the data - 2.44 TB/s with one block, 2.63 with another in the same process, every block of twenty launches alike
    (profiles/r04_ntt_leg_placement.txt); the kernels' HBM traffic is 2.015x the algorithmic bytes either way."""
    timer = S.HipTimer()
    polys = 2 * B
    assert xs.numel() == polys * K * n

    class _Buf:
        ptr = xs.data_ptr()
    for _ in range(warm):
        S.ntt_forward(ctx, _Buf, polys, K)
    timer.start()
    for _ in range(reps):
        S.ntt_forward(ctx, _Buf, polys, K)
    ms = timer.stop() / reps
    alg_bytes = 16.0 * n * K * polys
    achieved = alg_bytes / (ms * 1e-3) / 1e9
    return dict(bound="hbm", kernel="ntt_forward over %d transforms of 2^%d per launch" % (K * polys, n.bit_length() - 1),
                achieved=round(achieved, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(achieved / HBM_PEAK_GBS, 4),
                traffic=None, ms_per_launch=round(ms, 4), algorithmic_bytes_per_launch=alg_bytes)


def ntt_configs1(S, torch, device, polys=4096, reps=10):
    """forward / inverse NTT rate at BASELINE configs[1] (N = 8192, L = 4), batch of `polys` polynomials resident in HBM
    (1 GiB: four times the Infinity Cache), HIP events on the launch stream, algorithmic bytes = 16*N per component."""
    n = 8192
    out = []
    for label, bits in (("configs[1] {60,40,40,60}", [60, 40, 40, 60]), ("all primes < 2^50 {50,40,40,50}", [50, 40, 40, 50])):
        pr = S.CoeffModulus.Create(n, bits)
        p = S.EncryptionParameters("ckks")
        p.set_poly_modulus_degree(n)
        p.set_coeff_modulus(pr)
        ctx = S.SEALContext(p, True, 0)
        comps = len(pr)
        data = device_uniform(torch, pr, (polys,), n, device)

        class _Buf:
            ptr = data.data_ptr()
        timer = S.HipTimer()
        rates = {}
        for name, fn in (("forward", S.ntt_forward), ("inverse", S.ntt_inverse)):
            for _ in range(10):
                fn(ctx, _Buf, polys, comps)
            timer.start()
            for _ in range(reps):
                fn(ctx, _Buf, polys, comps)
            ms = timer.stop() / reps
            alg = 16.0 * n * comps * polys
            rates[name] = dict(achieved=round(alg / (ms * 1e-3) / 1e9, 1), frac=round(alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                               ms_per_launch=round(ms, 4), algorithmic_bytes_per_launch=alg)
        out.append(dict(chain=label, transforms_per_launch=comps * polys, **rates))
        del data
    return dict(bound="hbm", unit="GB/s", peak=HBM_PEAK_GBS, workload="CKKS N=8192, L=4: batched NTT / INTT over all RNS components", chains=out)


# ---- roofline.traffic: HBM bytes of one ntt_forward launch, measured by this run -----------------------------------
PMC_CALLS = 3


def pmc_child(args):
    """Run under `rocprofv3 --kernel-trace --pmc <counter>`: the roofline leg's launch (same shape), no torch, PMC_CALLS calls."""
    import seal_amd as S
    scheme, n, bits, tbits, default_batch = WORKLOADS[args.workload]
    primes = S.CoeffModulus.Create(n, bits)
    K = len(primes) - 1
    parms = S.EncryptionParameters(scheme)
    parms.set_poly_modulus_degree(n)
    parms.set_coeff_modulus(primes)
    if scheme != "ckks":
        parms.set_plain_modulus(S.PlainModulus.Batching(n, tbits))
    ctx = S.SEALContext(parms, True, 0)
    polys = 2 * (args.batch or default_batch)
    buf = S.DeviceBuffer(polys * K * n)  # contents do not matter for the byte counters
    for _ in range(PMC_CALLS):
        S.ntt_forward(ctx, buf, polys, K)
    S.device_synchronize()
    return 0


def pmc_traffic(args, B, K, n):
    """Two separate rocprofv3 passes (FETCH_SIZE, WRITE_SIZE: they do not fit one pass, MI355X_MICROARCH.md §PMC slots) over a
    child process that issues the roofline leg's launch; FETCH_SIZE is doubled (gfx950 tallies 128-B requests at 64 B, same
    guide §HBM).  Returns {'traffic': bytes per launch or None, 'traffic_source': how it was obtained}."""
    import glob
    import shutil
    import sqlite3
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return dict(traffic=None, traffic_source="rocprofv3 not found on this host")
    totals = {}
    per_kernel = {}
    tmp = tempfile.mkdtemp(prefix="sealhip_pmc_", dir="/tmp")
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, counter)
            cmd = [exe, "--kernel-trace", "--pmc", counter, "-d", out, "-o", "r", "--", sys.executable, BENCH_PY,
                   "--pmc-child", "--batch", str(B), "--workload", args.workload]
            env = dict(os.environ, TMPDIR="/tmp")
            p = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=420)
            dbs = glob.glob(os.path.join(out, "**", "*.db"), recursive=True)
            if p.returncode != 0 or not dbs:
                return dict(traffic=None, traffic_source="rocprofv3 --pmc %s failed (rc %d): %s" % (counter, p.returncode, (p.stderr or p.stdout)[-300:]))
            cur = sqlite3.connect(dbs[0]).cursor()
            rows = cur.execute("select kernel_name, count(*), sum(value) from counters_collection where counter_name = ? "
                               "group by kernel_name", (counter,)).fetchall()
            tot = 0.0
            for name, calls, val in rows:
                if "ntt" not in name:
                    continue
                kib = float(val) / PMC_CALLS
                short = name.replace("sealhip::(anonymous namespace)::", "").replace("void ", "").split("(")[0]
                per_kernel.setdefault(short, {})[counter] = round(kib * (2 if counter == "FETCH_SIZE" else 1), 1)
                tot += kib
            totals[counter] = tot * 1024.0
        traffic = int(round(2.0 * totals["FETCH_SIZE"] + totals["WRITE_SIZE"]))
        return dict(traffic=traffic, traffic_source="live: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this run "
                    "(KiB per launch per kernel, FETCH doubled): %s" % json.dumps(per_kernel, sort_keys=True),
                    traffic_over_algorithmic=round(traffic / (16.0 * n * K * 2 * B), 3))
    except Exception as e:  # the counters must never take the benchmark down
        return dict(traffic=None, traffic_source="PMC passes failed: %r" % (e,))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


# ---- roofline_step: the kernels that dominate the timed step, by this run's own counters --------------------------------
VALU_CYCLES_PER_WAVE_INST = 4      # a wave64 VALU instruction occupies a SIMD16 for four cycles (MI355X_MICROARCH.md; measured 4.2-5)
SIMDS, ENGINE_HZ = 256 * 4, 2.4e9


def step_counters(args, B):
    """One rocprofv3 --kernel-trace --pmc pass over a child that runs the timed step alone (a smaller batch: the per-dispatch
    figures scale with it, the utilisation does not once the chip is full).  Per kernel: share of the step's GPU time, wave
    instructions on the vector ALU per dispatch, and issue utilisation = those x 4 cycles / (duration x 1024 SIMDs x 2.4 GHz)."""
    import glob
    import shutil
    import sqlite3
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return dict(source="rocprofv3 not found on this host")
    child_batch = max(1, B)   # the timed batch itself (round 3 profiled a batch of 64 and left a 36 % per-item gap to explain)
    tmp = tempfile.mkdtemp(prefix="sealhip_step_", dir="/tmp")
    try:
        cmd = [exe, "--kernel-trace", "--pmc", "SQ_INSTS_VALU", "SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "-d", tmp, "-o", "r", "--",
               sys.executable, BENCH_PY, "--step-child", "--workload", args.workload, "--batch", str(child_batch),
               "--total-batch", str(child_batch), "--steps", "2", "--warmup", "1"]
        p = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=600)
        dbs = glob.glob(os.path.join(tmp, "**", "*.db"), recursive=True)
        if p.returncode != 0 or not dbs:
            return dict(source="rocprofv3 --pmc pass failed (rc %d): %s" % (p.returncode, (p.stderr or p.stdout)[-300:]))
        cur = sqlite3.connect(dbs[0]).cursor()
        rows = cur.execute("select kernel_name, counter_name, count(*), sum(value), sum(duration) from counters_collection "
                           "group by kernel_name, counter_name").fetchall()
        table = {}
        for name, ctr, cnt, val, dur in rows:
            if "sealhip" not in name:
                continue  # torch's input generation, copies
            short = name.replace("sealhip::(anonymous namespace)::", "").replace("void ", "").split("(")[0]
            e = table.setdefault(short, dict(dispatches=cnt, ns=float(dur)))
            e[ctr] = float(val)
        total_ns = sum(e["ns"] for e in table.values()) or 1.0
        out = []
        for k, e in sorted(table.items(), key=lambda kv: -kv[1]["ns"])[:8]:
            insts = e.get("SQ_INSTS_VALU", 0.0)
            util = insts * VALU_CYCLES_PER_WAVE_INST / (e["ns"] * 1e-9 * SIMDS * ENGINE_HZ) if e["ns"] else 0.0
            row = dict(kernel=k, share_of_gpu_time=round(e["ns"] / total_ns, 3), dispatches=e["dispatches"],
                       avg_ms=round(e["ns"] / e["dispatches"] / 1e6, 4), valu_wave_insts_per_dispatch=int(insts / e["dispatches"]),
                       valu_issue_utilisation=round(util, 3))
            if e.get("SQ_WAVE_CYCLES"):
                row["waiting_to_issue_frac_of_wave_cycles"] = round(e.get("SQ_WAIT_INST_ANY", 0.0) / e["SQ_WAVE_CYCLES"], 3)
            out.append(row)
        return dict(source="live: one rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY pass over "
                           "`bench.py --step-child --batch %d` (3 steps); utilisation = VALU wave instructions x %d cycles / (kernel time "
                           "x %d SIMDs x %.1f GHz).  The profiler SERIALISES the kernels: in the timed step the integer-class and the "
                           "double-precision key-switch kernels share the CUs on two streams, here each has the chip to itself - avg_ms "
                           "is the kernel alone, and the sum over the kernels exceeds the step's wall time by what the overlap saves "
                           "(~6 %%; profiles/r04_ks_handover.txt)" % (
                               child_batch, VALU_CYCLES_PER_WAVE_INST, SIMDS, ENGINE_HZ / 1e9),
                    kernels=out, serialised_gpu_ms_per_step=round(total_ns / 3e6, 3))
    except Exception as e:  # the counters must never take the benchmark down
        return dict(source="PMC pass failed: %r" % (e,))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def step_bytes(workload, K, L, n):
    """algorithmic bytes of one ciphertext through the timed step (words of 8 bytes; twiddles and scratch excluded),
    with the switching key read once per ciphertext / once per batch (what the kernels do: it stays in L2 for the whole batch)."""
    key = 2 * K * L * n * 8
    if workload == "rotate_c5":
        total = (2 * K * K + 10 * K - 2) * 8 * n          # apply_galois + key switch + rescale
    else:
        total = (2 * K * K + 18 * K - 2) * 8 * n          # multiply + relinearize + rescale / mod_switch
    out = dict(with_key=total, key_amortised=total - key)
    if workload == "bfv_c4":
        # BEHZ multiply is transform-heavy: (8K+4) forward + (6K+3) inverse transforms of 16 N bytes each, K(K+1) more in the key switch
        out["ntt_equivalent"] = ((8 * K + 4) + (6 * K + 3) + K * (K + 1) + 2 * K) * 16 * n
    return out
