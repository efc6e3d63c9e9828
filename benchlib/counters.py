"""bench.py, measurement side: the NTT roofline leg, its live PMC traffic, the per-kernel counters of the timed step and the
algorithmic-byte model of SURVEY section 8(d).  Everything here runs OUTSIDE the timed region."""
import json
import os
import subprocess
import sys

from .workloads import WORKLOADS, device_uniform

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
BENCH_PY = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py")


def ntt_leg(S, ctx, xs, B, K, n, reps=10, warm=25, blocks=3):
    """the roofline leg: the batched forward NTT over the resident batch (2*B polynomials x K components), HIP events on the stream
    the transform is launched on; achieved = algorithmic bytes (16*N per RNS-component transform) / time.
    `warm` untimed launches precede the timed ones (the leg follows ten seconds of host-side reference checking).
    The rate of one build moves by +-7 % with WHERE the scratch block of the intermediate happens to lie relative to the data
    (profiles/r04_ntt_leg_placement.txt: 2.44 TB/s with the block the pool recycles, 2.63 with a freshly allocated one, same
    process, every block of launches alike; the kernels' HBM traffic is 2.015x the algorithmic bytes either way).  So the leg is
    timed in `blocks` blocks of `reps` launches with the pool's block AND again after SealHip_ReleasePool() (fresh block):
    `frac` / `achieved` / `ms_per_launch` are the MEDIAN block, `frac_range` the slowest and the fastest, `frac_by_placement`
    the two populations."""
    timer = S.HipTimer()
    polys = 2 * B
    assert xs.numel() == polys * K * n
    alg_bytes = 16.0 * n * K * polys

    class _Buf:
        ptr = xs.data_ptr()

    def timed_blocks():
        out = []
        for _ in range(blocks):
            timer.start()
            for _ in range(reps):
                S.ntt_forward(ctx, _Buf, polys, K)
            out.append(timer.stop() / reps)
        return out
    for _ in range(warm):
        S.ntt_forward(ctx, _Buf, polys, K)
    pooled = timed_blocks()
    S.device_synchronize()
    S.release_pool()             # the next launch takes a newly allocated scratch block
    for _ in range(3):
        S.ntt_forward(ctx, _Buf, polys, K)
    fresh = timed_blocks()
    every = sorted(pooled + fresh)
    ms = every[len(every) // 2] if len(every) % 2 else 0.5 * (every[len(every) // 2 - 1] + every[len(every) // 2])
    gbs = lambda t: alg_bytes / (t * 1e-3) / 1e9  # noqa: E731
    achieved = gbs(ms)
    return dict(bound="hbm", kernel="ntt_forward over %d transforms of 2^%d per launch" % (K * polys, n.bit_length() - 1),
                achieved=round(achieved, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(achieved / HBM_PEAK_GBS, 4),
                frac_range=[round(gbs(every[-1]) / HBM_PEAK_GBS, 4), round(gbs(every[0]) / HBM_PEAK_GBS, 4)],
                frac_by_placement=dict(pool_block=[round(gbs(t) / HBM_PEAK_GBS, 4) for t in pooled],
                                       fresh_block=[round(gbs(t) / HBM_PEAK_GBS, 4) for t in fresh]),
                frac_is="median of %d blocks of %d launches, half with the pool's scratch block, half with a fresh one" % (2 * blocks, reps),
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
XCDS = 8                            # GRBM_GUI_ACTIVE is summed over the eight XCDs of an MI355X (calibrated: 18.5 G "cycles"/s on a 343 us kernel)


def _pmc_pass(exe, counters, child, tmp, tag):
    """one rocprofv3 --kernel-trace --pmc pass over `child`; {kernel: {dispatches, ns, counter: sum}} for this library's kernels"""
    import glob
    import sqlite3
    out = os.path.join(tmp, tag)
    cmd = [exe, "--kernel-trace", "--pmc"] + counters + ["-d", out, "-o", "r", "--"] + child
    p = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=600)
    dbs = glob.glob(os.path.join(out, "**", "*.db"), recursive=True)
    if p.returncode != 0 or not dbs:
        raise RuntimeError("rocprofv3 --pmc %s failed (rc %d): %s" % (" ".join(counters), p.returncode, (p.stderr or p.stdout)[-300:]))
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
    return table


def step_counters(args, B):
    """Two rocprofv3 --kernel-trace --pmc passes over a child that runs the timed step alone.  Per kernel: share of the step's GPU
    time, wave instructions on the vector ALU per dispatch, issue utilisation = those x 4 cycles / (duration x 1024 SIMDs x 2.4 GHz)
    - and, from the second pass (GRBM_GUI_ACTIVE = shader-engine clock cycles while the kernel ran), the clock the chip actually
    SUSTAINED under that kernel and the utilisation against it (under the fp64 load of the key switch the
    chip holds ~1.8 - 2.0 GHz, not the 2.4 GHz the first figure is quoted against)."""
    import shutil
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return dict(source="rocprofv3 not found on this host")
    child_batch = max(1, B)   # the timed batch itself (round 3 profiled a batch of 64 and left a 36 % per-item gap to explain)
    tmp = tempfile.mkdtemp(prefix="sealhip_step_", dir="/tmp")
    try:
        child = [sys.executable, BENCH_PY, "--step-child", "--workload", args.workload, "--batch", str(child_batch),
                 "--total-batch", str(child_batch), "--steps", "2", "--warmup", "1"]
        table = _pmc_pass(exe, ["SQ_INSTS_VALU", "SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY"], child, tmp, "sq")
        try:
            clocks = _pmc_pass(exe, ["GRBM_GUI_ACTIVE"], child, tmp, "grbm")
        except Exception as e:  # the second pass is an extra: without it the line simply lacks the clock
            clocks = {}
            clock_note = "GRBM_GUI_ACTIVE pass failed: %r" % (e,)
        else:
            clock_note = ("second pass: GRBM_GUI_ACTIVE (summed over the %d XCDs) / %d / kernel time; the counter brackets a dispatch "
                          "with a few tens of microseconds of its own, which is why it is only quoted for the long kernels" % (XCDS, XCDS))
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
            c = clocks.get(k)
            if c and c.get("GRBM_GUI_ACTIVE") and c["ns"] and c["ns"] / c["dispatches"] > 5e5:   # kernels of 0.5 ms and more
                cycles = c["GRBM_GUI_ACTIVE"] / XCDS            # shader clock cycles over the same dispatches of the second pass
                row["sclk_mhz_sustained"] = int(round(cycles / (c["ns"] * 1e-9) / 1e6))
                if c["dispatches"] == e["dispatches"]:
                    row["valu_issue_utilisation_at_sustained_clock"] = round(insts * VALU_CYCLES_PER_WAVE_INST / (cycles * SIMDS), 3)
            out.append(row)
        # what the step really moved: FETCH_SIZE (x2 on gfx950, MI355X_MICROARCH.md) and WRITE_SIZE over the same child, all kernels
        hbm = {}
        try:
            kib = 0.0
            for counter in ("FETCH_SIZE", "WRITE_SIZE"):
                t = _pmc_pass(exe, [counter], child, tmp, counter.lower())
                kib += sum(e.get(counter, 0.0) for e in t.values()) * (2.0 if counter == "FETCH_SIZE" else 1.0)
            per_step = kib * 1024.0 / 3.0   # the child runs 1 warm-up + 2 timed steps
            hbm = dict(hbm_counter_bytes_per_step=int(per_step), hbm_counter_bytes_per_ciphertext=int(per_step / child_batch),
                       hbm_counter_source="FETCH_SIZE x2 + WRITE_SIZE summed over every kernel of the step child, two more rocprofv3 --pmc passes")
        except Exception as e:
            hbm = dict(hbm_counter_source="FETCH_SIZE / WRITE_SIZE passes failed: %r" % (e,))
        weighted = [(r["sclk_mhz_sustained"], r["share_of_gpu_time"]) for r in out if "sclk_mhz_sustained" in r]
        sclk = int(round(sum(a * b for a, b in weighted) / sum(b for _, b in weighted))) if weighted else None
        return dict(source="live: rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY over "
                           "`bench.py --step-child --batch %d` (3 steps); utilisation = VALU wave instructions x %d cycles / (kernel time "
                           "x %d SIMDs x %.1f GHz).  The profiler SERIALISES the kernels: in the timed step the integer-class and the "
                           "double-precision key-switch kernels (and, round 5, the chunks of the key switch on their lanes) share the "
                           "CUs, here each has the chip to itself - avg_ms is the kernel alone; serialised, the four key-switch "
                           "kernels sum to about what they take overlapped (18.8 against 18.6 ms in round 4: the fork saves ~1 %%)" % (
                               child_batch, VALU_CYCLES_PER_WAVE_INST, SIMDS, ENGINE_HZ / 1e9),
                    clock_source=clock_note, sclk_mhz_sustained=sclk,
                    kernels=out, serialised_gpu_ms_per_step=round(total_ns / 3e6, 3), **hbm)
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
    # nominal_with_key charges the whole key to every ciphertext (SURVEY 8(d)'s formula); the kernels read it once per chunk per XCD, so
    # it is an accounting figure, not traffic - key_amortised is the honest algorithmic figure, hbm_counter_bytes_per_ciphertext
    # (roofline_step, from the FETCH_SIZE / WRITE_SIZE passes) what actually crossed the fabric
    out = dict(nominal_with_key=total, key_amortised=total - key)
    if workload == "bfv_c4":
        # BEHZ multiply is transform-heavy: (8K+4) forward + (6K+3) inverse transforms of 16 N bytes each, K(K+1) more in the key switch
        out["ntt_equivalent"] = ((8 * K + 4) + (6 * K + 3) + K * (K + 1) + 2 * K) * 16 * n
    return out
