"""CPU (emulated kernels): error behaviour and metadata semantics of the Evaluator mirror, following
the reference's negative tests (native/tests/seal/evaluator.cpp:2505-2632, 5590 and the throw sites in
native/src/seal/evaluator.cpp) and the C layer's exception -> HRESULT mapping
(native/src/seal/c/defines.h:75-97)."""
import os

import numpy as np
import pytest

from harness import DeviceSide
from oracle import Oracle, coeff_modulus_create, plain_modulus_batching, rand_ct


@pytest.fixture()
def ckks(emu):
    n, bits = 64, [40, 30, 30, 40]
    primes = coeff_modulus_create(n, bits)
    o = Oracle("ckks", n, primes, galois_elts=[3], kind="port")
    d = DeviceSide("ckks", n, primes)
    d.upload_keys(o)
    rng = np.random.default_rng(1)
    return emu, d, o, primes, rng


def test_multiply_requires_ntt_form(ckks):
    S, d, o, primes, rng = ckks
    x = d.ct(rand_ct(rng, primes, 3, 64), is_ntt=False)
    with pytest.raises(S.InvalidArgument):   # "encrypted1 or encrypted2 must be in NTT form" evaluator.cpp:572
        d.ev.multiply_inplace(x, x.copy())


def test_parameter_mismatch(ckks):
    S, d, o, primes, rng = ckks
    x = d.ct(rand_ct(rng, primes, 3, 64))
    y = d.ct(rand_ct(rng, primes, 2, 64))
    with pytest.raises(S.InvalidArgument):   # evaluator.cpp:363
        d.ev.multiply_inplace(x, y)
    with pytest.raises(S.InvalidArgument):
        d.ev.add_inplace(x, y)


def test_scale_out_of_bounds(ckks):
    S, d, o, primes, rng = ckks
    x = d.ct(rand_ct(rng, primes, 3, 64), scale=2.0 ** 60)
    with pytest.raises(S.InvalidArgument):   # scale^2 exceeds total_coeff_modulus_bit_count, evaluator.cpp:704
        d.ev.multiply_inplace(x, x.copy())
    z = d.ct(rand_ct(rng, primes, 3, 64), scale=-1.0)
    with pytest.raises(S.InvalidArgument):   # is_metadata_valid_for: scale must be positive normal
        d.ev.negate_inplace(z)


def test_scale_mismatch_on_add(ckks):
    S, d, o, primes, rng = ckks
    x = d.ct(rand_ct(rng, primes, 3, 64), scale=2.0 ** 20)
    y = d.ct(rand_ct(rng, primes, 3, 64), scale=2.0 ** 21)
    with pytest.raises(S.InvalidArgument):   # "scale mismatch" evaluator.cpp:172
        d.ev.add_inplace(x, y)


def test_rescale_at_end_of_chain(ckks):
    S, d, o, primes, rng = ckks
    x = d.ct(rand_ct(rng, primes, 1, 64))
    with pytest.raises(S.InvalidArgument):   # "end of modulus switching chain reached" evaluator.cpp:1512
        d.ev.rescale_to_next_inplace(x)
    with pytest.raises(S.InvalidArgument):
        d.ev.mod_switch_to_next_inplace(x)


def test_galois_key_missing_and_bad_element(ckks):
    S, d, o, primes, rng = ckks
    x = d.ct(rand_ct(rng, primes, 3, 64))
    with pytest.raises(S.InvalidArgument):   # "Galois key not present" evaluator.cpp:2416
        d.ev.apply_galois_inplace(x, 5, d.glk)
    with pytest.raises(S.InvalidArgument):   # even element: "Galois element is not valid"
        d.ev.apply_galois_inplace(x, 4, d.glk)
    y = d.ct(rand_ct(rng, primes, 3, 64, size=3))
    with pytest.raises(S.InvalidArgument):   # "encrypted size must be 2" evaluator.cpp:2427
        d.ev.apply_galois_inplace(y, 3, d.glk)


def test_apply_galois_rejects_wrong_ntt_form_without_mutating(ckks):
    """ApplyGaloisRejectsWrongNttFormWithoutMutating, native/tests/seal/evaluator.cpp:5590"""
    S, d, o, primes, rng = ckks
    data = rand_ct(rng, primes, 3, 64)
    x = d.ct(data, is_ntt=False)
    with pytest.raises(S.InvalidArgument):
        d.ev.apply_galois_inplace(x, 3, d.glk)
    assert np.array_equal(d.out(x)[0], data)


def test_rotate_scheme_checks(ckks):
    S, d, o, primes, rng = ckks
    x = d.ct(rand_ct(rng, primes, 3, 64))
    with pytest.raises(S.LogicError):        # rotate_rows on CKKS: "unsupported scheme" evaluator.h:1079
        d.ev.rotate_rows_inplace(x, 1, d.glk)
    with pytest.raises(S.InvalidArgument):   # step count too large, galois.cpp:68
        d.ev.rotate_vector_inplace(x, 32, d.glk)


def test_rotate_naf_fallback_without_exact_key(ckks):
    """rotate_internal decomposes the step in NAF form when the exact key is absent
    (evaluator.cpp:2536-2558); a lone power of two without a key is an error."""
    S, d, o, primes, rng = ckks
    x = d.ct(rand_ct(rng, primes, 3, 64))
    with pytest.raises(S.InvalidArgument):
        d.ev.rotate_vector_inplace(x, 2, d.glk)   # NAF(2) has one term and no key for it


def test_relinearize_needs_keys_and_index_range(ckks):
    S, d, o, primes, rng = ckks
    x = d.ct(rand_ct(rng, primes, 3, 64, size=3))
    empty = S.RelinKeys(d.ctx)
    with pytest.raises(S.InvalidArgument):
        d.ev.relinearize_inplace(x, empty)
    with pytest.raises(S.InvalidArgument):   # RelinKeys::get_index: key_power >= 2
        S.RelinKeys.get_index(1)
    with pytest.raises(S.InvalidArgument):   # GaloisKeys::get_index: odd elements only
        S.GaloisKeys.get_index(4)


def test_transparent_check_is_logic_error(ckks):
    """SEAL_THROW_ON_TRANSPARENT_CIPHERTEXT: a result whose c1.. are all zero is refused (evaluator.cpp:386)."""
    S, d, o, primes, rng = ckks
    data = rand_ct(rng, primes, 3, 64)
    data[1] = 0
    x = d.ct(data)
    assert x.is_transparent()
    d.ev.set_transparent_check(True)
    with pytest.raises(S.LogicError):
        d.ev.negate_inplace(x)
    d.ev.set_transparent_check(False)
    d.ev.negate_inplace(x)  # default for device-resident batches: not checked per op


def test_bfv_form_and_scheme_errors(emu):
    S = emu
    n = 32
    primes = coeff_modulus_create(n, [30, 30, 30])
    t = plain_modulus_batching(n, 12)
    d = DeviceSide("bfv", n, primes, t)
    rng = np.random.default_rng(2)
    x = d.ct(rand_ct(rng, primes, 2, n), is_ntt=True)
    with pytest.raises(S.InvalidArgument):   # "encrypted1 or encrypted2 cannot be in NTT form" evaluator.cpp:397
        d.ev.multiply_inplace(x, x.copy())
    y = d.ct(rand_ct(rng, primes, 2, n))
    with pytest.raises(S.InvalidArgument):   # rescale is CKKS-only, evaluator.cpp:1522
        d.ev.rescale_to_next_inplace(y)
    with pytest.raises(S.LogicError):        # rotate_vector is CKKS-only, evaluator.h:1215
        d.ev.rotate_vector_inplace(y, 1, S.GaloisKeys(d.ctx))


def test_invalid_parameters_are_rejected(emu):
    S = emu
    p = S.EncryptionParameters("ckks")
    p.set_poly_modulus_degree(64)
    p.set_coeff_modulus([17])            # 17 != 1 mod 128: no NTT (invalid_coeff_modulus_no_ntt)
    with pytest.raises(S.InvalidArgument):
        S.SEALContext(p)
    p.set_coeff_modulus([1 << 30])       # not prime
    with pytest.raises(S.InvalidArgument):
        S.SEALContext(p)
    p.set_poly_modulus_degree(48)        # not a power of two
    with pytest.raises(S.InvalidArgument):
        S.SEALContext(p)
    with pytest.raises(S.LogicError):    # CKKS has no plain modulus (encryptionparams.h)
        p.set_plain_modulus(65537)


def test_destination_forms_leave_operands_untouched(ckks):
    S, d, o, primes, rng = ckks
    a, b = rand_ct(rng, primes, 3, 64), rand_ct(rng, primes, 3, 64)
    x, y = d.ct(a, scale=2.0 ** 10), d.ct(b, scale=2.0 ** 10)
    dest = S.Ciphertext(d.ctx)
    d.ev.multiply(x, y, dest)
    assert np.array_equal(d.out(x)[0], a) and np.array_equal(d.out(y)[0], b)
    assert np.array_equal(d.out(dest)[0], o.multiply(a, b))
    d.ev.sub(x, y, y)   # destination aliases the second operand (evaluator.h sub())
    q = np.array(primes[:3], dtype=np.uint64)[None, :, None]
    assert np.array_equal(d.out(y)[0], (a + q - b) % q)


# ---- plaintext operands, many-operand forms, digit-parallel halves: the reference's throw sites
#      (native/src/seal/evaluator.cpp:242-261, 1649-1757, 1760-2020, 2196-2230; valcheck.cpp:28-79)
def test_plain_operand_errors(ckks):
    S, d, o, primes, rng = ckks
    pid3, pid2 = d.parms_id_for_K(3), d.parms_id_for_K(2)
    x = d.ct(rand_ct(rng, primes, 3, 64), scale=2.0 ** 20)
    good = S.Plaintext.from_numpy(d.ctx, np.stack([rng.integers(0, primes[i], 64, dtype=np.uint64) for i in range(3)]), pid3, 2.0 ** 20)
    d.ev.add_plain_inplace(x, good)
    with pytest.raises(S.InvalidArgument):   # "CKKS plain must be in NTT form" evaluator.cpp:1793
        d.ev.add_plain_inplace(x, S.Plaintext.from_numpy(d.ctx, np.arange(8, dtype=np.uint64)))
    lower = S.Plaintext.from_numpy(d.ctx, np.zeros((2, 64), dtype=np.uint64), pid2, 2.0 ** 20)
    with pytest.raises(S.InvalidArgument):   # "encrypted and plain parameter mismatch" evaluator.cpp:1797
        d.ev.add_plain_inplace(x, lower)
    other_scale = S.Plaintext.from_numpy(d.ctx, np.zeros((3, 64), dtype=np.uint64), pid3, 2.0 ** 21)
    with pytest.raises(S.InvalidArgument):   # "scale mismatch" evaluator.cpp:1801
        d.ev.sub_plain_inplace(x, other_scale)
    bad_count = S.Plaintext.from_numpy(d.ctx, np.zeros(65, dtype=np.uint64))
    bad_count.set_parms_id(pid3)
    with pytest.raises(S.InvalidArgument):   # is_metadata_valid_for: coeff_count != K*N, valcheck.cpp:56-59
        d.ev.multiply_plain_inplace(x, bad_count)
    with pytest.raises(S.InvalidArgument):   # "plain is already in NTT form" evaluator.cpp:2211
        d.ev.transform_plain_to_ntt_inplace(good, pid3)
    big = S.Plaintext.from_numpy(d.ctx, np.ones((3, 64), dtype=np.uint64), pid3, 2.0 ** 90)
    with pytest.raises(S.InvalidArgument):   # "scale out of bounds" evaluator.cpp:2191
        d.ev.multiply_plain_inplace(x, big)
    # mod switching a plaintext follows the chain and stops at its end (evaluator.cpp:1377-1380)
    p = good.copy()
    d.ev.mod_switch_plain_to_next_inplace(p)
    d.ev.mod_switch_plain_to_next_inplace(p)
    assert p.coeff_count() == 64
    with pytest.raises(S.InvalidArgument):
        d.ev.mod_switch_plain_to_next_inplace(p)


def test_many_operand_errors(ckks):
    S, d, o, primes, rng = ckks
    x = d.ct(rand_ct(rng, primes, 3, 64))
    dest = S.Ciphertext(d.ctx)
    with pytest.raises(S.InvalidArgument):   # "encrypteds cannot be empty" evaluator.cpp:244
        d.ev.add_many([], dest)
    with pytest.raises(S.InvalidArgument):   # "encrypteds must be different from destination" evaluator.cpp:250
        d.ev.add_many([x, x.copy()], x)
    with pytest.raises(S.LogicError):        # multiply_many is BFV/BGV only, evaluator.cpp:1682
        d.ev.multiply_many([x, x.copy()], d.rlk, dest)
    with pytest.raises(S.InvalidArgument):   # "exponent cannot be 0" evaluator.cpp:1743
        d.ev.exponentiate_inplace(x, 0, d.rlk)
    d.ev.exponentiate_inplace(x, 1, d.rlk)    # exponent 1 returns before the scheme check (evaluator.cpp:1747-1750)


def test_digit_parallel_argument_checks(ckks):
    S, d, o, primes, rng = ckks
    x3 = d.ct(rand_ct(rng, primes, 3, 64, size=3))
    words = d.ev.switch_key_acc_words(x3)
    assert words == 2 * (3 + 1) * 64
    acc = S.DeviceBuffer(words)
    with pytest.raises(S.InvalidArgument):   # digit range beyond the K = 3 digits of this level
        d.ev.relinearize_partial(x3, d.rlk, 2, 2, acc.ptr)
    sliced = S.RelinKeys(d.ctx)
    sliced.set_key_digits(0, 1, o.relin_key()[1:2])
    with pytest.raises(S.InvalidArgument):   # the resident key slice [1, 2) does not cover digits [0, 2)
        d.ev.relinearize_partial(x3, sliced, 0, 2, acc.ptr)
    d.ev.relinearize_partial(x3, sliced, 1, 1, acc.ptr)
    with pytest.raises(S.InvalidArgument):   # more than eight partial sums do not fit a 64-bit word
        d.ev.relinearize_finish(x3, acc.ptr, 9)
    x2 = d.ct(rand_ct(rng, primes, 3, 64))
    with pytest.raises(S.InvalidArgument):   # digit-parallel relinearization takes a size-3 ciphertext
        d.ev.relinearize_partial(x2, d.rlk, 0, 3, acc.ptr)


def test_bgv_correction_factor_validation(emu):
    S = emu
    n = 64
    primes = coeff_modulus_create(n, [40, 40, 41])
    t = plain_modulus_batching(n, 13)
    d = DeviceSide("bgv", n, primes, t)
    rng = np.random.default_rng(5)
    x = d.ct(rand_ct(rng, primes, 2, n), is_ntt=True)
    x.set_correction_factor(0)
    with pytest.raises(S.InvalidArgument):   # is_metadata_valid_for: correction factor must be in [1, t), valcheck.cpp:118-123
        d.ev.negate_inplace(x)
    x.set_correction_factor(t)
    with pytest.raises(S.InvalidArgument):
        d.ev.negate_inplace(x)
    y = d.ct(rand_ct(rng, primes, 2, n), is_ntt=False)
    with pytest.raises(S.InvalidArgument):   # "encrypted1 or encrypted2 must be in NTT form" evaluator.cpp:712
        d.ev.multiply_inplace(y, y.copy())


def test_abort_leaves_a_call_stack_behind(emu, tmp_path):
    """SealHip_InstallAbortTrace (sealhip.h, diagnostics.cpp): a process that aborts - here by hand; on a GPU box the ROCm runtime does it on
    a device memory fault - appends the aborting thread's call stack to the named file before it dies, and an exception that reaches
    std::terminate prints its message first (round 3's one aborted process left nothing behind: pytest keeps stderr in a file that dies
    with the process)."""
    import subprocess
    import sys
    trace = tmp_path / "abort_trace.txt"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import os, sys\n"
            "sys.path.insert(0, %r)\n"
            "import seal_amd as S\n"
            "S.load(%r)\n"
            "S.install_abort_trace(%r)\n"
            "os.abort()\n" % (root, os.path.join(root, "tests", "hipemu", "libsealhip_emu.so"), str(trace)))
    run = subprocess.run([sys.executable, "-X", "faulthandler=0", "-c", code], capture_output=True, text=True, timeout=120,
                         env=dict(os.environ, SEALHIP_COMM_NO_RCCL="1"))
    assert run.returncode != 0
    text = trace.read_text()
    assert "sealhip: SIGABRT in process" in text and "end of call stack" in text, text
    # the environment form installs it at load time
    trace2 = tmp_path / "abort_trace_env.txt"
    run = subprocess.run([sys.executable, "-X", "faulthandler=0", "-c", code.replace("S.install_abort_trace(%r)\n" % str(trace), "")],
                         capture_output=True, text=True, timeout=120, env=dict(os.environ, SEALHIP_COMM_NO_RCCL="1", SEALHIP_ABORT_TRACE=str(trace2)))
    assert run.returncode != 0 and "SIGABRT" in trace2.read_text()
