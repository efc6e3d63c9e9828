import sys, time, os
ROOT=os.path.join(os.path.dirname(os.path.abspath(__file__)),'..','..'); sys.path.insert(0,os.path.join(ROOT,'tests')); sys.path.insert(0,ROOT); os.chdir(ROOT)
import numpy as np, sealref as R
import seal_amd as S
if len(sys.argv) > 1 and sys.argv[1] == 'emu':
    S.load('tests/hipemu/libsealhip_emu.so')
def rand_ct(rng, primes, K, n, B, size=2):
    return np.stack([np.stack([np.stack([rng.integers(0,primes[i],n,dtype=np.uint64) for i in range(K)]) for _ in range(B)]) for _ in range(size)])
def run(n, pr, t, B=2):
    ref = R.RefContext('bfv', n, pr, t)
    ref.keygen_relin(); ref.keygen_galois_elts([ref.galois_elt_from_step(1), 2*n-1])
    p = S.EncryptionParameters('bfv'); p.set_poly_modulus_degree(n); p.set_coeff_modulus(pr); p.set_plain_modulus(t)
    ctx = S.SEALContext(p); ev = S.Evaluator(ctx)
    L=len(pr); K=L-1
    fc = ref.first_chain_index
    bsk, mt, gamma = ref.behz_bases(fc)
    assert bsk == ctx.base_bsk(fc), (bsk, ctx.base_bsk(fc))
    rlk = S.RelinKeys(ctx); rlk.set_key(0, ref.key('relin', 0))
    glk = S.GaloisKeys(ctx)
    for elt in [ref.galois_elt_from_step(1), 2*n-1]:
        glk.set_key(S.GaloisKeys.get_index(elt), ref.key('galois', S.GaloisKeys.get_index(elt)))
    rng = np.random.default_rng(7)
    a = rand_ct(rng, pr, K, n, B); b = rand_ct(rng, pr, K, n, B)
    ca = S.Ciphertext.from_numpy(ctx, a, ctx.first_parms_id(), False)
    cb = S.Ciphertext.from_numpy(ctx, b, ctx.first_parms_id(), False)
    ras = [ref.ct(fc, a[:,i], False) for i in range(B)]
    rbs = [ref.ct(fc, b[:,i], False) for i in range(B)]
    def cmp(tag, c, rs):
        g = c.to_numpy()
        ok = all(np.array_equal(g[:,i], rs[i].data()) for i in range(B))
        info = rs[0].info()
        meta = (c.size()==info['size'] and c.coeff_modulus_size()==info['coeff_modulus_size'] and c.is_ntt_form()==info['is_ntt_form'] and c.scale()==info['scale'] and c.chain_index()==info['chain_index'])
        print(n, tag, 'data', ok, 'meta', meta, flush=True)
        return ok and meta
    ok = True
    # RNS stages
    nBsk = len(bsk)
    x = a[0]  # [B][K][n]
    exp0 = np.stack([ref.rns_stage(fc, 'fastbconv_m_tilde', x[i], nBsk+1) for i in range(B)])
    src = S.DeviceBuffer.from_numpy(x); dst = S.DeviceBuffer(B*(nBsk+1)*n)
    S.rns_stage(ctx, fc, 'fastbconv_m_tilde', src, dst, B); got0 = dst.to_numpy(exp0.shape); print(n,'fastbconv_m_tilde', np.array_equal(got0,exp0)); ok &= np.array_equal(got0,exp0)
    exp1 = np.stack([ref.rns_stage(fc, 'sm_mrq', exp0[i], nBsk) for i in range(B)])
    src = S.DeviceBuffer.from_numpy(exp0); dst = S.DeviceBuffer(B*nBsk*n)
    S.rns_stage(ctx, fc, 'sm_mrq', src, dst, B); got1 = dst.to_numpy(exp1.shape); print(n,'sm_mrq', np.array_equal(got1,exp1)); ok &= np.array_equal(got1,exp1)
    qb = np.concatenate([x, exp1], axis=1)
    exp2 = np.stack([ref.rns_stage(fc, 'fast_floor', qb[i], nBsk) for i in range(B)])
    src = S.DeviceBuffer.from_numpy(qb); dst = S.DeviceBuffer(B*nBsk*n)
    S.rns_stage(ctx, fc, 'fast_floor', src, dst, B); got2 = dst.to_numpy(exp2.shape); print(n,'fast_floor', np.array_equal(got2,exp2)); ok &= np.array_equal(got2,exp2)
    exp3 = np.stack([ref.rns_stage(fc, 'fastbconv_sk', exp2[i], K) for i in range(B)])
    src = S.DeviceBuffer.from_numpy(exp2); dst = S.DeviceBuffer(B*K*n)
    S.rns_stage(ctx, fc, 'fastbconv_sk', src, dst, B); got3 = dst.to_numpy(exp3.shape); print(n,'fastbconv_sk', np.array_equal(got3,exp3)); ok &= np.array_equal(got3,exp3)
    if K >= 2:
        for which in ['divide_and_round_q_last', 'divide_and_round_q_last_ntt']:
            e = np.stack([ref.rns_stage(fc, which, x[i], K)[:K-1] for i in range(B)])
            src = S.DeviceBuffer.from_numpy(x); dst = S.DeviceBuffer(B*(K-1)*n)
            S.rns_stage(ctx, fc, which, src, dst, B); g = dst.to_numpy(e.shape); print(n, which, np.array_equal(g,e)); ok &= np.array_equal(g,e)
    ev.multiply_inplace(ca, cb); [ref.multiply_inplace(x,y) for x,y in zip(ras,rbs)]; ok &= cmp('multiply', ca, ras)
    ev.relinearize_inplace(ca, rlk); [ref.relinearize_inplace(x) for x in ras]; ok &= cmp('relinearize', ca, ras)
    ev.rotate_rows_inplace(ca, 1, glk); [ref.rotate_rows_inplace(x,1) for x in ras]; ok &= cmp('rotate_rows(1)', ca, ras)
    ev.rotate_columns_inplace(ca, glk); [ref.rotate_columns_inplace(x) for x in ras]; ok &= cmp('rotate_columns', ca, ras)
    if K >= 2:
        ev.mod_switch_to_next_inplace(ca); [ref.mod_switch_to_next_inplace(x) for x in ras]; ok &= cmp('mod_switch', ca, ras)
    ev.square_inplace(ca); [ref.square_inplace(x) for x in ras]; ok &= cmp('square', ca, ras)
    ev.multiply_inplace(ca, ca.copy()); [ref.multiply_inplace(x, x.copy()) for x in ras]; ok &= cmp('multiply 3x3', ca, ras)
    ev.add_inplace(cb, cb.copy()); [ref.add_inplace(x, x.copy()) for x in rbs]; ok &= cmp('add', cb, rbs)
    ev.negate_inplace(cb); [ref.negate_inplace(x) for x in rbs]; ok &= cmp('negate', cb, rbs)
    return ok
allok = True
cases = [(16, R.coeff_modulus_create(16,[30,30,30,30]), R.plain_modulus_batching(16, 12)),
         (4096, R.bfv_default(4096), R.plain_modulus_batching(4096, 20))]
if 'big' in sys.argv:
    cases.append((8192, R.coeff_modulus_create(8192,[55]*6), R.plain_modulus_batching(8192, 20)))
for n, pr, t in cases:
    t0=time.time(); allok &= run(n,pr,t); print('  time', time.time()-t0)
print('ALL OK' if allok else 'FAILURES')
