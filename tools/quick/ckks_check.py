import sys, time
import os; ROOT=os.path.join(os.path.dirname(os.path.abspath(__file__)),'..','..'); sys.path.insert(0,os.path.join(ROOT,'tests')); sys.path.insert(0,ROOT); os.chdir(ROOT)
import numpy as np, sealref as R
import seal_amd as S
if len(sys.argv) > 1 and sys.argv[1] == 'emu':
    S.load('tests/hipemu/libsealhip_emu.so')
def rand_ct(rng, primes, K, n, B, size=2):
    return np.stack([np.stack([np.stack([rng.integers(0,primes[i],n,dtype=np.uint64) for i in range(K)]) for _ in range(B)]) for _ in range(size)])
def run(n, bits, B=2):
    pr = R.coeff_modulus_create(n, bits)
    ref = R.RefContext('ckks', n, pr)
    ref.keygen_relin(); ref.keygen_galois_steps([1, -2]);
    p = S.EncryptionParameters('ckks'); p.set_poly_modulus_degree(n); p.set_coeff_modulus(pr)
    ctx = S.SEALContext(p); ev = S.Evaluator(ctx)
    L=len(pr); K=L-1
    rlk = S.RelinKeys(ctx); rlk.set_key(0, ref.key('relin', 0))
    glk = S.GaloisKeys(ctx)
    for step in [1,-2]:
        elt = ref.galois_elt_from_step(step); assert elt == ctx.galois_elt_from_step(step)
        glk.set_key(S.GaloisKeys.get_index(elt), ref.key('galois', S.GaloisKeys.get_index(elt)))
    rng = np.random.default_rng(5)
    a = rand_ct(rng, pr, K, n, B); b = rand_ct(rng, pr, K, n, B)
    scale = 2.0**(bits[-2]//2-1)
    ca = S.Ciphertext.from_numpy(ctx, a, ctx.first_parms_id(), True, scale)
    cb = S.Ciphertext.from_numpy(ctx, b, ctx.first_parms_id(), True, scale)
    ras = [ref.ct(ref.first_chain_index, a[:,i], True, scale) for i in range(B)]
    rbs = [ref.ct(ref.first_chain_index, b[:,i], True, scale) for i in range(B)]
    def cmp(tag, c, rs):
        g = c.to_numpy()
        ok = all(np.array_equal(g[:,i], rs[i].data()) for i in range(B))
        info = rs[0].info()
        meta = (c.size()==info['size'] and c.coeff_modulus_size()==info['coeff_modulus_size'] and c.is_ntt_form()==info['is_ntt_form'] and c.scale()==info['scale'] and c.chain_index()==info['chain_index'])
        print(n, tag, 'data', ok, 'meta', meta, flush=True)
        return ok and meta
    ok = True
    ev.multiply_inplace(ca, cb); [ref.multiply_inplace(x,y) for x,y in zip(ras,rbs)]; ok &= cmp('multiply', ca, ras)
    ev.relinearize_inplace(ca, rlk); [ref.relinearize_inplace(x) for x in ras]; ok &= cmp('relinearize', ca, ras)
    ev.rescale_to_next_inplace(ca); [ref.rescale_to_next_inplace(x) for x in ras]; ok &= cmp('rescale', ca, ras)
    ev.rotate_vector_inplace(ca, 1, glk); [ref.rotate_vector_inplace(x,1) for x in ras]; ok &= cmp('rotate(1)', ca, ras)
    ev.rotate_vector_inplace(ca, -2, glk); [ref.rotate_vector_inplace(x,-2) for x in ras]; ok &= cmp('rotate(-2)', ca, ras)
    if K > 2:
        ev.mod_switch_to_next_inplace(ca); [ref.mod_switch_to_next_inplace(x) for x in ras]; ok &= cmp('mod_switch', ca, ras)
    ev.square_inplace(ca); [ref.square_inplace(x) for x in ras]; ok &= cmp('square', ca, ras)
    ev.transform_from_ntt_inplace(cb); [ref.transform_from_ntt_inplace(x) for x in rbs]; ok &= cmp('from_ntt', cb, rbs)
    ev.transform_to_ntt_inplace(cb); [ref.transform_to_ntt_inplace(x) for x in rbs]; ok &= cmp('to_ntt', cb, rbs)
    return ok
allok = True
for n, bits in [(16,[30,30,30,30]), (1024,[50,40,40,50]), (4096,[60,40,40,40,60])] + ([(16384,[60,50,50,50,60])] if 'big' in sys.argv else []):
    t=time.time(); allok &= run(n,bits); print('  time', time.time()-t)
print('ALL OK' if allok else 'FAILURES')
