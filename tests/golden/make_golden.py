"""Generates tests/golden/*.npz from the float64 oracle (oracle/).  The reference itself cannot run
here (Chainer 3.1.0 is not installed / installable: SURVEY 8c), so these vectors pin the oracle
against drift and give the GPU tests a committed target; they are NOT outputs of the reference.

    python tests/golden/make_golden.py

Inputs are regenerated from np.random.RandomState(seed) (legacy generator: bit-stable across NumPy
versions), so only the expected outputs are stored.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import functions as F          # noqa: E402
from oracle import net as onet             # noqa: E402
from oracle import updater as oupd         # noqa: E402
from oracle import philox                  # noqa: E402

F64 = np.float64
# Seeds chosen by `python tests/golden/make_golden.py --search [model:dim_zl]` (round 6): per variant the first seed >= the round-1 seed
# whose THREE iterations all keep every pre-activation at least MARGIN away from its kink (leaky_relu / relu at 0) -- the gradient
# is discontinuous there, and an fp32 device result on the other side of a kink differs from the float64 oracle by a whole
# activation slope, not by rounding.  With such seeds the GPU test holds every iteration to the tight gradient tolerances
# (tests/test_gpu_golden.py); rounds 1-5 had a 0.15 fallback for the four iterations whose margin was below 2e-6.
# Searched: normal/0 303..1267, normal/6 311..3342, cgan/6 320 (the round-1 seed qualifies); infogan/6 (seven logits, more
# activations: one iteration in ten clears 3e-6) has no seed below 4313 at 3e-6 -- 481 is its first at 2.3e-6 (margins 4.6e-6,
# 2.4e-6, 3.6e-6), still above the 2e-6 the device's rounding needs.
MARGIN = {"normal": 3e-6, "cgan": 3e-6, "infogan": 2.3e-6}
STEP_CASES = [("normal", 0, 1267), ("normal", 6, 3342), ("infogan", 6, 481), ("cgan", 6, 320)]
SEARCH_FROM = {("normal", 0): 303, ("normal", 6): 311, ("infogan", 6): 313, ("cgan", 6): 320}


def f64(p):
    return {k: (v.astype(F64) if v.dtype.kind == 'f' else v) for k, v in p.items()}


def step_inputs(model, dim_zl, seed, nf=4, n=2):
    """The exact construction the step tests use (tests/test_gpu_step.py::_run_steps)."""
    rng = np.random.RandomState(seed)
    out_c = 7 if model == 'infogan' else 1
    c_d = 3 + (dim_zl if model == 'cgan' else 0)
    gen = f64(onet.init_generator(rng, dim_zl=dim_zl, n_filters=nf))
    di = f64(onet.init_discriminator(rng, 2, c_d, out_c, nf))
    dv = f64(onet.init_discriminator(rng, 3, c_d, out_c, nf))
    return rng, gen, di, dv


def step_case(model, dim_zl, seed, steps=3, nf=4, n=2, stop_below=None):
    """stop_below (the seed search): give up -- return None -- at the first iteration whose margin is below it"""
    rng, gen, di, dv = step_inputs(model, dim_zl, seed, nf, n)
    og, oi, ov = (oupd.new_adam_state(q) for q in (gen, di, dv))
    rec = {}
    for s in range(steps):
        x_real = rng.uniform(-1, 1, (n, 3, 16, 64, 64))
        t_real = rng.randint(0, 6, n)
        rnd = oupd.draw_step_randomness(rng, model, n, 3, nf, dim_zl=dim_zl, dtype=F64)
        ref = oupd.update_core(model, gen, di, dv, og, oi, ov, x_real, t_real, rnd, dim_zl=dim_zl, keep=True)
        rec['s%d/losses' % s] = np.array([ref['loss_dis_i'], ref['loss_dis_v'], ref['loss_gen']])
        rec['s%d/min_margin' % s] = np.array(ref['min_margin'])
        if stop_below is not None and float(ref['min_margin']) < stop_below:
            return None
        rec['s%d/t' % s] = np.array(rnd['t'])
        rec['s%d/x_fake_slice' % s] = ref['x_fake'][:, :3, ::5, ::16, ::16].copy()
        rec['s%d/gx_fake_slice' % s] = ref['gx_fake'][:, :, ::5, ::16, ::16].copy()
        for name, g in (('di', ref['grads_dis_i']), ('dv', ref['grads_dis_v']), ('g', ref['grads_gen'])):
            rec['s%d/gradnorm_%s' % (s, name)] = np.array([np.linalg.norm(g[k]) for k in sorted(g)])
        for name, p in (('di', di), ('dv', dv), ('g', gen)):
            rec['s%d/paramnorm_%s' % (s, name)] = np.array([np.linalg.norm(p[k]) for k in sorted(p) if not k.endswith('/N')])
    return rec


def op_cases():
    rng = np.random.RandomState(2024)
    rec = {}
    x = rng.uniform(-1, 1, (2, 3, 7, 16, 16))
    W = rng.randn(8, 3, 4, 4, 4) * 0.1
    b = rng.randn(8)
    y = F.conv3d_fwd(x, W, b, (1, 2, 2), (0, 1, 1))
    gy = rng.randn(*y.shape)
    gx, gW, gb = F.conv3d_bwd(x, W, gy, (1, 2, 2), (0, 1, 1))
    rec.update({'conv3d/y': y, 'conv3d/gx': gx, 'conv3d/gW': gW, 'conv3d/gb': gb})
    xd = rng.randn(3, 8, 4, 4)
    Wd = rng.randn(8, 3, 4, 4) * 0.1
    bd = rng.randn(3)
    yd = F.deconv2d_fwd(xd, Wd, bd, 2, 1)
    gyd = rng.randn(*yd.shape)
    gxd, gWd, gbd = F.deconv2d_bwd(xd, Wd, gyd, 2, 1)
    rec.update({'deconv2d/y': yd, 'deconv2d/gx': gxd, 'deconv2d/gW': gWd, 'deconv2d/gb': gbd})
    xb = rng.randn(4, 8, 3, 5, 5) * 2 + 1
    gamma, beta = 1 + 0.1 * rng.randn(8), 0.1 * rng.randn(8)
    am, av = np.zeros(8), np.ones(8)
    yb, cache = F.bn_train_fwd(xb, gamma, beta, am, av)
    gxb, gg, gbb = F.bn_train_bwd(cache, gamma, rng.randn(*yb.shape))
    rec.update({'bn/y': yb, 'bn/avg_mean': am, 'bn/avg_var': av, 'bn/gx': gxb, 'bn/ggamma': gg, 'bn/gbeta': gbb})
    p = {k[3:]: v.astype(F64) for k, v in onet.init_generator(rng, dim_zl=6, n_filters=2).items() if k.startswith('g0/')}
    h, xx = rng.randn(3, 10), rng.randn(3, 16)
    h1, c = F.gru_step_fwd(p, h, xx)
    grads = {k: np.zeros_like(v) for k, v in p.items()}
    gh, gxx = F.gru_step_bwd(p, c, rng.randn(3, 10), grads)
    rec.update({'gru/h1': h1, 'gru/gh': gh, 'gru/gx': gxx, 'gru/gW_r': grads['W_r/W'], 'gru/gU': grads['U/W']})
    rec['philox/randn'] = philox.randn(64, 0.2, 0x1234567887654321, 42)
    # Random123 known-answer vectors for Philox4x32-10 (published with the library)
    rec['philox/kat'] = np.array([philox.philox4x32_10([c0], [c1], [c2], [c3], k0, k1) for c0, c1, c2, c3, k0, k1 in
                                  ((0, 0, 0, 0, 0, 0), (0xffffffff,) * 6,
                                   (0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344, 0xa4093822, 0x299f31d0))],
                                 dtype=np.uint32).reshape(3, 4)
    return rec


def search(only=None, limit=4000):
    """prints, per variant (or for the one named `model:dim_zl`), the first seed >= SEARCH_FROM whose three iterations all clear MARGIN[model]"""
    found = []
    for (model, dim_zl), first in sorted(SEARCH_FROM.items()):
        if only and only != '%s:%d' % (model, dim_zl):
            continue
        for seed in range(first, first + limit):
            rec = step_case(model, dim_zl, seed, stop_below=MARGIN[model])
            if rec is not None:
                print(model, dim_zl, seed, ['%.2e' % float(rec['s%d/min_margin' % s]) for s in range(3)], flush=True)
                found.append((model, dim_zl, seed))
                break
        else:
            print(model, dim_zl, 'no seed below', first + limit, flush=True)
    print('STEP_CASES +=', found, flush=True)


def main():
    if '--search' in sys.argv:                                       # [--search model:dim_zl] one variant (run the four side by side)
        i = sys.argv.index('--search')
        return search(sys.argv[i + 1] if len(sys.argv) > i + 1 else None)
    for f in os.listdir(HERE):                                       # (fixtures of seeds that are no longer in STEP_CASES)
        if f.startswith('step_') and f.endswith('.npz'):
            os.remove(os.path.join(HERE, f))
    np.savez_compressed(os.path.join(HERE, 'ops.npz'), **{k.replace('/', '.'): v for k, v in op_cases().items()})
    for model, dim_zl, seed in STEP_CASES:
        rec = step_case(model, dim_zl, seed)
        np.savez_compressed(os.path.join(HERE, 'step_%s_zl%d_seed%d.npz' % (model, dim_zl, seed)),
                            **{k.replace('/', '.'): v for k, v in rec.items()})
    print('written to', HERE)


if __name__ == '__main__':
    main()
