"""Detect corruption of the generator's saved forward state between G.forward and G.backward."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np, torch
from oracle import net as onet, updater as oupd
import mocogan_chainer_amd.hiplib as hl, mocogan_chainer_amd.layout as lay, mocogan_chainer_amd.nets as nets, mocogan_chainer_amd.step as step
from test_gpu_step import dev, rel_l2, _f64, noise_to_dev, draw_to_dev
F64 = np.float64
model, dim_zl, nf, n, steps, seed = 'normal', 0, 8, 3, 3, 306
rng = np.random.RandomState(seed)
gen = _f64(onet.init_generator(rng, dim_zl=dim_zl, n_filters=nf)); di = _f64(onet.init_discriminator(rng, 2, 3, 1, nf)); dv = _f64(onet.init_discriminator(rng, 3, 3, 1, nf))
G = nets.GenNet(dim_zl=dim_zl, n_filters=nf); DI = nets.DisNet(2, 3, 1, nf, use_noise=True); DV = nets.DisNet(3, 3, 1, nf, use_noise=True)
G.load_reference_params(gen), DI.load_reference_params(di), DV.load_reference_params(dv)
ts = step.TrainStep(model, G, DI, DV)
og, oi, ov = (oupd.new_adam_state(q) for q in (gen, di, dv))
def flat(d, pre=''):
    out = {}
    if isinstance(d, torch.Tensor): out[pre] = d
    elif isinstance(d, dict):
        for k, v in d.items(): out.update(flat(v, pre + '/' + str(k)))
    return out
stash = {}
of, ob = G.forward, G.backward
def fwd(n_, draw, **kw):
    x, saved = of(n_, draw, **kw)
    stash['clones'] = {k: v.clone() for k, v in flat(saved).items()}
    stash['p'] = G.fp.p.clone()
    return x, saved
def bwd(saved, gx):
    torch.cuda.synchronize()
    for k, v in flat(saved).items():
        c = stash['clones'][k]
        if not torch.equal(v, c):
            d = (v != c)
            idx = d.reshape(-1).nonzero().reshape(-1)
            print('   CORRUPTED saved%s shape %s: %d elems differ, first flat idx %d..%d ptr %x' % (k, tuple(v.shape), int(d.sum()), int(idx[0]), int(idx[-1]), v.data_ptr()))
    if not torch.equal(G.fp.p, stash['p']): print('   G params changed between fwd and bwd!')
    return ob(saved, gx)
G.forward, G.backward = fwd, bwd
for s in range(steps):
    x_real = rng.uniform(-1, 1, (n, 3, 16, 64, 64)); t_real = rng.randint(0, 6, n)
    rnd = oupd.draw_step_randomness(rng, model, n, 3, nf, dim_zl=dim_zl, dtype=F64)
    ref = oupd.update_core(model, gen, di, dv, og, oi, ov, x_real, t_real, rnd, dim_zl=dim_zl, keep=True)
    inject = {'t': rnd['t'], 'gen': draw_to_dev(rnd['gen'])}
    for k in ('noise_i_real', 'noise_v_real', 'noise_i_fake', 'noise_v_fake'):
        inject[k] = noise_to_dev(lay, rnd[k])
    out = ts.run(dev(x_real), dev(t_real, torch.int32), inject)
    gg = G.export_reference_grads()
    for name, net, refp, refg in (('DI', DI, di, ref['grads_dis_i']), ('DV', DV, dv, ref['grads_dis_v']), ('G', G, gen, ref['grads_gen'])):
        got = net.export_reference_params(); ggot = net.export_reference_grads()
        for k in refg:
            if k.startswith('dc') and k.endswith('/b') and k not in ('dc5/b',) and not (name != 'G' and k == 'dc1/b'): continue
            d = np.abs(got[k] - refp[k]).reshape(-1)
            idx = np.argsort(-d)[:3]
            for i in idx:
                if d[i] > 2e-6:
                    print('   step %d %s %s[%d] param diff %.2e  oracle g %.3e device g %.3e' % (s, name, k, i, d[i], refg[k].reshape(-1)[i], ggot[k].reshape(-1)[i]))
    print('step', s, 'CHK oracle %.10e device %.10e gx oracle %.10e device %.10e' % (np.abs(ref['grads_gen']['dc5/W']).sum(), np.abs(gg['dc5/W']).sum(), np.abs(ref['gx_fake']).sum(), float(out['gx_fake'].abs().sum())))
    print('step', s, 'G grad errs', ['%s %.1e' % (k, rel_l2(gg[k], ref['grads_gen'][k])) for k in ('dc5/W','dc4/W','dc3/W','dc1/W','g0/W/W')])
