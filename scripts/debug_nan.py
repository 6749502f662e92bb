"""Poison every torch.empty with NaN: any kernel that reads memory it (or a predecessor) did not write shows up."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np, torch
_empty, _empty_like = torch.empty, torch.empty_like
def empty(*a, **k):
    t = _empty(*a, **k)
    if t.is_floating_point(): t.fill_(float('nan'))
    return t
def empty_like(*a, **k):
    t = _empty_like(*a, **k)
    if t.is_floating_point(): t.fill_(float('nan'))
    return t
torch.empty, torch.empty_like = empty, empty_like
from oracle import net as onet, updater as oupd
import mocogan_chainer_amd.hiplib as hl, mocogan_chainer_amd.layout as lay, mocogan_chainer_amd.nets as nets, mocogan_chainer_amd.step as step
from test_gpu_step import dev, rel_l2, _f64, noise_to_dev, draw_to_dev
F64 = np.float64
model, dim_zl, nf, n, steps, seed = 'normal', 0, 8, 3, 2, 306
rng = np.random.RandomState(seed)
gen = _f64(onet.init_generator(rng, dim_zl=dim_zl, n_filters=nf)); di = _f64(onet.init_discriminator(rng, 2, 3, 1, nf)); dv = _f64(onet.init_discriminator(rng, 3, 3, 1, nf))
G = nets.GenNet(dim_zl=dim_zl, n_filters=nf); DI = nets.DisNet(2, 3, 1, nf, use_noise=True); DV = nets.DisNet(3, 3, 1, nf, use_noise=True)
for netx in (G, DI, DV):
    netx.ws.fill_(float('nan'))
G.load_reference_params(gen), DI.load_reference_params(di), DV.load_reference_params(dv)
ts = step.TrainStep(model, G, DI, DV)
og, oi, ov = (oupd.new_adam_state(q) for q in (gen, di, dv))
def nan_report(tag, d, depth=0):
    if isinstance(d, torch.Tensor):
        if d.is_floating_point():
            c = int(torch.isnan(d).sum())
            if c: print('   NaN in', tag, tuple(d.shape), c)
    elif isinstance(d, dict):
        for k, v in d.items(): nan_report('%s[%s]' % (tag, k), v)
    elif isinstance(d, (list, tuple)):
        for i, v in enumerate(d): nan_report('%s[%d]' % (tag, i), v)
for s in range(steps):
    x_real = rng.uniform(-1, 1, (n, 3, 16, 64, 64)); t_real = rng.randint(0, 6, n)
    rnd = oupd.draw_step_randomness(rng, model, n, 3, nf, dim_zl=dim_zl, dtype=F64)
    ref = oupd.update_core(model, gen, di, dv, og, oi, ov, x_real, t_real, rnd, dim_zl=dim_zl, keep=True)
    inject = {'t': rnd['t'], 'gen': draw_to_dev(rnd['gen'])}
    for k in ('noise_i_real', 'noise_v_real', 'noise_i_fake', 'noise_v_fake'):
        inject[k] = noise_to_dev(lay, rnd[k])
    out = ts.run(dev(x_real), dev(t_real, torch.int32), inject)
    print('step', s, ts.losses())
    nan_report('out', out)
    for name, net in (('G', G), ('DI', DI), ('DV', DV)):
        nan_report(name + '.g', net.fp.g); nan_report(name + '.p', net.fp.p)
    gg = G.export_reference_grads()
    for k in ('dc5/W', 'dc3/W', 'bn2/gamma', 'g0/W/W'):
        print('   ', k, rel_l2(gg[k], ref['grads_gen'][k]))
