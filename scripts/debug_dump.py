import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np, torch
from oracle import net as onet, updater as oupd
import mocogan_chainer_amd.hiplib as hl, mocogan_chainer_amd.layout as lay, mocogan_chainer_amd.nets as nets, mocogan_chainer_amd.step as step
from test_gpu_step import dev, rel_l2, _f64, noise_to_dev, draw_to_dev
F64 = np.float64
tag = sys.argv[1]
model, dim_zl, nf, n, seed = 'normal', 0, 8, 3, 306
rng = np.random.RandomState(seed)
gen = _f64(onet.init_generator(rng, dim_zl=dim_zl, n_filters=nf)); di = _f64(onet.init_discriminator(rng, 2, 3, 1, nf)); dv = _f64(onet.init_discriminator(rng, 3, 3, 1, nf))
G = nets.GenNet(dim_zl=dim_zl, n_filters=nf); DI = nets.DisNet(2, 3, 1, nf, use_noise=True); DV = nets.DisNet(3, 3, 1, nf, use_noise=True)
G.load_reference_params(gen), DI.load_reference_params(di), DV.load_reference_params(dv)
ts = step.TrainStep(model, G, DI, DV)
def flat(d, pre=''):
    out = {}
    if isinstance(d, torch.Tensor): out[pre] = d
    elif isinstance(d, dict):
        for k, v in d.items(): out.update(flat(v, pre + '/' + str(k)))
    return out
for s in range(2):
    x_real = rng.uniform(-1, 1, (n, 3, 16, 64, 64)); t_real = rng.randint(0, 6, n)
    rnd = oupd.draw_step_randomness(rng, model, n, 3, nf, dim_zl=dim_zl, dtype=F64)
    inject = {'t': rnd['t'], 'gen': draw_to_dev(rnd['gen'])}
    for k in ('noise_i_real', 'noise_v_real', 'noise_i_fake', 'noise_v_fake'):
        inject[k] = noise_to_dev(lay, rnd[k])
    pre = {'DIp': DI.fp.p.clone(), 'DVp': DV.fp.p.clone(), 'Gp': G.fp.p.clone(), 'DIm': DI.fp.m.clone(), 'DIv': DI.fp.v.clone(), 'DVm': DV.fp.m.clone(), 'DVv': DV.fp.v.clone()}
    out = ts.run(dev(x_real), dev(t_real, torch.int32), inject)
rec = {'pre/' + k: v for k, v in pre.items()}
rec.update({'post/DIp': DI.fp.p, 'post/DVp': DV.fp.p, 'post/Gp': G.fp.p, 'DIg': DI.fp.g, 'DVg': DV.fp.g, 'Gg': G.fp.g, 'gx': out['gx_fake'], 'x_fake': out['x_fake']})
for nm in ('saved_gen', 'saved_fake_i', 'saved_fake_v'):
    rec.update({nm + k: v for k, v in flat(out[nm]).items()})
for k in ('y_fake_i', 'y_fake_v', 'y_real_i', 'y_real_v'): rec[k] = out[k]
np.savez_compressed(os.path.join('/tmp/mcgdump', 'dump_%s.npz' % tag), **{k.replace('/', '.'): v.detach().cpu().numpy() for k, v in rec.items() if isinstance(v, torch.Tensor)})
import json
json.dump({'DI': DI.fp.offsets, 'DV': DV.fp.offsets, 'G': G.fp.offsets}, open('/tmp/mcgdump/offsets.json', 'w'))
print('dumped', tag, float(out['gx_fake'].abs().sum()))
