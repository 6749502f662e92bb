import sys, os, copy
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np, torch
from oracle import net as onet, updater as oupd
import mocogan_chainer_amd.hiplib as hl, mocogan_chainer_amd.layout as lay, mocogan_chainer_amd.nets as nets, mocogan_chainer_amd.step as step
from test_gpu_step import dev, rel_l2, _f64, _perturb, draw_to_dev
F64 = np.float64
rng = np.random.RandomState(5)
n, nf, dim_zl = 3, 8, 0
p = _f64(onet.init_generator(rng, dim_zl=dim_zl, n_filters=nf))
if len(sys.argv) > 1 and sys.argv[1] == 'perturb':
    p = _perturb(p, rng)
g = nets.GenNet(dim_zl=dim_zl, n_filters=nf)
g.load_reference_params(p)
st = oupd.new_adam_state(p)
for rnd in range(3):
    draw = onet.gen_draw(rng, n, dim_zl=dim_zl, dtype=F64)
    x_ref, _, cache = onet.gen_forward(p, draw)
    gx = rng.randn(*x_ref.shape)
    grads = oupd.zero_grads(p)
    onet.gen_backward(p, cache, gx, grads)
    xd, saved = g.forward(n, draw_to_dev(draw))
    g.zero_grad()
    g.backward(saved, lay.act_to_dev(dev(gx.transpose(1, 2, 0, 3, 4))))
    got = g.export_reference_grads()
    print('round', rnd, 'x err', rel_l2(lay.act_from_dev(xd, 3), x_ref.transpose(1, 2, 0, 3, 4)))
    for k in grads:
        print('   %-12s %.2e' % (k, rel_l2(got[k], grads[k])))
    oupd.adam_wd_update(p, grads, st)
    step.adam_update(g, step.AdamHyper())
