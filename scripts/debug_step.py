"""Debug helper: per-step / per-tensor deviation of the HIP path from the float64 oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
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
for s in range(steps):
    x_real = rng.uniform(-1, 1, (n, 3, 16, 64, 64)); t_real = rng.randint(0, 6, n)
    rnd = oupd.draw_step_randomness(rng, model, n, 3, nf, dim_zl=dim_zl, dtype=F64)
    ref = oupd.update_core(model, gen, di, dv, og, oi, ov, x_real, t_real, rnd, dim_zl=dim_zl, keep=True)
    inject = {'t': rnd['t'], 'gen': draw_to_dev(rnd['gen'])}
    for k in ('noise_i_real', 'noise_v_real', 'noise_i_fake', 'noise_v_fake'):
        inject[k] = noise_to_dev(lay, rnd[k])
    # grads before adam are overwritten; capture by exporting grads right after run (flat g still holds them)
    out = ts.run(dev(x_real), dev(t_real, torch.int32), inject)
    L = ts.losses()
    print('   x_fake err', rel_l2(lay.act_from_dev(out['x_fake'], 3), ref['x_fake'][:, :3]), 'gx_fake err', rel_l2(lay.act_from_dev(out['gx_fake'], 3), ref['gx_fake']), 't', rnd['t'])
    for k in ('y_real_i','y_real_v','y_fake_i','y_fake_v'):
        print('   ', k, rel_l2(out[k], ref[k].reshape(out[k].shape)))
    print('step', s, 'loss diffs', L['image_dis/loss'] - ref['loss_dis_i'], L['video_dis/loss'] - ref['loss_dis_v'], L['image_gen/loss'] - ref['loss_gen'])
    print('   stream', torch.cuda.current_stream().cuda_stream)
    gg = G.export_reference_grads()
    print('   G grad errs', ['%s %.1e' % (k, rel_l2(gg[k], ref['grads_gen'][k])) for k in ('dc5/W','dc4/W','dc3/W','dc1/W','g0/W/W')])
    for name, net, refp, refg in (('DI', DI, di, ref['grads_dis_i']), ('DV', DV, dv, ref['grads_dis_v']), ('G', G, gen, ref['grads_gen'])):
        got = net.export_reference_params(); gg = net.export_reference_grads()
        for k in refp:
            if k.endswith('/N'): continue
            e = rel_l2(got[k], refp[k])
            ge = rel_l2(gg[k], refg[k]) if k in refg else float('nan')
            flag = ' <<<' if e > 1e-5 else ''
            if os.environ.get('VERBOSE'): print('  %-3s %-14s param %.2e  grad %.2e  |g|max %.2e%s' % (name, k, e, ge, np.abs(refg[k]).max() if k in refg else 0, flag))
