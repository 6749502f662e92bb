#!/usr/bin/env python
"""How far a FREE-RUNNING fp32 HIP trajectory drifts from the float64 oracle: both start from the same parameters
and see the same randomness every iteration, but (unlike the teacher-forced parity tests) the device keeps its own
parameters and Adam state from iteration to iteration.  Prints one line per iteration: |loss differences|, rel-L2 of the
generated clip, rel-L2 of each network's parameters, and the relative size of the parameter UPDATE error.
    python tests/trajectory_drift.py [--iters 20] [--nf 8] [--n 4] [--model infogan]"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))      # repo root (this file lives in tests/)
sys.path.insert(0, ROOT)
from oracle import net as onet, updater as oupd                  # noqa: E402
import mocogan_chainer_amd.hiplib as hl                           # noqa: E402
import mocogan_chainer_amd.layout as lay                          # noqa: E402
import mocogan_chainer_amd.nets as nets                           # noqa: E402
import mocogan_chainer_amd.step as step                           # noqa: E402


def dev(a, dt=torch.float32):
    return torch.tensor(np.asarray(a), dtype=dt, device='cuda')


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--nf', type=int, default=8)
    ap.add_argument('--n', type=int, default=4)
    ap.add_argument('--model', default='infogan')
    ap.add_argument('--seed', type=int, default=3)
    args = ap.parse_args()
    hl.load()
    model, nf, n, dim_zl = args.model, args.nf, args.n, 6
    out_c = 7 if model == 'infogan' else 1
    rng = np.random.RandomState(args.seed)
    f64 = lambda p: {k: (v.astype(np.float64) if v.dtype.kind == 'f' else v) for k, v in p.items()}
    gen, di, dv = f64(onet.init_generator(rng, dim_zl=dim_zl, n_filters=nf)), f64(onet.init_discriminator(rng, 2, 3, out_c, nf)), \
        f64(onet.init_discriminator(rng, 3, 3, out_c, nf))
    G, DI, DV = nets.GenNet(dim_zl=dim_zl, n_filters=nf), nets.DisNet(2, 3, out_c, nf, use_noise=True), nets.DisNet(3, 3, out_c, nf, use_noise=True)
    og, oi, ov = (oupd.new_adam_state(q) for q in (gen, di, dv))
    for net, p, st in ((G, gen, og), (DI, di, oi), (DV, dv, ov)):
        net.load_reference_params(p)
        net.load_adam_state(st)
    ts = step.TrainStep(model, G, DI, DV)
    print('%4s %10s %10s %10s %10s   %9s %9s %9s   %s' % ('iter', '|dL_DI|', '|dL_DV|', '|dL_G|', 'x_fake', 'par G', 'par D_I', 'par D_V', 'min margin'))
    for it in range(args.iters):
        x_real = rng.uniform(-1, 1, (n, 3, 16, 64, 64))
        t_real = rng.randint(0, 6, n)
        rnd = oupd.draw_step_randomness(rng, model, n, 3, nf, dim_zl=dim_zl, dtype=np.float64)
        ref = oupd.update_core(model, gen, di, dv, og, oi, ov, x_real, t_real, rnd, dim_zl=dim_zl, keep=True)
        d = rnd['gen']
        inject = {'t': rnd['t'], 'gen': {'h0': dev(d['h0']), 'e': dev(d['e']), 'zc': dev(d['zc']), 'labels': dev(d['labels'], torch.int32)}}
        for k in ('noise_i_real', 'noise_v_real', 'noise_i_fake', 'noise_v_fake'):
            inject[k] = [lay.act_to_dev(dev(a)) for a in rnd[k]]
        out = ts.run(dev(x_real), dev(t_real, torch.int32), inject)
        l = ts.losses()
        pr = []
        for net, p in ((G, gen), (DI, di), (DV, dv)):
            got = net.export_reference_params()
            num = den = 0.0
            for k, v in p.items():
                if v.dtype.kind != 'f' or 'avg_' in k:
                    continue
                a = np.asarray(got[k].cpu() if torch.is_tensor(got[k]) else got[k], np.float64)
                num += float(((a - v) ** 2).sum())
                den += float((v ** 2).sum())
            pr.append((num / den) ** 0.5)
        print('%4d %10.2e %10.2e %10.2e %10.2e   %9.2e %9.2e %9.2e   %.1e' % (
            it, abs(l['image_dis/loss'] - ref['loss_dis_i']), abs(l['video_dis/loss'] - ref['loss_dis_v']),
            abs(l['image_gen/loss'] - ref['loss_gen']), rel(lay.act_from_dev(out['x_fake'], 3).cpu().numpy(), ref['x_fake'][:, :3]),
            pr[0], pr[1], pr[2], ref['min_margin']), flush=True)


if __name__ == '__main__':
    main()
