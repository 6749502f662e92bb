#!/usr/bin/env python
"""Where the HOST time of a product-path iteration goes (tools/bench_train.py's object graph, cached uint8 clips): seconds spent in
PrefetchIterator.next_device_batch and in TrainStep.run (enqueue only) per iteration, against the wall time of the iteration.
    python tools/diag_train_host.py [--mfma bf16] [--batchsize 256] [--iters 40]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--mfma', default='bf16')
    ap.add_argument('--batchsize', type=int, default=256)
    ap.add_argument('--iters', type=int, default=40)
    ap.add_argument('--loader_workers', type=int, default=8)
    args = ap.parse_args()
    import torch
    from bench_train import CachedClips
    from model.net import ImageGenerator, ImageDiscriminator, VideoDiscriminator
    from model.updater import Updater
    from mocogan_chainer_amd import trainer as T
    import mocogan_chainer_amd.hiplib as hl
    hl.set_autotune(True)
    np.random.seed(0)
    ds = CachedClips(max(1024, 16 * args.batchsize), 6, 3, 16)
    gen, di, dv = ImageGenerator(50, 10, 6, 3, 64, 16), ImageDiscriminator(3, 1, 64, True, 0.2), VideoDiscriminator(3, 1, 64, True, 0.2)
    it = T.PrefetchIterator(ds, args.batchsize, n_workers=args.loader_workers, seed=0)

    def opt(model):
        o = T.Adam(alpha=2e-4, beta1=5e-5)
        o.setup(model)
        o.add_hook(T.WeightDecay(1e-5), 'hook_dec')
        return o
    up = Updater(model='normal', models=(gen, di, dv), video_length=16, img_size=64, channel=3, dim_zl=6, iterator=it,
                 tensorboard_writer=T.NullWriter(), optimizer={'image_gen': opt(gen), 'image_dis': opt(di), 'video_dis': opt(dv)},
                 device=0, seed=0, overlap=True, precision=args.mfma)
    acc = {'loader': 0.0, 'run': 0.0}
    nd, run = it.next_device_batch, up._step.run

    def nd_t(*a, **k):
        t0 = time.perf_counter()
        r = nd(*a, **k)
        acc['loader'] += time.perf_counter() - t0
        return r

    def run_t(*a, **k):
        t0 = time.perf_counter()
        r = run(*a, **k)
        acc['run'] += time.perf_counter() - t0
        return r
    it.next_device_batch, up._step.run = nd_t, run_t
    for _ in range(15):
        up.update()
    torch.cuda.synchronize()
    acc['loader'] = acc['run'] = 0.0
    prof = None
    if os.environ.get('MCG_DIAG_PROFILE') == '1':
        import cProfile
        prof = cProfile.Profile()
        prof.enable()
    t0 = time.perf_counter()
    for _ in range(args.iters):
        up.update()
    host = time.perf_counter() - t0
    if prof is not None:
        prof.disable()
        import pstats
        st = pstats.Stats(prof).sort_stats('tottime')
        st.print_stats(12)
        st.print_callers('synchronize')
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    n = args.iters
    print('%s batch %d: wall %.2f ms/iter (%.0f clips/s); host loop %.2f ms/iter of which next_device_batch %.2f, TrainStep.run %.2f'
          % (args.mfma, args.batchsize, wall / n * 1e3, args.batchsize * n / wall, host / n * 1e3, acc['loader'] / n * 1e3, acc['run'] / n * 1e3))
    it.close()


if __name__ == '__main__':
    main()
