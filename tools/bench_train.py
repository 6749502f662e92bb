#!/usr/bin/env python
"""Throughput of the PRODUCT path: train.py's object graph -- model.net networks, trainer.Adam + WeightDecay, model.updater.Updater,
an iterator, trainer.Trainer with its report extensions -- timed over K iterations, next to what bench.py times (TrainStep on a
resident batch).  (/root/reference/train.py:107-183, model/updater.py:78-113.)

    python tools/bench_train.py [--mfma f32] [--batchsize 32] [--data cached|synthetic|jpeg] [--loader_workers 8] [--iters 50]

--data cached    : `--size` clips drawn once, held as uint8 frames (what a decoded video file is): the loader's cost is indexing +
                   the pipe + the H2D copy -- this isolates the Updater / Trainer / iterator overhead from decoding speed
--data synthetic : train.py --dataset_type synthetic (every item is drawn from a NumPy generator in the worker: ~2.5 ms per clip)
--data jpeg      : a MUG-shaped tree of random 64x64 JPEGs under /tmp read by datasets.MugDataset (PIL decode in the workers)
--loader_workers 0 is the reference-style SerialIterator (decode + stack on the training process).
Prints one JSON line per run and appends it to --out (default bench_train.json)."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))


class CachedClips:
    """`size` random clips held as uint8 (T,H,W,C) frames, like a decoded video; get_example_raw hands them out as they are."""

    def __init__(self, size, num_labels=6, channel=3, video_length=16, img_size=64, seed=0):
        rng = np.random.RandomState(seed)
        self.frames = rng.randint(0, 256, (size, video_length, img_size, img_size, channel), dtype=np.uint8)
        self.labels = rng.randint(0, num_labels, size)

    def __len__(self):
        return len(self.frames)

    def get_example_raw(self, i):
        return self.frames[i], int(self.labels[i])

    def get_example(self, i, raw=False):
        if raw:
            return self.get_example_raw(i)
        return ((self.frames[i].astype(np.float32) - 128.) / 128.).transpose(3, 0, 1, 2), int(self.labels[i])

    __getitem__ = get_example


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--mfma', default='f32', choices=['f32', 'bf16', 'f32x3'])
    ap.add_argument('--batchsize', type=int, default=32)
    ap.add_argument('--model', default='normal', choices=['normal', 'cgan', 'infogan'])
    ap.add_argument('--data', default='cached', choices=['cached', 'synthetic', 'jpeg'])
    ap.add_argument('--size', type=int, default=0, help='clips in the dataset (default: max(1024, 16 batches) -- the per-epoch report reads the '
                                                        'losses back, a host synchronisation, so a 4-iteration epoch measures that instead)')
    ap.add_argument('--loader_workers', type=int, default=8)
    ap.add_argument('--iters', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=15)
    ap.add_argument('--overlap', type=int, default=1)
    ap.add_argument('--out', default=os.path.join(ROOT, 'bench_train.json'))
    args = ap.parse_args()

    import torch
    from model.net import ImageGenerator, ImageDiscriminator, VideoDiscriminator
    from model.updater import Updater
    from mocogan_chainer_amd import trainer as T
    import mocogan_chainer_amd.hiplib as hl
    import mocogan_chainer_amd.step as mstep
    if not torch.cuda.is_available():
        raise SystemExit('bench_train.py needs an MI355X')
    hl.set_autotune(True)
    np.random.seed(0)
    if args.size <= 0:
        args.size = max(1024, 16 * args.batchsize)
    num_labels, channel, T_ = 6, 3, 16
    if args.data == 'cached':
        ds = CachedClips(args.size, num_labels, channel, T_)
    elif args.data == 'synthetic':
        from datasets import SyntheticDataset
        ds = SyntheticDataset(args.size, num_labels, channel, T_, 64)
    else:
        import bench_loader
        from datasets import MugDataset
        root = '/tmp/mcg_fake_mug_%d_%d' % (max(args.size // 4, 96), 40)
        bench_loader.make_tree(root, max(args.size // 4, 96), 40)
        ds = MugDataset(root, T_)
    c_d = channel + (num_labels if args.model == 'cgan' else 0)
    out_d = 1 + (num_labels if args.model == 'infogan' else 0)
    gen = ImageGenerator(50, 10, num_labels, channel, 64, T_)
    di = ImageDiscriminator(c_d, out_d, 64, True, 0.2)
    dv = VideoDiscriminator(c_d, out_d, 64, True, 0.2)
    if args.loader_workers > 0:
        it = T.PrefetchIterator(ds, args.batchsize, n_workers=args.loader_workers, seed=0)
    else:
        it = T.SerialIterator(ds, args.batchsize)

    def opt(model):
        o = T.Adam(alpha=2e-4, beta1=5e-5)
        o.setup(model)
        o.add_hook(T.WeightDecay(1e-5), 'hook_dec')
        return o
    updater = Updater(model=args.model, models=(gen, di, dv), video_length=T_, img_size=64, channel=channel, dim_zl=num_labels,
                      iterator=it, tensorboard_writer=T.NullWriter(), optimizer={'image_gen': opt(gen), 'image_dis': opt(di), 'video_dis': opt(dv)},
                      device=0, seed=0, overlap=bool(args.overlap), precision=args.mfma)
    out_dir = '/tmp/mcg_bench_train'

    def run_to(n):
        tr = T.Trainer(updater, (n, 'iteration'), out=out_dir)
        tr.extend(T.extensions.LogReport(trigger=(1, 'epoch')), trigger=(1, 'epoch'))
        tr.extend(T.extensions.PrintReport(['epoch', 'iteration', 'image_gen/loss', 'image_dis/loss', 'video_dis/loss']), trigger=(1, 'epoch'))
        tr.run()
    run_to(args.warmup)
    torch.cuda.synchronize()
    chains0 = mstep.chain_iterations
    t0 = time.perf_counter()
    run_to(args.warmup + args.iters)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    rec = {"path": "train.py object graph: Trainer + Updater.update_core + %s" % type(it).__name__, "data": args.data,
           "loader_workers": args.loader_workers, "dtype": args.mfma, "model": args.model, "batch": args.batchsize, "iters": args.iters,
           "side_streams": bool(args.overlap), "two_chain_iterations": mstep.chain_iterations - chains0,
           "clips_per_s": round(args.batchsize * args.iters / dt, 1), "ms_per_iteration": round(dt / args.iters * 1e3, 3),
           "epochs_seen": updater.epoch, "dataset_clips": args.size}
    print(json.dumps(rec), flush=True)
    try:
        prev = json.load(open(args.out))
    except Exception:
        prev = []
    prev.append(rec)
    with open(args.out, 'w') as f:
        json.dump(prev, f, indent=1)
    if hasattr(it, 'close'):
        it.close()


if __name__ == '__main__':
    main()
