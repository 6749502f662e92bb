#!/usr/bin/env python
"""Input-pipeline throughput: builds a MUG-shaped tree of random 64x64 JPEGs under /tmp and
times the reference-style SerialIterator against trainer.PrefetchIterator (CPU only, no GPU needed).
    python tools/bench_loader.py [--videos 192] [--frames 40] [--batch 32] [--workers 8]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def make_tree(root, videos, frames):
    from PIL import Image
    rng = np.random.RandomState(0)
    cats = ["anger", "disgust", "happiness", "fear", "sadness", "surprise"]
    for v in range(videos):
        d = os.path.join(root, cats[v % 6], "%04d" % v)
        if os.path.isdir(d):
            continue
        os.makedirs(d)
        base = rng.randint(0, 255, (64, 64, 3))
        for f in range(frames):
            img = np.clip(base + rng.randint(-20, 20, (64, 64, 3)), 0, 255).astype(np.uint8)
            Image.fromarray(img).save(os.path.join(d, "%03d.jpg" % f), quality=90)


def rate(it, batches, batch, raw=False):
    """raw=True times PrefetchIterator._pop(): the uint8 batch that next_device_batch() hands to the GPU (the float
    conversion and transpose then run on the device); raw=False times next(), the reference-style float list."""
    step = it._pop if raw else it.next
    step()                                      # first batch: worker start-up / file cache
    t0 = time.perf_counter()
    for _ in range(batches):
        step()
    return batches * batch / (time.perf_counter() - t0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--videos', type=int, default=192)
    ap.add_argument('--frames', type=int, default=40)
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--batches', type=int, default=12)
    ap.add_argument('--workers', type=int, default=8)
    args = ap.parse_args()
    root = '/tmp/mcg_fake_mug_%d_%d' % (args.videos, args.frames)
    make_tree(root, args.videos, args.frames)
    from datasets import MugDataset
    from mocogan_chainer_amd.trainer import SerialIterator, PrefetchIterator
    ds = MugDataset(root)
    print('dataset: %d videos x %d frames' % (len(ds), args.frames))
    print('SerialIterator           : %7.1f clips/s' % rate(SerialIterator(ds, args.batch), args.batches, args.batch))
    for w in sorted({2, 4, args.workers}):
        it = PrefetchIterator(ds, args.batch, n_workers=w, prefetch=4)
        r1 = rate(it, args.batches * 2, args.batch)
        r2 = rate(it, args.batches * 2, args.batch, raw=True)
        print('PrefetchIterator %2d procs : %7.1f clips/s (float list, next())   %7.1f clips/s (uint8 batch for the GPU)' % (w, r1, r2))
        it.close()


if __name__ == '__main__':
    main()
