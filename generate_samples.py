#!/usr/bin/env python
"""Sample clips from a trained generator (raahii/mocogan-chainer generate_samples.py:17-58).
Like the reference it builds ``ImageGenerator()`` with default arguments and runs it in train mode
(batch-statistic BatchNorm, quirk Q11); ``--dim_zl`` lets a label-conditioned (MUG-trained) generator load."""
import argparse
from pathlib import Path

import numpy as np

from model.net import ImageGenerator
from util import to_grid, save_video
from mocogan_chainer_amd.trainer import load_npz


def main(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument('model_weight')
    parser.add_argument('save_path')
    parser.add_argument('--num', '-n', type=int, default=36)
    parser.add_argument('--gpu', '-g', type=int, default=-1)
    parser.add_argument('--dim_zl', type=int, default=0, help='label dimension the generator was trained with (extension)')
    parser.add_argument('--n_filters', type=int, default=64)
    args = parser.parse_args(argv)
    if np.sqrt(args.num) % 1.0 != 0:
        raise ValueError('--num must be n^2 (n: natural number).')
    n = int(np.sqrt(args.num))

    gen = ImageGenerator(dim_zl=args.dim_zl, n_filters=args.n_filters)
    load_npz(args.model_weight, gen)
    print(">>> generating...")
    videos = gen(args.num)[0].detach().cpu().numpy()                 # (T, N, C, H, W)
    videos = ((videos / 2. + 0.5) * 255).astype(np.uint8)
    print(">>> saving...")
    save_path = Path(args.save_path)
    save_path.mkdir(parents=True, exist_ok=True)
    save_video(to_grid(videos, n).transpose(0, 2, 3, 1), save_path / 'grid.mp4', True, save_path / 'grid')
    for i, video in enumerate(videos.transpose(1, 0, 3, 4, 2)):
        save_video(video, save_path / '{:03d}.mp4'.format(i), True, save_path / '{:03d}'.format(i))


if __name__ == "__main__":
    main()
