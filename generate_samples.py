#!/usr/bin/env python
"""Sample clips from a trained generator (raahii/mocogan-chainer generate_samples.py:17-58).
Like the reference it builds ``ImageGenerator()`` with default arguments and runs it in train mode
(batch-statistic BatchNorm, quirk Q11); ``--dim_zl`` lets a label-conditioned (MUG-trained) generator load."""
import argparse
import os
from pathlib import Path

import numpy as np

from model.net import ImageGenerator
from util import to_grid, save_video
from mocogan_chainer_amd.trainer import load_npz


def main(argv=None):
    cli = argparse.ArgumentParser(description='sample videos from a trained ImageGenerator (reference generate_samples.py)')
    for positional in ('model_weight', 'save_path'):
        cli.add_argument(positional)
    for flags, default, text in ((('--num', '-n'), 36, 'number of videos, a square number'),
                                 (('--gpu', '-g'), -1, 'kept for command-line compatibility; the MI355X path always runs on the GPU'),
                                 (('--dim_zl',), 0, 'label dimension the generator was trained with (extension)'),
                                 (('--n_filters',), 64, 'generator width (extension)')):
        cli.add_argument(*flags, type=int, default=default, help=text)
    args = cli.parse_args(argv)
    n = int(round(np.sqrt(args.num)))
    if n * n != args.num:
        raise ValueError('--num must be n^2 (n: natural number).')

    gen = ImageGenerator(dim_zl=args.dim_zl, n_filters=args.n_filters)
    load_npz(args.model_weight, gen)
    print(">>> generating...")
    videos = gen(args.num)[0].detach().cpu().numpy()                 # (T, N, C, H, W) in [-1, 1]
    videos = (255 * (0.5 * videos + 0.5)).astype(np.uint8)           # truncating cast, as the reference (:41)
    print(">>> saving...")
    save_path = Path(args.save_path)
    os.makedirs(save_path, exist_ok=True)
    save_video(to_grid(videos, n).transpose(0, 2, 3, 1), save_path / 'grid.mp4', True, save_path / 'grid')
    for i, video in enumerate(videos.transpose(1, 0, 3, 4, 2)):
        save_video(video, save_path / '{:03d}.mp4'.format(i), True, save_path / '{:03d}'.format(i))


if __name__ == "__main__":
    main()
