"""Visualisation / IO helpers with the reference's names (raahii/mocogan-chainer util.py:13-115)."""
import shutil
import subprocess as sp
from pathlib import Path

import numpy as np


def to_sequence(video, horizontally=True):
    """(num, C, H, W) -> one image with the frames side by side (or stacked)."""
    return np.concatenate(list(video), axis=2 if horizontally else 1)


def to_grid(videos, size):
    """(T, batch, C, H, W) -> (T, C, size*H, size*W); missing cells are black."""
    t, bs, c, h, w = videos.shape
    grid = np.zeros((t, c, size * h, size * w), dtype=videos.dtype)
    for k in range(min(bs, size * size)):
        i, j = divmod(k, size)
        grid[:, :, i * h:(i + 1) * h, j * w:(j + 1) * w] = videos[:, k]
    return grid


def save_frames(video, save_path):
    from PIL import Image
    save_path = Path(save_path)
    save_path.mkdir(parents=True, exist_ok=True)
    for i, v in enumerate(video):
        Image.fromarray(v).save(save_path / "{:02d}.jpg".format(i))


def save_video(video, save_path, save_frame=False, frame_path=Path("/tmp/mocogan-chainer")):
    """(T,H,W,C) uint8 -> mp4 through ffmpeg (frames are kept when ffmpeg is unavailable)."""
    frame_path = Path(frame_path)
    save_frames(video, frame_path)
    if shutil.which('ffmpeg') is None:
        print('ffmpeg not found: frames left in {}'.format(frame_path))
        return False
    cmd = ['ffmpeg', '-y', '-r', '16', '-i', str(frame_path / '%02d.jpg'), '-vcodec', 'libx264', '-pix_fmt', 'yuv420p',
           '-vf', 'setpts=PTS/0.5', str(save_path)]
    sp.call(cmd)
    if not save_frame:
        shutil.rmtree(frame_path, ignore_errors=True)
    return True


def log_tensorboard(image_gen, num, video_length, writer):
    """Trainer extension: sample `num` clips in test mode and log 4 grid frames + 10 frame strips."""
    from model.net import config

    def log(trainer):
        updater = trainer.updater
        prev, config.train = config.train, False
        try:
            videos, _ = image_gen(num)
        finally:
            config.train = prev
        videos = videos.detach().cpu().numpy() / 2. + 0.5                 # (T,N,C,H,W) in [0,1]
        grid = to_grid(videos, int(np.sqrt(num)))
        for i in np.linspace(0, video_length, 4, endpoint=False, dtype=int):
            writer.add_image('{:02d}th frame'.format(i), grid[i], updater.epoch)
        for i in range(min(10, videos.shape[1])):
            writer.add_image('video_{:02d}'.format(i), to_sequence(videos[:, i]), updater.epoch)
    return log
