"""Datasets with the reference's interface (raahii/mocogan-chainer datasets.py: MugDataset :29-107,
MovingMnistDataset :109-167): ``len(ds)`` and ``ds[i] -> (video float32 (C,T,H,W) in [-1,1), label)``.
``SyntheticDataset`` produces clips of the same shape and range without any files (benchmarks, smoke
tests: there is no dataset in this environment)."""
import glob
import os
import re
from pathlib import Path

import numpy as np

_FRAME = re.compile(r'([0-9]+).jpg')
MUG_CATEGORIES = {"anger": 0, "disgust": 1, "happiness": 2, "fear": 3, "sadness": 4, "surprise": 5}


def _frame_number(name):
    """Sort key of a frame file.  Deliberately NUMERIC: the reference's frame_number (datasets.py:12-13) returns the
    matched digit string and so sorts frames lexicographically, which equals the numeric order for its zero-padded
    '{:02d}.jpg' names below 100 frames and scrambles longer clips ('100.jpg' < '11.jpg'); numeric order is the
    temporal order the reference intends."""
    return int(_FRAME.search(str(name)).group(1))


frame_number = _frame_number          # the reference's public name (datasets.py:12)


def read_video(paths, dtype=np.float32):
    from PIL import Image
    frames = []
    for p in paths:
        with Image.open(p) as f:
            frames.append(np.asarray(f, dtype=dtype))
    return np.asarray(frames, dtype=dtype)


def load_examples(dataset, indices, raw):
    """Worker entry point of trainer.PrefetchIterator (this module imports neither torch nor the HIP library,
    so spawned worker processes stay light).  raw=True returns the decoded uint8 frames (T,H,W,C) -- a quarter
    of the bytes; normalisation and the (C,T,H,W) transpose then happen on the GPU."""
    if raw:
        out = [dataset.get_example_raw(int(i)) for i in indices]
    else:
        out = [dataset.get_example(int(i)) for i in indices]
    return np.stack([o[0] for o in out]), [o[1] for o in out]


_WORKER = {'seed': 0, 'dataset': None}


def worker_init(seed, dataset):
    """initializer of the PrefetchIterator's worker processes: the dataset travels once, not per task"""
    _WORKER['seed'], _WORKER['dataset'] = seed, dataset


def _seed_sample(batch_no, pos):
    """the worker's NumPy generator for ONE sample: (loader seed, batch number, position in the batch) -- the sampled sub-sequence
    offsets then do not depend on how a batch is cut into worker tasks (--loader_workers, the chunk size), so a resumed run with
    another worker count sees the same data stream (round 5's advice)"""
    np.random.seed((int(_WORKER['seed']) * 1000003 + batch_no * 4099 + pos) % (2 ** 32))


def worker_load(indices, raw, batch_no, first):
    """`first`: position in the batch of indices[0]"""
    ds = _WORKER['dataset']
    out = []
    for k, i in enumerate(indices):
        _seed_sample(batch_no, first + k)
        out.append(ds.get_example_raw(int(i)) if raw else ds.get_example(int(i)))
    return np.stack([o[0] for o in out]), [o[1] for o in out]


_SHM = {}


def worker_load_shm(indices, batch_no, chunk_no, shm_name, first, clip_shape):
    """As worker_load(raw=True), but the decoded uint8 frames go straight into clips [first, first + len(indices)) of the shared-memory
    segment `shm_name` (one batch slot of trainer.PrefetchIterator) instead of back through the result pipe -- at 256 clips per batch
    the pickled results (50 MB per batch) were what the loader spent its time on.  Returns the labels only."""
    shm = _SHM.get(shm_name)
    if shm is None:
        from multiprocessing import shared_memory
        # (spawned workers share the parent's resource tracker: attaching re-registers the same name there, nothing is unlinked
        #  when a worker exits; the parent unlinks in PrefetchIterator.close())
        shm = shared_memory.SharedMemory(name=shm_name)
        _SHM[shm_name] = shm
    per = int(np.prod(clip_shape))
    arr = np.ndarray((shm.size // per,) + tuple(clip_shape), dtype=np.uint8, buffer=shm.buf)
    labels = []
    for k, i in enumerate(indices):
        _seed_sample(batch_no, first + k)
        v, l = _WORKER['dataset'].get_example_raw(int(i))
        arr[first + k] = v
        labels.append(l)
    return labels


class _FrameDirDataset:
    video_length = 16
    channels = 3            # 1: keep only the first colour plane (the grey-scale Moving-MNIST shape 16x1x64x64)

    def __len__(self):
        return len(self.videos)

    def __getitem__(self, i):
        return self.get_example(i)

    def _load(self, frame_paths, extract_speed=None, raw=False):
        n, T = len(frame_paths), self.video_length
        if n < T:
            raise ValueError('invalid video length: {} < {}'.format(n, T))
        if extract_speed and n > T * extract_speed:          # MUG: sample every 2nd frame when long enough
            needed = extract_speed * (T - 1)
            gap = n - needed
            start = 0 if gap == 0 else np.random.randint(0, gap, 1)[0]
            idx = np.linspace(start, start + needed, T, endpoint=True, dtype=np.int32)
        else:
            gap = n - T
            start = 0 if gap == 0 else np.random.randint(0, gap, 1)[0]
            idx = np.arange(start, start + T)
        video = read_video(frame_paths[idx], np.uint8 if raw else np.float32)
        if video.ndim != 4:
            raise ValueError('invalid video shape: {}'.format(video.shape))
        if self.channels != video.shape[3]:
            video = np.ascontiguousarray(video[..., :self.channels])
        if raw:
            return video                                              # (T,H,W,C) uint8
        video = (video - 128.) / 128.
        return video.astype(np.float32).transpose(3, 0, 1, 2)         # (C,T,H,W)


class MugDataset(_FrameDirDataset):
    def __init__(self, root_path, video_length=16):
        self.root_path, self.video_length, self.extract_speed = Path(root_path), video_length, 2
        self.video_categories = list(self.root_path.glob("*"))
        self.num_labels = len(self.video_categories)
        self.videos = []
        for cat in self.video_categories:
            if not cat.is_dir():
                continue
            for vp in cat.glob("*"):
                if vp.is_dir() and len(list(vp.glob("*.jpg"))) >= video_length:
                    self.videos.append((vp, MUG_CATEGORIES[cat.name]))

    def get_example(self, i, raw=False):
        vp, categ = self.videos[i]
        paths = np.array(sorted(glob.glob(os.path.join(vp, '*.jpg')), key=_frame_number))
        return self._load(paths, self.extract_speed, raw), categ

    def get_example_raw(self, i):
        return self.get_example(i, raw=True)


class MovingMnistDataset(_FrameDirDataset):
    def __init__(self, dataset_path, video_length=16, save_path="data/dataset/moving_mnist/preprocessed", channels=3):
        """channels=3 is the reference (datasets.py:127 tiles the grey frames to RGB); channels=1 yields the
        single-plane clips of BASELINE configs[0] from the same preprocessed JPEG tree."""
        self.video_length, self.channels = video_length, channels
        save_path = Path(save_path)
        if not save_path.exists():
            self.preprocess(dataset_path, save_path)
        self.videos = [p for p in save_path.glob("*") if p.is_dir()]

    def preprocess(self, dataset_path, save_path):
        from PIL import Image
        videos = np.load(dataset_path)                                  # (T, N, 64, 64) uint8
        videos = np.tile(videos[:, :, :, :, None], (1, 1, 1, 1, 3)).transpose(1, 0, 2, 3, 4)
        for i, video in enumerate(videos):
            path = save_path / "{:05d}".format(i)
            path.mkdir(parents=True, exist_ok=True)
            for j, img in enumerate(video):
                Image.fromarray(img).save(path / "{:02d}.jpg".format(j))

    def get_example(self, i, raw=False):
        paths = np.array(sorted(self.videos[i].glob("*.jpg"), key=_frame_number))
        return self._load(paths, raw=raw), None

    def get_example_raw(self, i):
        return self.get_example(i, raw=True)


class SyntheticDataset:
    """`size` clips ~ U(-1,1) of shape (channel, video_length, 64, 64); labels uniform in [0,num_labels)
    (None when num_labels == 0).  Deterministic per index."""

    def __init__(self, size=256, num_labels=6, channel=3, video_length=16, img_size=64, seed=0):
        self.size, self.num_labels, self.shape, self.seed = size, num_labels, (channel, video_length, img_size, img_size), seed

    def __len__(self):
        return self.size

    def __getitem__(self, i):
        rng = np.random.RandomState(self.seed * 1000003 + int(i))
        video = rng.uniform(-1, 1, self.shape).astype(np.float32)
        return video, (int(rng.randint(0, self.num_labels)) if self.num_labels else None)

    get_example = __getitem__


class ShardedDataset:
    """Data parallel: rank r of `world` sees items r, r + world, r + 2 world, ... of `dataset` (ChainerMN's scatter_dataset without
    the shuffle) -- the ranks' iterators together make ONE pass over the data per epoch.  EVERY shard has the same length
    ceil(N / world) (scatter_dataset's force_equal_length=True): when world does not divide N the last index of the short shards
    wraps around to the front of the dataset, (rank + i * world) % N.  Equal lengths are what keeps the ranks in step -- each rank
    stops on its own iterator's epoch count, and a rank that ran out of batches first would leave the others waiting in the
    gradient all-reduce forever.  No reference counterpart (the reference is single-device, train.py:87-91).  Forwards the
    raw-frame accessor the prefetching loader uses when the dataset has one."""

    def __init__(self, dataset, rank, world):
        if not 0 <= rank < world:
            raise ValueError('rank %d outside world %d' % (rank, world))
        if len(dataset) == 0:
            raise ValueError('cannot shard an empty dataset')
        self.dataset, self.rank, self.world = dataset, rank, world

    def __getattr__(self, name):                     # (only reached for names the instance does not have)
        if name == 'get_example_raw' and 'dataset' in self.__dict__ and hasattr(self.dataset, 'get_example_raw'):
            return self._raw
        raise AttributeError(name)

    def _index(self, i):
        i = int(i)
        if not 0 <= i < len(self):
            raise IndexError(i)
        return (self.rank + i * self.world) % len(self.dataset)

    def _raw(self, i):
        return self.dataset.get_example_raw(self._index(i))

    def __len__(self):
        return (len(self.dataset) + self.world - 1) // self.world

    def __getitem__(self, i):
        return self.dataset[self._index(i)]

    def get_example(self, i, *a, **kw):
        get = getattr(self.dataset, 'get_example', None)
        j = self._index(i)
        return get(j, *a, **kw) if get is not None else self.dataset[j]
