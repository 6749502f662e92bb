"""Host-side logic of the reference-facing surface (model.net / model.updater / train.py / util.py /
datasets.py) that needs no GPU: construction, attributes, checkpoint key scheme, iterator, CLI."""
import os

import numpy as np
import pytest
import torch


def test_network_constructors_mirror_the_reference_signatures():
    from model.net import ImageGenerator, ImageDiscriminator, VideoDiscriminator
    g = ImageGenerator(dim_zl=6, n_filters=4, device='cpu')
    assert (g.dim_zc, g.dim_zm, g.dim_zl, g.out_channels, g.n_filters, g.video_len) == (50, 10, 6, 3, 4, 16)
    assert g.n_hidden == 60 and g.use_label and g.name == 'ImageGenerator'
    assert g.make_hidden(5, 10).shape == (5, 10) and g.make_hidden(5, 10).dtype == np.float32
    assert np.array_equal(g.to_one_hot([2, 0], np), np.eye(6, dtype=np.float32)[[2, 0]])
    d = ImageDiscriminator(3, 7, 4, True, 0.2, device='cpu')
    assert (d.in_channels, d.out_channels, d.n_filters, d.use_noise, d.noise_sigma, d.name) == (3, 7, 4, True, 0.2, 'ImageDiscriminator')
    v = VideoDiscriminator(n_filters=4, device='cpu')
    assert v.name == 'VideoDiscriminator' and v.use_noise is False
    # Chainer child / parameter names and shapes (also the checkpoint keys)
    gp, dp, vp = g.serialize_dict(), d.serialize_dict(), v.serialize_dict()
    assert gp['g0/W_r/W'].shape == (10, 16) and gp['g0/U/W'].shape == (10, 10) and gp['dc1/W'].shape == (60, 32, 4, 4)
    assert gp['dc5/W'].shape == (4, 3, 4, 4) and gp['dc5/b'].shape == (3,) and gp['bn1/avg_var'].shape == (32,)
    assert dp['dc1/W'].shape == (4, 3, 4, 4) and dp['dc5/W'].shape == (7, 32, 4, 4) and 'bn1/gamma' not in dp
    assert vp['dc1/W'].shape == (4, 3, 4, 4, 4) and vp['dc5/W'].shape == (1, 32, 4, 4, 4) and vp['bn4/N'].shape == ()
    # GlorotNormal scale (model/net.py:35): std = sqrt(2 / (fan_in + fan_out))
    w = ImageDiscriminator(3, 1, 64, device='cpu').serialize_dict()['dc4/W']
    assert abs(w.std() / np.sqrt(2.0 / ((256 + 512) * 16)) - 1) < 0.02
    assert np.all(dp['dc3/b'] == 0) and np.all(dp['bn2/gamma'] == 1)


def test_product_path_has_no_cpu_fallback():
    from model.net import ImageDiscriminator
    import mocogan_chainer_amd.hiplib as hl
    d = ImageDiscriminator(n_filters=4, device='cpu')
    with pytest.raises(hl.McgError):
        d(np.zeros((1, 3, 64, 64), np.float32))


def test_npz_round_trip_uses_chainer_keys(tmp_path):
    from model.net import ImageGenerator
    from mocogan_chainer_amd.trainer import save_npz, load_npz
    g = ImageGenerator(n_filters=4, device='cpu')
    save_npz(tmp_path / 'gen.npz', g)
    with np.load(tmp_path / 'gen.npz') as f:
        keys = set(f.files)
    assert {'dc1/W', 'dc1/b', 'bn1/gamma', 'bn1/beta', 'bn1/avg_mean', 'bn1/avg_var', 'bn1/N', 'g0/W_r/W', 'g0/U/b'} <= keys
    g2 = ImageGenerator(n_filters=4, device='cpu')
    load_npz(tmp_path / 'gen.npz', g2)
    a, b = g.serialize_dict(), g2.serialize_dict()
    assert all(np.array_equal(a[k], b[k]) for k in a)


def test_serial_iterator_epochs_and_batches():
    from mocogan_chainer_amd.trainer import SerialIterator
    from datasets import SyntheticDataset
    ds = SyntheticDataset(10, num_labels=6)
    v, l = ds[0]
    assert v.shape == (3, 16, 64, 64) and v.dtype == np.float32 and -1 <= v.min() and v.max() < 1 and 0 <= l < 6
    assert SyntheticDataset(3, num_labels=0)[1][1] is None
    it = SerialIterator(ds, 4)
    flags = []
    for _ in range(5):
        assert len(it.next()) == 4
        flags.append((it.epoch, it.is_new_epoch))
    assert flags == [(0, False), (0, False), (1, True), (1, False), (2, True)]


def _fake_frame_tree(root, videos=5, frames=20):
    from PIL import Image
    rng = np.random.RandomState(1)
    for v in range(videos):
        d = root / ("anger", "happiness")[v % 2] / ("%03d" % v)
        d.mkdir(parents=True)
        for f in range(frames):
            Image.fromarray(rng.randint(0, 255, (64, 64, 3)).astype(np.uint8)).save(d / ("%03d.jpg" % f))


def test_prefetch_iterator_keeps_serial_order_and_epoch_bookkeeping(tmp_path):
    """PrefetchIterator (worker processes decode ahead) must show exactly SerialIterator's sample order, wrap-around
    batches and epoch / is_new_epoch / epoch_detail for the same NumPy seed; the uint8 'raw' transport must
    reproduce the float path bit for bit."""
    from mocogan_chainer_amd.trainer import SerialIterator, PrefetchIterator
    from datasets import SyntheticDataset, MugDataset
    ds = SyntheticDataset(10, num_labels=6)
    np.random.seed(7)
    a = SerialIterator(ds, 4)
    ref = []
    for _ in range(7):
        b = a.next()
        ref.append(([x[1] for x in b], float(np.sum([x[0][0, 0, 0, 0] for x in b])), a.epoch, a.is_new_epoch, a.epoch_detail))
    np.random.seed(7)
    p = PrefetchIterator(ds, 4, n_workers=2, prefetch=3, chunk=2)
    try:
        for r in ref:
            b = p.next()
            assert ([x[1] for x in b], float(np.sum([x[0][0, 0, 0, 0] for x in b])), p.epoch, p.is_new_epoch, p.epoch_detail) == r
    finally:
        p.close()
    # frame directories: raw (uint8 through the pipes) == float path, shapes and range as the reference's
    _fake_frame_tree(tmp_path / "mug")
    mug = MugDataset(tmp_path / "mug")
    assert len(mug) == 5 and mug.num_labels == 2
    np.random.seed(3)
    v_f, l_f = mug.get_example(2)
    np.random.seed(3)
    v_r, l_r = mug.get_example_raw(2)
    assert v_r.dtype == np.uint8 and v_r.shape == (16, 64, 64, 3) and l_f == l_r
    assert np.array_equal(((v_r.astype(np.float32) - 128.) / 128.).transpose(3, 0, 1, 2), v_f)
    p = PrefetchIterator(mug, 3, n_workers=2, prefetch=2, chunk=2, shuffle=False)
    try:
        for _ in range(3):
            b = p.next()
            assert len(b) == 3 and b[0][0].shape == (3, 16, 64, 64) and b[0][0].dtype == np.float32
            assert -1 <= b[0][0].min() and b[0][0].max() < 1
        assert (p.epoch, p.is_new_epoch) == (1, False)          # 5 videos, batches of 3: the 2nd batch wrapped
    finally:
        p.close()


def test_prefetch_iterator_shared_memory_slots_equal_the_result_pipes(tmp_path, monkeypatch):
    """Round 5: raw (uint8) datasets reach the training process through shared-memory batch slots the workers write into; the
    batches must be those of the pickled-result transport (MCG_LOADER_SHM=0) and of SerialIterator, sample for sample, across a
    wrap-around and a reset; the staging callback of next_device_batch (`_pop(out=...)`) receives the same frames; close() leaves
    no segment behind."""
    import os
    from mocogan_chainer_amd.trainer import SerialIterator, PrefetchIterator
    from datasets import MugDataset
    _fake_frame_tree(tmp_path / "mug")
    mug = MugDataset(tmp_path / "mug")

    def run(make, staged=False):
        np.random.seed(11)
        it = make()
        out = []
        try:
            for k in range(5):
                if staged:
                    v, l = it._pop(out=lambda shape, dt: np.empty(shape, dt))
                    b = [(((v[i].astype(np.float32) - 128.) / 128.).transpose(3, 0, 1, 2), l[i]) for i in range(len(l))]
                else:
                    b = it.next()
                out.append(([x[1] for x in b], [x[0].copy() for x in b], it.epoch, it.is_new_epoch, it.epoch_detail))
                if k == 2 and getattr(it, '_slots', None):
                    assert len(it._free_slots) + len(it._queue) == len(it._slots)          # no slot is lost
        finally:
            if hasattr(it, 'close'):
                it.close()
        return out
    ref = run(lambda: SerialIterator(mug, 3, shuffle=False))
    shm = run(lambda: PrefetchIterator(mug, 3, n_workers=2, prefetch=2, chunk=2, shuffle=False))
    staged = run(lambda: PrefetchIterator(mug, 3, n_workers=2, prefetch=2, chunk=2, shuffle=False), staged=True)
    monkeypatch.setenv('MCG_LOADER_SHM', '0')
    pipes = run(lambda: PrefetchIterator(mug, 3, n_workers=2, prefetch=2, chunk=2, shuffle=False))
    for got in (shm, staged, pipes):
        for a, b in zip(ref, got):
            assert a[0] == b[0] and a[2:] == b[2:]                  # labels, epoch bookkeeping: SerialIterator's
        for a, b in zip(shm, got):                                  # frames: the workers' sub-sequence offsets are seeded per (seed, batch, position)
            assert all(np.array_equal(x, y) for x, y in zip(a[1], b[1]))
    # ... and so do not depend on how a batch is cut into worker tasks (round 5's advice: a resumed run with another worker count)
    other = run(lambda: PrefetchIterator(mug, 3, n_workers=3, prefetch=2, chunk=1, shuffle=False))
    for a, b in zip(shm, other):
        assert a[0] == b[0] and all(np.array_equal(x, y) for x, y in zip(a[1], b[1]))
    assert not [f for f in os.listdir('/dev/shm') if f.startswith('mcg_') and ('_%d_' % os.getpid()) in f]


def test_stale_slot_reaping_stays_inside_this_pid_namespace(tmp_path, monkeypatch):
    """Round 5's advice: batch slots are named after (PID namespace, pid); only segments of THIS namespace whose pid is gone are
    reaped -- a foreign namespace's pid means nothing here (two containers sharing /dev/shm), and a slot that cannot be backed by
    /dev/shm makes the iterator fall back to the result pipes instead of dying with SIGBUS in a worker."""
    import os
    import subprocess
    import sys
    from mocogan_chainer_amd import trainer as T
    ns = T._pid_namespace()
    dead = subprocess.Popen([sys.executable, '-c', 'pass'])
    dead.wait()
    mine_dead = tmp_path / ('mcg_%s_%d_abc_0' % (ns, dead.pid))
    mine_live = tmp_path / ('mcg_%s_%d_abc_0' % (ns, os.getpid()))
    foreign = tmp_path / ('mcg_%s_%d_abc_0' % ('deadbeef' if ns != 'deadbeef' else 'feedface', dead.pid))
    old_style = tmp_path / ('mcg_%d_abc_0' % dead.pid)              # (round 5's names: no namespace in them -- left alone)
    for f in (mine_dead, mine_live, foreign, old_style):
        f.write_bytes(b'x')
    T._remove_stale_slots(str(tmp_path))
    assert not mine_dead.exists() and mine_live.exists() and foreign.exists() and old_style.exists()
    assert T._slot_name(0x1234567, 3) == 'mcg_%s_%d_%x_3' % (ns, os.getpid(), 0x234567)

    # no room in /dev/shm: the slots are given up in __init__, nothing is left behind, the batches still arrive
    from datasets import MugDataset

    def no_space(fd, off, n):
        raise OSError(28, 'No space left on device')
    monkeypatch.setattr(os, 'posix_fallocate', no_space)
    _fake_frame_tree(tmp_path / "mug")
    ds = MugDataset(tmp_path / "mug")
    assert hasattr(ds, 'get_example_raw')
    it = T.PrefetchIterator(ds, 2, n_workers=1, prefetch=1, shuffle=False)
    try:
        assert it._slots == [] and it._free_slots == []
        b = it.next()
        assert len(b) == 2 and b[0][0].shape == (3, 16, 64, 64)
    finally:
        it.close()
    assert not [f for f in os.listdir('/dev/shm') if f.startswith('mcg_') and ('_%d_' % os.getpid()) in f]


def test_grid_and_sequence_helpers():
    from util import to_grid, to_sequence
    v = np.arange(2 * 3 * 1 * 2 * 2, dtype=np.uint8).reshape(2, 3, 1, 2, 2)
    g = to_grid(v, 2)
    assert g.shape == (2, 1, 4, 4)
    assert np.array_equal(g[:, :, :2, 2:], v[:, 1]) and np.all(g[:, :, 2:, 2:] == 0)
    assert to_sequence(v[:, 0]).shape == (1, 2, 4)


def test_cli_flags_match_the_reference():
    import train
    a = train.parse_args([])
    assert (a.gpu, a.dataset_type, a.batchsize, a.max_epoch, a.model, a.dim_zc, a.dim_zm) == (-1, 'mug', 100, 1000, 'normal', 50, 10)
    assert (a.display_interval, a.snapshot_interval, a.log_tensorboard_interval, a.num_gen_samples) == (1, 10, 10, 36)
    assert (a.n_filters_gen, a.n_filters_idis, a.n_filters_vdis, a.resume) == (64, 64, 64, '')
    a = train.parse_args(['-g', '0', '--model', 'infogan', '-r', 'snap.npz'])
    assert a.gpu == 0 and a.model == 'infogan' and a.resume == 'snap.npz'
    with pytest.raises(SystemExit):
        train.parse_args(['--model', 'wgan'])


def test_optimizer_objects_carry_the_reference_hyperparameters():
    from mocogan_chainer_amd import trainer as T
    from model.net import ImageDiscriminator
    d = ImageDiscriminator(n_filters=4, device='cpu')
    opt = T.Adam(alpha=2e-4, beta1=5e-5)
    opt.setup(d)
    opt.add_hook(T.WeightDecay(1e-5), 'hook_dec')
    h = opt.hyper()
    assert (h.alpha, h.beta1, h.beta2, h.eps, h.weight_decay) == (2e-4, 5e-5, 0.999, 1e-8, 1e-5)
    assert opt.t == 0


def test_shipped_tile_table_is_well_formed():
    """mocogan-chainer_amd/tuned_tiles_mi355x.json: [[key, code], ...] with keys as hiplib._geom_key builds them and
    codes the C ABI accepts (mcg_conv_geom.tile)."""
    import json
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'mocogan-chainer_amd', 'tuned_tiles_mi355x.json')
    table = json.load(open(path))
    assert len(table) >= 40
    seen = set()
    for key, code in table:
        kind, N, Ti, Hi, Wi, Ci, Co, kt, perm, prec = key[:10]
        assert Hi == Wi and Hi in (8, 16, 32, 64) and kt in (1, 4) and prec in (0, 1, 2, 3, 4) and Ci % 4 == 0 and N > 0
        # (4 = MCG_PREC_BF16_Y16, round 6: the first layer's weight / input gradient of bf16 networks, y bf16 beside the fp32 clip)
        assert prec != 4 or (kind in ('wgrad', 'dgrad') and Ci == 4 and Co == 64), (key, code)
        if kind.startswith('split-'):
            # 'f32x3' networks: does the split form (fp32 values as three bf16 terms, bf16 MFMA) of this fp32 launch pay?  (hiplib.split_pays)
            assert kind[6:] in ('fprop', 'dgrad', 'wgrad') and len(key) == 10 and prec == 0 and code in (0, 1), (key, code)
            assert all(isinstance(v, int) for v in key[1:])
        else:
            base = 12 if kind == 'dgrad' else 10                            # dgrad keys carry (act, accumulate)
            # bf16-stored geometries (fprop / dgrad) are tuned per launch form: + ('ep', sums, mask, bf16 output)  (hiplib._ep_key)
            form = 4 if (prec == 2 and kind != 'wgrad') else 0
            assert kind in ('fprop', 'dgrad', 'wgrad') and len(key) == base + form, key
            assert all(isinstance(v, int) for v in key[1:base]) and (not form or (key[base] == 'ep' and all(isinstance(v, int) for v in key[base + 1:]))), key
            assert isinstance(code, int) and 0 <= code % 100 <= 10 and (code // 100) % 10 <= 2 and code // 1000 <= 2, code
            # 7 / 8: the LDS-DMA kernels, 9: the patch-stationary input gradient -- bf16-stored operands (no K split) or split
            # operands (prec 3: only these kernels; + 1000 / 2000 K splits of fprop / dgrad)
            if prec == 3:
                assert code % 1000 in (7, 8, 9, 10) and (code < 1000 or kind != 'wgrad') and (code % 1000 != 9 or (kind == 'dgrad' and Ci == 64 and Hi == 32)), (key, code)
            else:
                assert code % 100 < 7 or (prec in (0, 2) and code == 10) or (prec == 2 and code in (7, 8)) or (prec == 2 and code == 9 and kind == 'dgrad' and Ci == 64 and Hi == 32), (key, code)
        assert tuple(key) not in seen
        seen.add(tuple(key))


class _RawSynthetic(__import__('datasets').SyntheticDataset):        # (module level: the shard is pickled with its dataset)
    def get_example_raw(self, i):
        return ('raw', i)


def test_sharded_dataset_partitions_the_data():
    """datasets.ShardedDataset (train.py --dp_shard, data parallel): the ranks' shards cover the dataset, ALL have the same length
    ceil(N / world) (the short shards wrap around to the front: ChainerMN scatter_dataset(force_equal_length=True)) so that every
    rank runs the same number of iterations per epoch, survive pickling (the prefetching loader ships the dataset to its workers)
    and forward the raw accessor."""
    import pickle
    from datasets import ShardedDataset, SyntheticDataset
    ds = SyntheticDataset(11, num_labels=6, video_length=2, img_size=8)
    shards = [ShardedDataset(ds, r, 3) for r in range(3)]
    assert [len(s_) for s_ in shards] == [4, 4, 4]
    seen = []
    for r, sh in enumerate(shards):
        for i in range(len(sh)):
            v, lab = sh[i]
            j = (r + 3 * i) % 11
            assert np.array_equal(v, ds[j][0]) and lab == ds[j][1] and np.array_equal(sh.get_example(i)[0], v)
            seen.append(j)
        with pytest.raises(IndexError):
            sh[len(sh)]
    assert sorted(set(seen)) == list(range(11)) and len(seen) == 12            # one item (index 0, via rank 2) is seen twice
    assert not hasattr(shards[0], 'get_example_raw')              # (the synthetic dataset has no raw frames)
    for n, world in ((1001, 8), (8, 8), (3, 8), (16, 4)):                     # the advisor's example; exact division; fewer items than ranks
        lens = {len(ShardedDataset(SyntheticDataset(n, video_length=1, img_size=8), r, world)) for r in range(world)}
        assert lens == {-(-n // world)}

    sh = pickle.loads(pickle.dumps(ShardedDataset(_RawSynthetic(11, video_length=2, img_size=8), 2, 3)))
    assert hasattr(sh, 'get_example_raw') and sh.get_example_raw(1) == ('raw', 5) and len(sh) == 4
    assert sh.get_example_raw(3) == ('raw', 0)
    with pytest.raises(ValueError):
        ShardedDataset(ds, 3, 3)


def test_split_form_predicates(monkeypatch):
    """hiplib.split_covers / split_decided (host logic of the 'f32x3' networks): which launches have a MCG_PREC_SPLIT form, and when
    a producer may rely on it (MCG_SPLIT=always / never, else the table entry 'split-<pass>' == 1)."""
    import mocogan_chainer_amd.hiplib as hl
    wide = hl.make_geom(64, 13, 32, 32, 64, 128, 4)             # D_V dc2 at 64 clips
    clip = hl.make_geom(64, 16, 64, 64, 4, 64, 4, ci_valid=3)   # the 4-channel first layer
    narrow = hl.make_geom(4, 5, 16, 16, 16, 32, 4)
    assert hl.split_covers('fprop', wide) and hl.split_covers('dgrad', wide) and hl.split_covers('wgrad', wide)
    assert not any(hl.split_covers(k, clip) for k in ('fprop', 'dgrad', 'wgrad'))
    assert hl.split_covers('fprop', narrow) and not hl.split_covers('dgrad', narrow) and not hl.split_covers('wgrad', narrow)
    huge = hl.make_geom(512, 13, 32, 32, 64, 128, 4)            # 8 bytes per value must stay below the 2 GiB buffer range
    assert not hl.split_covers('fprop', huge)
    monkeypatch.setenv('MCG_SPLIT', 'always')
    assert hl.split_decided('fprop', wide) and not hl.split_decided('fprop', clip)
    monkeypatch.setenv('MCG_SPLIT', 'never')
    assert not hl.split_decided('fprop', wide)
    monkeypatch.setenv('MCG_SPLIT', 'auto')
    key = hl._geom_key('split-fprop', wide)
    old = hl._tile_cache.pop(key, None)
    try:
        assert not hl.split_decided('fprop', wide)              # undecided: the producers keep writing fp32
        hl._tile_cache[key] = 1
        assert hl.split_decided('fprop', wide)
        hl._tile_cache[key] = 0
        assert not hl.split_decided('fprop', wide)
    finally:
        hl._tile_cache.pop(key, None)
        if old is not None:
            hl._tile_cache[key] = old
    g3 = hl.with_precision(wide, 'f32x3')
    assert g3.precision == hl.PREC_SPLIT and g3.tile == 0 and wide.precision == hl.PREC_F32


def test_prefetch_iterator_resumes_from_a_restored_position():
    """load_state must drop the look-ahead queued from the old position; a resumed PrefetchIterator has to
    show what a SerialIterator restored to the same state shows (reference: --resume, train.py:162-163)."""
    from mocogan_chainer_amd.trainer import SerialIterator, PrefetchIterator
    from datasets import SyntheticDataset
    ds = SyntheticDataset(12, num_labels=6)
    np.random.seed(11)
    a = SerialIterator(ds, 4)
    for _ in range(4):                                       # into epoch 1
        a.next()
    saved = (a.epoch, a.current_position, a._order.copy(), a.is_new_epoch, a._previous_epoch_detail)
    rs = np.random.get_state()
    ref = []
    for _ in range(5):
        b = a.next()
        ref.append(([x[1] for x in b], a.epoch, a.is_new_epoch, a.current_position))
    np.random.seed(99)                                       # a fresh process: different permutation, position 0
    p = PrefetchIterator(ds, 4, n_workers=2, prefetch=3, chunk=2)
    try:
        p.next()                                             # the look-ahead is now queued from the wrong place
        p.load_state(*saved)
        np.random.set_state(rs)                              # (Chainer restores the order, not NumPy's global generator)
        assert (p.epoch, p.current_position) == saved[:2]
        for r in ref:
            b = p.next()
            assert ([x[1] for x in b], p.epoch, p.is_new_epoch, p.current_position) == r
    finally:
        p.close()


def test_trainer_state_round_trips_the_iterator_and_the_step_counter():
    """Trainer.state()/load_state(): Chainer's SerialIterator keys (epoch, current_position, order, ...) and the device
    step's own iteration counter, which keys the Philox streams and the frame index."""
    from mocogan_chainer_amd import trainer as T
    from datasets import SyntheticDataset

    class _Impl:
        t = 3

        def export_reference_params(self):
            return {'dc1/W': np.ones(2, np.float32)}

        def export_adam_state(self):
            return {'t': 3, 'm': {'dc1/W': np.zeros(2, np.float32)}, 'v': {'dc1/W': np.zeros(2, np.float32)}}

        def trainable_keys(self):
            return ['dc1/W']

        def load_reference_params(self, d):
            self.loaded = d

        def load_adam_state(self, st):
            self.t = st['t']

    class _Link:
        def __init__(self):
            self.impl = _Impl()

    class _Step:
        iteration = 0

    class _U:
        def __init__(self, it):
            self.iteration, self._it, self._step, self._links = 0, it, _Step(), {'image_gen': _Link()}

        def get_iterator(self, name):
            return self._it

        def links(self):
            return self._links

    np.random.seed(5)
    it = T.SerialIterator(SyntheticDataset(10, 6), 4)
    for _ in range(4):
        it.next()
    u = _U(it)
    u.iteration = 4
    d = T.Trainer(u, (1, 'epoch')).state()
    assert {'updater/iterator:main/order', 'updater/iterator:main/is_new_epoch', 'updater/iterator:main/previous_epoch_detail'} <= set(d)
    np.random.seed(6)
    it2 = T.SerialIterator(SyntheticDataset(10, 6), 4)
    u2 = _U(it2)
    T.Trainer(u2, (1, 'epoch')).load_state(d)
    assert u2.iteration == 4 and u2._step.iteration == 4
    assert (it2.epoch, it2.current_position, it2.is_new_epoch) == (it.epoch, it.current_position, it.is_new_epoch)
    assert np.array_equal(it2._order, it._order)


def test_bench_self_launches_its_ranks(tmp_path):
    """`python bench.py --gpus N` with no WORLD_SIZE starts the N ranks itself (torch.distributed.run as a child
    process), relays rank 0's single JSON line and the exit code; fewer visible GPUs than ranks is a clear error, never
    a hang.  MCG_BENCH_DRYRUN=1 swaps the GPU work for a gloo rendezvous + MAX all-reduce, so the launcher runs here."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MCG_BENCH_DRYRUN='1')
    env.pop('WORLD_SIZE', None)
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1'],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line['n_gpus'] == 2 and line['steps'] == 3 and line['dry_run'] is True and line['max_over_ranks_check'] == 2.0
    if not torch.cuda.is_available():                        # this container: no GPU -> the real launch refuses, quickly
        env.pop('MCG_BENCH_DRYRUN')
        r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2'], capture_output=True, text=True,
                           env=env, timeout=600)
        assert r.returncode == 2 and 'GPU(s) visible' in r.stderr


def test_drivers_eight_rank_launch_line_reaches_one_result_line():
    """The driver's own command for N = 8 -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1
    --master-port P bench.py --gpus 8 --steps K --warmup W` -- with MCG_BENCH_DRYRUN=1 (no GPU here: the ranks rendezvous over gloo and
    run the timed region's MAX all-reduce): eight ranks parse RANK / WORLD_SIZE / MASTER_*, agree on the world size, and rank 0 alone
    prints the line.  (The GPU side of the eight-rank branch is walked by four ranks on one card: tests/test_gpu_dp.py.)"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MCG_BENCH_DRYRUN='1')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    port = 34500 + os.getpid() % 2000
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '8', '--master-addr', '127.0.0.1',
                        '--master-port', str(port), os.path.join(root, 'bench.py'), '--gpus', '8', '--steps', '4', '--warmup', '2'],
                       capture_output=True, text=True, env=env, timeout=900, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line['n_gpus'] == 8 and line['steps'] == 4 and line['warmup'] == 2 and line['dry_run'] is True and line['max_over_ranks_check'] == 8.0


def test_bench_stdout_line_stays_small():
    """The driver parses ONE stdout line; round 3's grew to 51 KB (tile table, per-layer tables of four workloads) and was not
    parsed.  compact_line() keeps the contract fields + config + roofline + cpu_baseline + one short record per secondary
    workload within bench.LINE_LIMIT bytes, whatever the size of the detailed record (which goes to bench_detail.json)."""
    import json
    import bench
    by_layer = {"D_V.fprop N=%d T=16 H=64 Ci=%d Co=%d" % (n, c, 2 * c): {"launches_per_step": 2.0, "ms_per_launch": 0.123456789, "tflops": 123.456789}
                for n in (32, 64) for c in (4, 64, 128, 256) for _ in range(1)}

    def rec(dtype, wl, value):
        return {"metric": "training clips/sec (16×3×64×64)", "value": value, "unit": "clips/s", "n_gpus": 1, "steps": 20, "warmup": 5,
                "ms_per_step": 14.987654321, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": dtype, "data": "synthetic",
                "config": {"workload": "MUG-shape synthetic (B,3,16,64,64) U(-1,1), one update_core iteration per step (BASELINE.json %s)" % wl,
                           "variant": "normal", "per_gpu_batch": 32, "global_batch": 32, "n_filters": 64, "dim_zl": 6, "parallelism": "dp1",
                           "side_streams": True, "sync_bn": False},
                "roofline": {"bound": "mfma", "achieved": 123.0123456, "peak": 157.3, "unit": "TFLOP/s", "frac": 0.782123456, "traffic": 7.43e9,
                             "algorithmic_bytes": 2.47e9, "traffic_source": "profiles/r03_dv_conv_traffic.json (PMC, batch 32)",
                             "kernel": "VideoDiscriminator Conv3d implicit-GEMM family (gemm_kernel<FpropP|DgradP|WgradP>), dc1..dc4, all launches of one step",
                             "algorithmic_gflop_per_step": 1315.3, "kernel_ms_per_step": 10.7,
                             "by_pass": {k: {"launches_per_step": 8.0, "ms_per_step": 3.3, "tflops": 120.123456} for k in ("fprop", "wgrad", "dgrad")},
                             "by_layer": by_layer, "by_network_and_pass_ms_per_step": {"x" * 20 + str(i): 1.0 for i in range(40)},
                             "measured": "m" * 400},
                "dist": {"backend": None, "world_size": 1, "per_rank_ms_per_step": {"min": 1.0, "median": 1.0, "max": 1.0}},
                "losses": {"image_dis/loss": 0.71234567, "video_dis/loss": 0.6923456, "image_gen/loss": 1.4123456}}
    out = rec("f32", "configs[1]", 2142.123456)
    secondary = [rec("bf16", "configs[2]", 11000.5), rec("f32", "configs[3]", 2100.25), rec("f32x3", "configs[1], fp32 products on the bf16 pipe", 2840.0),
                 {"config": {"workload": "normal f32 batch 128"}, "dtype": "f32", "error": "E" * 1000}]
    cpu = {"value": 1.234567, "unit": "clips/s", "cores": 8, "kind": "port", "batch": 8, "timed_iterations": 5, "sample": "s" * 600,
           "c1_moving_mnist_shape": {"value": 2.0}}
    text = bench.compact_line(out, secondary, cpu, "/somewhere/bench_detail.json")
    assert len(text) <= bench.LINE_LIMIT <= 4096 and "\n" not in text
    line = json.loads(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(line["roofline"]) and "by_layer" not in line["roofline"]
    assert {"value", "unit", "cores", "kind", "sample"} <= set(line["cpu_baseline"])
    assert "configs[1]" in line["config"]["workload"] and line["dtype"] == "f32"
    assert [s["dtype"] for s in line["secondary"]] == ["bf16", "f32", "f32x3", "f32"] and "error" in line["secondary"][3]
    assert all({"value", "ms_per_step", "frac"} <= set(s) for s in line["secondary"][:3])
    assert line["detail"] == "bench_detail.json"
    # an absurdly long record still respects the bound (optional parts are dropped, never the contract fields)
    big = [rec("bf16", "w" * 3000, 1.0) for _ in range(6)]
    text = bench.compact_line(out, big, cpu, None)
    assert len(text) <= bench.LINE_LIMIT and json.loads(text)["value"] == round(2142.123456, 4)
    assert bench.PEAK_F32X3_TFLOPS == pytest.approx(2516.8 / 6)
    # round 4's advice: a string that grows must shrink the line, never cost it (an assert there meant NO stdout line at all)
    grown = rec("f32", "configs[1]" + "w" * 3000, 2142.0)
    grown["roofline"]["kernel"] = "k" * 3000
    text = bench.compact_line(grown, secondary, dict(cpu, sample="s" * 5000), "/somewhere/bench_detail.json")
    line = json.loads(text)
    assert len(text) <= bench.LINE_LIMIT and line["value"] == 2142.0 and line["metric"] and line["unit"] == "clips/s" and line["n_gpus"] == 1

