"""The reference-facing classes on the GPU: call conventions / shapes, NumPy-generator parity of the
latent and noise draws with the oracle (np.random.seed selects the draws a Chainer CPU run would
consume), Updater.update_core through the iterator, trainer snapshot / resume."""
import numpy as np
import pytest
import torch

from oracle import net as onet

pytestmark = pytest.mark.gpu


def rel_l2(a, b):
    a = a.detach().cpu().double().numpy() if isinstance(a, torch.Tensor) else np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)


def _f64(p):
    return {k: (v.astype(np.float64) if v.dtype.kind == 'f' else v) for k, v in p.items()}


def test_generator_call_matches_oracle_under_the_numpy_generator():
    from model.net import ImageGenerator
    for dim_zl in (0, 6):
        g = ImageGenerator(dim_zl=dim_zl, n_filters=8)
        p = _f64(g.serialize_dict())
        np.random.seed(11)
        x, labels = g(3)
        assert tuple(x.shape) == (16, 3, 3, 64, 64) and x.is_cuda
        rng = np.random.RandomState(11)
        draw = onet.gen_draw(rng, 3, dim_zl=dim_zl)
        x_ref, l_ref, _ = onet.gen_forward(p, {k: (v.astype(np.float64) if k != 'labels' and v is not None else v) for k, v in draw.items()})
        assert (labels is None) == (dim_zl == 0) and (labels is None or np.array_equal(labels, l_ref))
        assert rel_l2(x, x_ref) < 1e-5
        np.random.seed(12)
        zm = g.make_zm(3, None if dim_zl == 0 else g.to_one_hot([1, 2, 3]))
        assert tuple(zm.shape) == (16, 3, 10)


@pytest.mark.parametrize("ndim", [2, 3])
def test_discriminator_call_matches_oracle_under_the_numpy_generator(ndim):
    from model.net import ImageDiscriminator, VideoDiscriminator, config
    cls = ImageDiscriminator if ndim == 2 else VideoDiscriminator
    d = cls(3, 1, 8, True, 0.2)
    p = _f64(d.serialize_dict())
    shp = (2, 3, 64, 64) if ndim == 2 else (2, 3, 16, 64, 64)
    x = np.random.RandomState(0).uniform(-1, 1, shp).astype(np.float32)
    np.random.seed(21)
    y = d(torch.as_tensor(x).cuda())
    assert tuple(y.shape) == (2, 1) + (1,) * ndim
    rng = np.random.RandomState(21)
    noise = [0.2 * rng.randn(*s) for s in onet.dis_noise_shapes(ndim, 2, 3, 8)]
    y_ref, _ = onet.dis_forward(p, x.astype(np.float64), noise)
    assert rel_l2(y, y_ref) < 1e-5
    # test mode: no noise, running statistics (reference util.py:92)
    config.train = False
    try:
        y_t = d(torch.as_tensor(x).cuda())
        p2 = _f64(d.serialize_dict())
        y_t_ref, _ = onet.dis_forward(p2, x.astype(np.float64), None, train=False)
        assert rel_l2(y_t, y_t_ref) < 1e-5
    finally:
        config.train = True


def _make_updater(batch=4, nf=8, model='infogan', seed=0):
    from model.net import ImageGenerator, ImageDiscriminator, VideoDiscriminator
    from model.updater import Updater
    from datasets import SyntheticDataset
    from mocogan_chainer_amd import trainer as T
    np.random.seed(seed)
    out_d = 7 if model == 'infogan' else 1
    g, di, dv = ImageGenerator(dim_zl=6, n_filters=nf), ImageDiscriminator(3, out_d, nf, True, 0.2), VideoDiscriminator(3, out_d, nf, True, 0.2)
    opts = {}
    for name, link in (('image_gen', g), ('image_dis', di), ('video_dis', dv)):
        o = T.Adam(alpha=2e-4, beta1=5e-5)
        o.setup(link)
        o.add_hook(T.WeightDecay(1e-5), 'hook_dec')
        opts[name] = o
    it = T.SerialIterator(SyntheticDataset(8, 6), batch)
    return Updater(model=model, models=(g, di, dv), video_length=16, img_size=64, channel=3, dim_zl=6,
                   tensorboard_writer=T.NullWriter(), iterator=it, optimizer=opts, device=0)


def test_updater_runs_iterations_and_reports_per_epoch():
    u = _make_updater()
    assert u.get_optimizer('image_gen').alpha == 2e-4 and u.get_iterator('main').batch_size == 4
    before = u.image_gen.serialize_dict()['dc3/W'].copy()
    u.update()
    assert u.iteration == 1 and u.epoch == 0 and not u.is_new_epoch and u.observation == {}
    u.update()
    assert u.iteration == 2 and u.epoch == 1 and u.is_new_epoch
    assert set(u.observation) == {'image_gen/loss', 'image_dis/loss', 'video_dis/loss'}
    assert all(np.isfinite(v) for v in u.observation.values())
    after = u.image_gen.serialize_dict()['dc3/W']
    assert np.abs(after - before).max() > 1e-5 and u.get_optimizer('video_dis').t == 2
    # the loss methods keep the reference's signatures
    y = torch.randn(4, 7, 1, 1, 1, device='cuda')
    l = u.loss_dis(u.video_dis, y, y * 0.5, np.array([0, 1, 2, 3]), np.array([3, 2, 1, 0]))
    assert l.dim() == 0 and np.isfinite(float(l))
    l = u.loss_gen(u.image_gen, y[..., 0], y, np.array([0, 1, 2, 3]))
    assert np.isfinite(float(l))
    v = u.concat_label_video(torch.zeros(2, 3, 16, 8, 8), np.array([5, 0]))
    assert tuple(v.shape) == (2, 9, 16, 8, 8) and float(v[0, 3 + 5].min()) == 1 and float(v[0, 3].max()) == -1


def test_trainer_snapshot_resume_round_trip(tmp_path):
    from mocogan_chainer_amd import trainer as T
    u = _make_updater(model='normal')
    tr = T.Trainer(u, (2, 'iteration'), out=tmp_path)
    tr.extend(T.extensions.snapshot(filename='snap_{.updater.iteration}.npz'), trigger=(2, 'iteration'))
    tr.extend(T.extensions.snapshot_object(u.image_gen, 'gen_{.updater.iteration}.npz'), trigger=(2, 'iteration'))
    tr.run()
    assert (tmp_path / 'snap_2.npz').exists() and (tmp_path / 'gen_2.npz').exists()
    u2 = _make_updater(model='normal', seed=5)
    tr2 = T.Trainer(u2, (3, 'iteration'), out=tmp_path)
    T.load_npz(tmp_path / 'snap_2.npz', tr2)
    assert u2.iteration == 2 and u2.get_optimizer('image_dis').t == 2
    a, b = u.video_dis.serialize_dict(), u2.video_dis.serialize_dict()
    assert all(np.array_equal(a[k], b[k]) for k in a)
    sa, sb = u.video_dis.impl.export_adam_state(), u2.video_dis.impl.export_adam_state()
    assert all(np.array_equal(sa['v'][k], sb['v'][k]) for k in sa['v'])
    tr2.run()
    assert u2.iteration == 3


def test_train_and_generate_entry_points(tmp_path, monkeypatch):
    """train.py (synthetic data, 2 epochs) writes Chainer-keyed checkpoints and a log; generate_samples.py loads
    the generator back and writes frames (reference train.py:132-192, generate_samples.py:17-58)."""
    import json
    import train
    import generate_samples
    monkeypatch.chdir(tmp_path)
    tr = train.main(['--dataset_type', 'synthetic', '--synthetic_size', '8', '--batchsize', '4', '--max_epoch', '2',
                     '--n_filters_gen', '8', '--snapshot_interval', '1', '--log_tensorboard_interval', '1',
                     '--num_gen_samples', '4', '--save_name', 'run', '--model', 'infogan'])
    out = tmp_path / 'result' / 'run'
    assert tr.updater.iteration == 4 and tr.updater.epoch == 2
    for f in ('snapshot_epoch_2.npz', 'image_gen_epoch_1.npz', 'video_dis_epoch_2.npz', 'image_gen_epoch_fianl.npz', 'log'):
        assert (out / f).exists(), f
    log = json.load(open(out / 'log'))
    assert len(log) == 2 and {'epoch', 'iteration', 'image_gen/loss', 'image_dis/loss', 'video_dis/loss'} <= set(log[-1])
    assert all(np.isfinite(e['image_gen/loss']) for e in log)
    with np.load(out / 'image_gen_epoch_fianl.npz') as f:
        assert f['dc1/W'].shape == (60, 64, 4, 4) and f['g0/W_r/W'].shape == (10, 16)
    # resume: one more epoch from the snapshot
    tr2 = train.main(['--dataset_type', 'synthetic', '--synthetic_size', '8', '--batchsize', '4', '--max_epoch', '3',
                      '--n_filters_gen', '8', '--save_name', 'run2', '--model', 'infogan',
                      '--resume', str(out / 'snapshot_epoch_2.npz')])
    assert tr2.updater.iteration == 6 and tr2.updater.epoch == 3
    generate_samples.main([str(out / 'image_gen_epoch_fianl.npz'), str(tmp_path / 'samples'), '--num', '4',
                           '--dim_zl', '6', '--n_filters', '8'])
    frames = sorted((tmp_path / 'samples' / 'grid').glob('*.jpg'))
    assert len(frames) == 16
    from PIL import Image
    assert Image.open(frames[0]).size == (128, 128)


def test_prefetching_loader_feeds_the_updater(tmp_path, monkeypatch):
    """--loader_workers: worker processes decode uint8 frames, the batch is copied from pinned memory on a side
    stream and normalised / transposed on the GPU; the device batch equals the reference-style float batch, and
    train.py runs with it end to end on a MUG-shaped JPEG tree."""
    from PIL import Image
    from datasets import MugDataset
    from mocogan_chainer_amd import trainer as T
    import train
    rng = np.random.RandomState(2)
    for v in range(6):
        d = tmp_path / 'mug' / ('anger', 'disgust', 'happiness', 'fear', 'sadness', 'surprise')[v] / ('%03d' % v)
        d.mkdir(parents=True)
        for f in range(16):                                          # exactly 16 frames: no random crop offset
            Image.fromarray(rng.randint(0, 255, (64, 64, 3)).astype(np.uint8)).save(d / ('%03d.jpg' % f))
    ds = MugDataset(tmp_path / 'mug')
    assert len(ds) == 6 and ds.num_labels == 6
    ser = T.SerialIterator(ds, 4, shuffle=False)
    pre = T.PrefetchIterator(ds, 4, shuffle=False, n_workers=2, prefetch=2, chunk=2)
    try:
        for _ in range(3):
            b = ser.next()
            x_ref = np.stack([e[0] for e in b])
            if _ == 1:                                               # the Updater's form: + the copy's event and the labels on the device
                x_dev, labels, ready, lab_dev = pre.next_device_batch(torch.device('cuda'), with_event=True)
                ready.synchronize()
                assert lab_dev.dtype == torch.int32 and lab_dev.cpu().tolist() == labels
            elif _ == 2:                                             # round 6: the uint8 batch itself (TrainStep.run normalises it)
                x_u8, labels = pre.next_device_batch(torch.device('cuda'), as_uint8=True)
                torch.cuda.synchronize()
                assert x_u8.shape == (4, 16, 64, 64, 3) and x_u8.dtype == torch.uint8 and x_u8.is_contiguous()
                x_dev = ((x_u8.float() - 128.) / 128.).permute(0, 4, 1, 2, 3).contiguous()
            else:
                x_dev, labels = pre.next_device_batch(torch.device('cuda'))
            torch.cuda.synchronize()
            assert x_dev.shape == (4, 3, 16, 64, 64) and x_dev.dtype == torch.float32 and x_dev.is_contiguous()
            assert np.array_equal(x_dev.cpu().numpy(), x_ref) and labels == [e[1] for e in b]
            assert (pre.epoch, pre.is_new_epoch) == (ser.epoch, ser.is_new_epoch)
    finally:
        pre.close()
    # round 6: with ahead=True the copy of the NEXT batch is queued in the same call; batches, labels and the epoch bookkeeping are
    # still those of the batch returned (SerialIterator's), across a wrap-around, and a snapshot position counts returned batches only
    ser = T.SerialIterator(ds, 4, shuffle=False)
    pre = T.PrefetchIterator(ds, 4, shuffle=False, n_workers=2, prefetch=2, chunk=2)
    try:
        for k in range(4):
            b = ser.next()
            x_u8, labels, ready, lab_dev = pre.next_device_batch(torch.device('cuda'), with_event=True, as_uint8=True, ahead=True)
            ready.synchronize()
            x_dev = ((x_u8.float() - 128.) / 128.).permute(0, 4, 1, 2, 3)
            assert np.array_equal(x_dev.cpu().numpy(), np.stack([e[0] for e in b])) and labels == [e[1] for e in b], k
            assert (pre.epoch, pre.is_new_epoch, pre.epoch_detail) == (ser.epoch, ser.is_new_epoch, ser.epoch_detail), k
            assert pre.consumed_batches() == k + 1 and isinstance(pre._dev_next, dict)
        with pytest.raises(ValueError):
            pre.next_device_batch(torch.device('cuda'), with_event=False, ahead=True)
    finally:
        pre.close()
    monkeypatch.chdir(tmp_path)
    tr = train.main(['--dataset_type', 'mug', '--dataset', str(tmp_path / 'mug'), '--batchsize', '3', '--max_epoch', '2',
                     '--n_filters_gen', '8', '--save_name', 'pf', '--loader_workers', '2', '--snapshot_interval', '5'])
    assert tr.updater.iteration == 4 and tr.updater.epoch == 2
    tr.updater.get_iterator('main').close()


@pytest.mark.parametrize("model", ['normal', 'cgan'])
def test_train_step_takes_the_loaders_uint8_clips(model):
    """TrainStep.run on the loader's uint8 (N,T,H,W,C) batch (mcg_pack_clip_u8 in front of both discriminators) against the same
    iteration on the reference's float (N,C,T,H,W) batch (model/updater.py:87-92): same Philox streams, same frame index -- losses
    and updated parameters agree to fp32 summation order (the weight gradients' float atomics)."""
    import mocogan_chainer_amd.step as step
    rng = np.random.RandomState(5)
    n, nf, dim_zl = 3, 4, 6
    u8 = rng.randint(0, 256, (n, 16, 64, 64, 3)).astype(np.uint8)
    t_real = torch.tensor(rng.randint(0, 6, n), dtype=torch.int32, device='cuda')
    xs = {'u8': torch.tensor(u8, device='cuda'),
          'f32': torch.tensor(((u8.astype(np.float32) - 128.) / 128.).transpose(0, 4, 1, 2, 3).copy(), device='cuda')}
    res = {}
    for form, x in xs.items():
        gen, di, dv = step.make_models(model, num_labels=dim_zl, seed=3, n_filters=nf)
        ts = step.TrainStep(model, gen, di, dv, seed=21, rank=0)
        ts.run(x, t_real)
        ts.run(x, t_real)
        torch.cuda.synchronize()
        res[form] = (ts.losses(), {'%s/%s' % (name, kk): np.asarray(vv.cpu() if torch.is_tensor(vv) else vv)
                                   for name, net in (('gen', gen), ('di', di), ('dv', dv)) for kk, vv in net.export_reference_params().items()})
    la, lb = res['u8'][0], res['f32'][0]
    assert all(abs(la[k] - lb[k]) < 1e-5 for k in la), (la, lb)
    for k, a in res['u8'][1].items():
        b = res['f32'][1][k]
        if a.dtype.kind == 'f' and a.size:
            assert np.linalg.norm(a - b) <= 2e-4 * (np.linalg.norm(b) + 1e-12), k


def test_train_on_a_moving_mnist_file(tmp_path, monkeypatch):
    """BASELINE configs[0] plumbing: --dataset_type mnist reads the (T, N, 64, 64) uint8 .npy of Moving MNIST, writes the
    frame directories the reference writes (datasets.py:124-139), and trains label-free (dim_zl = 0) on 16-frame crops."""
    import train
    rng = np.random.RandomState(0)
    np.save(tmp_path / 'mnist_test_seq.npy', rng.randint(0, 255, (20, 6, 64, 64)).astype(np.uint8))
    monkeypatch.chdir(tmp_path)
    tr = train.main(['--dataset_type', 'mnist', '--dataset', str(tmp_path / 'mnist_test_seq.npy'), '--batchsize', '3', '--max_epoch', '2',
                     '--n_filters_gen', '8', '--save_name', 'mm', '--snapshot_interval', '5'])
    assert tr.updater.iteration == 4 and tr.updater.epoch == 2
    assert tr.updater.image_gen.dim_zl == 0
    frames = sorted((tmp_path / 'data' / 'dataset' / 'moving_mnist' / 'preprocessed' / '00000').glob('*.jpg'))
    assert len(frames) == 20
    assert all(np.isfinite(v) for v in tr.updater.observation.values())


def test_train_on_single_channel_moving_mnist(tmp_path, monkeypatch):
    """BASELINE configs[0] names 16x1x64x64 clips: --channel 1 trains the single-plane networks (in/out_channels = 1)
    on the same preprocessed JPEG tree (the reference tiles the grey frames to RGB, datasets.py:127)."""
    import train
    rng = np.random.RandomState(0)
    np.save(tmp_path / 'mnist_test_seq.npy', rng.randint(0, 255, (20, 6, 64, 64)).astype(np.uint8))
    monkeypatch.chdir(tmp_path)
    tr = train.main(['--dataset_type', 'mnist', '--dataset', str(tmp_path / 'mnist_test_seq.npy'), '--batchsize', '3', '--max_epoch', '1',
                     '--n_filters_gen', '8', '--save_name', 'mm1', '--snapshot_interval', '5', '--channel', '1',
                     '--log_tensorboard_interval', '100'])
    assert tr.updater.iteration == 2 and tr.updater.epoch == 1
    assert tr.updater.image_gen.out_channels == 1 and tr.updater.video_dis.in_channels == 1
    assert tr.updater.get_iterator('main').dataset[0][0].shape == (1, 16, 64, 64)
    assert all(np.isfinite(v) for v in tr.updater.observation.values())
    with np.load(tmp_path / 'result' / 'mm1' / 'image_gen_epoch_fianl.npz') as f:
        assert f['dc5/W'].shape == (8, 1, 4, 4)
