#!/usr/bin/env python
"""Training entry point with the reference's flags (raahii/mocogan-chainer train.py:25-45).

Extensions over the reference: ``--dataset_type synthetic`` (no files needed) and, when launched
through ``python -m torch.distributed.run``, data-parallel training with one process per GPU
(gradients averaged over RCCL)."""
import argparse
import os
from datetime import datetime
from pathlib import Path

import numpy as np

from datasets import MugDataset, MovingMnistDataset, SyntheticDataset      # torch-free: all a loader worker needs
# (model.net, model.updater, util and the trainer -- torch and the HIP binding -- are imported inside main(): the
#  loader's spawned worker processes re-import this module and must stay light)


def parse_args(argv=None):
    try:
        from pytz import timezone
        default_name = datetime.now(timezone('Asia/Tokyo')).strftime("%Y_%m%d_%H%M")
    except Exception:
        default_name = datetime.now().strftime("%Y_%m%d_%H%M")
    p = argparse.ArgumentParser(description='MoCoGAN training on MI355X; flags and defaults of the reference train.py:21-47')
    # (flags, type, default, help): names, short forms and defaults are the reference's interface, the texts are ours
    reference_flags = [
        (('--gpu', '-g'), int, -1, 'device index; this build always runs on an MI355X (the flag keeps old command lines valid)'),
        (('--dataset',), str, 'data/dataset/train', 'root directory of the frame-directory dataset'),
        (('--batchsize',), int, 100, 'clips per iteration (per GPU under torch.distributed.run)'),
        (('--max_epoch',), int, 1000, 'stop after this many passes over the dataset'),
        (('--save_name',), str, default_name, 'sub-directory of result/ and runs/ for log, snapshots and TensorBoard files'),
        (('--display_interval',), int, 1, 'epochs between console reports'),
        (('--snapshot_interval',), int, 10, 'epochs between .npz snapshots'),
        (('--log_tensorboard_interval',), int, 10, 'epochs between TensorBoard sample videos'),
        (('--num_gen_samples',), int, 36, 'videos per TensorBoard sample grid (a square number)'),
        (('--dim_zc',), int, 50, 'size of the content code z_c'),
        (('--dim_zm',), int, 10, 'size of the motion code z_m (GRU state).  dim_zm <= 16 and dim_zm + num_labels <= 32 (the '
                                 'reference\'s default 10, every BASELINE config) run the fused register-resident GRU kernels; up to '
                                 'dim_zm = 64 / dim_zm + num_labels = 128 a slower kernel pair with the weights in memory; beyond that '
                                 'MCG_ERR_UNSUPPORTED (the reference, model/net.py:38-41, accepts any)'),
        (('--n_filters_gen',), int, 64, 'base width; as in the reference it is used for all three networks'),
        (('--n_filters_idis',), int, 64, 'accepted and reported, not used (reference quirk)'),
        (('--n_filters_vdis',), int, 64, 'accepted and reported, not used (reference quirk)'),
        (('--resume', '-r'), str, '', 'trainer snapshot (.npz) to continue from'),
    ]
    for flags, typ, default, text in reference_flags:
        p.add_argument(*flags, type=typ, default=default, help=text)
    p.add_argument('--dataset_type', default='mug', choices=['mug', 'mnist', 'synthetic'],
                   help="'synthetic' (this build only) needs no files")
    p.add_argument('--model', default='normal', choices=['normal', 'cgan', 'infogan'], help='model variant')
    p.add_argument('--synthetic_size', type=int, default=256, help='clips in the synthetic dataset')
    p.add_argument('--seed', type=int, default=0)
    p.add_argument('--channel', type=int, default=3, choices=[1, 3],
                   help="colour planes of the clips: 3 = the reference's constant (train.py:49); 1 = grey-scale clips "
                        "(the Moving-MNIST shape 16x1x64x64; mnist and synthetic datasets)")
    # MI355X-path options (no counterpart in the reference's train.py)
    p.add_argument('--mfma', choices=['f32', 'bf16', 'f32x3'], default='f32',
                   help="operand type of the convolution GEMMs (accumulation, parameters, Adam: always fp32); f32x3: fp32 products "
                        "of the wide layers on the bf16 matrix pipe (operands as three bf16 terms, six bf16 products each)")
    p.add_argument('--overlap', type=int, default=1, help="side HIP streams for independent kernels")
    p.add_argument('--dp_shard', type=int, default=1,
                   help="data parallel: 1 (default) = rank r trains on items r, r + world, ... (an epoch of all ranks is one pass over "
                        "the data); 0 = every rank walks the whole dataset in its own order (an epoch is world passes)")
    p.add_argument('--autotune', type=int, default=1,
                   help="1: time the GEMM tile candidates once per layer geometry not in the shipped table; 0: the library's tile "
                        "heuristic, and for --mfma f32x3 only the shipped table's split / fp32 decisions (geometries it does not "
                        "hold run the fp32-MFMA form; a warning says so once)")
    p.add_argument('--sync_bn', type=int, default=0,
                   help="data parallel only: 1 = BatchNorm statistics over the global batch (all-reduced sums) instead of per rank")
    p.add_argument('--loader_workers', type=int, default=0,
                   help="0: the reference's serial in-process loading (SerialIterator); N > 0: N worker processes decode the "
                        "next batches while the GPU trains (PrefetchIterator)")
    return p.parse_args(argv)


def main(argv=None):
    args = parse_args(argv)
    size, channel, video_length = 64, args.channel, 16           # train.py:48-50 (channel is the constant 3 there)
    use_noise, noise_sigma = True, 0.2                           # train.py:56-57
    nf = args.n_filters_gen                                      # the reference passes n_filters_gen to all three nets

    import torch
    from model.net import ImageGenerator, ImageDiscriminator, VideoDiscriminator
    from model.updater import Updater
    from util import log_tensorboard
    from mocogan_chainer_amd import trainer as T
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')     # (RCCL between processes needs dmabuf IPC on this driver; before HIP starts)
    if not torch.cuda.is_available():
        raise SystemExit('train.py needs an MI355X: the HIP path has no CPU fallback (the reference\'s --gpu -1 CPU mode '
                         'is what oracle/ restates for tests)')
    torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', max(args.gpu, 0))))
    import mocogan_chainer_amd.hiplib as hl
    hl.set_autotune(bool(args.autotune))
    exchange = None
    if world > 1:
        import torch.distributed as dist
        from mocogan_chainer_amd.step import GradExchange
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', rank=rank, world_size=world)
        exchange = GradExchange()
    np.random.seed(args.seed)                                    # identical initial weights on every rank

    if args.dataset_type == "mug":
        if channel != 3:
            raise ValueError('--channel 1 needs --dataset_type mnist or synthetic (MUG clips are RGB)')
        num_labels, train_dataset = 6, MugDataset(args.dataset, video_length)
    elif args.dataset_type == "mnist":
        num_labels, train_dataset = 0, MovingMnistDataset(args.dataset, video_length, channels=channel)
    else:
        num_labels, train_dataset = 6, SyntheticDataset(args.synthetic_size, 6, channel, video_length, size,
                                                        seed=0 if (world > 1 and args.dp_shard) else rank)   # (shards partition ONE dataset)

    if args.model == "normal":
        use_label, c_d, out_d = False, channel, 1
    elif args.model == "cgan":
        if num_labels == 0:
            raise ValueError("Called cgan model, but dataset has no label.")
        use_label, c_d, out_d = True, channel + num_labels, 1
    else:
        if num_labels == 0:
            raise ValueError("Called cgan model, but dataset has no label.")
        use_label, c_d, out_d = True, channel, 1 + num_labels
    if exchange is not None and args.autotune:
        # every rank runs the same GEMM tile codes: geometries the shipped table does not hold are tuned by one local
        # iteration on throw-away networks, then rank 0's table replaces everyone's (a straggler would set the step time)
        from mocogan_chainer_amd.step import pretune_and_share_tiles
        pretune_and_share_tiles(exchange, args.model, args.mfma, args.batchsize, rank, num_labels=num_labels, channel=channel,
                                dim_zc=args.dim_zc, dim_zm=args.dim_zm, n_filters=nf, video_length=video_length)
    image_gen = ImageGenerator(args.dim_zc, args.dim_zm, num_labels, channel, nf, video_length)
    image_dis = ImageDiscriminator(c_d, out_d, nf, use_noise, noise_sigma)
    video_dis = VideoDiscriminator(c_d, out_d, nf, use_noise, noise_sigma)
    if world > 1 and args.dp_shard:
        from datasets import ShardedDataset
        train_dataset = ShardedDataset(train_dataset, rank, world)     # the ranks together make one pass over the data per epoch
    np.random.seed(args.seed + 1 + rank)                         # data order / sub-sequence offsets differ per rank
    if args.loader_workers > 0:
        train_iter = T.PrefetchIterator(train_dataset, args.batchsize, n_workers=args.loader_workers, seed=args.seed + rank)
    else:
        train_iter = T.SerialIterator(train_dataset, args.batchsize)

    def make_optimizer(model, alpha=1e-3, beta1=0.9, beta2=0.999):
        optimizer = T.Adam(alpha=alpha, beta1=beta1)              # beta2 is not forwarded (train.py:94)
        optimizer.setup(model)
        optimizer.add_hook(T.WeightDecay(1e-5), 'hook_dec')
        return optimizer

    opts = {'image_gen': make_optimizer(image_gen, 2e-4, 5e-5, 0.999),
            'image_dis': make_optimizer(image_dis, 2e-4, 5e-5, 0.999),
            'video_dis': make_optimizer(video_dis, 2e-4, 5e-5, 0.999)}
    writer = T.make_summary_writer(Path('runs') / args.save_name) if rank == 0 else T.NullWriter()
    updater = Updater(model=args.model, models=(image_gen, image_dis, video_dis), video_length=video_length,
                      img_size=size, channel=channel, dim_zl=num_labels, iterator=train_iter,
                      tensorboard_writer=writer, optimizer=opts, device=args.gpu, seed=args.seed, exchange=exchange, rank=rank,
                      overlap=bool(args.overlap), precision=args.mfma, sync_bn=bool(args.sync_bn))

    save_path = Path('result') / args.save_name
    trainer = T.Trainer(updater, (args.max_epoch, 'epoch'), out=save_path)
    if rank == 0:
        snap = (args.snapshot_interval, 'epoch')
        trainer.extend(T.extensions.snapshot(filename='snapshot_epoch_{.updater.epoch}.npz'), trigger=snap)
        trainer.extend(T.extensions.snapshot_object(image_gen, 'image_gen_epoch_{.updater.epoch}.npz'), trigger=snap)
        trainer.extend(T.extensions.snapshot_object(image_dis, 'image_dis_epoch_{.updater.epoch}.npz'), trigger=snap)
        trainer.extend(T.extensions.snapshot_object(video_dis, 'video_dis_epoch_{.updater.epoch}.npz'), trigger=snap)
        disp = (args.display_interval, 'epoch')
        trainer.extend(T.extensions.LogReport(trigger=disp), trigger=disp)
        trainer.extend(T.extensions.PrintReport(['epoch', 'iteration', 'image_gen/loss', 'image_dis/loss', 'video_dis/loss']),
                       trigger=disp)
        if np.sqrt(args.num_gen_samples) % 1.0 != 0:
            raise ValueError('--num_gen_samples must be n^2 (n: natural number).')
        trainer.extend(log_tensorboard(image_gen, args.num_gen_samples, video_length, writer),
                       trigger=(args.log_tensorboard_interval, 'epoch'))
    if args.resume:
        T.load_npz(args.resume, trainer)
    if exchange is not None:
        # Data parallel (no reference counterpart).  Replicas are identical by construction (same seed) and stay so
        # because every rank applies the same averaged gradient; rank 0's parameters, Adam moments and running
        # statistics are still broadcast once, after construction / resume, so that nothing depends on that.
        # Epoch semantics: with --dp_shard 1 (default) rank r walks items r, r + world, ... in its own order, so one epoch of all
        # ranks is ONE pass over the data; with --dp_shard 0 every rank walks the WHOLE dataset (an "epoch" is world-size passes).
        # BatchNorm running statistics are per rank and rank 0's are the ones saved.
        import torch
        for link in (image_gen, image_dis, video_dis):
            net = link.impl
            exchange.broadcast_params([net.fp.p, net.fp.m, net.fp.v] + list(net.running.values()))
            # the Adam step counter (it sets lr_t) and the BatchNorm call counts travel too; the bf16 copy of the
            # weights is rebuilt from what arrived
            names = sorted(net.bn_count)
            cnt = torch.tensor([net.t] + [net.bn_count[k] for k in names], dtype=torch.int64, device=net.fp.p.device)
            exchange.broadcast_params([cnt])
            cnt = cnt.tolist()
            net.t = int(cnt[0])
            for k, v in zip(names, cnt[1:]):
                net.bn_count[k] = int(v)
            net.fp.touch()
            if net.precision == 'bf16':
                net.fp.refresh16()

    if rank == 0:
        # the reference's start-up banner (train.py:165-187), same lines and order
        banner = (('gpu', '{}  (world size {})'.format(args.gpu, world)), ('minibatch size', args.batchsize),
                  ('max epoch', args.max_epoch), ('num batches', len(train_dataset) // args.batchsize),
                  ('data size', len(train_dataset)), ('data shape', train_dataset[0][0].shape),
                  ('num filters igen', nf), ('num filters idis', args.n_filters_idis), ('num filters vdis', args.n_filters_vdis),
                  ('use noise', '{}(sigma={})'.format(use_noise, noise_sigma)), ('use label', use_label),
                  ('snapshot interval', args.snapshot_interval), ('log tensorboard interval', args.log_tensorboard_interval),
                  ('num generate samples', args.num_gen_samples))
        print('\n'.join(['[ Training configuration ]'] + ['# %s: %s' % kv for kv in banner]) + '\n')
    trainer.run()
    if rank == 0:
        T.save_npz(save_path / 'image_gen_epoch_fianl.npz', image_gen)     # (sic) train.py:190-192
        T.save_npz(save_path / 'image_dis_epoch_fianl.npz', image_dis)
        T.save_npz(save_path / 'video_dis_epoch_fianl.npz', video_dis)
    return trainer


if __name__ == '__main__':
    main()
