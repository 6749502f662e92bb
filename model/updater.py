"""Reference-facing ``Updater``: same keyword interface and methods as raahii/mocogan-chainer
``model/updater.py`` (Updater :9-19, loss_dis :21-44, loss_gen :46-63, concat_label_video :65-76,
update_core :78-113).  ``update_core`` hands the batch to the device step (mocogan-chainer_amd/step.py),
which reproduces the reference's kernel-visible ordering without an autograd graph."""
import os

import numpy as np
import torch

import mocogan_chainer_amd.hiplib as _hl
import mocogan_chainer_amd.step as _step


class Updater:
    def __init__(self, *args, **kwargs):
        self.model = kwargs.pop('model')
        self.image_gen, self.image_dis, self.video_dis = kwargs.pop('models')
        self.video_length = kwargs.pop('video_length')
        self.img_size = kwargs.pop('img_size')
        self.channel = kwargs.pop('channel')
        self.dim_zl = kwargs.pop('dim_zl')
        self.tf_writer = kwargs.pop('tensorboard_writer')
        # chainer.training.StandardUpdater(iterator, optimizer, device=...)
        iterator = kwargs.pop('iterator')
        self._iterators = iterator if isinstance(iterator, dict) else {'main': iterator}
        self._optimizers = kwargs.pop('optimizer')
        self.device = kwargs.pop('device', None)
        seed = kwargs.pop('seed', 0)
        exchange = kwargs.pop('exchange', None)
        rank = kwargs.pop('rank', 0)
        # placement / arithmetic options of the MI355X path (no reference counterpart): side HIP streams,
        # MFMA operand type of the conv GEMMs ('f32' | 'bf16')
        overlap = kwargs.pop('overlap', False)
        precision = kwargs.pop('precision', None)
        sync_bn = kwargs.pop('sync_bn', False)
        if kwargs:
            raise TypeError('unexpected arguments: %s' % sorted(kwargs))
        self.iteration = 0
        self.observation = {}
        hyper = {k: self._optimizers[k].hyper() for k in ('image_gen', 'image_dis', 'video_dis')}
        self._step = _step.TrainStep(self.model, self.image_gen.impl, self.image_dis.impl, self.video_dis.impl,
                                     hyper=hyper, exchange=exchange, seed=seed, rank=rank, overlap=overlap,
                                     precision=precision, sync_bn=sync_bn)

    # ---- StandardUpdater surface -------------------------------------------------------------------
    def get_optimizer(self, name):
        return self._optimizers[name]

    def get_iterator(self, name):
        return self._iterators[name]

    def links(self):
        return {'image_gen': self.image_gen, 'image_dis': self.image_dis, 'video_dis': self.video_dis}

    @property
    def epoch(self):
        return self._iterators['main'].epoch

    @property
    def epoch_detail(self):
        return self._iterators['main'].epoch_detail

    @property
    def is_new_epoch(self):
        return self._iterators['main'].is_new_epoch

    def update(self):
        self.update_core()
        self.iteration += 1

    # ---- losses ----------------------------------------------------------------------------------
    def _logits2d(self, y):
        y = torch.as_tensor(y, dtype=torch.float32, device=self._step.device)
        return y.reshape(y.shape[0], -1).contiguous()

    def _labels(self, t):
        return None if t is None else torch.as_tensor(np.asarray(t) if not torch.is_tensor(t) else t,
                                                      dtype=torch.int32, device=self._step.device)

    def loss_dis(self, dis, y_real, y_fake, t_real, t_fake):
        """softplus GAN criterion on batch sample 0 (divided by the batch size) plus, for infogan's
        VideoDiscriminator, the two categorical terms.  Returns the loss (0-dim tensor); the gradients
        w.r.t. the logits are left in ``self.last_loss_grads``."""
        yr, yf = self._logits2d(y_real), self._logits2d(y_fake)
        n, c = yf.shape
        with_ce = self.model == 'infogan' and dis.name == 'VideoDiscriminator'
        loss = torch.empty(1, device=yr.device)
        gr, gf = torch.empty_like(yr), torch.empty_like(yf)
        _hl.loss_dis(n, c, yr, yf, self._labels(t_real), self._labels(t_fake), with_ce, loss, gr, gf)
        self.last_loss_grads = (gr, gf)
        self._report(dis, loss)
        return loss[0]

    def loss_gen(self, gen, y_fake_i, y_fake_v, t_fake):
        yi, yv = self._logits2d(y_fake_i), self._logits2d(y_fake_v)
        n, c = yi.shape
        loss = torch.empty(1, device=yi.device)
        gi, gv = torch.empty_like(yi), torch.empty_like(yv)
        _hl.loss_gen(n, c, yi, yv, self._labels(t_fake), self.model == 'infogan', loss, gi, gv)
        self.last_loss_grads = (gi, gv)
        self._report(gen, loss)
        return loss[0]

    def _report(self, link, loss):
        if self.is_new_epoch:
            v = float(loss)
            self.observation['%s/loss' % self._name_of(link)] = v
            self.tf_writer.add_scalar('loss:{}'.format(link.name), v, self.epoch)

    def _name_of(self, link):
        for k, v in self.links().items():
            if v is link:
                return k
        return link.name

    def concat_label_video(self, video, label, xp=None):
        """(N,C,T,H,W) -> (N,C+dim_zl,T,H,W): dim_zl planes of -1 with the label's plane set to +1."""
        video = torch.as_tensor(video)
        n, c, t, h, w = video.shape
        lv = -torch.ones((n, self.dim_zl, t, h, w), dtype=video.dtype, device=video.device)
        lv[torch.arange(n), torch.as_tensor(np.asarray(label) if not torch.is_tensor(label) else label).long()] = 1.0
        return torch.cat((video, lv), dim=1)

    # ---- one iteration ---------------------------------------------------------------------------
    def update_core(self):
        it = self.get_iterator('main')
        ready = t_real = None
        if hasattr(it, 'next_device_batch'):                                         # prefetching loader: uint8 from pinned memory on
            # a copy stream; raw (uint8) datasets stay uint8 (N,T,H,W,C): TrainStep.run normalises in its first kernels (MCG_LOADER_U8=0:
            # the float (N,C,T,H,W) batch of rounds 1-5, five torch passes on the copy stream)
            # ahead: the copy of the NEXT batch is queued now as well (MCG_LOADER_AHEAD=0: when it is needed, as in rounds 1-5)
            x_real, labels, ready, t_real = it.next_device_batch(self._step.device, with_event=True,
                                                                 as_uint8=os.environ.get('MCG_LOADER_U8', '1') == '1',
                                                                 ahead=os.environ.get('MCG_LOADER_AHEAD', '1') == '1')
        else:
            batch = it.next()
            labels = [b[1] for b in batch]
            # concat_examples straight into a reused PINNED staging buffer: one host copy, and the H2D copy is asynchronous
            shape = (len(batch),) + tuple(np.shape(batch[0][0]))
            pin = getattr(self, '_pin', None)
            if pin is None or tuple(pin.shape) != shape:
                pin = self._pin = torch.empty(shape, dtype=torch.float32, pin_memory=torch.cuda.is_available())
            evt = getattr(self, '_pin_evt', None)
            if evt is not None:
                evt.synchronize()                                                    # the previous batch has left the buffer
            np.stack([b[0] for b in batch], out=pin.numpy(), casting='unsafe')
            x_real = pin.to(self._step.device, non_blocking=True)
            if x_real.is_cuda:
                self._pin_evt = torch.cuda.Event()
                self._pin_evt.record()
        if t_real is None and labels[0] is not None:
            # (pinned + non-blocking: a copy from pageable memory blocks the host until everything queued before it has run)
            lab = getattr(self, '_pin_lab', None)
            if lab is None or lab.numel() < len(labels):
                lab = self._pin_lab = torch.empty(len(labels), dtype=torch.int32, pin_memory=torch.cuda.is_available())
            evt = getattr(self, '_pin_lab_evt', None)
            if evt is not None:
                evt.synchronize()
            lab[:len(labels)] = torch.from_numpy(np.asarray(labels, dtype=np.int32))
            t_real = lab[:len(labels)].to(self._step.device, non_blocking=True)
            if t_real.is_cuda:
                self._pin_lab_evt = torch.cuda.Event()
                self._pin_lab_evt.record()
        # (the loader is a batch ahead: its copy's event lets the two-chain schedule start the next real chain under this iteration's tail)
        self._step.run(x_real, t_real, input_event=ready)
        if self.is_new_epoch:
            l = self._step.losses()
            self.observation = dict(l)
            for name, link in self.links().items():
                self.tf_writer.add_scalar('loss:{}'.format(link.name), l['%s/loss' % name], self.epoch)
