"""Host-side training-loop conveniences with the names the reference's train.py uses from Chainer
(optimizers.Adam, optimizer.WeightDecay, iterators.SerialIterator, training.Trainer, extensions.*,
serializers.save_npz/load_npz).  Thin and framework-free: the hot path lives in step.py."""
import json
import os
import sys
import time

import numpy as np

from . import step as _step


# ---- optimizers (train.py:93-97) ---------------------------------------------------------------
class WeightDecay:
    name = 'WeightDecay'

    def __init__(self, rate):
        self.rate = rate


class Adam:
    """chainer.optimizers.Adam hyper-parameters; the update itself is the fused kernel (step.adam_update)."""

    def __init__(self, alpha=0.001, beta1=0.9, beta2=0.999, eps=1e-8):
        self.alpha, self.beta1, self.beta2, self.eps = alpha, beta1, beta2, eps
        self.target = None
        self._hooks = {}

    def setup(self, link):
        self.target = link
        return self

    def add_hook(self, hook, name=None):
        self._hooks[name or getattr(hook, 'name', 'hook')] = hook

    @property
    def t(self):
        return self.target.impl.t

    def hyper(self):
        wd = sum(h.rate for h in self._hooks.values() if isinstance(h, WeightDecay))
        return _step.AdamHyper(self.alpha, self.beta1, self.beta2, self.eps, wd)


# ---- iterator (train.py:66) -----------------------------------------------------------------------
class SerialIterator:
    """chainer.iterators.SerialIterator(dataset, batch_size, repeat=True, shuffle=True)."""

    def __init__(self, dataset, batch_size, repeat=True, shuffle=True):
        self.dataset, self.batch_size, self._repeat, self._shuffle = dataset, batch_size, repeat, shuffle
        self.reset()

    def reset(self):
        n = len(self.dataset)
        self._order = np.random.permutation(n) if self._shuffle else np.arange(n)
        self.current_position, self.epoch, self.is_new_epoch = 0, 0, False
        self._previous_epoch_detail = -1.0

    @property
    def epoch_detail(self):
        return self.epoch + self.current_position / len(self.dataset)

    def next(self):
        n = len(self.dataset)
        if not self._repeat and self.epoch > 0:
            raise StopIteration
        self._previous_epoch_detail = self.epoch_detail
        i, i_end = self.current_position, self.current_position + self.batch_size
        batch = [self.dataset[int(j)] for j in self._order[i:i_end]]
        if i_end >= n:
            if self._repeat:
                rest = i_end - n
                if self._shuffle:
                    self._order = np.random.permutation(n)
                if rest > 0:
                    batch.extend(self.dataset[int(j)] for j in self._order[:rest])
                self.current_position = rest
            else:
                self.current_position = 0
            self.epoch += 1
            self.is_new_epoch = True
        else:
            self.is_new_epoch = False
            self.current_position = i_end
        return batch

    __next__ = next

    def __iter__(self):
        return self


# ---- serializers (train.py:139-144,162-163,190-192; generate_samples.py:34) -------------------------
def save_npz(path, link):
    """One network in Chainer's npz key scheme (dc1/W, bn2/avg_var, g0/W_r/W, ...)."""
    np.savez_compressed(str(path), **link.serialize_dict())


def load_npz(path, obj):
    with np.load(str(path)) as f:
        d = {k: f[k] for k in f.files}
    if hasattr(obj, 'load_state'):
        obj.load_state(d)          # a Trainer snapshot
    else:
        obj.load_dict(d)


# ---- trainer + extensions (train.py:132-160) -----------------------------------------------------
class Trainer:
    def __init__(self, updater, stop_trigger, out='result'):
        self.updater, self.out = updater, str(out)
        self.stop_n, self.stop_unit = stop_trigger
        self._ext = []
        self.observation = {}
        self.start = None

    def extend(self, ext, trigger=(1, 'epoch'), name=None):
        self._ext.append((ext, trigger))

    def _fires(self, trigger):
        n, unit = trigger
        u = self.updater
        if unit == 'iteration':
            return u.iteration % n == 0
        return u.is_new_epoch and u.epoch % n == 0

    def _done(self):
        u = self.updater
        return (u.iteration if self.stop_unit == 'iteration' else u.epoch) >= self.stop_n

    @property
    def elapsed_time(self):
        return time.time() - self.start

    def run(self):
        os.makedirs(self.out, exist_ok=True)
        self.start = time.time()
        while not self._done():
            self.observation = {}
            self.updater.update()
            self.observation.update(self.updater.observation)
            for ext, trig in self._ext:
                if self._fires(trig):
                    ext(self)

    # whole-run snapshot (extensions.snapshot): models, optimizers, counters
    def state(self):
        u = self.updater
        d = {'updater/iteration': np.asarray(u.iteration), 'updater/iterator:main/epoch': np.asarray(u.get_iterator('main').epoch),
             'updater/iterator:main/current_position': np.asarray(u.get_iterator('main').current_position)}
        for name, link in u.links().items():
            for k, v in link.impl.export_reference_params().items():
                d['updater/model:%s/%s' % (name, k)] = v
            st = link.impl.export_adam_state()
            d['updater/optimizer:%s/t' % name] = np.asarray(st['t'])
            for k in st['m']:
                d['updater/optimizer:%s/%s/m' % (name, k)] = st['m'][k]
                d['updater/optimizer:%s/%s/v' % (name, k)] = st['v'][k]
        return d

    def load_state(self, d):
        u = self.updater
        u.iteration = int(d['updater/iteration'])
        it = u.get_iterator('main')
        it.epoch = int(d['updater/iterator:main/epoch'])
        it.current_position = int(d['updater/iterator:main/current_position'])
        for name, link in u.links().items():
            pre = 'updater/model:%s/' % name
            link.impl.load_reference_params({k[len(pre):]: v for k, v in d.items() if k.startswith(pre)})
            opre = 'updater/optimizer:%s/' % name
            keys = link.impl.trainable_keys()
            link.impl.load_adam_state({'t': int(d[opre + 't']), 'm': {k: d[opre + k + '/m'] for k in keys},
                                       'v': {k: d[opre + k + '/v'] for k in keys}})


class extensions:
    @staticmethod
    def snapshot(filename='snapshot_epoch_{.updater.epoch}.npz'):
        def ext(trainer):
            np.savez_compressed(os.path.join(trainer.out, filename.format(trainer)), **trainer.state())
        return ext

    @staticmethod
    def snapshot_object(target, filename):
        def ext(trainer):
            save_npz(os.path.join(trainer.out, filename.format(trainer)), target)
        return ext

    class LogReport:
        def __init__(self, trigger=(1, 'epoch'), log_name='log'):
            self.log, self.log_name = [], log_name

        def __call__(self, trainer):
            u = trainer.updater
            entry = dict(trainer.observation, epoch=u.epoch, iteration=u.iteration, elapsed_time=trainer.elapsed_time)
            self.log.append(entry)
            trainer.last_log = entry
            with open(os.path.join(trainer.out, self.log_name), 'w') as f:
                json.dump(self.log, f, indent=4)

    class PrintReport:
        def __init__(self, entries, out=sys.stdout):
            self.entries, self.out, self._header = entries, out, False

        def __call__(self, trainer):
            if not self._header:
                self.out.write('  '.join('%-16s' % e for e in self.entries) + '\n')
                self._header = True
            obs = dict(trainer.observation, epoch=trainer.updater.epoch, iteration=trainer.updater.iteration)
            self.out.write('  '.join('%-16s' % ('%.6g' % obs[e] if isinstance(obs.get(e), float) else obs.get(e, '')) for e in self.entries) + '\n')
            self.out.flush()

    class ProgressBar:
        def __init__(self, update_interval=100, out=sys.stdout):
            self.interval, self.out = update_interval, out

        def __call__(self, trainer):
            u = trainer.updater
            self.out.write('\riter %d  epoch %.3f  %.2f iters/sec' % (u.iteration, u.epoch_detail,
                                                                       u.iteration / max(trainer.elapsed_time, 1e-9)))
            self.out.flush()


class NullWriter:
    """Stand-in for tb_chainer.SummaryWriter when no TensorBoard writer is available."""

    def add_scalar(self, *a, **k):
        pass

    def add_image(self, *a, **k):
        pass


def make_summary_writer(path):
    try:
        from torch.utils.tensorboard import SummaryWriter
        return SummaryWriter(str(path))
    except Exception:
        return NullWriter()
