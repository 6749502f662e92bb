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

    def load_state(self, epoch, current_position, order=None, is_new_epoch=None, previous_epoch_detail=None):
        """Restore the position a snapshot recorded (Chainer's SerialIterator.serialize: current_position, epoch,
        is_new_epoch, order, previous_epoch_detail).  ``order`` None keeps the current permutation."""
        self.epoch, self.current_position = int(epoch), int(current_position)
        if order is not None and len(order) == len(self.dataset):
            self._order = np.asarray(order, dtype=np.int64).copy()
        if is_new_epoch is not None:
            self.is_new_epoch = bool(is_new_epoch)
        if previous_epoch_detail is not None:
            self._previous_epoch_detail = float(previous_epoch_detail)

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


def _pid_namespace():
    """an id of this process's PID namespace (the inode of /proc/self/ns/pid): pids are only meaningful inside one namespace, and
    containers that share /dev/shm (`--ipc=host`) usually do not share it"""
    try:
        return '%x' % os.stat('/proc/self/ns/pid').st_ino
    except OSError:
        return '0'


def _slot_name(owner_id, k):
    return 'mcg_%s_%d_%x_%d' % (_pid_namespace(), os.getpid(), owner_id & 0xffffff, k)


def _remove_stale_slots(root='/dev/shm'):
    """batch slots ('mcg_<pid namespace>_<pid>_...') of training processes OF THIS PID NAMESPACE that no longer exist -- a run that was
    killed could not unlink its own, and /dev/shm is memory.  Segments of another namespace are never touched: their pid says nothing
    here (round 5's advice: two containers sharing /dev/shm would unlink each other's live slots)."""
    try:
        names = os.listdir(root)
    except OSError:
        return
    ns = _pid_namespace()
    for name in names:
        parts = name.split('_')
        if len(parts) == 5 and parts[0] == 'mcg' and parts[1] == ns and ns != '0' and parts[2].isdigit():
            pid = int(parts[2])
            try:
                os.kill(pid, 0)                                       # (signal 0: existence check only)
            except ProcessLookupError:
                try:
                    os.unlink(os.path.join(root, name))
                except OSError:
                    pass
            except OSError:
                pass                                                  # (exists, not ours to judge)


class PrefetchIterator(SerialIterator):
    """SerialIterator's order and epoch bookkeeping, with the samples of the next `prefetch` batches being
    decoded by `n_workers` worker processes while the GPU trains (the reference's
    single-threaded PIL loop delivers a few hundred clips/s, the GPU consumes ~2000).

    * index order, wrap-around batches, ``epoch`` / ``is_new_epoch`` / ``epoch_detail`` are exactly those
      of SerialIterator for the same NumPy seed: the index lists are drawn by the same code, only earlier;
    * workers are *spawned* (never forked from a process that may have touched the GPU); their work functions live
      in the torch-free ``datasets.py`` (spawn also re-imports the parent's main module, so a script that wants
      light workers keeps its heavy imports inside ``main()``, as train.py does); random crop offsets are drawn in the workers, each seeded from (seed, batch number), so
      a run is reproducible but not sample-identical to the serial loop;
    * datasets that offer ``get_example_raw`` ship uint8 frames (a quarter of the bytes through the pipes
      and over PCIe); ``next_device_batch`` copies them from pinned memory on a side stream and normalises /
      transposes on the GPU;
    * ``next()`` still returns the reference's list of ``(video float32 (C,T,H,W), label)``."""

    def __init__(self, dataset, batch_size, repeat=True, shuffle=True, n_workers=8, prefetch=4, chunk=None, seed=0):
        import concurrent.futures as cf
        import multiprocessing as mp
        self._raw = hasattr(dataset, 'get_example_raw')
        import datasets as _ds                                        # torch-free module: all a worker imports
        self._pool = cf.ProcessPoolExecutor(max_workers=n_workers, mp_context=mp.get_context('spawn'),
                                            initializer=_ds.worker_init, initargs=(seed, dataset))
        if chunk is None:                                             # clips per worker task: two tasks per worker and batch, at least 4 clips
            chunk = max(4, -(-batch_size // (2 * max(n_workers, 1))))  # (64 tasks per 256-clip batch cost the training thread 3-6 ms in submit / result)
        self._depth, self._chunk, self._queue, self._batch_no = prefetch, chunk, [], 0
        self._copy_stream = None
        # raw (uint8) datasets: the workers write the decoded frames into shared-memory batch slots and return only the labels
        # (MCG_LOADER_SHM=0: back through the result pipes, as before round 5)
        self._slots, self._free_slots, self._clip_shape = [], [], None
        if self._raw and os.environ.get('MCG_LOADER_SHM', '1') == '1':
            from multiprocessing import shared_memory
            _remove_stale_slots()
            st = np.random.get_state()                                # (a dataset may draw sub-sequence offsets: the order stays SerialIterator's)
            self._clip_shape = tuple(np.asarray(dataset.get_example_raw(0)[0]).shape)
            np.random.set_state(st)
            nbytes = batch_size * int(np.prod(self._clip_shape))
            try:
                for k in range(prefetch + 2):
                    shm = shared_memory.SharedMemory(create=True, size=nbytes, name=_slot_name(id(self), k))
                    self._slots.append((shm, np.ndarray((batch_size,) + self._clip_shape, dtype=np.uint8, buffer=shm.buf)))
                    # create=True only ftruncates: tmpfs pages are not reserved, and a worker's first write into a slot that
                    # /dev/shm cannot back dies with SIGBUS (the 64 MB Docker default against ~300 MB at 256 clips).  Reserve now.
                    os.posix_fallocate(shm._fd, 0, nbytes)
            except OSError as exc:                                    # ENOSPC, or a name left behind by an un-closed iterator (EEXIST)
                sys.stderr.write('PrefetchIterator: no shared-memory batch slots (%r): the workers return the frames through the '
                                 'result pipes (MCG_LOADER_SHM=0 does the same without trying)\n' % (exc,))
                while self._slots:
                    shm = self._slots.pop()[0]
                    for fn in (shm.unlink, shm.close):
                        try:
                            fn()
                        except Exception:
                            pass
            self._free_slots = list(range(len(self._slots)))
        super().__init__(dataset, batch_size, repeat, shuffle)

    def _reclaim(self, block=False):
        """page-locked slots whose device copy has completed go back to the free list (block: wait for the oldest if none is free)"""
        busy = getattr(self, '_busy', [])
        while busy and (busy[0][1].query() or (block and not self._free_slots)):
            slot, ev = busy.pop(0)
            ev.synchronize()
            self._free_slots.append(slot)

    def _drop_queue(self):
        """forget the look-ahead: cancel what has not started, wait for what is running (it writes into a slot), free the slots"""
        self._dev_next = None                                         # (a batch already on the device: its slot returns through _busy)
        for _, futs, slot in getattr(self, '_queue', []):
            for f in futs:
                if not f.cancel():
                    try:
                        f.result()
                    except Exception:
                        pass
            if slot is not None:
                self._free_slots.append(slot)
        self._queue = []

    def reset(self):
        super().reset()
        self._drop_queue()
        self._head = self._state()

    def load_state(self, epoch, current_position, order=None, is_new_epoch=None, previous_epoch_detail=None):
        """As SerialIterator.load_state; the look-ahead (batches already queued from the old position) is dropped
        and restarts from the restored position."""
        self._drop_queue()
        super().load_state(epoch, current_position, order, is_new_epoch, previous_epoch_detail)
        self._head = self._state()

    def consumed_batches(self):
        """number of batches handed out so far (the look-ahead's are not counted): what a snapshot stores, so that a
        resumed run seeds the workers' per-batch draws (sub-sequence offsets) where this one stopped"""
        return self._batch_no - len(self._queue) - (1 if isinstance(getattr(self, '_dev_next', None), dict) else 0)

    # -- bookkeeping: SerialIterator.next() minus the loading ------------------------------------------
    def _state(self):
        return (self.current_position, self.epoch, self.is_new_epoch, self._previous_epoch_detail, self._order)

    def _set_state(self, st):
        self.current_position, self.epoch, self.is_new_epoch, self._previous_epoch_detail, self._order = st

    def _advance(self):
        """Draws the index list of the next batch and moves the bookkeeping; returns the indices."""
        n = len(self.dataset)
        if not self._repeat and self.epoch > 0:
            return None
        self._previous_epoch_detail = self.epoch_detail
        i, i_end = self.current_position, self.current_position + self.batch_size
        idx = [int(j) for j in self._order[i:i_end]]
        if i_end >= n:
            if self._repeat:
                rest = i_end - n
                if self._shuffle:
                    self._order = np.random.permutation(n)
                if rest > 0:
                    idx.extend(int(j) for j in self._order[:rest])
                self.current_position = rest
            else:
                self.current_position = 0
            self.epoch += 1
            self.is_new_epoch = True
        else:
            self.is_new_epoch = False
            self.current_position = i_end
        return idx

    def _fill(self):
        import datasets as _ds
        user = self._state()
        self._set_state(self._head)                                   # continue where the look-ahead stopped
        while len(self._queue) < self._depth:
            idx = self._advance()
            if idx is None:
                break
            slot = None
            if self._slots and len(idx) <= self.batch_size:
                if not self._free_slots:
                    self._reclaim(block=True)
                if not self._free_slots:
                    raise RuntimeError('PrefetchIterator: every batch slot is in use and none is waiting for a device copy '
                                       '(a slot was lost: %d slots, look-ahead %d)' % (len(self._slots), len(self._queue)))
                slot = self._free_slots.pop()
                futs = [self._pool.submit(_ds.worker_load_shm, idx[c:c + self._chunk], self._batch_no, c, self._slots[slot][0].name, c,
                                          self._clip_shape) for c in range(0, len(idx), self._chunk)]
            else:
                futs = [self._pool.submit(_ds.worker_load, idx[c:c + self._chunk], self._raw, self._batch_no, c)
                        for c in range(0, len(idx), self._chunk)]
            self._batch_no += 1
            self._queue.append((self._state(), futs, slot))
        self._head = self._state()
        self._set_state(user)

    def _pop(self, out=None, keep_slot=False):
        """out: a callable (shape, dtype) -> NumPy array the batch is assembled IN (a pinned staging buffer), or None.
        keep_slot (page-locked slots): no copy at all -- returns (view of the batch slot, labels, slot); the caller releases the slot
        (`_busy`) once its device copy has read it."""
        self._fill()
        if not self._queue:
            raise StopIteration
        st, futs, slot = self._queue.pop(0)
        try:
            parts = [f.result() for f in futs]
        except BaseException:
            for f in futs:                                            # (the others may still be writing into the slot)
                if not f.cancel():
                    try:
                        f.result()
                    except BaseException:
                        pass
            if slot is not None:
                self._free_slots.append(slot)
            raise
        self._set_state(st)                                           # what SerialIterator shows after this batch
        if slot is not None:                                          # the frames are in the batch slot; the workers returned the labels
            labels = [l for p in parts for l in p]
            view = self._slots[slot][1][:len(labels)]
            if keep_slot:
                return view, labels, slot
            if out is not None:
                videos = out(view.shape, view.dtype)
                np.copyto(videos, view)
            else:
                videos = view.copy()
            self._free_slots.append(slot)
            self._fill()
            return videos, labels
        self._fill()
        arrs = [p[0] for p in parts]
        if out is not None:
            shape = (sum(a.shape[0] for a in arrs),) + arrs[0].shape[1:]
            videos = np.concatenate(arrs, out=out(shape, arrs[0].dtype))
        else:
            videos = np.concatenate(arrs)
        labels = [l for p in parts for l in p[1]]
        return videos, labels

    def next(self):
        videos, labels = self._pop()
        if self._raw:
            videos = ((videos.astype(np.float32) - 128.) / 128.).transpose(0, 4, 1, 2, 3)
        return [(videos[i], labels[i]) for i in range(len(labels))]

    __next__ = next

    def next_device_batch(self, device, with_event=False, as_uint8=False, ahead=False):
        """-> (x_real float32 (N,C,T,H,W) on `device`, labels list).  The H2D copy runs from pinned memory on a
        side stream; the caller's stream waits for it.  as_uint8 (raw datasets only; otherwise ignored): x_real is the copied
        uint8 batch (N,T,H,W,C) itself -- step.TrainStep.run normalises it in its first kernels (mcg_pack_clip_u8), which saves the
        five torch passes below (1.4 GB of traffic per 256-clip batch beside the GEMMs).  with_event: -> (x_real, labels, event, labels_dev) -- the event recorded on
        the copy stream when the batch was complete (`TrainStep.run(input_event=...)` lets a stream that needs nothing but the batch,
        the VideoDiscriminator's real chain, wait for exactly that instead of for everything queued on the caller's stream) and the
        labels as an int32 device tensor copied on that stream too (None for an unlabelled dataset).
        ahead (round 6; every call of an iterator must then use the same arguments): the copy of the batch AFTER the one returned is
        issued in this call too, one iteration early.  HIP maps streams onto four hardware queues in creation order and TrainStep's
        six side streams exist already, so the copy stream shares a queue with a compute stream and a copy waits behind that
        stream's queued kernels: issued when it is needed, it completed up to an iteration late (the host waited 5 ms per
        iteration at batch 256 for a batch slot whose copy had not run; the next real chain for its input).  Epoch bookkeeping
        (`epoch`, `is_new_epoch`, `epoch_detail`) stays that of the batch RETURNED."""
        key = (str(device), bool(with_event), bool(as_uint8))
        rec, self._dev_next = getattr(self, '_dev_next', None), None
        if rec is StopIteration:
            raise StopIteration
        if rec is not None and rec['key'] != key:
            raise ValueError('next_device_batch(ahead=True): the look-ahead batch was issued with other arguments %r' % (rec['key'],))
        if rec is None:
            rec = self._issue_device_batch(device, with_event, as_uint8, key)
        out = self._consume_device_batch(rec, with_event)
        if ahead:
            try:
                self._dev_next = self._issue_device_batch(device, with_event, as_uint8, key)
            except StopIteration:
                self._dev_next = StopIteration
        return out

    def _consume_device_batch(self, rec, with_event):
        """hand an issued batch to the caller's stream: that stream waits for the copy, the iterator shows the batch's bookkeeping"""
        import torch
        cur = torch.cuda.current_stream()
        cur.wait_event(rec['event'])
        rec['dev'].record_stream(cur)
        if rec['lab_dev'] is not None:
            rec['lab_dev'].record_stream(cur)
        self._set_state(rec['state'])
        return (rec['dev'], rec['labels'], rec['event'], rec['lab_dev']) if with_event else (rec['dev'], rec['labels'])

    def _issue_device_batch(self, device, with_event, as_uint8, key):
        """pop the next batch and queue its H2D copy on the copy stream; the iterator's visible bookkeeping does not move"""
        import torch
        if self._copy_stream is None:
            # (MCG_LOADER_STREAM_SKIP throw-away streams created first choose WHICH hardware queue the copy stream shares;
            #  MCG_LOADER_STREAM_PRIO=-1 asks for a high-priority stream: measured in round 6 -- no placement avoids the sharing, and the
            #  high-priority stream costs the iteration 8 %; see `ahead` above for what helps)
            skip = int(os.environ.get('MCG_LOADER_STREAM_SKIP', '0'))
            self._skipped_streams = [torch.cuda.Stream(device=device) for _ in range(skip)]
            self._copy_stream = torch.cuda.Stream(device=device, priority=int(os.environ.get('MCG_LOADER_STREAM_PRIO', '0')))
            self._pin, self._pin_free, self._pin_i, self._pin_lab = [None] * 3, [None] * 3, 0, [None] * 3
            # The batch slots themselves are page-locked when that is possible (hipHostRegister of the shared-memory segments): the
            # workers then decode straight into pinned memory and the H2D copy reads the slot -- no host copy at all (50 MB per batch
            # at 256 clips: 8-10 ms of the training thread, which at bf16 batch 256 made the HOST the bound: 22.7 against 20.2 ms)
            self._busy, self._slots_pinned = [], False
            if self._slots and os.environ.get('MCG_LOADER_PIN_SLOTS', '1') == '1':
                try:
                    from torch.cuda._pin_memory_utils import pin_memory as _pin
                    for shm, arr in self._slots:
                        _pin(arr.ctypes.data, arr.nbytes)
                    self._slots_pinned = torch.from_numpy(self._slots[0][1]).is_pinned()
                except Exception as exc:                              # (fall back to the staging ring)
                    sys.stderr.write('PrefetchIterator: batch slots not page-locked (%r): staging through pinned buffers\n' % (exc,))
        user = self._state()
        try:
            return self._issue_popped(device, with_event, as_uint8, key)
        finally:
            self._set_state(user)                                     # (_pop moved the bookkeeping to the popped batch: rec['state'] holds it)

    def _issue_popped(self, device, with_event, as_uint8, key):
        import torch
        if getattr(self, '_slots_pinned', False):
            self._reclaim()
            view, labels, slot = self._pop(keep_slot=True)
            host = torch.from_numpy(view)
            i = self._pin_i = (self._pin_i + 1) % 3
            with torch.cuda.stream(self._copy_stream):
                dev = host.to(device, non_blocking=True)
                if self._raw and not as_uint8:
                    dev = ((dev.float() - 128.) / 128.).permute(0, 4, 1, 2, 3).contiguous()
                lab_dev = None
                if with_event and labels and labels[0] is not None:
                    if self._pin_lab[i] is None or self._pin_lab[i].numel() < len(labels):
                        self._pin_lab[i] = torch.empty(max(len(labels), self.batch_size), dtype=torch.int32, pin_memory=True)
                    elif self._pin_free[i] is not None:
                        self._pin_free[i].synchronize()
                    self._pin_lab[i][:len(labels)] = torch.from_numpy(np.asarray(labels, dtype=np.int32))
                    lab_dev = self._pin_lab[i][:len(labels)].to(device, non_blocking=True)
                done = torch.cuda.Event()
                done.record(self._copy_stream)
                self._pin_free[i] = done
                self._busy.append((slot, done))
            return dict(key=key, dev=dev, labels=labels, event=done, lab_dev=lab_dev, state=self._state())
        # Otherwise: a ring of three PINNED staging buffers, allocated once; the batch is assembled in one of them and copied from there.
        # (Round 5, found by timing the product path -- tools/bench_train.py: `torch.from_numpy(videos).pin_memory()` allocated and
        # freed pinned memory every iteration, and freeing pinned memory waits for the device -- 35-40 ms per batch whatever its
        # size, the loader SLOWER than the serial loop: 900 against 1900 clips/s at batch 32.)
        i = self._pin_i = (self._pin_i + 1) % 3

        def staging(shape, dtype):
            t = self._pin[i]
            tdt = torch.from_numpy(np.empty(0, dtype)).dtype
            if t is None or tuple(t.shape) != tuple(shape) or t.dtype != tdt:
                t = self._pin[i] = torch.empty(shape, dtype=tdt, pin_memory=True)
            elif self._pin_free[i] is not None:
                self._pin_free[i].synchronize()                       # the copy that last read this buffer (three batches ago)
            return t.numpy()
        videos, labels = self._pop(out=staging)
        host = self._pin[i]
        with torch.cuda.stream(self._copy_stream):
            dev = host.to(device, non_blocking=True)
            if self._raw and not as_uint8:
                dev = ((dev.float() - 128.) / 128.).permute(0, 4, 1, 2, 3).contiguous()     # (N,T,H,W,C) u8 -> (N,C,T,H,W)
            lab_dev = None
            if with_event and labels and labels[0] is not None:
                # the labels travel the same way (pinned, copy stream): `torch.as_tensor(labels).to(device)` from pageable memory is a
                # BLOCKING copy ordered behind everything queued on the caller's stream -- the host then cannot run ahead of the GPU
                if self._pin_lab[i] is None or self._pin_lab[i].numel() < len(labels):
                    self._pin_lab[i] = torch.empty(max(len(labels), self.batch_size), dtype=torch.int32, pin_memory=True)
                self._pin_lab[i][:len(labels)] = torch.from_numpy(np.asarray(labels, dtype=np.int32))
                lab_dev = self._pin_lab[i][:len(labels)].to(device, non_blocking=True)
            ready = torch.cuda.Event()                                # the batch (and both staging buffers of ring position i) has been read
            ready.record(self._copy_stream)
            self._pin_free[i] = ready
        return dict(key=key, dev=dev, labels=labels, event=ready, lab_dev=lab_dev, state=self._state())

    def close(self):
        self._pool.shutdown(wait=bool(getattr(self, '_slots', None)), cancel_futures=True)      # (workers may still be writing into a slot)
        slots, self._slots = getattr(self, '_slots', []), []
        if getattr(self, '_slots_pinned', False):
            try:
                import torch
                torch.cuda.synchronize()
                from torch.cuda._pin_memory_utils import unpin_memory
                for _, arr in slots:
                    unpin_memory(arr.ctypes.data)
            except Exception:
                pass
            self._slots_pinned = False
        while slots:
            shm = slots.pop()[0]                                      # (the NumPy view of the slot goes first: a mapped buffer cannot be closed)
            for fn in (shm.unlink, shm.close):                        # unlink FIRST and on its own: close() raises BufferError while any
                try:                                                  # view of the slot is alive, and the name must leave /dev/shm anyway
                    fn()
                except Exception:
                    pass

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


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
        it = u.get_iterator('main')
        d = {'updater/iteration': np.asarray(u.iteration), 'updater/iterator:main/epoch': np.asarray(it.epoch),
             'updater/iterator:main/current_position': np.asarray(it.current_position),
             'updater/iterator:main/is_new_epoch': np.asarray(bool(it.is_new_epoch)),
             'updater/iterator:main/previous_epoch_detail': np.asarray(float(getattr(it, '_previous_epoch_detail', -1.0)))}
        if getattr(it, '_order', None) is not None:                 # Chainer's SerialIterator serializes its permutation too
            d['updater/iterator:main/order'] = np.asarray(it._order)
        if hasattr(it, '_batch_no'):                                # PrefetchIterator: seeds the workers' sub-sequence offsets per batch
            d['updater/iterator:main/batch_no'] = np.asarray(it.consumed_batches())
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
        if hasattr(u, '_step'):
            # the device step's own counter keys the Philox streams (add_noise, latent codes) and the frame index:
            # a resumed run must continue that sequence, not replay iterations 0..k
            u._step.iteration = u.iteration
        it = u.get_iterator('main')
        opt = lambda k: d['updater/iterator:main/' + k] if 'updater/iterator:main/' + k in d else None
        # Data parallel: only rank 0 writes snapshots, so the stored permutation is rank 0's.  Every rank keeps ITS OWN
        # permutation (seeded per rank in train.py) and restores position and epoch only -- restoring rank 0's order on
        # every rank would make all ranks read identical clips for the rest of the resumed epoch.
        import torch.distributed as dist
        world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        order = opt('order') if world == 1 else None
        args = (int(d['updater/iterator:main/epoch']), int(d['updater/iterator:main/current_position']), order,
                opt('is_new_epoch'), opt('previous_epoch_detail'))
        if hasattr(it, '_batch_no') and opt('batch_no') is not None:
            it._batch_no = int(opt('batch_no'))                     # before load_state: the look-ahead restarts from this number
        if hasattr(it, 'load_state'):
            it.load_state(*args)
        else:
            it.epoch, it.current_position = args[0], args[1]
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
