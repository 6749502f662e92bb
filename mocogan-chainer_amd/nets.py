"""The three MoCoGAN networks on the gfx950 kernel library.

Mirrors the architecture of reference model/net.py (ImageGenerator :17-117,
ImageDiscriminator :119-158, VideoDiscriminator :160-199) but owns its own execution plan:
parameters live in one flat fp32 buffer per network (so Adam and the gradient all-reduce are
single launches), activations are channels-last, and forward / backward are explicit kernel
sequences -- there is no autograd graph.  Every saved tensor a backward pass needs is kept in
the dict returned by forward, which is what lets the step reproduce Chainer's "backward through
already-updated weights" ordering (quirk Q5) exactly.
"""
import math
import os

import numpy as np
import torch

from . import hiplib as hl
from . import layout as lay

NOISE_SIGMA_Z = 0.33          # make_hidden: np.random.normal(0, 0.33), model/net.py:55-56
IMG = 64                      # output size hard-coded in the reference, model/net.py:115

# What runs in the convolutions' epilogues instead of in passes of its own (same arithmetic either way; the
# stand-alone passes remain the fallback for split-K tiles and synchronised BatchNorm).  MCG_FUSE="" switches all
# of it off, MCG_FUSE="stats,dc1" a subset (A/B timing, tests of both paths):
#   stats: BatchNorm statistics (sum, sum of squares per channel) from the producing fprop / deconvolution
#   dc1  : D's first layer: leaky_relu + add_noise (+ sign bits for backward) in dc1's epilogue, the leaky_relu mask and
#          dc1's bias gradient in the epilogue of dc2's input-gradient GEMM
#   bwd  : the per-channel sums of BatchNorm's backward pass from the GEMM that produces the incoming gradient, fp32-MFMA kernels
#          (off by default: measured on MI355X it costs the producing GEMMs more -- they read the saved BatchNorm input
#          in their per-element epilogue -- than the removed reduction pass took: +0.20 ms against -0.18 ms per iteration at batch 32)
#   bwd2 : the same sums from the ROW-WISE epilogue of the LDS-DMA kernels (bf16 networks, 'f32x3' split launches): 16-byte reads of
#          the saved BatchNorm input next to the 16-byte stores; a launch whose kernel cannot carry them falls back to the pass.
#          Off by default as well: measured at bf16 batch 256 the GEMMs grow by 0.58 ms and the removed pass took 0.6 (10883 against
#          10894 clips/s); 'f32x3' batch 32: 2693 against 2681 -- with one block per CU nothing overlaps an epilogue's reads
OUT16 = os.environ.get('MCG_OUT16', '1') == '1'        # bf16 networks: GEMM outputs in bf16 where the schedule allows (A/B switch)
Y16 = os.environ.get('MCG_Y16', '1') == '1'            # bf16 networks: the 64-channel neighbours of the clip (dc1's output gradient in D, the last
                                                       # layer's input and its gradient in G) stored in bf16, 'bf16y' launches read them (A/B switch)
FUSE = set(filter(None, os.environ.get('MCG_FUSE', 'stats,dc1').split(',')))
DC1_SPLIT_OUT = os.environ.get('MCG_DC1_SPLIT_OUT', '1') == '1'   # 'f32x3': D's first-layer epilogue writes the split form layer 2 reads (A/B switch)
WGRAD_AFTER = os.environ.get('MCG_WGRAD_AFTER', '0') == '1'    # side-stream weight gradients start AFTER the layer's input-gradient GEMM (A/B switch, off: see _Net._hold_wgrads)


class Config:
    """Stand-in for chainer.config: ``train`` selects batch-stat BN + add_noise (model/net.py:12)."""
    train = True


config = Config()


class FlatParams:
    """Named views into flat parameter / gradient / Adam-moment buffers (device layout)."""

    def __init__(self, specs, device):
        self.names, self.offsets, self.shapes = [], {}, {}
        off = 0
        for name, shape in specs:
            self.names.append(name)
            self.offsets[name] = off
            self.shapes[name] = tuple(shape)
            off += (int(np.prod(shape)) + 3) // 4 * 4          # keep every view 16-byte aligned
        self.size = off
        self.p = torch.zeros(off, dtype=torch.float32, device=device)
        self.g = torch.zeros_like(self.p)
        self.m = torch.zeros_like(self.p)
        self.v = torch.zeros_like(self.p)
        self.p16 = None                 # bf16 copy of p (same offsets): the weight operand of the bf16-storage GEMMs
        self.version = 0                # bumped by whatever writes p (touch): derived copies (the split filters of 'f32x3') follow it

    def touch(self):
        self.version += 1

    def _view(self, buf, name):
        o, s = self.offsets[name], self.shapes[name]
        return buf[o:o + int(np.prod(s))].view(s)

    def param(self, name):
        return self._view(self.p, name)

    def param16(self, name):
        return self._view(self.p16, name)

    def refresh16(self):
        """(re)build the bf16 copy from the fp32 master parameters; afterwards the Adam kernel keeps it current"""
        if self.p16 is None:
            self.p16 = torch.empty(self.size, dtype=torch.bfloat16, device=self.p.device)
        self.p16.copy_(self.p)

    def grad(self, name):
        return self._view(self.g, name)

    def view(self, buf, name):
        """buf in 'p' (parameters), 'g' (gradients), 'm' / 'v' (Adam moments)."""
        return self._view(getattr(self, buf), name)


def _np_to(t, device):
    return torch.as_tensor(np.asarray(t), dtype=torch.float32).to(device)


class _Net:
    """Shared parameter plumbing.  ``ref_shapes`` maps Chainer keys to Chainer shapes."""

    # The gradient of a conv/deconv bias that feeds train-mode BatchNorm is exactly zero: with
    # gx = gamma*inv_std*(g - (x_hat*ggamma + gbeta)/m),  sum_m gx = -gamma*inv_std*ggamma*sum_m(x_hat)/m and
    # sum_m x_hat = 0.  Chainer (and oracle/) compute rounding noise there (~1e-9 in fp32), which Adam's
    # sign-like update turns into an O(alpha) random walk of a parameter no output depends on.  The
    # backward passes below leave that gradient at its exact value 0 instead of reducing gx again.
    BIAS_NOTE = "pre-BatchNorm bias gradients are exactly zero"

    # MFMA operand type of this network's convolution GEMMs: 'f32' (default, the parity configuration) or 'bf16'
    # (BASELINE.json configs[2]): bf16 MFMA with fp32 accumulation; master parameters, BatchNorm statistics, Adam and
    # every GEMM OUTPUT stay fp32, while the tensors that are only ever read as GEMM operands -- the activations behind
    # BatchNorm / the activation function, the gradients behind their backward, and a copy of the weights -- are
    # STORED in bf16 (hl 'bf16s' launches: half the operand traffic, no conversion in the K loop).  Rounding a value
    # when it is stored or when it is loaded gives the same bf16 number, so this equals rounding in the kernel.
    # Layers whose channel counts are not multiples of 8 (the 4-channel clip side) keep fp32 tensors ('bf16' launches).
    precision = 'f32'

    # 'f32x3': fp32 networks whose wide convolutions may run on the bf16 matrix pipe with fp32 accuracy -- operands split into
    # three bf16 terms, six bf16 products per fp32 product (hl PREC_SPLIT, include/mocogan_hip.h).  Every tensor stays fp32; a
    # launch that pays for it (hl.split_pays: timed once per geometry) splits its two operands on the way in.
    def set_precision(self, precision):
        assert precision in ('f32', 'bf16', 'f32x3')
        self.precision = precision
        if precision == 'bf16':
            self.fp.refresh16()

    @property
    def gemm_precision(self):
        """what the geometries of this network's launches carry ('f32x3' launches are fp32 launches that MAY take the split form)"""
        return 'f32' if self.precision == 'f32x3' else self.precision

    def _wsplit(self, name, form):
        """the split form of a filter: 'f' as the forward GEMM of a convolution reads it (groups of 16 input channels), 'd' as the
        input-gradient GEMM does (planes of 16 filters).  Rebuilt when the parameters have changed since (FlatParams.version)."""
        cache = self.__dict__.setdefault('_wsplits', {})
        ent = cache.get((name, form))
        if ent is None or ent[0] != self.fp.version:
            w = self.fp.param(name)
            run = 16 if form == 'f' else 16 * (w.numel() // w.shape[0])
            out = hl.split_planes(w, run=run, out=ent[1] if ent else None)
            cache[(name, form)] = ent = (self.fp.version, out)
        self.__dict__.setdefault('_wsplit_used', set()).add((name, form))     # read since the last refresh: worth refreshing again
        return ent[1]

    def refresh_wsplits(self):
        """Right after this network's Adam update ('f32x3' networks): every (filter, form) pair its launches have used so far is
        re-split in ONE launch (mcg_split_planes_multi) instead of one launch per pair on first use -- 15 launches per iteration
        became 2-3.  Pairs not yet in the cache (the first iteration) are still split on first use by _wsplit."""
        cache = self.__dict__.get('_wsplits')
        if not cache or self.precision != 'f32x3' or os.environ.get('MCG_WSPLIT_MULTI', '1') != '1':
            return
        # only the pairs a launch has READ since the last refresh: hl.split_pays' timed trial of a geometry where the split form
        # then lost also fills the cache, and such a filter would be re-split every iteration and never read (round 5's advice);
        # a pair that was not refreshed keeps its old version and is rebuilt by _wsplit if a launch ever asks for it again
        used, self._wsplit_used = self.__dict__.get('_wsplit_used', set()), set()
        items, keys = [], []
        for (name, form), (ver, out) in cache.items():
            if ver != self.fp.version and (name, form) in used:
                w = self.fp.param(name)
                items.append((w, 16 if form == 'f' else 16 * (w.numel() // w.shape[0]), out))
                keys.append((name, form))
        if not items:
            return
        saved_tag = hl.get_tag()                                   # (called from inside adam_update: the caller's timing tag stays)
        hl.set_tag(getattr(self, 'tag', 'G'))
        try:
            for i in range(0, len(items), 32):
                hl.split_planes_multi(items[i:i + 32])
        finally:
            hl.set_tag(saved_tag)
        for key in keys:
            cache[key] = (self.fp.version, cache[key][1])

    def _cfprop(self, g, x, wname, w, b, y, ep=None, must_fuse=False, xs=None, force=False):
        """hl.conv_fprop of this network (w = the filter operand the caller would pass; xs: callable returning the split form of x).
        force: x exists in its split form ONLY (its producer relied on _split_only): the split launch is the only correct one."""
        if force:
            assert self.precision == 'f32x3' and hl.split_covers('fprop', g) and xs is not None, "an operand written in the split form only needs the split launch"
            return hl.conv_fprop(hl.with_precision(g, 'f32x3'), xs(), self._wsplit(wname, 'f'), b, y, ep=ep, must_fuse=must_fuse)
        if self.precision == 'f32x3' and hl.split_covers('fprop', g):
            gs = hl.with_precision(g, 'f32x3')

            def plain():
                return hl.conv_fprop(g, x, w, b, y, ep=ep, must_fuse=must_fuse)

            def split():
                return hl.conv_fprop(gs, xs() if xs else hl.split_planes(x), self._wsplit(wname, 'f'), b, y, ep=ep, must_fuse=must_fuse)
            if hl.split_pays('fprop', g, plain, split):
                return split()
        return hl.conv_fprop(g, x, w, b, y, ep=ep, must_fuse=must_fuse)

    def _fuse_bwd_sums(self, s16, kind, g):
        """should the GEMM (kind, g) that produces a gradient also produce the sums of the BatchNorm backward pass that reads it?"""
        if self.sync_bn is not None:
            return False
        if self.precision == 'f32x3':
            return 'bwd2' in FUSE and hl.split_decided(kind, g)
        return ('bwd2' in FUSE) if s16 else ('bwd' in FUSE)

    @staticmethod
    def _with_bwd_sums(launch, ep):
        """launch(ep) -> fused?  A kernel that cannot carry the sums (an fp32-MFMA-family tile around bf16 tensors, the patch kernel)
        refuses before anything is queued: the plain launch follows and the stand-alone pass does the sums."""
        try:
            return launch(ep)
        except hl.McgError:
            launch(None)
            return False

    def _split_only(self, *launches):
        """'f32x3': may the producer of a tensor write its split form ONLY?  Yes when every GEMM that reads it -- launches:
        (pass, geometry) pairs -- is known to take the split form (hl.split_decided); the fp32 tensor then stays allocated (shapes,
        slicing) but is never written nor read.  The caller RECORDS the decision (saved['only'] / a local flag) and passes it to
        _cfprop / _cdgrad / _cwgrad as force=True: a reader never consults the table again, so nothing that happens between the
        producer and the reader (another geometry key for a selected group, a cleared or replaced table, MCG_SPLIT changing) can make
        it read the unwritten fp32 tensor."""
        return (self.precision == 'f32x3' and self.sync_bn is None and os.environ.get('MCG_SPLIT_ONLY', '1') == '1'
                and all(hl.split_decided(kind, g) for kind, g in launches))

    @staticmethod
    def _sp(store, key, t):
        """the split form of tensor t, made once per (store, key): forward keeps what the weight gradient reads again"""
        if store is None:
            return hl.split_planes(t)
        s = store.get(key)
        if s is None:
            s = store[key] = hl.split_planes(t)
        return s

    def _cwgrad(self, g, x, y, dw, xs=None, ys=None, force=False):
        """self._wgrad (dw += ...); xs / ys: callables returning the split form of x / y; force: as _cfprop (x or y has no fp32 form)"""
        if force:
            assert self.precision == 'f32x3' and hl.split_covers('wgrad', g) and xs is not None and ys is not None, \
                "an operand written in the split form only needs the split launch"
            return self._wgrad(hl.with_precision(g, 'f32x3'), xs(), ys(), dw)
        if self.precision == 'f32x3' and hl.split_covers('wgrad', g):
            gs = hl.with_precision(g, 'f32x3')
            xs = xs or (lambda: hl.split_planes(x))
            ys = ys or (lambda: hl.split_planes(y))

            def plain():
                hl.conv_wgrad(g, x, y, self._wg_scratch(dw))

            def split():
                hl.conv_wgrad(gs, xs(), ys(), self._wg_scratch(dw))
            if hl.split_pays('wgrad', g, plain, split):
                return self._wgrad(gs, xs(), ys(), dw)
        return self._wgrad(g, x, y, dw)

    def _wg_scratch(self, dw):
        """a stand-in for dw while the two forms of a weight gradient are timed (the launch ADDS into its output)"""
        sc = self.__dict__.get('_wgs')
        if sc is None or sc.numel() < dw.numel():
            sc = self.__dict__['_wgs'] = torch.zeros(dw.numel(), device=dw.device)
        return sc[:dw.numel()].view(dw.shape)

    def _cdgrad(self, g, y, wname, w, b, x, ep=None, must_fuse=False, ys=None, force=False):
        """hl.conv_dgrad of this network (no activation, not accumulating: the launches that have a split form); force: as _cfprop"""
        if force:
            assert self.precision == 'f32x3' and hl.split_covers('dgrad', g) and ys is not None, "an operand written in the split form only needs the split launch"
            return hl.conv_dgrad(hl.with_precision(g, 'f32x3'), ys(), self._wsplit(wname, 'd'), b, x, ep=ep, must_fuse=must_fuse)
        if self.precision == 'f32x3' and hl.split_covers('dgrad', g):
            gs = hl.with_precision(g, 'f32x3')

            def plain():
                return hl.conv_dgrad(g, y, w, b, x, ep=ep, must_fuse=must_fuse)

            def split():
                return hl.conv_dgrad(gs, ys() if ys else hl.split_planes(y), self._wsplit(wname, 'd'), b, x, ep=ep, must_fuse=must_fuse)
            if hl.split_pays('dgrad', g, plain, split):
                return split()
        return hl.conv_dgrad(g, y, w, b, x, ep=ep, must_fuse=must_fuse)

    def _stored16(self, ci, co):
        """does a layer with these channel counts run on bf16-stored operands?"""
        return self.precision == 'bf16' and ci % 8 == 0 and co % 8 == 0 and ci > 4

    def _w(self, name, stored16):
        return self.fp.param16(name) if stored16 else self.fp.param(name)

    # Optional second HIP stream for the weight-gradient GEMMs of the backward pass.  wgrad(l) and dgrad(l)
    # both only read the layer's output gradient, so they may run side by side: the blocks of one fill the
    # CUs the other leaves idle in its last, partial round of tiles.  None = everything on the caller's stream.
    wgrad_stream = None
    sync_bn = None                 # step.GradExchange when BatchNorm is synchronised over the data-parallel ranks

    _held = None                   # weight-gradient launches held back until the layer's input-gradient GEMM is queued (_hold_wgrads)

    def _hold_wgrads(self):
        """Round 4: two big GEMMs side by side only share the machine (a trace of the batch-256 bf16 iteration: 11.8 of 19.7 ms with
        two or more GEMMs running, the SUM of their durations twice the one-stream sum), while the element-wise passes of the main
        chain -- BatchNorm's backward of the next layer, which depends on the input gradient -- ran alone for 2.5 ms.  So a layer's
        weight gradient can be queued behind its input-gradient GEMM (MCG_WGRAD_AFTER=1): that GEMM runs alone, the weight gradient
        starts together with the element-wise passes that follow it.  Measured (alternating, one box): bf16 batch 256 12.51-12.67 k
        against 12.70-12.71 k clips/s, fp32 batch 32 2099-2105 against 2131-2140, 'f32x3' 2747-2789 against 2752-2799 -- the one-block-
        per-CU weight gradient then holds the CUs the NEXT input-gradient GEMM of the main chain wants, and that costs what the hidden
        element-wise passes gain.  Off by default."""
        if WGRAD_AFTER and self.wgrad_stream is not None:
            self._held = []

    def _flush_wgrads(self):
        held, self._held = self._held, None
        for args in held or ():
            self._wgrad(*args)

    def _wgrad(self, geom, x, y, dw):
        """conv_wgrad on wgrad_stream (after everything queued so far on the current stream)."""
        if self._held is not None:
            self._held.append((geom, x, y, dw))
            return
        ws = self.wgrad_stream
        if ws is None:
            hl.conv_wgrad(geom, x, y, dw)
            return
        cur = torch.cuda.current_stream()
        ws.wait_stream(cur)
        with torch.cuda.stream(ws):
            hl.conv_wgrad(geom, x, y, dw)
        for t in (x, y):
            t.record_stream(ws)            # the caching allocator must not hand the block out before ws is done with it

    def _after_wgrads(self, fn):
        """Runs fn (the start of a gradient bucket's all-reduce) once everything queued so far -- on the current stream
        and on wgrad_stream -- is done, without making the current stream wait: fn is issued on wgrad_stream."""
        ws = self.wgrad_stream
        if ws is None:
            fn()
            return
        ws.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(ws):
            fn()

    def _wgrad_join(self):
        """the current stream waits for the weight gradients queued on wgrad_stream"""
        if self.wgrad_stream is not None:
            torch.cuda.current_stream().wait_stream(self.wgrad_stream)

    # ---- two chains of one network on two streams (round 4) ----------------------------------------------------------------
    # The real and the fake call of a discriminator are independent except for what they ACCUMULATE INTO: the running BatchNorm
    # statistics (two updates per iteration, real first: model/updater.py:97-98,107-108), the gradients of gamma / beta / the
    # logit layer / dc1's bias (read-modify-write kernels) and the weight gradients (float atomics: no order needed).  step.py may
    # therefore run the two calls as two chains on two HIP streams -- one chain's element-wise passes then run beside the other's
    # GEMMs -- if (a) each chain has its own scratch (BatchNorm workspace, partial sums, weight-gradient stream) and (b) the
    # read-modify-write kernels keep the reference's order.  `with net.chain(i):` swaps the scratch in while chain i's launches are
    # ENQUEUED (enqueueing is sequential on the host) and `_ordered(key, fn)` gives (b): chain 0 records an event behind fn, chain 1
    # (enqueued after chain 0) makes its stream wait for that event in front of its own fn.  Without a chain: plain calls.
    _chain = None

    class _Chain:
        def __init__(self, net, idx, events, wgrad_stream):
            self.net, self.idx, self.events, self.wgrad_stream = net, idx, events, wgrad_stream
            self.ws = torch.empty_like(net.ws)
            self.part = None

        def __enter__(self):
            n = self.net
            assert n._chain is None
            self._saved = (n.ws, n._part, n.wgrad_stream)
            n.ws, n._part, n.wgrad_stream, n._chain = self.ws, self.part, self.wgrad_stream, self
            return self

        def __exit__(self, *exc):
            n = self.net
            self.part = n._part                                  # (grown on demand while the chain ran)
            n.ws, n._part, n.wgrad_stream = self._saved
            n._chain = None
            return False

    def make_chains(self, wgrad_streams):
        """two chain contexts (see above) sharing one table of ordering events; wgrad_streams: a stream (or None) per chain"""
        events = {}
        return [self._Chain(self, i, events, wgrad_streams[i]) for i in range(2)]

    def _ordered(self, key, fn):
        ch = self._chain
        if ch is None:
            return fn()
        if ch.idx == 0:
            r = fn()
            ev = ch.events.get(key)
            if ev is None:
                ev = ch.events[key] = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            return r
        torch.cuda.current_stream().wait_event(ch.events[key])    # (recorded: chain 0 of this pass has been enqueued)
        return fn()

    def _alloc(self, specs, device):
        self.device = torch.device(device)
        self.fp = FlatParams(specs, self.device)
        self.t = 0                                             # Adam step counter of this net's optimizer
        self.ws = torch.empty(hl.bn_workspace_floats(1024), dtype=torch.float32, device=self.device)
        self._part = None                                      # per-tile partial sums of fused conv epilogues (grown on demand)

    def _part_buf(self, geom, kind, groups):
        n = hl.epilogue_part_floats(geom, kind, groups)
        if self._part is None or self._part.numel() < n:
            self._part = torch.empty(n, dtype=torch.float32, device=self.device)
        return self._part

    # ---- reference-layout import / export (also the .npz checkpoint format, train.py:139-144) ----
    def load_reference_params(self, params):
        for key in self.ref_shapes:
            if key.endswith('/N'):
                if key in params:
                    self.bn_count[key[:-2]] = int(params[key])
                continue
            v = _np_to(params[key], self.device)
            assert tuple(v.shape) == tuple(self.ref_shapes[key]), (key, v.shape, self.ref_shapes[key])
            self._set_from_ref(key, v)
        self.fp.touch()
        if self.precision == 'bf16':
            self.fp.refresh16()

    def export_reference_params(self):
        out = {}
        for key in self.ref_shapes:
            if key.endswith('/N'):
                out[key] = np.asarray(self.bn_count.get(key[:-2], 0), dtype=np.int64)
            else:
                out[key] = self._get_as_ref(key).detach().cpu().numpy()
        return out

    def export_reference_grads(self):
        return {k: self._get_as_ref(k, 'g').detach().cpu().numpy() for k in self.trainable_keys()}

    def trainable_keys(self):
        return [k for k in self.ref_shapes if k.endswith(('/W', '/b', '/gamma', '/beta'))]

    # ---- optimizer state in the reference layout (part of a Chainer trainer snapshot, train.py:135-138) ----
    def export_adam_state(self):
        return {'t': self.t,
                'm': {k: self._get_as_ref(k, 'm').detach().cpu().numpy() for k in self.trainable_keys()},
                'v': {k: self._get_as_ref(k, 'v').detach().cpu().numpy() for k in self.trainable_keys()}}

    def load_adam_state(self, state):
        self.t = int(state['t'])
        for k in self.trainable_keys():
            self._set_from_ref(k, _np_to(state['m'][k], self.device), 'm')
            self._set_from_ref(k, _np_to(state['v'][k], self.device), 'v')

    def zero_grad(self):
        self.fp.g.zero_()

    def to(self, device):
        """Move every buffer (chainer's to_gpu / to_cpu, train.py:87-91,185-188).  Only a 'cuda' device can
        run the kernels; host placement exists for construction and checkpoint IO."""
        device = torch.device(device)
        for b in ('p', 'g', 'm', 'v'):
            setattr(self.fp, b, getattr(self.fp, b).to(device))
        if self.fp.p16 is not None:
            self.fp.p16 = self.fp.p16.to(device)
        self.fp.touch()
        self.__dict__.pop('_wsplits', None)
        self.running = {k: v.to(device) for k, v in self.running.items()}
        self.ws = self.ws.to(device)
        self._part = None
        self.device = device
        return self

    def _bn_keys(self, name, c):
        return {name + '/gamma': (c,), name + '/beta': (c,), name + '/avg_mean': (c,), name + '/avg_var': (c,),
                name + '/N': ()}

    def _init_bn(self, name):
        self.fp.param(name + '/gamma').fill_(1.0)
        self.running[name + '/avg_var'].fill_(1.0)


# ==========================================================================================
# Discriminators
# ==========================================================================================
class DisNet(_Net):
    """ndim=2: ImageDiscriminator (model/net.py:119-158); ndim=3: VideoDiscriminator (:160-199)."""

    def __init__(self, ndim, in_channels=3, out_channels=1, n_filters=64, use_noise=False, noise_sigma=0.2,
                 video_len=16, device='cuda', seed=None):
        self.ndim, self.in_channels, self.out_channels = ndim, in_channels, out_channels
        self.n_filters, self.use_noise, self.noise_sigma = n_filters, use_noise, noise_sigma
        self.kt = 4 if ndim == 3 else 1
        self.T0 = video_len if ndim == 3 else 1
        self.tag = 'D_V' if ndim == 3 else 'D_I'
        nf = n_filters
        self.chans = [in_channels, nf, nf * 2, nf * 4, nf * 8, out_channels]
        self.cp0 = lay.pad4(in_channels)
        kshape = (4,) * ndim
        self.ref_shapes = {}
        specs = []
        for l in range(1, 6):
            co, ci = self.chans[l], self.chans[l - 1]
            self.ref_shapes['dc%d/W' % l] = (co, ci) + kshape
            self.ref_shapes['dc%d/b' % l] = (co,)
            specs.append(('dc%d/W' % l, (co, self.kt, 4, 4, lay.pad4(ci))))
            specs.append(('dc%d/b' % l, (co,)))
        for l in (2, 3, 4):
            self.ref_shapes.update(self._bn_keys('bn%d' % l, self.chans[l]))
            specs.append(('bn%d/gamma' % l, (self.chans[l],)))
            specs.append(('bn%d/beta' % l, (self.chans[l],)))
        self._alloc(specs, device)
        self.running = {}
        self.bn_count = {}
        for l in (2, 3, 4):
            c = self.chans[l]
            self.running['bn%d/avg_mean' % l] = torch.zeros(c, device=self.device)
            self.running['bn%d/avg_var' % l] = torch.ones(c, device=self.device)
            self.bn_count['bn%d' % l] = 0
            self._init_bn('bn%d' % l)
        if seed is not None:
            self.init_weights(np.random.RandomState(seed))

    def init_weights(self, rng):
        """GlorotNormal conv weights, zero biases (model/net.py:131,172)."""
        for l in range(1, 6):
            shape = self.ref_shapes['dc%d/W' % l]
            rf = int(np.prod(shape[2:]))
            std = math.sqrt(2.0 / (shape[1] * rf + shape[0] * rf))
            self._set_from_ref('dc%d/W' % l, _np_to(rng.normal(0, std, size=shape), self.device))

    def _set_from_ref(self, key, v, buf='p'):
        self.fp.touch()
        if key.endswith('/W'):
            self.fp.view(buf, key).copy_(lay.conv_w_to_dev(v))
        elif key in self.running:
            self.running[key].copy_(v)
        else:
            self.fp.view(buf, key).copy_(v)

    def _get_as_ref(self, key, buf='p'):
        if key in self.running:
            return self.running[key]
        t = self.fp.view(buf, key)
        if key.endswith('/W'):
            return lay.conv_w_from_dev(t, self.ref_shapes[key][1], self.ndim)
        return t

    # ---- geometry -------------------------------------------------------------------------
    def _extents(self, l):
        """(T, H) of the INPUT of layer l (1..5)."""
        return self.T0 - (self.kt - 1) * (l - 1), IMG >> (l - 1)

    def _s16(self, l):
        """layer l (1..4) runs on bf16-stored operands"""
        return 1 <= l <= 4 and self._stored16(lay.pad4(self.chans[l - 1]), self.chans[l])

    def _geom(self, l, n, x_stride0=None):
        t, h = self._extents(l)
        return hl.make_geom(n, t, h, h, lay.pad4(self.chans[l - 1]), self.chans[l], self.kt, x_stride0=x_stride0,
                            precision='bf16s' if self._s16(l) else self.gemm_precision, ci_valid=self.chans[l - 1])

    # ---- forward ---------------------------------------------------------------------------
    def forward(self, n, first_input, noise=None, rng=None, update_stats=True):
        """One call of the discriminator on n samples (reference __call__).
        first_input: callable(out, noise_kwargs) that writes a1 = x + noise into `out`
        ([n][T][64][64][cp0]).  noise: list of 4 device tensors in device layout (parity mode) or
        None; rng: (seed, base_stream_id) for in-kernel Philox noise (perf mode) or None.
        Returns (logits [n][out], saved)."""
        return self.forward_groups(n, [dict(first_input=first_input, noise=noise, rng=rng)], update_stats)

    def forward_groups(self, n, groups, update_stats=True):
        """Several independent calls (e.g. the real and the fake batch of one iteration) run as ONE
        batch of len(groups)*n samples through the convolutions -- twice the rows per launch fills the GPU
        better on the small late layers -- while everything that is per call in the reference stays per
        group: BatchNorm statistics (and the order of their running-average updates), noise, activations.
        groups: list of dict(first_input=..., noise=..., rng=...) as in forward().
        Returns (logits [G*n][out], saved); group i owns rows [i*n, (i+1)*n)."""
        train = config.train
        noisy = train and self.use_noise
        dev = self.device
        hl.set_tag(self.tag)
        G = len(groups)
        N = G * n

        def noise_args(grp, l):
            if not noisy:
                return dict()
            if grp.get('noise') is not None:
                return dict(addend=grp['noise'][l - 1])
            if grp.get('rng') is not None:
                return dict(sigma=self.noise_sigma, seed=grp['rng'][0], stream_id=grp['rng'][1] + l - 1)
            return dict()

        fuse_stats = train and 'stats' in FUSE and self.sync_bn is None
        fuse_dc1 = 'dc1' in FUSE
        saved = {'n': n, 'G': G, 'a': {}, 'y': {}, 'stats': {}, 'mask1': None, 'split': {}, 'only': set()}
        t, h = self._extents(1)
        a = torch.empty((N, t, h, h, self.cp0), device=dev)
        for gi, grp in enumerate(groups):
            grp['first_input'](a[gi * n:(gi + 1) * n], noise_args(grp, 1))
        saved['a'][1] = a
        for l in (1, 2, 3, 4):
            g = self._geom(l, N)
            co = self.chans[l]
            w, b = self._w('dc%d/W' % l, self._s16(l)), self.fp.param('dc%d/b' % l)
            m = n * g.To * g.Ho * g.Wo                       # rows of ONE group
            adt = torch.bfloat16 if self._s16(l + 1) else torch.float32      # what the NEXT layer's GEMMs read
            if l == 1 and fuse_dc1:
                # dc1 has no BatchNorm: leaky_relu + add_noise run in its epilogue (model/net.py:148-149,189-190); what
                # backward needs of the pre-activation -- its sign -- is kept as one bit per element
                a = torch.empty((N, g.To, g.Ho, g.Wo, co), device=dev, dtype=adt)
                mask = torch.empty((G * m, (co + 31) // 32), dtype=torch.int32, device=dev)
                # 'f32x3' (round 6): when both GEMMs of layer 2 that read this tensor take the split form, the epilogue writes the
                # three bf16 terms directly (MCG_IO_OUT_SPLIT) -- the fp32 tensor is never written, and the pass that read it back
                # and split it (0.65 GB of traffic at 64 clips, the largest operand split of the iteration) is gone
                a_split = None
                g2 = self._geom(2, N)
                if DC1_SPLIT_OUT and self._split_only(('fprop', g2), ('wgrad', g2)) and co % 16 == 0:
                    a_split = saved['split'][2] = torch.empty(a.shape[:-1] + (4 * co,), device=dev, dtype=torch.bfloat16)
                    saved['only'].add(2)                         # saved['a'][2] stays unwritten
                na = [noise_args(grp, 2) for grp in groups]
                kw = {}
                if na[0]:
                    assert all(('addend' in x) == ('addend' in na[0]) for x in na)
                    if 'addend' in na[0]:
                        kw = dict(addend=[x['addend'] for x in na])
                    else:
                        assert all(x['seed'] == na[0]['seed'] for x in na)
                        kw = dict(sigma=na[0]['sigma'], seed=na[0]['seed'], stream_id=[x['stream_id'] for x in na])
                hl.conv_fprop(g, saved['a'][1], w, b, a_split if a_split is not None else a, must_fuse=True,
                              ep=hl.epilogue(act=hl.ACT_LRELU, groups=G, mask_out=mask, out_bf16=adt == torch.bfloat16,
                                             out_split=a_split is not None, **kw))
                saved['y'][1], saved['mask1'], saved['a'][2] = None, mask, a
                continue
            # bf16 networks: the pre-BatchNorm values are bf16 too (the element-wise passes that read them are HBM-bound)
            # -- unless the tuned tile splits K (partial tiles are added in fp32) or the statistics need the stand-alone pass
            y16 = (OUT16 and l >= 2 and self._s16(l) and self.sync_bn is None and (fuse_stats or not train)
                   and hl.fprop_tile(g, a, w, b) < 1000)
            y = torch.empty((N, g.To, g.Ho, g.Wo, co), device=dev, dtype=torch.bfloat16 if y16 else torch.float32)
            ep = None
            if l >= 2 and fuse_stats:
                part = self._part_buf(g, 'fprop', G)
                ep = hl.epilogue(sums=hl.SUMS_STATS, groups=G, part=part, out_bf16=y16)
                if not self._cfprop(g, a, 'dc%d/W' % l, w, b, y, ep=ep, xs=lambda: self._sp(saved['split'], l, saved['a'][l]), force=l in saved['only']):
                    ep = None                                    # split-K tile: the stand-alone statistics pass below
            else:
                self._cfprop(g, a, 'dc%d/W' % l, w, b, y, xs=lambda: self._sp(saved['split'], l, saved['a'][l]), force=l in saved['only'])
            saved['y'][l] = y
            a = torch.empty_like(y, dtype=adt)
            a_split = None                                       # 'f32x3': layer l + 1's GEMMs both read the split form -> written directly
            if l < 4:
                gn = self._geom(l + 1, N)
                if self._split_only(('fprop', gn), ('wgrad', gn)):
                    a_split = saved['split'][l + 1] = torch.empty(y.shape[:-1] + (4 * co,), device=dev, dtype=torch.bfloat16)
                    saved['only'].add(l + 1)                     # saved['a'][l + 1] stays unwritten
            if l >= 2:
                saved['stats'][l] = []
            for gi, grp in enumerate(groups):
                yg, ag = y[gi * n:(gi + 1) * n], a[gi * n:(gi + 1) * n]
                ss = None
                if l >= 2:
                    name = 'bn%d' % l
                    if train:
                        stats = torch.empty(4 * co, device=dev)
                        rm = self.running[name + '/avg_mean'] if update_stats else None
                        rv = self.running[name + '/avg_var'] if update_stats else None
                        # (the running averages are read-modify-write state shared by the two calls of an iteration: ordered)
                        if ep is not None:
                            self._ordered(('stats', l), lambda: hl.bn_stats_from_partials(
                                m, co, part[gi * 2 * co:], ep.n_slots, ep.slot_stride, self.fp.param(name + '/gamma'),
                                self.fp.param(name + '/beta'), stats, rm, rv, self.ws))
                        else:
                            self._ordered(('stats', l), lambda: hl.bn_stats(
                                m, co, yg, self.fp.param(name + '/gamma'), self.fp.param(name + '/beta'), stats, rm, rv,
                                self.ws, sync=self.sync_bn))
                        if update_stats:
                            self.bn_count[name] += 1
                        saved['stats'][l].append(stats)
                        ss = stats[2 * co:]
                    else:
                        ss = self._test_scale_shift(name)
                if a_split is not None:
                    hl.bn_act_fwd(m, co, yg, ss, hl.ACT_LRELU, a_split[gi * n:(gi + 1) * n], split_out=True, **noise_args(grp, l + 1))
                else:
                    hl.bn_act_fwd(m, co, yg, ss, hl.ACT_LRELU, ag, **(noise_args(grp, l + 1) if l < 4 else {}))
            saved['a'][l + 1] = a
        k = a[0].numel()
        logits = torch.empty((N, self.out_channels), device=dev)
        hl.fc_fprop(N, k, self.out_channels, a.view(N, k), self.fp.param('dc5/W').view(self.out_channels, k),
                    self.fp.param('dc5/b'), logits)
        return logits, saved

    @staticmethod
    def select_group(saved, gi):
        """The saved state of group gi alone (views, no copies)."""
        if 'chains' in saved:                                    # (two-chain schedule: each group was a forward of its own)
            return saved['chains'][gi]
        n = saved['n']
        sl = slice(gi * n, (gi + 1) * n)
        mask = saved.get('mask1')
        if mask is not None:
            rows = mask.shape[0] // saved['G']
            mask = mask[gi * rows:(gi + 1) * rows]
        return {'n': n, 'G': 1, 'a': {l: v[sl] for l, v in saved['a'].items()}, 'split': {l: v[sl] for l, v in saved.get('split', {}).items()},
                'y': {l: (None if v is None else v[sl]) for l, v in saved['y'].items()},
                'stats': {l: [v[gi]] for l, v in saved['stats'].items()}, 'mask1': mask, 'only': set(saved.get('only', ()))}

    def _test_scale_shift(self, name):
        """fixed_batch_normalization (test mode, reference util.py:92): scale/shift from running stats."""
        inv = torch.rsqrt(self.running[name + '/avg_var'] + 2e-5)
        scale = self.fp.param(name + '/gamma') * inv
        return torch.cat((scale, self.fp.param(name + '/beta') - self.running[name + '/avg_mean'] * scale))

    # ---- backward --------------------------------------------------------------------------
    def grad_bucket_late(self):
        """(start, end) of the flat-gradient range holding dc4/W .. dc5/b: 76 % of the bytes, and complete
        right after layer 4 of the backward pass -- the data-parallel exchange starts it early."""
        o = self.fp.offsets
        return o['dc4/W'], o['bn2/gamma']

    def backward(self, saved, g_logits, param_grads, gx=None, gx_geom=None, gx_accumulate=False, on_late_bucket=None, defer_gx=False):
        """g_logits [G*n][out] for the G groups of `saved`.  param_grads: accumulate dW/db/dgamma/dbeta into
        the flat gradient (D's own loss: all groups in one set of launches).  gx (single group only): the
        gradient w.r.t. the first conv's input is written (or accumulated) there through gx_geom (G's loss
        through this D, current weights: Q5).  on_late_bucket: called once the gradients of dc5 and dc4
        are final (see grad_bucket_late).  defer_gx: the launch that writes gx is not queued but returned as a
        callable -- the caller runs it on the stream (and after the work) gx belongs to."""
        n, G = saved['n'], saved['G']
        N = n * G
        fp = self.fp
        hl.set_tag(self.tag)
        self._held = None
        assert gx is None or G == 1
        a5 = saved['a'][5]
        k = a5[0].numel()
        co5 = self.out_channels
        if param_grads:
            self._ordered(('g', 5), lambda: hl.fc_wgrad(N, k, co5, a5.view(N, k), g_logits, fp.grad('dc5/W').view(co5, k), fp.grad('dc5/b')))
        g = torch.empty_like(a5)
        hl.fc_dgrad(N, k, co5, g_logits, fp.param('dc5/W').view(co5, k), None, 0, g.view(N, k))
        mask1 = saved.get('mask1')
        pending = None            # (epilogue, partial sums) the GEMM that produced g left for the layer processed next
        deferred = None
        for l in (4, 3, 2, 1):
            geom = self._geom(l, N)
            co = self.chans[l]
            y = saved['y'][l]
            m = n * geom.To * geom.Ho * geom.Wo                  # rows of ONE group
            s16 = self._s16(l)
            if l >= 2:
                name = 'bn%d' % l
                gy = torch.empty_like(g, dtype=torch.bfloat16) if s16 else g       # bf16-stored operand of wgrad / dgrad, else in place
                # 'f32x3': both GEMMs that read gy take the split form -> written in that form only (g keeps its shape, not its meaning)
                gy_split = None
                if self._split_only(('dgrad', geom), *((('wgrad', geom),) if param_grads else ())):
                    gy_split = torch.empty(g.shape[:-1] + (4 * co,), device=g.device, dtype=torch.bfloat16)
                for gi in range(G):
                    gg, yg, go = g[gi * n:(gi + 1) * n], y[gi * n:(gi + 1) * n], gy[gi * n:(gi + 1) * n]
                    if gy_split is not None:
                        go = gy_split[gi * n:(gi + 1) * n]
                    dg = fp.grad(name + '/gamma') if param_grads else None
                    db = fp.grad(name + '/beta') if param_grads else None
                    # (gamma's / beta's gradients are added by a read-modify-write kernel: ordered between the chains when they are asked for)
                    if pending is not None:
                        ep, part = pending
                        bwd = lambda: hl.bn_act_bwd_from_partials(m, co, gg, yg, saved['stats'][l][gi], fp.param(name + '/gamma'), hl.ACT_LRELU,
                                                                  part[gi * 2 * co:], ep.n_slots, ep.slot_stride, go, dg, db, self.ws,
                                                                  split_out=gy_split is not None)
                    else:
                        bwd = lambda: hl.bn_act_bwd(m, co, gg, yg, saved['stats'][l][gi], fp.param(name + '/gamma'), hl.ACT_LRELU, go, dg, db,
                                                    self.ws, sync=self.sync_bn, split_out=gy_split is not None)
                    if param_grads:
                        self._ordered(('g', l), bwd)
                    else:
                        bwd()
                g = gy
            elif mask1 is None:
                for gi in range(G):
                    gg, yg = g[gi * n:(gi + 1) * n], y[gi * n:(gi + 1) * n]
                    hl.bn_act_bwd(m, co, gg, yg, None, None, hl.ACT_LRELU, gg, None, None, self.ws)
            # (l == 1 with a stored mask: dc2's input-gradient GEMM applied leaky_relu's backward in its epilogue)
            gsp = {}                                             # 'f32x3': the split form of g, shared by the two GEMMs that read it
            g_only = l >= 2 and gy_split is not None             # ... which is the ONLY form of g that was written
            if g_only:
                gsp[0] = gy_split

            def gys():
                return self._sp(gsp, 0, g)
            if param_grads:
                if l == 1:
                    if pending is not None:
                        self._ordered(('g', 1), lambda: hl.colsum_from_partials(co, pending[1], pending[0].n_slots, pending[0].slot_stride,
                                                                                fp.grad('dc1/b'), self.ws))
                    else:
                        self._ordered(('g', 1), lambda: hl.colsum_acc(m * G, co, g, fp.grad('dc1/b'), self.ws))
                # dc2..dc4 feed BatchNorm: sum_m gx == 0 exactly (see _Net.BIAS_NOTE), nothing to add
                wgeom = hl.with_precision(geom, 'bf16y') if (l == 1 and g.dtype == torch.bfloat16) else geom      # (a bf16 y beside the fp32 clip)
                if l > 1:
                    self._hold_wgrads()                          # ... queued once the input-gradient GEMM below is (see _hold_wgrads)
                self._cwgrad(wgeom, saved['a'][l], g, fp.grad('dc%d/W' % l), xs=lambda: self._sp(saved.get('split'), l, saved['a'][l]), ys=gys,
                             force=g_only or l in saved.get('only', ()))
                if l == 4 and on_late_bucket is not None and self._held is None:
                    self._after_wgrads(on_late_bucket)
            pending = None
            if l > 1:
                w = self._w('dc%d/W' % l, s16)
                # bf16 networks: the gradient BatchNorm's backward of layer l - 1 reads is bf16 as well (see forward_groups)
                g16 = OUT16 and l > 2 and s16 and self.sync_bn is None and hl.dgrad_tile(geom, g, w, None) < 1000
                if l == 2 and mask1 is not None and OUT16 and Y16 and s16 and self.precision == 'bf16':
                    # ... and so is dc1's output gradient (16x the clip's size): its readers -- dc1's weight gradient and the MFMA
                    # input-gradient kernel of the first layer -- take a bf16 y beside the fp32 clip ('bf16y' launches).  Not when
                    # this pass ACCUMULATES onto a frame of the clip gradient (D_I under G's loss: the VALU kernel reads fp32).
                    g1 = self._geom(1, N)
                    ok_gx = gx is None or (not gx_accumulate and hl.dgrad_c4_mfma_covers(hl.with_precision(gx_geom if gx_geom is not None else g1, 'bf16y')))
                    g16 = ok_gx and g1.Co % 8 == 0 and hl.dgrad_tile(geom, g, w, None) < 1000
                ga = torch.empty_like(saved['a'][l], dtype=torch.bfloat16 if g16 else torch.float32)
                if l == 2 and mask1 is not None:
                    part = self._part_buf(geom, 'dgrad', 1) if param_grads else None
                    ep = hl.epilogue(mask_in=mask1, sums=hl.SUMS_COL if param_grads else hl.SUMS_NONE, groups=1, part=part, out_bf16=g16)
                    self._cdgrad(geom, g, 'dc%d/W' % l, w, None, ga, ep=ep, must_fuse=True, ys=gys, force=g_only)
                    pending = (ep, part) if param_grads else None
                elif l > 2 and self._fuse_bwd_sums(s16, 'dgrad', geom):
                    part = self._part_buf(geom, 'dgrad', G)
                    ep = hl.epilogue(sums=hl.SUMS_BN_BWD, groups=G, part=part, bn_y=saved['y'][l - 1], bn_stats=saved['stats'][l - 1],
                                     bn_act=hl.ACT_LRELU, out_bf16=g16)
                    if self._with_bwd_sums(lambda e: self._cdgrad(geom, g, 'dc%d/W' % l, w, None, ga, ep=e, ys=gys, force=g_only), ep):
                        pending = (ep, part)
                else:
                    self._cdgrad(geom, g, 'dc%d/W' % l, w, None, ga, ys=gys, force=g_only)
                if self._held is not None:
                    self._flush_wgrads()
                    if l == 4 and on_late_bucket is not None:
                        self._after_wgrads(on_late_bucket)
                g = ga
            elif gx is not None:
                def write_gx(g=g, geom=geom):
                    hl.set_tag(self.tag)
                    gg = gx_geom if gx_geom is not None else geom
                    if g.dtype == torch.bfloat16:
                        gg = hl.with_precision(gg, 'bf16y')
                    hl.conv_dgrad(gg, g, fp.param('dc1/W'), None, gx, accumulate=gx_accumulate)
                    if defer_gx:                                 # g was allocated under the stream this pass ran on: the allocator must
                        g.record_stream(torch.cuda.current_stream())     # not hand its block out before the caller's stream is done
                if defer_gx:
                    deferred = write_gx
                else:
                    write_gx()
        if param_grads:
            self._wgrad_join()
        return deferred


# ==========================================================================================
# Generator
# ==========================================================================================
class GenNet(_Net):
    """ImageGenerator (model/net.py:17-117): GRU motion codes + 5 transposed convolutions."""

    def __init__(self, dim_zc=50, dim_zm=10, dim_zl=0, out_channels=3, n_filters=64, video_len=16,
                 device='cuda', seed=None):
        self.dim_zc, self.dim_zm, self.dim_zl = dim_zc, dim_zm, dim_zl
        self.out_channels, self.n_filters, self.video_len = out_channels, n_filters, video_len
        self.n_hidden = dim_zc + dim_zm
        nf = n_filters
        self.chans = [self.n_hidden, nf * 8, nf * 4, nf * 2, nf, out_channels]
        self.cp_out = lay.pad4(out_channels)
        self.ref_shapes = {}
        nin = dim_zm + dim_zl
        self.gru_size = 0
        for k in lay.GRU_LINKS:
            cols = dim_zm if k.startswith('U') else nin
            self.ref_shapes['g0/%s/W' % k] = (dim_zm, cols)
            self.ref_shapes['g0/%s/b' % k] = (dim_zm,)
            self.gru_size += dim_zm * cols + dim_zm
        specs = [('g0', (self.gru_size,))]
        for l in range(1, 6):
            ci, co = self.chans[l - 1], self.chans[l]
            self.ref_shapes['dc%d/W' % l] = (ci, co, 4, 4)
            self.ref_shapes['dc%d/b' % l] = (co,)
            specs.append(('dc%d/W' % l, (ci, 1, 4, 4, lay.pad4(co))))
            specs.append(('dc%d/b' % l, (lay.pad4(co),)))
        for l in (1, 2, 3, 4):
            self.ref_shapes.update(self._bn_keys('bn%d' % l, self.chans[l]))
            specs.append(('bn%d/gamma' % l, (self.chans[l],)))
            specs.append(('bn%d/beta' % l, (self.chans[l],)))
        self._alloc(specs, device)
        self.running, self.bn_count = {}, {}
        for l in (1, 2, 3, 4):
            c = self.chans[l]
            self.running['bn%d/avg_mean' % l] = torch.zeros(c, device=self.device)
            self.running['bn%d/avg_var' % l] = torch.ones(c, device=self.device)
            self.bn_count['bn%d' % l] = 0
            self._init_bn('bn%d' % l)
        if seed is not None:
            self.init_weights(np.random.RandomState(seed))

    def init_weights(self, rng):
        """GlorotNormal deconv weights (model/net.py:35), LeCunNormal GRU Linear weights (Chainer
        Linear default), zero biases."""
        for k in lay.GRU_LINKS:
            shape = self.ref_shapes['g0/%s/W' % k]
            self._set_from_ref('g0/%s/W' % k, _np_to(rng.normal(0, math.sqrt(1.0 / shape[1]), size=shape), self.device))
        for l in range(1, 6):
            shape = self.ref_shapes['dc%d/W' % l]
            std = math.sqrt(2.0 / (shape[1] * 16 + shape[0] * 16))
            self._set_from_ref('dc%d/W' % l, _np_to(rng.normal(0, std, size=shape), self.device))

    def _gru_views(self, buf='p'):
        return lay.gru_from_dev(self.fp.view(buf, 'g0'), self.dim_zm, self.dim_zl)

    def _set_from_ref(self, key, v, buf='p'):
        self.fp.touch()
        if key.startswith('g0/'):
            self._gru_views(buf)[key].copy_(v)
        elif key.endswith('/W'):
            self.fp.view(buf, key).copy_(lay.deconv_w_to_dev(v))
        elif key in self.running:
            self.running[key].copy_(v)
        elif key.startswith('dc') and key.endswith('/b'):
            self.fp.view(buf, key).copy_(lay.vec_to_dev(v))
        else:
            self.fp.view(buf, key).copy_(v)

    def _get_as_ref(self, key, buf='p'):
        if key in self.running:
            return self.running[key]
        if key.startswith('g0/'):
            return self._gru_views(buf)[key]
        t = self.fp.view(buf, key)
        if key.endswith('/W'):
            return lay.deconv_w_from_dev(t, self.ref_shapes[key][1])
        if key.startswith('dc') and key.endswith('/b'):
            return t[:self.ref_shapes[key][0]]
        return t

    def _geom(self, l, frames, clip_order_n=0):
        """Conv-form geometry of deconv layer l (2..5): x side = its OUTPUT, y side = its input.
        clip_order_n = N makes the x side the clip tensor [N][T][H][W][C] (frame f = t*N + n lands at
        clip n, time t): the (T,N)->(N,T) transpose of model/updater.py:102 costs nothing."""
        h = 4 << (l - 1)
        ci = lay.pad4(self.chans[l])
        prec = 'bf16s' if self._s16(l) else self.gemm_precision
        if clip_order_n:
            T = frames // clip_order_n
            return hl.make_geom(frames, 1, h, h, ci, self.chans[l - 1], 1, x_stride0=T * h * h * ci,
                                x_perm_n=clip_order_n, x_stride1=h * h * ci, precision=prec, ci_valid=self.chans[l])
        return hl.make_geom(frames, 1, h, h, ci, self.chans[l - 1], 1, precision=prec, ci_valid=self.chans[l])

    def _s16(self, l):
        """deconvolution layer l (2..5) runs on bf16-stored operands (conv form: Ci = its output channels, Co = its input's)"""
        return 2 <= l <= 5 and self._stored16(lay.pad4(self.chans[l]), self.chans[l - 1])

    # ---- latent draws (model/net.py:55-56,66,71,92,102) ------------------------------------------
    def draw(self, n, rng):
        """Perf-mode latent draw on the device: rng = (seed, base_stream_id).  Philox streams base .. base + 3 hold h0, e, zc
        and the labels (oracle.philox states the same draws)."""
        dev = self.device
        T, dz = self.video_len, self.dim_zm
        d = {'labels': None}
        if self.dim_zl:
            d['labels'] = torch.empty(n, device=dev, dtype=torch.int32)
            hl.randint(d['labels'], self.dim_zl, rng[0], rng[1] + 3)
        d['h0'] = torch.empty((n, dz), device=dev)
        d['e'] = torch.empty((T, n, dz), device=dev)
        d['zc'] = torch.empty((n, self.dim_zc), device=dev)
        hl.randn(d['h0'], NOISE_SIGMA_Z, rng[0], rng[1])
        hl.randn(d['e'], NOISE_SIGMA_Z, rng[0], rng[1] + 1)
        hl.randn(d['zc'], NOISE_SIGMA_Z, rng[0], rng[1] + 2)
        return d

    # ---- forward ---------------------------------------------------------------------------
    def forward(self, n, draw, update_stats=True):
        """draw: dict(h0 [n][dz], e [T][n][dz], zc [n][dc], labels int32 [n] | None) on the device.
        Returns (x_fake clip tensor [n][T][64][64][cp_out], saved)."""
        dev = self.device
        T, dz, dl, dc = self.video_len, self.dim_zm, self.dim_zl, self.dim_zc
        frames = T * n
        fp = self.fp
        train = config.train
        hl.set_tag('G')
        saved = {'n': n, 'draw': draw, 'y': {}, 'a': {}, 'stats': {}, 'split': {}, 'only': set()}
        z = torch.empty((frames, dc + dz), device=dev)
        gsaved = torch.empty((T, n, 4 * dz), device=dev)
        hl.gru_seq_fwd(n, T, dz, dl, dc, fp.param('g0'), draw['h0'], draw['e'], draw['labels'], draw['zc'], z, gsaved)
        saved['z'], saved['gru'] = z, gsaved
        c1 = self.chans[1]
        k1 = 16 * c1
        y = torch.empty((frames, 4, 4, c1), device=dev)
        hl.fc_dgrad(frames, k1, self.n_hidden, z, fp.param('dc1/W').view(self.n_hidden, k1), fp.param('dc1/b'), c1,
                    y.view(frames, k1))
        fuse_stats = train and 'stats' in FUSE and self.sync_bn is None
        pending = None               # (epilogue, partial sums) of the deconvolution that produced y
        for l in (1, 2, 3, 4):
            co = self.chans[l]
            m = y.numel() // co
            name = 'bn%d' % l
            if train:
                stats = torch.empty(4 * co, device=dev)
                rm = self.running[name + '/avg_mean'] if update_stats else None
                rv = self.running[name + '/avg_var'] if update_stats else None
                if pending is not None:
                    ep, part = pending
                    hl.bn_stats_from_partials(m, co, part, ep.n_slots, ep.slot_stride, fp.param(name + '/gamma'), fp.param(name + '/beta'),
                                              stats, rm, rv, self.ws)
                else:
                    hl.bn_stats(m, co, y, fp.param(name + '/gamma'), fp.param(name + '/beta'), stats, rm, rv, self.ws, sync=self.sync_bn)
                if update_stats:
                    self.bn_count[name] += 1
                saved['stats'][l] = stats
                ss = stats[2 * co:]
            else:
                inv = torch.rsqrt(self.running[name + '/avg_var'] + 2e-5)
                scale = fp.param(name + '/gamma') * inv
                ss = torch.cat((scale, fp.param(name + '/beta') - self.running[name + '/avg_mean'] * scale))
            saved['y'][l] = y
            a16 = self._s16(l + 1)
            if l == 4 and OUT16 and Y16 and self.precision == 'bf16' and self._s16(4) and self.sync_bn is None:
                # the last layer's input (64 channels, 16x the clip's size) is bf16 too when its readers take a bf16 y beside the
                # fp32 clip: the MFMA kernel of the 4-channel layers forward, 'bf16y' weight gradient backward
                a16 = hl.dgrad_c4_mfma_covers(hl.with_precision(self._geom(5, frames, clip_order_n=n), 'bf16y'))
            a = torch.empty_like(y, dtype=torch.bfloat16 if a16 else torch.float32)   # the operand of layer l + 1's GEMMs
            gn = self._geom(l + 1, frames) if l < 4 else None
            if gn is not None and self._split_only(('dgrad', gn), ('wgrad', gn)):      # 'f32x3': both readers take the split form
                saved['split'][l + 1] = torch.empty(y.shape[:-1] + (4 * co,), device=dev, dtype=torch.bfloat16)
                saved['only'].add(l + 1)                         # saved['a'][l + 1] stays unwritten
                hl.bn_act_fwd(m, co, y, ss, hl.ACT_RELU, saved['split'][l + 1], split_out=True)
            else:
                hl.bn_act_fwd(m, co, y, ss, hl.ACT_RELU, a)
            saved['a'][l + 1] = a
            h = 4 << l
            pending = None
            if l < 4:
                y = torch.empty((frames, h, h, self.chans[l + 1]), device=dev)
                geom = self._geom(l + 1, frames)
                w, b = self._w('dc%d/W' % (l + 1), self._s16(l + 1)), fp.param('dc%d/b' % (l + 1))
                y16 = (OUT16 and self._s16(l + 1) and self.sync_bn is None and (fuse_stats or not train)
                       and hl.dgrad_tile(geom, a, w, b) < 1000)                    # (as DisNet.forward_groups)
                if y16:
                    y = torch.empty_like(y, dtype=torch.bfloat16)
                if fuse_stats:
                    part = self._part_buf(geom, 'dgrad', 1)
                    ep = hl.epilogue(sums=hl.SUMS_STATS, groups=1, part=part, out_bf16=y16)
                    if self._cdgrad(geom, a, 'dc%d/W' % (l + 1), w, b, y, ep=ep, ys=lambda: self._sp(saved['split'], l + 1, saved['a'][l + 1]),
                                    force=l + 1 in saved['only']):
                        pending = (ep, part)
                else:
                    self._cdgrad(geom, a, 'dc%d/W' % (l + 1), w, b, y, ys=lambda: self._sp(saved['split'], l + 1, saved['a'][l + 1]),
                                 force=l + 1 in saved['only'])
        x = torch.empty((n, T, IMG, IMG, self.cp_out), device=dev)
        g5 = self._geom(5, frames, clip_order_n=n)
        if saved['a'][5].dtype == torch.bfloat16:
            g5 = hl.with_precision(g5, 'bf16y')
        if hl.dgrad_c4_mfma_covers(g5):
            # the last deconvolution (64 -> 3 channels) on the matrix pipe; bias + tanh (model/net.py:114) follow as an
            # element-wise pass over the 4-channel clip (x = tanh(1 * x + b)), which the MFMA kernel cannot carry
            if getattr(self, '_one_bias', None) is None or self._one_bias.device != dev:
                self._one_bias = torch.ones(2 * self.cp_out, device=dev)
            self._one_bias[self.cp_out:].copy_(fp.param('dc5/b'))
            hl.conv_dgrad(g5, saved['a'][5], fp.param('dc5/W'), None, x)
            hl.bn_act_fwd(x.numel() // self.cp_out, self.cp_out, x, self._one_bias, hl.ACT_TANH, x)
        else:
            hl.conv_dgrad(g5, saved['a'][5], fp.param('dc5/W'), fp.param('dc5/b'), x, act=hl.ACT_TANH)
        saved['x'] = x
        return x, saved

    # ---- backward --------------------------------------------------------------------------
    def grad_bucket_late(self):
        """(start, end) of the flat-gradient range dc2/W .. dc5/b: 85 % of the bytes, final once layer 2's weight
        gradient is queued -- its exchange then overlaps the rest of the backward pass (dc2's input gradient, dc1, GRU)."""
        o = self.fp.offsets
        return o['dc2/W'], o['bn1/gamma']

    def backward(self, saved, gx_clip, on_late_bucket=None):
        """gx_clip: gradient w.r.t. the clip tensor [n][T][64][64][cp_out].  Accumulates into the flat gradient."""
        n = saved['n']
        T, dz, dl, dc = self.video_len, self.dim_zm, self.dim_zl, self.dim_zc
        frames = T * n
        fp = self.fp
        dev = self.device
        hl.set_tag('G')
        self._held = None
        g = torch.empty((frames, IMG, IMG, self.cp_out), device=dev)
        hl.tanh_bwd_to_frames(n, T, IMG * IMG * self.cp_out, gx_clip, saved['x'], g)
        pending = None               # (epilogue, partial sums) the GEMM that produced g left for BatchNorm's backward
        for l in (5, 4, 3, 2):
            geom = self._geom(l, frames)
            ci = lay.pad4(self.chans[l])
            m = g.numel() // ci
            s16 = self._s16(l)
            gy_split = None
            if l < 5:
                name = 'bn%d' % l
                gy = torch.empty_like(g, dtype=torch.bfloat16) if s16 else g      # bf16-stored operand of wgrad / fprop, else in place
                if self._split_only(('wgrad', geom), ('fprop', geom)):            # 'f32x3': written in the split form only
                    gy_split = torch.empty(g.shape[:-1] + (4 * ci,), device=g.device, dtype=torch.bfloat16)
                go = gy_split if gy_split is not None else gy
                if pending is not None:
                    ep, part = pending
                    hl.bn_act_bwd_from_partials(m, ci, g, saved['y'][l], saved['stats'][l], fp.param(name + '/gamma'), hl.ACT_RELU, part,
                                                ep.n_slots, ep.slot_stride, go, fp.grad(name + '/gamma'), fp.grad(name + '/beta'), self.ws,
                                                split_out=gy_split is not None)
                else:
                    hl.bn_act_bwd(m, ci, g, saved['y'][l], saved['stats'][l], fp.param(name + '/gamma'), hl.ACT_RELU, go,
                                  fp.grad(name + '/gamma'), fp.grad(name + '/beta'), self.ws, sync=self.sync_bn, split_out=gy_split is not None)
                g = gy
            if l == 5:
                hl.colsum_acc(m, ci, g, fp.grad('dc5/b'), self.ws)         # dc1..dc4 feed BatchNorm: exact zero
            gsp = {} if gy_split is None else {0: gy_split}      # 'f32x3': the split form of g, shared by the two GEMMs that read it

            def gxs():
                return self._sp(gsp, 0, g)
            g_only = gy_split is not None                        # g exists in its split form only
            a5_16 = l == 5 and saved['a'][5].dtype == torch.bfloat16      # (the 64-channel side of the last layer stored in bf16)
            self._hold_wgrads()                                  # (queued once the input-gradient GEMM below is: see _hold_wgrads)
            self._cwgrad(hl.with_precision(geom, 'bf16y') if a5_16 else geom, g, saved['a'][l], fp.grad('dc%d/W' % l), xs=gxs,
                         ys=lambda: self._sp(saved.get('split'), l, saved['a'][l]), force=g_only or l in saved.get('only', ()))
            if l == 2 and on_late_bucket is not None and self._held is None:
                self._after_wgrads(on_late_bucket)
            wl = self._w('dc%d/W' % l, s16)
            g16 = (OUT16 and 2 < l < 5 and s16 and self.sync_bn is None                        # (layer 1's gradient feeds the fp32 fully-connected layer)
                   and hl.fprop_tile(geom, g, wl, None) < 1000)                            # (as DisNet.backward)
            g16 = g16 or (a5_16 and OUT16 and self.sync_bn is None)     # ... and its gradient likewise (BatchNorm's backward reads bf16)
            ga = torch.empty_like(saved['a'][l], dtype=torch.bfloat16 if g16 else torch.float32)
            pending = None
            if self._fuse_bwd_sums(s16, 'fprop', geom):     # ga is the gradient w.r.t. relu(bn_{l-1}(y_{l-1})): the sums of that BatchNorm's backward
                part = self._part_buf(geom, 'fprop', 1)
                ep = hl.epilogue(sums=hl.SUMS_BN_BWD, groups=1, part=part, bn_y=saved['y'][l - 1], bn_stats=[saved['stats'][l - 1]],
                                 bn_act=hl.ACT_RELU, out_bf16=g16)
                if self._with_bwd_sums(lambda e: self._cfprop(geom, g, 'dc%d/W' % l, wl, None, ga, ep=e, xs=gxs, force=g_only), ep):
                    pending = (ep, part)
            else:
                self._cfprop(geom, g, 'dc%d/W' % l, wl, None, ga, xs=gxs, force=g_only)
            if self._held is not None:
                self._flush_wgrads()
                if l == 2 and on_late_bucket is not None:
                    self._after_wgrads(on_late_bucket)
            g = ga
        c1 = self.chans[1]
        k1 = 16 * c1
        if pending is not None:
            ep, part = pending
            hl.bn_act_bwd_from_partials(frames * 16, c1, g, saved['y'][1], saved['stats'][1], fp.param('bn1/gamma'), hl.ACT_RELU, part,
                                        ep.n_slots, ep.slot_stride, g, fp.grad('bn1/gamma'), fp.grad('bn1/beta'), self.ws)
        else:
            hl.bn_act_bwd(frames * 16, c1, g, saved['y'][1], saved['stats'][1], fp.param('bn1/gamma'), hl.ACT_RELU, g,
                          fp.grad('bn1/gamma'), fp.grad('bn1/beta'), self.ws, sync=self.sync_bn)
        hl.fc_wgrad(frames, k1, self.n_hidden, g.view(frames, k1), saved['z'], fp.grad('dc1/W').view(self.n_hidden, k1))
        gz = torch.empty_like(saved['z'])
        hl.fc_fprop(frames, k1, self.n_hidden, g.view(frames, k1), fp.param('dc1/W').view(self.n_hidden, k1), None, gz)
        d = saved['draw']
        hl.gru_seq_bwd(n, T, dz, dl, dc, fp.param('g0'), d['e'], d['labels'], saved['gru'], gz, fp.grad('g0'))
        self._wgrad_join()
