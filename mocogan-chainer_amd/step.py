"""One MoCoGAN training iteration on the device (reference model/updater.py:78-113) and the
data-parallel gradient exchange.

The kernel schedule reproduces the reference's observable ordering:
  forwards with the OLD parameters: D_I(real), D_V(real), G, D_I(fake), D_V(fake)      (:97-108)
  D_I: loss_dis -> backward(real)+backward(fake) -> [all-reduce] -> Adam              (:111)
  D_V: the same                                                                         (:112)
  G  : loss_gen -> gradient through the ALREADY UPDATED D_V and D_I (saved activations,
       new weights / gamma: quirk Q5) -> G backward -> [all-reduce] -> Adam             (:113)
Gradients Chainer computes into the other networks and then discards (Q6) are not computed.
"""
import math

import torch

from . import hiplib as hl
from . import layout as lay
from . import nets


import os

CHAINS = os.environ.get('MCG_CHAINS', '1') == '1'          # the VideoDiscriminator's real / fake calls as two chains on two streams (A/B switch)
CHAINS_MIN_N = int(os.environ.get('MCG_CHAINS_MIN_N', '64'))  # ... from this many clips per call on (measured on one MI355X, clips/s with / without:
                                                              # fp32 batch 32 2120 / 2138, 64 2206 / 2140, 128 2284 / 2257; bf16 128 11.89 / 11.80 k, 256 12.19 / 11.94 k)
chain_iterations = 0                                          # iterations that took the two-chain schedule (tests assert that it really ran)


_STREAMS = {}


def _side_streams(device):
    """the process's six side streams on `device`: [D_I's, three weight-gradient streams (G, D_I, D_V), D_V's chain, its weight-gradient]"""
    key = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    if key not in _STREAMS:
        # (MCG_STREAM_MAP="a,b,c,d,e,f": tuning aid -- which of twelve streams, in creation order, plays each role)
        m = [int(v) for v in os.environ.get('MCG_STREAM_MAP', '0,1,2,3,4,5').split(',')]
        pool = [torch.cuda.Stream(device=device) for _ in range(max(m) + 1)]
        _STREAMS[key] = [pool[i] for i in m]
    return _STREAMS[key]


class AdamHyper:
    """train.py:93-101 -- Chainer Adam(alpha=2e-4, beta1=5e-5) (beta2 0.999 / eps 1e-8 defaults)
    plus the WeightDecay(1e-5) hook."""

    def __init__(self, alpha=2e-4, beta1=5e-5, beta2=0.999, eps=1e-8, weight_decay=1e-5):
        self.alpha, self.beta1, self.beta2, self.eps, self.weight_decay = alpha, beta1, beta2, eps, weight_decay

    def lr(self, t):
        return self.alpha * math.sqrt(1.0 - self.beta2 ** t) / (1.0 - self.beta1 ** t)


def adam_update(net, hyper, grad_scale=1.0):
    """optimizer.update() tail: hooks, t += 1, per-parameter Adam -- one launch over the flat buffers.
    grad_scale: 1 / world under data parallelism (the flat gradient then holds the SUM over the ranks)."""
    net.t += 1
    fp = net.fp
    hl.adam_wd(fp.p, fp.g, fp.m, fp.v, hyper.lr(net.t), hyper.beta1, hyper.beta2, hyper.eps, hyper.weight_decay, grad_scale,
               p16=fp.p16 if net.precision == 'bf16' else None)
    fp.touch()
    net.refresh_wsplits()                                         # ('f32x3': the split forms of the filters follow in one launch)


class GradExchange:
    """Data-parallel averaging of one network's flat gradient over the ranks of `group`
    (RCCL over xGMI on the GPU box, gloo in the CPU tests).  No reference counterpart: the
    reference is single-device (train.py:87-91).  Each rank runs the step on its own shard of
    the batch with its own BatchNorm statistics; gradients are averaged, so every rank applies
    the same Adam update and the replicas stay bit-identical.  The collective SUMS; the division by the world
    size happens inside the Adam kernel (``grad_scale``), not in a pass of its own."""

    def __init__(self, group=None, force=False):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.grad_scale = 1.0 / self.world
        # force: a world of ONE still goes through the collectives (a one-GPU rehearsal of the RCCL path: bench.py MCG_DP_REHEARSE_NCCL)
        self.active = self.world > 1 or (force and dist.is_available() and dist.is_initialized())

    def start(self, flat_grad):
        """Asynchronous SUM all-reduce of a (slice of a) flat gradient; returns a handle for finish()."""
        if not self.active:
            return None
        return (self.dist.all_reduce(flat_grad, op=self.dist.ReduceOp.SUM, group=self.group, async_op=True), flat_grad)

    def finish(self, handle):
        """Wait for the collective start() returned the handle of (stream-ordered on RCCL: the current stream waits, the host
        does not).  Its buffer then holds the SUM over the ranks; adam_update(..., grad_scale=self.grad_scale) makes it the mean."""
        if handle is None:
            return
        work, _ = handle
        work.wait()

    def all_reduce_sum(self, t):
        """In-place SUM over the ranks, ordered on the current stream (synchronised BatchNorm's per-channel sums)."""
        if self.active:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)

    def broadcast_params(self, tensors, src=0):
        if not self.active:
            return
        for t in tensors:
            self.dist.broadcast(t, src=src, group=self.group)


class TrainStep:
    """Holds the three networks, their Adam hyper-parameters and runs update_core on device data."""

    def __init__(self, model, gen, dis_i, dis_v, hyper=None, exchange=None, seed=0, rank=0, precision=None, overlap=False, sync_bn=False,
                 input_ready_early=False):
        assert model in ('normal', 'cgan', 'infogan')
        self.model, self.gen, self.dis_i, self.dis_v = model, gen, dis_i, dis_v
        if precision is not None:                                 # 'f32' | 'bf16': MFMA operand type of every conv GEMM
            assert precision in ('f32', 'bf16', 'f32x3')
            for net in (gen, dis_i, dis_v):
                net.set_precision(precision)
        self.hyper = hyper or {'image_gen': AdamHyper(), 'image_dis': AdamHyper(), 'video_dis': AdamHyper()}
        self.exchange = exchange
        # sync_bn (opt-in): BatchNorm statistics and their backward sums are all-reduced over the ranks, i.e.
        # taken over the GLOBAL batch; without it every rank normalises with the statistics of its own shard.
        for net in (gen, dis_i, dis_v):
            net.sync_bn = exchange if (sync_bn and exchange is not None) else None
        self.seed, self.rank = seed, rank
        # input_ready_early (two-chain schedule only): the caller guarantees that x_real / t_real of run() were complete BEFORE the
        # previous run() was queued (resident synthetic data; a loader that stays one batch ahead).  The real chain of the
        # VideoDiscriminator then waits only for what it really needs -- the VideoDiscriminator's Adam update of the previous
        # iteration -- instead of for everything queued so far, and runs beside the END of the previous iteration (G's backward pass):
        # the host is several milliseconds ahead of the GPU, so the launches are there to overlap.  Off by default: a caller that
        # writes parameters or inputs on the current stream between two iterations (the teacher-forced tests do) needs the full wait.
        self.input_ready_early = bool(input_ready_early)
        self._ev_dv_updated = None
        self.iteration = 0
        self.device = gen.device
        self.loss = torch.zeros(3, device=self.device)            # loss_dis_i, loss_dis_v, loss_gen
        # overlap: independent work goes to side HIP streams -- the whole ImageDiscriminator update (small
        # kernels that cannot fill 256 CUs) beside the VideoDiscriminator's, and every weight-gradient GEMM
        # beside the input-gradient GEMM of the same layer.  Same kernels, same results; only placement in time.
        self.side = None
        self._wstreams = None
        self._chain_stream = None
        self._dv_chains = None
        self.set_overlap(overlap)

    def set_overlap(self, on):
        """Switch the side-stream placement on or off (between iterations)."""
        if on and self._wstreams is None:
            # ONE set of side streams per process and device (not per TrainStep): the runtime deals hardware queues to streams in
            # creation order (four queues by default), and which streams share a queue decides what really overlaps -- a second
            # TrainStep with fresh streams got another assignment and ran 4-5 % slower than the first (bench.py's secondary
            # workloads against the same workloads run alone).  (More hardware queues are not better: GPU_MAX_HW_QUEUES=8 cost
            # the batch-32 iteration 22 %.)
            st = _side_streams(self.device)
            self._side = st[0]
            self._wstreams = st[1:4]
            # the VideoDiscriminator's real and fake calls as two chains (nets._Net.chain): a second compute stream and a
            # weight-gradient stream of its own for the real chain
            self._chain_stream = st[4]
            self._dv_chains = self.dis_v.make_chains([st[5], self._wstreams[2]])
        self.side = self._side if on else None
        for i, net in enumerate((self.gen, self.dis_i, self.dis_v)):
            net.wgrad_stream = self._wstreams[i] if on else None

    # ---- cgan label planes (model/updater.py:65-76) -------------------------------------------------
    def _concat_label_clip(self, x_dev, labels):
        """x_dev [n][T][H][W][4] with C=3 -> [n][T][H][W][pad4(3+dim_zl)] with -1/+1 label planes (one launch)."""
        n, T, H, W, _ = x_dev.shape
        c, dl = self.gen.out_channels, self.gen.dim_zl
        out = torch.empty((n, T, H, W, lay.pad4(c + dl)), device=x_dev.device)
        return hl.concat_label_planes(x_dev, c, dl, labels, out)

    # ---- perf-mode randomness bookkeeping ----------------------------------------------------------
    STREAMS_PER_RANK = 64        # Philox stream ids one rank may consume per iteration (uses 5 x 8)
    MAX_RANKS = 64

    def frame_index(self, it, T):
        """The frame D_I looks at (model/updater.py:96).  One draw per iteration, shared by the whole
        batch (quirk Q7) -- and, under data parallelism, by every rank: seeded by (seed, iteration) only."""
        g = torch.Generator()
        g.manual_seed(self.seed * 7919 + it)
        return int(torch.randint(0, T, (1,), generator=g))

    @classmethod
    def stream_base(cls, it, rank):
        """First Philox stream id of (iteration, rank): disjoint across both, so every rank adds
        independent noise / latent codes while sharing the seed."""
        assert 0 <= rank < cls.MAX_RANKS
        return (it * cls.MAX_RANKS + rank + 1) * cls.STREAMS_PER_RANK

    # ---- one iteration -------------------------------------------------------------------------
    def run(self, x_real, t_real=None, inject=None, input_event=None):
        """x_real: device tensor in the reference layout (N,C,T,H,W) float32 (model/updater.py:89-90) -- or the loader's own
        form, uint8 (N,T,H,W,C) as datasets.py decodes it: the first kernel of each discriminator then normalises (v - 128) / 128
        while it writes the device layout (mcg_pack_clip_u8), and no float copy of the batch is ever made.
        t_real: int32 device tensor (N,) or None.
        input_event: an event recorded (on whatever stream produced x_real -- a loader's copy stream) when x_real was complete; the
        caller's current stream must be ordered behind it as well.  With it the two-chain schedule starts the VideoDiscriminator's
        real chain as soon as that event and the previous iteration's Adam(D_V) have happened, beside the end of the previous
        iteration (what `input_ready_early` promises for resident data, here per call and checked by the hardware).
        inject: parity mode -- dict with 't', 'noise_{i,v}_{real,fake}' (lists of 4 device tensors in
        device layout, pre-scaled) and 'gen' (latent draw dict); None = perf mode (Philox in-kernel,
        frame index from a seeded host generator shared by all ranks, quirk Q7)."""
        gen, di, dv = self.gen, self.dis_i, self.dis_v
        u8 = x_real.dtype == torch.uint8
        if u8:
            n, T, H, W, c_img = x_real.shape
            if not x_real.is_contiguous():
                raise hl.McgError("uint8 clips must be dense (N,T,H,W,C)")
        else:
            n, c_img, T, H, W = x_real.shape
        hw = H * W
        it = self.iteration
        base = self.stream_base(it, self.rank)
        seed = self.seed
        t = int(inject['t']) if inject is not None else self.frame_index(it, T)

        def nz(key):
            return inject[key] if inject is not None else None

        def rngs(k):
            return None if inject is not None else (seed, base + 8 * k)

        cgan = self.model == 'cgan'
        with_ce = self.model == 'infogan'
        cp = dv.cp0

        # ------------------------------------------------ forward: real
        if cgan:
            # label planes are part of D's input: build the device-layout clip once, then add noise
            tmp = torch.empty((n, T, H, W, lay.pad4(c_img)), device=self.device)
            (hl.pack_clip_u8 if u8 else hl.pack_clip)(n, c_img, lay.pad4(c_img), T, hw, x_real, tmp)
            xr = self._concat_label_clip(tmp, t_real)
            c_valid = c_img + gen.dim_zl

            def first_real_v(out, na):
                hl.bn_act_fwd(n * T * hw, cp, xr, None, hl.ACT_NONE, out, c_valid=c_valid, **na)

            def first_real_i(out, na):
                hl.bn_act_fwd(n * hw, cp, xr[:, t], None, hl.ACT_NONE, out, c_valid=c_valid, rows_per_item=hw,
                              item_stride=T * hw * cp, **na)
        else:
            def first_real_v(out, na):
                (hl.pack_clip_u8 if u8 else hl.pack_clip)(n, c_img, cp, T, hw, x_real, out, **na)

            def first_real_i(out, na):
                if u8:                                               # frame t of every clip: the clip's item stride, one frame
                    hl.pack_clip_u8(n, c_img, cp, 1, hw, x_real[:, t], out, stride_n=T * hw * c_img, **na)
                else:
                    hl.pack_clip(n, c_img, cp, 1, hw, x_real[:, :, t], out, stride_n=c_img * T * hw, stride_c=T * hw, **na)

        # ------------------------------------------------ forward: D_V on the real clips, as a chain of its own beside G's forward
        ex = self.exchange
        main = torch.cuda.current_stream()
        real_chain = None
        # (not for 'f32x3': measured 2-3 % SLOWER at 32 / 64 / 128 clips -- its launches are tuned, form by form, at the 2n batch)
        if CHAINS and self.side is not None and ex is None and dv.sync_bn is None and dv.precision != 'f32x3' and n >= CHAINS_MIN_N:
            cs = self._chain_stream
            if (self.input_ready_early or input_event is not None) and self._ev_dv_updated is not None and not cgan:
                cs.wait_event(self._ev_dv_updated)                   # the previous iteration's Adam(D_V)
                if input_event is not None:
                    cs.wait_event(input_event)                       # x_real (otherwise ready by contract)
            else:
                cs.wait_stream(main)                                 # x_real (and whatever produced it)
            with torch.cuda.stream(cs), self._dv_chains[0]:
                real_chain = dv.forward(n, first_real_v, noise=nz('noise_v_real'), rng=rngs(1))
            x_real.record_stream(cs)
        # ------------------------------------------------ forward: fake (G needs nothing from D)
        draw = inject['gen'] if inject is not None else gen.draw(n, (seed, base + 8 * 2))
        x_fake, s_gen = gen.forward(n, draw)
        t_fake = draw['labels']
        xf = self._concat_label_clip(x_fake, t_fake) if cgan else x_fake
        c_valid = c_img + (gen.dim_zl if cgan else 0)

        def first_fake_v(out, na):
            hl.bn_act_fwd(n * T * hw, cp, xf, None, hl.ACT_NONE, out, c_valid=c_valid, **na)

        def first_fake_i(out, na):
            hl.bn_act_fwd(n * hw, cp, xf[:, t], None, hl.ACT_NONE, out, c_valid=c_valid, rows_per_item=hw,
                          item_stride=T * hw * cp, **na)

        # ------------------------------------------------ forward: both discriminators on [real | fake]
        # (model/updater.py:97-98,107-108 as one 2n batch per net; per-call BatchNorm statistics, real first)
        cd = di.out_channels
        gs = ex.grad_scale if ex else 1.0
        side = self.side if self.side is not None else main
        side.wait_stream(main)                                        # x_fake is ready
        # ------------------------------------------------ image_dis_optimizer.update(loss_dis, ...)   :111
        with torch.cuda.stream(side):
            y_i, s_i = di.forward_groups(n, [dict(first_input=first_real_i, noise=nz('noise_i_real'), rng=rngs(0)),
                                             dict(first_input=first_fake_i, noise=nz('noise_i_fake'), rng=rngs(3))])
            y_real_i, y_fake_i = y_i[:n], y_i[n:]
            g_i = torch.empty((2 * n, cd), device=self.device)      # [d loss / d logits] of (real | fake)
            di.zero_grad()
            hl.loss_dis(n, cd, y_real_i, y_fake_i, t_real, t_fake, False, self.loss[0:1], g_i[:n], g_i[n:])
            di.backward(s_i, g_i, True)
            work_i = ex.start(di.fp.g) if ex else None
        # ------------------------------------------------ video_dis_optimizer.update(loss_dis, ...)   :112
        two_chains = CHAINS and self.side is not None and ex is None and dv.sync_bn is None and real_chain is not None
        late = []
        lo, hi = dv.grad_bucket_late()
        if two_chains:
            # The real and the fake call as TWO CHAINS on two streams (round 4): the real call needs nothing from G and was queued on
            # the chain stream BEFORE G's forward (real_chain below); the fake call follows G on the main stream.  One chain's
            # BatchNorm / activation passes (HBM-bound, 3.3 of 21.5 ms exposed at batch 256 in the one-batch schedule) run beside the
            # other's GEMMs.  Same kernels on n instead of 2n samples; what the calls share is kept in the reference's order by events
            # (nets._Net._ordered: running statistics real -> fake, gradient accumulators real -> fake).
            global chain_iterations
            chain_iterations += 1
            cs, (ch_r, ch_f) = self._chain_stream, self._dv_chains
            y_real_v, s_real = real_chain
            with ch_f:
                y_fake_v, s_fake = dv.forward(n, first_fake_v, noise=nz('noise_v_fake'), rng=rngs(4))
            main.wait_stream(cs)                                     # the real logits
            y_real_v.record_stream(main)
            g_v = torch.empty((2 * n, cd), device=self.device)
            dv.zero_grad()
            hl.loss_dis(n, cd, y_real_v, y_fake_v, t_real, t_fake, with_ce, self.loss[1:2], g_v[:n], g_v[n:])
            cs.wait_stream(main)                                     # the loss gradients (and the cleared gradient buffer)
            g_v.record_stream(cs)
            with torch.cuda.stream(cs), ch_r:
                dv.backward(s_real, g_v[:n], True)
            with ch_f:
                dv.backward(s_fake, g_v[n:], True)
            main.wait_stream(cs)
            s_v = {'n': n, 'G': 2, 'chains': [s_real, s_fake]}
            del s_real
        else:
            if real_chain is not None:                              # (decided before G's forward; cannot happen)
                raise RuntimeError('the real chain ran but the two-chain schedule was not taken')
            y_v, s_v = dv.forward_groups(n, [dict(first_input=first_real_v, noise=nz('noise_v_real'), rng=rngs(1)),
                                             dict(first_input=first_fake_v, noise=nz('noise_v_fake'), rng=rngs(4))])
            y_real_v, y_fake_v = y_v[:n], y_v[n:]
            g_v = torch.empty((2 * n, cd), device=self.device)
            dv.zero_grad()
            hl.loss_dis(n, cd, y_real_v, y_fake_v, t_real, t_fake, with_ce, self.loss[1:2], g_v[:n], g_v[n:])
            # D_V's gradient is exchanged in two buckets: dc4/W..dc5/b (76 % of the bytes) is final after the
            # first two layers of the backward pass and travels while dc3..dc1 are still being computed
            dv.backward(s_v, g_v, True, on_late_bucket=(lambda: late.append(ex.start(dv.fp.g[lo:hi]))) if ex else None)
        with torch.cuda.stream(side):
            if ex:
                ex.finish(work_i)                                    # D_I's exchange overlapped D_V's backward
            adam_update(di, self.hyper['image_dis'], gs)
        if ex:
            rest = [ex.start(dv.fp.g[:lo]), ex.start(dv.fp.g[hi:])]
            for h in late + rest:
                ex.finish(h)
        adam_update(dv, self.hyper['video_dis'], gs)
        if self._dv_chains is not None:                               # (what the next iteration's real chain has to wait for)
            if self._ev_dv_updated is None:
                self._ev_dv_updated = torch.cuda.Event()
            self._ev_dv_updated.record(main)
        main.wait_stream(side)                                        # D_I's logits and updated weights (Q5)
        # ------------------------------------------------ image_gen_optimizer.update(loss_gen, ...)   :113
        gen.zero_grad()
        gi, gv = g_i[:n], g_v[:n]                                    # buffers reused: D's backward has consumed them
        hl.loss_gen(n, cd, y_fake_i, y_fake_v, t_fake, with_ce, self.loss[2:3], gi, gv)
        gx = torch.empty_like(xf)
        s_fake_v, s_fake_i = dv.select_group(s_v, 1), di.select_group(s_i, 1)
        gi_geom = hl.make_geom(n, 1, H, W, cp, di.chans[1], 1, x_stride0=T * hw * cp, precision=di.gemm_precision, ci_valid=di.chans[0])
        if self.side is not None:
            # the two discriminators' input gradients are independent until they meet in frame t of the clip gradient:
            # D_I's layers 5..2 (small kernels) run on the side stream beside D_V's; its LAST launch -- the input gradient of
            # dc1, accumulated onto frame t of what D_V's pass wrote -- follows on the main stream once both are done
            self.side.wait_stream(main)                              # loss_gen's gradients
            with torch.cuda.stream(self.side):
                last_i = di.backward(s_fake_i, gi, False, gx=gx[:, t], gx_geom=gi_geom, gx_accumulate=True, defer_gx=True)
            gi.record_stream(self.side)
            dv.backward(s_fake_v, gv, False, gx=gx)                  # new D_V weights, old activations (Q5)
            main.wait_stream(self.side)
            last_i()
        else:
            dv.backward(s_fake_v, gv, False, gx=gx)                  # new D_V weights, old activations (Q5)
            di.backward(s_fake_i, gi, False, gx=gx[:, t], gx_geom=gi_geom, gx_accumulate=True)
        if cgan:                                                     # label planes carry no gradient to G: the clip's channels alone
            gx = hl.concat_label_planes(gx, c_img, 0, None, torch.empty_like(x_fake))
        late_g = []
        lo_g, hi_g = gen.grad_bucket_late()
        gen.backward(s_gen, gx, on_late_bucket=(lambda: late_g.append(ex.start(gen.fp.g[lo_g:hi_g]))) if ex else None)
        if ex:
            rest_g = [ex.start(gen.fp.g[:lo_g]), ex.start(gen.fp.g[hi_g:])]
            for h in late_g + rest_g:
                ex.finish(h)
        adam_update(gen, self.hyper['image_gen'], gs)
        self.iteration += 1
        return {'x_fake': x_fake, 't_fake': t_fake, 't': t, 'gx_fake': gx, 'saved_gen': s_gen, 'saved_fake_i': s_fake_i, 'saved_fake_v': s_fake_v,
                'saved_i': s_i, 'saved_v': s_v,
                'y_real_i': y_real_i, 'y_real_v': y_real_v, 'y_fake_i': y_fake_i, 'y_fake_v': y_fake_v}

    def losses(self):
        """(loss_dis_i, loss_dis_v, loss_gen) of the last iteration -- forces a host sync."""
        l = self.loss.cpu().tolist()
        return {'image_dis/loss': l[0], 'video_dis/loss': l[1], 'image_gen/loss': l[2]}


def make_models(model='normal', num_labels=6, channel=3, dim_zc=50, dim_zm=10, n_filters=64, video_length=16,
                use_noise=True, noise_sigma=0.2, device='cuda', seed=0):
    """The three networks exactly as train.py:69-85 configures them for --model normal|cgan|infogan
    (note: n_filters_gen is used for all three nets there, and on MUG num_labels = 6 makes the GRU
    label-conditioned even for 'normal': quirk Q9)."""
    if model not in ('normal', 'cgan', 'infogan'):
        raise ValueError('unknown model %r' % model)
    if model in ('cgan', 'infogan') and num_labels == 0:
        raise ValueError('Called %s model, but dataset has no label.' % model)       # train.py:75,81
    c_d = channel + (num_labels if model == 'cgan' else 0)
    out_d = 1 + (num_labels if model == 'infogan' else 0)
    gen = nets.GenNet(dim_zc, dim_zm, num_labels, channel, n_filters, video_length, device=device, seed=seed)
    dis_i = nets.DisNet(2, c_d, out_d, n_filters, use_noise, noise_sigma, video_length, device=device, seed=seed + 1)
    dis_v = nets.DisNet(3, c_d, out_d, n_filters, use_noise, noise_sigma, video_length, device=device, seed=seed + 2)
    return gen, dis_i, dis_v


def pretune_and_share_tiles(exchange, model, precision, batch, rank, **model_kw):
    """Data parallel: make every rank run the SAME GEMM tile codes.  With autotune on, a geometry the shipped table does
    not hold is timed at its first launch -- per rank, so two ranks may keep different winners (results stay identical:
    Adam consumes the all-reduced gradient; but a rank with a slower choice sets the step time).  Here every rank runs ONE
    local iteration (no exchange) on throw-away networks, which tunes whatever is missing, then rank 0's table replaces
    everyone's.  No reference counterpart (the reference is single-device, train.py:87-91)."""
    if exchange is None or not exchange.active:
        return
    if hl._autotune:
        model_kw.setdefault('num_labels', 6)
        gen, di, dv = make_models(model, seed=0, **model_kw)           # (same widths / channels as the run: same geometries)
        ts = TrainStep(model, gen, di, dv, seed=0, rank=rank, precision=precision)
        x = torch.zeros((batch, gen.out_channels, gen.video_len, nets.IMG, nets.IMG), device=gen.device)
        t = torch.zeros(batch, dtype=torch.int32, device=gen.device)
        ts.run(x, t)
        torch.cuda.synchronize()
        del ts, gen, di, dv, x
    table = [[[list(k), v] for k, v in hl.tile_choices().items()]] if rank == 0 else [None]
    exchange.dist.broadcast_object_list(table, src=0, group=exchange.group)
    hl._tile_cache.clear()
    for k, v in table[0]:
        hl._tile_cache[tuple(k)] = int(v)
