"""Net-level and step-level parity of the HIP path against the float64 oracle with injected
randomness (test pyramid levels 2 and 3).  Tolerances: forward rel-L2 <= 1e-5,
gradients <= 1e-4, losses abs <= 1e-5, parameters after 3 steps rel-L2 <= 1e-4."""
import copy

import numpy as np
import pytest
import torch

from oracle import net as onet
from oracle import updater as oupd

pytestmark = pytest.mark.gpu

F64 = np.float64


@pytest.fixture(scope="module")
def pkg():
    assert torch.cuda.is_available()
    import mocogan_chainer_amd.hiplib as hl
    import mocogan_chainer_amd.layout as lay
    import mocogan_chainer_amd.nets as nets
    import mocogan_chainer_amd.step as step
    hl.load()
    return hl, lay, nets, step


def dev(a, dtype=torch.float32):
    return torch.tensor(np.asarray(a), dtype=dtype, device="cuda")


def rel_l2(a, b):
    a = a.detach().cpu().double().numpy() if isinstance(a, torch.Tensor) else np.asarray(a, F64)
    b = np.asarray(b, F64)
    assert a.shape == b.shape, (a.shape, b.shape)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)


def _f64(p):
    return {k: (v.astype(F64) if v.dtype.kind == 'f' else v) for k, v in p.items()}


def _perturb(p, rng):
    for k in p:
        if k.endswith(('/b', '/beta')):
            p[k] = rng.randn(*p[k].shape) * 0.1
        if k.endswith('/gamma'):
            p[k] = 1 + rng.randn(*p[k].shape) * 0.1
    return p


def is_pre_bn_bias(key, net_kind):
    """Bias of a conv/deconv that feeds BatchNorm: its true gradient is exactly zero (BN removes the
    mean) so both sides hold rounding noise, which Adam's sign-like first steps turn into O(alpha)
    drifts that never reach any output.  Compared with an absolute tolerance instead."""
    if not key.endswith('/b') or not key.startswith('dc'):
        return False
    l = int(key[2])
    return (l in (2, 3, 4)) if net_kind == 'dis' else (l in (1, 2, 3, 4))


def check_params(got, ref, net_kind, tol, what, grads=None, gtol=0.0):
    """grads: the oracle's gradients of this iteration.  Adam with beta1 = 5e-5 (train.py:99-101) moves every parameter by
    ~alpha * sign(g): an element whose gradient (incl. the weight-decay term) is within the device / oracle difference of
    zero -- |g| below 1e-3 of the tensor's rms -- may step the other way on either side, an O(alpha) difference of one
    element that no tolerance on the gradient excludes.  Those elements (~0.1 %) are compared through their gradients
    only; all others must agree to `tol`."""
    for k, v in ref.items():
        if k.endswith('/N'):
            continue
        if is_pre_bn_bias(k, net_kind) or k.endswith('/avg_mean'):
            # avg_mean inherits the drift of the bias it absorbs
            assert np.abs(got[k] - v).max() < 1e-3, (what, k)
            continue
        g, r = np.asarray(got[k], F64), np.asarray(v, F64)
        if grads is not None and k in grads:
            gt = grads[k] + 1e-5 * r                                   # WeightDecay(1e-5) hook (train.py:96)
            # (gtol: the bound the gradients themselves are held to -- 0.15 on an iteration with an activation on its kink:
            #  an element smaller than twice that share of the tensor's rms may change sign within the bound)
            keep = np.abs(gt) > max(1e-3, 2 * gtol) * np.sqrt(np.mean(gt * gt))
            assert keep.mean() > 0.3, (what, k, keep.mean())     # (GRU columns of labels that do not occur have g = wd * p only)
            g, r = g[keep], r[keep]
        err = np.linalg.norm(g - r) / max(np.linalg.norm(r), 1e-30)
        assert err < tol, (what, k, err)


def noise_to_dev(lay, lst):
    return [lay.act_to_dev(dev(a)) for a in lst]


def draw_to_dev(d):
    return {'h0': dev(d['h0']), 'e': dev(d['e']), 'zc': dev(d['zc']),
            'labels': None if d['labels'] is None else dev(d['labels'], torch.int32)}


@pytest.mark.parametrize("ndim,out,nf", [(2, 1, 8), (3, 1, 8), (3, 7, 16)])
def test_discriminator_forward_backward(pkg, ndim, out, nf):
    hl, lay, nets, _ = pkg
    rng = np.random.RandomState(100 + ndim + out)
    n = 3
    p = _perturb(_f64(onet.init_discriminator(rng, ndim, 3, out, nf)), rng)
    shp = (n, 3, 64, 64) if ndim == 2 else (n, 3, 16, 64, 64)
    x = rng.uniform(-1, 1, shp)
    noise = [0.2 * rng.randn(*s) for s in onet.dis_noise_shapes(ndim, n, 3, nf)]
    p_run = copy.deepcopy(p)
    y_ref, cache = onet.dis_forward(p_run, x, noise)
    gy = rng.randn(*y_ref.shape)
    grads = oupd.zero_grads(p)
    gx_ref = onet.dis_backward(p, cache, gy, grads, need_gx=True)

    d = nets.DisNet(ndim, 3, out, nf, use_noise=True)
    d.load_reference_params(p)
    xd = lay.act_to_dev(dev(x))

    def first(outp, na):
        hl.bn_act_fwd(outp.numel() // 4, 4, xd, None, hl.ACT_NONE, outp, c_valid=3, **na)
    logits, saved = d.forward(n, first, noise_to_dev(lay, noise))
    assert rel_l2(logits, y_ref.reshape(n, out)) < 1e-5
    d.zero_grad()
    gxd = torch.empty_like(xd)
    d.backward(saved, dev(gy.reshape(n, out)), True, gx=gxd)
    assert rel_l2(lay.act_from_dev(gxd, 3, ndim), gx_ref) < 1e-4
    got = d.export_reference_grads()
    for k in grads:
        if is_pre_bn_bias(k, 'dis'):
            assert np.abs(got[k]).max() < 1e-4 * max(1.0, np.abs(got[k[:-2] + '/W']).max())
        else:
            assert rel_l2(got[k], grads[k]) < 1e-4, k
    # running statistics (two quantities Chainer updates in forward)
    st = d.export_reference_params()
    for l in (2, 3, 4):
        assert rel_l2(st['bn%d/avg_mean' % l], p_run['bn%d/avg_mean' % l]) < 1e-5
        assert rel_l2(st['bn%d/avg_var' % l], p_run['bn%d/avg_var' % l]) < 1e-5


@pytest.mark.parametrize("dim_zl,nf,dim_zm", [(0, 8, 10), (6, 16, 10), (6, 8, 40)])        # (40: the wide-state GRU kernels, --dim_zm > 16)
def test_generator_forward_backward(pkg, dim_zl, nf, dim_zm):
    hl, lay, nets, _ = pkg
    rng = np.random.RandomState(200 + dim_zl)
    n = 3
    p = _perturb(_f64(onet.init_generator(rng, dim_zl=dim_zl, dim_zm=dim_zm, n_filters=nf)), rng)
    draw = onet.gen_draw(rng, n, dim_zl=dim_zl, dim_zm=dim_zm, dtype=F64)
    x_ref, _, cache = onet.gen_forward(copy.deepcopy(p), draw)
    gx = rng.randn(*x_ref.shape)
    grads = oupd.zero_grads(p)
    onet.gen_backward(p, cache, gx, grads)

    g = nets.GenNet(dim_zl=dim_zl, dim_zm=dim_zm, n_filters=nf)
    g.load_reference_params(p)
    xd, saved = g.forward(n, draw_to_dev(draw))                    # [n][T][64][64][4]
    x_clip_ref = x_ref.transpose(1, 2, 0, 3, 4)                     # (N,C,T,H,W)
    assert rel_l2(lay.act_from_dev(xd, 3), x_clip_ref) < 1e-5
    assert float(xd[..., 3].abs().max()) == 0.0
    g.zero_grad()
    g.backward(saved, lay.act_to_dev(dev(gx.transpose(1, 2, 0, 3, 4))))
    got = g.export_reference_grads()
    for k in grads:
        if is_pre_bn_bias(k, 'gen'):
            assert np.abs(got[k]).max() < 1e-4 * max(1.0, np.abs(got[k[:-2] + '/W']).max())
        else:
            assert rel_l2(got[k], grads[k]) < 1e-4, k


TIGHT_MARGIN = 2e-6     # see oracle.updater.update_core: min |pre-activation| over every ReLU / LeakyReLU decision


# ---- perf mode: the randomness of the BENCHMARKED schedule, restated from its specification -----------------------
# TrainStep.run(inject=None) draws every random tensor in-kernel from Philox4x32-10 keyed by (seed, stream id).  The id
# arithmetic below is written out independently of step.py / nets.py (DESIGN.md section 1 is the specification):
#   base(it, rank) = (it * 64 + rank + 1) * 64
#   call k of the iteration owns ids base + 8 k ..: k = 0 D_I(real), 1 D_V(real), 2 the generator's latent draw,
#   3 D_I(fake), 4 D_V(fake); a discriminator call uses id + l - 1 for the add_noise in front of layer l = 1..4
#   (model/net.py:148-154,189-195), the latent draw id + 0 / 1 / 2 / 3 for h0 / e / zc / labels (model/net.py:66,71,102,92).
# Element order: flat device layout [n][T][H][W][C] in groups of four channels per Philox counter, except the tensor
# behind D's first layer, which the fused epilogue draws one counter per (row quad, channel) (oracle.philox.randn_rowquad).
from oracle.philox import perf_mode_randomness      # noqa: E402  (the specification above as code, shared with __graft_entry__.smoke)


def perf_mode_stream_ids(it, rank):
    """every Philox stream id iteration `it` of rank `rank` consumes (the specification above)"""
    base = (it * 64 + rank + 1) * 64
    return [base + 8 * k + j for k in (0, 1, 3, 4) for j in range(4)] + [base + 16 + j for j in range(4)]


def _run_steps(pkg, model, dim_zl, nf, n, steps, seed, min_tight_steps=1, overlap=False, perf=None, precision=None):
    """Teacher-forced multi-step parity: before every iteration the device state (parameters, Adam
    moments and step counters, BN running statistics) is loaded from the oracle, so each iteration
    is compared on identical inputs and errors cannot compound through Adam's sign-like early steps.

    Forward quantities (losses, logits, generated clip) are continuous in the inputs and are always
    held to the tight tolerance.  Gradients are not: a pre-activation within fp32 rounding of 0 may
    take the other ReLU/LeakyReLU branch on the device, which with n=2..3 samples per BatchNorm
    channel moves everything behind it by O(1e-2).  The oracle reports the distance of the closest
    pre-activation to its kink; iterations with margin > TIGHT_MARGIN are held to 1e-4, the others
    to 0.15 (still far below the O(1) error of any wrong formula), and every case must contain
    tight iterations."""
    hl, lay, nets, step = pkg
    rng = np.random.RandomState(seed)
    out_c = 7 if model == 'infogan' else 1
    c_d = 3 + (dim_zl if model == 'cgan' else 0)
    gen = _f64(onet.init_generator(rng, dim_zl=dim_zl, n_filters=nf))
    di = _f64(onet.init_discriminator(rng, 2, c_d, out_c, nf))
    dv = _f64(onet.init_discriminator(rng, 3, c_d, out_c, nf))
    G = nets.GenNet(dim_zl=dim_zl, n_filters=nf)
    DI = nets.DisNet(2, c_d, out_c, nf, use_noise=True)
    DV = nets.DisNet(3, c_d, out_c, nf, use_noise=True)
    ts = step.TrainStep(model, G, DI, DV, overlap=overlap, precision=precision, **({'seed': perf[0], 'rank': perf[1]} if perf else {}))
    og, oi, ov = (oupd.new_adam_state(q) for q in (gen, di, dv))
    tight_steps = 0
    for s in range(steps):
        for net, p, st in ((G, gen, og), (DI, di, oi), (DV, dv, ov)):      # teacher forcing
            net.load_reference_params(p)
            net.load_adam_state(st)
        x_real = rng.uniform(-1, 1, (n, 3, 16, 64, 64))
        t_real = rng.randint(0, 6, n)
        if perf:
            # perf mode (what bench.py and Updater.update_core run): nothing is injected; the oracle is fed the draws
            # oracle.philox states for the ids of this (seed, iteration, rank)
            assert ts.iteration == s
            rnd = perf_mode_randomness(perf[0], s, perf[1], model, n, nf, dim_zl)
            inject = None
        else:
            rnd = oupd.draw_step_randomness(rng, model, n, 3, nf, dim_zl=dim_zl, dtype=F64)
            inject = {'t': rnd['t'], 'gen': draw_to_dev(rnd['gen'])}
            for k in ('noise_i_real', 'noise_v_real', 'noise_i_fake', 'noise_v_fake'):
                inject[k] = noise_to_dev(lay, rnd[k])
        ref = oupd.update_core(model, gen, di, dv, og, oi, ov, x_real, t_real, rnd, dim_zl=dim_zl, keep=True)
        out = ts.run(dev(x_real), dev(t_real, torch.int32), inject)
        losses = ts.losses()
        if perf:
            assert out['t'] == rnd['t']
            if dim_zl:
                assert np.array_equal(out['t_fake'].cpu().numpy(), rnd['gen']['labels'])
        # ---- forward: always tight
        assert abs(losses['image_dis/loss'] - ref['loss_dis_i']) < 1e-5, s
        assert abs(losses['video_dis/loss'] - ref['loss_dis_v']) < 1e-5, s
        assert abs(losses['image_gen/loss'] - ref['loss_gen']) < 1e-5, s
        assert rel_l2(lay.act_from_dev(out['x_fake'], 3), ref['x_fake'][:, :3]) < 1e-5, s
        for k in ('y_real_i', 'y_real_v', 'y_fake_i', 'y_fake_v'):
            assert rel_l2(out[k], ref[k].reshape(out[k].shape)) < 2e-5, (s, k)
        # ---- backward / update: tight when no activation sits on a kink
        tight = ref['min_margin'] > TIGHT_MARGIN
        tight_steps += tight
        gtol, ptol = (1e-4, 1e-4) if tight else (0.15, 1e-2)
        assert rel_l2(lay.act_from_dev(out['gx_fake'], 3), ref['gx_fake']) < gtol, (s, ref['min_margin'])
        for name, net, kind, refg in (('D_I', DI, 'dis', ref['grads_dis_i']), ('D_V', DV, 'dis', ref['grads_dis_v']),
                                      ('G', G, 'gen', ref['grads_gen'])):
            got = net.export_reference_grads()
            for k in refg:
                if not is_pre_bn_bias(k, kind):
                    # (the bias of the logit layer is 1..7 numbers, each the sum of a few loss gradients of either sign with
                    # magnitude <= 1/n: after cancellation fp32 rounding of the TERMS, ~6e-8 each, is all that is left to
                    # compare -- held to an absolute bound where the relative one is below that floor)
                    tiny = refg[k].size <= 8 and np.abs(np.asarray(got[k], F64) - refg[k]).max() < 5e-7
                    assert tiny or rel_l2(got[k], refg[k]) < gtol, (s, name, k, ref['min_margin'])
        check_params(DI.export_reference_params(), di, 'dis', ptol, 'D_I step %d' % s, ref['grads_dis_i'], 0.0 if tight else gtol)
        check_params(DV.export_reference_params(), dv, 'dis', ptol, 'D_V step %d' % s, ref['grads_dis_v'], 0.0 if tight else gtol)
        check_params(G.export_reference_params(), gen, 'gen', ptol, 'G step %d' % s, ref['grads_gen'], 0.0 if tight else gtol)
        assert G.t == DI.t == DV.t == s + 1
    assert tight_steps >= min_tight_steps, "seed no longer yields a well-conditioned iteration"


@pytest.mark.parametrize("model,dim_zl,seed", [("normal", 0, 1267), ("normal", 6, 3342), ("infogan", 6, 481), ("cgan", 6, 320)])
def test_update_core_three_steps(pkg, model, dim_zl, seed):
    """the seeds of the committed golden iterations (tests/golden/make_golden.py searched them in round 6): all three iterations of
    every variant keep their pre-activations clear of the kinks, so every gradient and update is held to the tight tolerances"""
    _run_steps(pkg, model, dim_zl, nf=4, n=2, steps=3, seed=seed, min_tight_steps=3)


def test_update_core_with_side_streams(pkg):
    """overlap=True only moves launches onto side HIP streams (D_I's update beside D_V's, wgrad beside dgrad);
    the same teacher-forced parity must hold."""
    _run_steps(pkg, "infogan", 6, nf=4, n=2, steps=3, seed=313, overlap=True)
    _run_steps(pkg, "normal", 6, nf=16, n=3, steps=2, seed=77, min_tight_steps=0, overlap=True)


def test_update_core_with_the_two_chain_schedule(pkg, monkeypatch):
    """The VideoDiscriminator's real and fake calls as two chains on two HIP streams (step.CHAINS; default from 64 clips per call on):
    the real call is queued before G's forward, both backward passes run side by side with their own scratch and weight-gradient
    streams, and what the calls share -- running BatchNorm statistics, the gradient accumulators -- keeps the reference's order
    (real, then fake: model/updater.py:97-98,107-108,112) through events.  Same kernels on n instead of 2n samples: the teacher-forced
    and the perf-mode parity must hold unchanged, for fp32 and bf16 networks, and the running statistics (two updates per iteration,
    order-dependent) are part of the compared state."""
    step = pkg[3]
    monkeypatch.setattr(step, 'CHAINS_MIN_N', 1)
    before = step.chain_iterations
    _run_steps(pkg, "infogan", 6, nf=4, n=2, steps=3, seed=313, overlap=True)
    _run_steps(pkg, "normal", 6, nf=16, n=3, steps=2, seed=77, min_tight_steps=0, overlap=True)
    _run_steps(pkg, "cgan", 6, nf=4, n=2, steps=2, seed=320, min_tight_steps=0, overlap=True)
    _run_steps(pkg, "normal", 6, nf=4, n=2, steps=3, seed=1311, overlap=True, perf=(PERF_SEEDS[("normal", 6, 0)], 0), min_tight_steps=2)
    assert step.chain_iterations - before == 3 + 2 + 2 + 3, "the two-chain schedule did not run"
    # ('f32x3' networks keep the one-batch schedule: measured slower in two chains)
    monkeypatch.setenv('MCG_SPLIT', 'always')
    _run_steps(pkg, "normal", 6, nf=16, n=3, steps=1, seed=77, min_tight_steps=0, overlap=True, precision='f32x3')
    monkeypatch.delenv('MCG_SPLIT')
    assert step.chain_iterations - before == 10
    monkeypatch.setattr(step, 'CHAINS', False)                     # ... and the switch really switches
    _run_steps(pkg, "normal", 6, nf=4, n=2, steps=1, seed=1311, overlap=True, min_tight_steps=0)
    assert step.chain_iterations - before == 10


def test_update_core_with_split_fp32_convolutions(pkg, monkeypatch):
    """precision 'f32x3': the wide convolutions' forward and input-gradient GEMMs on the bf16 matrix pipe, operands as three bf16
    terms (six bf16 products per fp32 product, fp32 accumulation) -- an fp32 computation, held to the SAME tolerances as the
    fp32-MFMA iteration.  MCG_SPLIT=always: every launch that has a split form takes it (no timing decides)."""
    hl = pkg[0]
    monkeypatch.setenv('MCG_SPLIT', 'always')
    before, before_only = hl.split_launches, hl.split_only_outputs
    _run_steps(pkg, "normal", 6, nf=16, n=3, steps=2, seed=77, min_tight_steps=0, precision='f32x3')
    _run_steps(pkg, "infogan", 6, nf=16, n=3, steps=2, seed=78, min_tight_steps=0, overlap=True, precision='f32x3')
    assert hl.split_launches - before >= 2 * 2 * 8, "the split form did not run"
    # ... and where every reader of an activation / gradient tensor is a split launch, BatchNorm's passes wrote the split form only
    assert hl.split_only_outputs > before_only
    monkeypatch.setenv('MCG_SPLIT_ONLY', '0')                      # the other path: fp32 tensors, split on the way into the GEMMs
    _run_steps(pkg, "normal", 6, nf=16, n=3, steps=1, seed=77, min_tight_steps=0, precision='f32x3')
    # BatchNorm's backward sums from the row-wise epilogue of the split launches (nets.FUSE 'bwd2': off by default, no gain measured)
    nets = pkg[2]
    monkeypatch.setattr(nets, 'FUSE', nets.FUSE | {'bwd2'})
    _run_steps(pkg, "normal", 6, nf=16, n=3, steps=1, seed=77, min_tight_steps=0, precision='f32x3')


PERF_CASES = [("normal", 0, 1303), ("normal", 6, 1311), ("infogan", 6, 1313), ("cgan", 6, 1320)]
# Philox seeds (searched on the CPU with the oracle) for which at least two of the three iterations keep every pre-activation
# more than TIGHT_MARGIN away from its kink: {(model, dim_zl, rank): seed}
PERF_SEEDS = {("normal", 0, 3): 85, ("normal", 0, 0): 78, ("normal", 6, 3): 77, ("normal", 6, 0): 87,
              ("infogan", 6, 3): 100, ("infogan", 6, 0): 81, ("cgan", 6, 3): 79, ("cgan", 6, 0): 80}


@pytest.mark.parametrize("overlap", [False, True])
@pytest.mark.parametrize("model,dim_zl,seed", PERF_CASES)
def test_update_core_perf_mode_matches_oracle_philox(pkg, model, dim_zl, seed, overlap):
    """The schedule bench.py times and Updater.update_core runs (inject=None: in-kernel Philox for the eight noise
    tensors, GenNet.draw, the seeded frame index) against the oracle fed with oracle.philox draws of the same
    (seed, stream id)s -- three iterations, every model variant, with and without side streams, on rank 0 and on a
    non-zero rank.  Same tolerances as the injected-randomness test (model/updater.py:78-113, model/net.py:10-15,
    55-56,90-93)."""
    rank = 0 if overlap else 3
    _run_steps(pkg, model, dim_zl, nf=4, n=2, steps=3, seed=seed, overlap=overlap, perf=(PERF_SEEDS[(model, dim_zl, rank)], rank),
               min_tight_steps=2)


def test_perf_mode_parity_notices_a_wrong_stream_id(pkg, monkeypatch):
    """the check above has teeth: shift the ids the DEVICE uses by one and it must fail"""
    _, _, _, step = pkg
    orig = step.TrainStep.stream_base.__func__
    monkeypatch.setattr(step.TrainStep, 'stream_base', classmethod(lambda cls, it, rank: orig(cls, it, rank) + 1))
    with pytest.raises(AssertionError):
        _run_steps(pkg, "normal", 6, nf=4, n=2, steps=1, seed=1311, perf=(83, 0), min_tight_steps=0)


# ---- the device's ReLU / LeakyReLU decisions, read back from what its forward pass saved ---------------------------
def _to_ref_bool(a):
    """[n][T][H][W][C] (or [frames][H][W][C]) -> (n,C,T,H,W) / (frames,C,H,W)"""
    return np.ascontiguousarray(np.moveaxis(a, -1, 1))


def _bn_decisions(y, stats, strict):
    """sign of fma(y, scale, shift) for a saved conv output y [..][C] and its BatchNorm stats [mean | istd | scale | shift]:
    float64 evaluates y * scale + shift with ONE rounding of an exact product plus a summand -- its sign is that of the exact
    value, which is also the sign of the device's fp32 fma (both round the same real number)."""
    C = y.shape[-1]
    st = stats.double().cpu().numpy()
    pre = y.double().cpu().numpy() * st[2 * C:3 * C] + st[3 * C:4 * C]
    return _to_ref_bool(pre > 0 if strict else pre >= 0)


def device_decisions(out, G, DI, DV):
    """{'real_i' | 'real_v' | 'fake_i' | 'fake_v' | 'gen': {layer: boolean array}} of one TrainStep.run."""
    k = {}
    for net, key, tag in ((DI, 'saved_i', 'i'), (DV, 'saved_v', 'v')):
        for gi, grp in enumerate(('real', 'fake')):
            sg = net.select_group(out[key], gi)
            n, co = sg['n'], net.chans[1]
            g1 = net._geom(1, n)
            words = sg['mask1'].cpu().numpy().astype(np.uint32)              # [rows][(co + 31) / 32], bit c & 31 of word c >> 5
            bits = ((words[:, :, None] >> np.arange(32, dtype=np.uint32)) & 1).reshape(words.shape[0], -1)[:, :co].astype(bool)
            d = {1: _to_ref_bool(bits.reshape(n, g1.To, g1.Ho, g1.Wo, co))}
            for l in (2, 3, 4):
                d[l] = _bn_decisions(sg['y'][l], sg['stats'][l][0], strict=False)
            k['%s_%s' % (grp, tag)] = d
    sg = out['saved_gen']
    k['gen'] = {l: _bn_decisions(sg['y'][l], sg['stats'][l], strict=True) for l in (1, 2, 3, 4)}
    return k


KINK_BAND = 1e-4     # |pre-activation| below which the device's branch is taken over (its own error there is ~1e-6)


@pytest.mark.parametrize("precision", ['f32', 'f32x3'])
def test_update_core_full_width_with_the_devices_activation_decisions(pkg, precision, monkeypatch):
    """One iteration at the reference's width (n_filters = 64: the 128x128 tiles, the split-K weight gradients, K = 4096 ..
    16384) held to the TIGHT tolerances.  With ~2e7 pre-activations some always lie within fp32 rounding of their kink, where
    an fp32 implementation may take the other branch than float64 -- no seed avoids that (expected count within 2e-6 of a kink:
    ~30).  So the oracle is run with the device's decisions inside the band |pre-activation| < 1e-4 (a few hundred
    elements; everywhere else the two must agree, asserted) -- both sides then differentiate the SAME piecewise-linear
    function and every gradient must match to 1e-4, every forward quantity to 1e-5 (model/updater.py:78-113)."""
    hl, lay, nets, step = pkg
    model, dim_zl, nf, n = 'infogan', 6, 64, 2
    rng = np.random.RandomState(999)
    gen = _f64(onet.init_generator(rng, dim_zl=dim_zl, n_filters=nf))
    di = _f64(onet.init_discriminator(rng, 2, 3, 7, nf))
    dv = _f64(onet.init_discriminator(rng, 3, 3, 7, nf))
    G, DI, DV = nets.GenNet(dim_zl=dim_zl, n_filters=nf), nets.DisNet(2, 3, 7, nf, use_noise=True), nets.DisNet(3, 3, 7, nf, use_noise=True)
    monkeypatch.setenv('MCG_SPLIT', 'always')                      # ('f32x3': every launch that has a split form takes it)
    split_before = hl.split_launches
    ts = step.TrainStep(model, G, DI, DV, precision=precision)
    og, oi, ov = (oupd.new_adam_state(q) for q in (gen, di, dv))
    for net, p, st in ((G, gen, og), (DI, di, oi), (DV, dv, ov)):
        net.load_reference_params(p)
        net.load_adam_state(st)
    x_real = rng.uniform(-1, 1, (n, 3, 16, 64, 64))
    t_real = rng.randint(0, 6, n)
    rnd = oupd.draw_step_randomness(rng, model, n, 3, nf, dim_zl=dim_zl, dtype=F64)
    inject = {'t': rnd['t'], 'gen': draw_to_dev(rnd['gen'])}
    for k in ('noise_i_real', 'noise_v_real', 'noise_i_fake', 'noise_v_fake'):
        inject[k] = noise_to_dev(lay, rnd[k])
    out = ts.run(dev(x_real), dev(t_real, torch.int32), inject)
    losses = ts.losses()
    assert (hl.split_launches - split_before >= 24) == (precision == 'f32x3')
    kinks = device_decisions(out, G, DI, DV)
    kinks['eps'] = KINK_BAND
    ref = oupd.update_core(model, gen, di, dv, og, oi, ov, x_real, t_real, rnd, dim_zl=dim_zl, keep=True, kinks=kinks)
    print('full-width step: min margin %.1e, %d decisions taken from the device inside the band, %d disagreements outside'
          % (ref['min_margin'], ref['kink_forced'], ref['kink_disagree']))
    assert ref['kink_disagree'] == 0
    assert abs(losses['image_dis/loss'] - ref['loss_dis_i']) < 1e-5
    assert abs(losses['video_dis/loss'] - ref['loss_dis_v']) < 1e-5
    assert abs(losses['image_gen/loss'] - ref['loss_gen']) < 1e-5
    assert rel_l2(lay.act_from_dev(out['x_fake'], 3), ref['x_fake'][:, :3]) < 1e-5
    for k in ('y_real_i', 'y_real_v', 'y_fake_i', 'y_fake_v'):
        assert rel_l2(out[k], ref[k].reshape(out[k].shape)) < 2e-5, k
    assert rel_l2(lay.act_from_dev(out['gx_fake'], 3), ref['gx_fake']) < 1e-4
    for name, net, kind, refg in (('D_I', DI, 'dis', ref['grads_dis_i']), ('D_V', DV, 'dis', ref['grads_dis_v']),
                                  ('G', G, 'gen', ref['grads_gen'])):
        got = net.export_reference_grads()
        for k in refg:
            if not is_pre_bn_bias(k, kind):
                tiny = refg[k].size <= 8 and np.abs(np.asarray(got[k], F64) - refg[k]).max() < 5e-7
                assert tiny or rel_l2(got[k], refg[k]) < 1e-4, (name, k, rel_l2(got[k], refg[k]))
    check_params(DI.export_reference_params(), di, 'dis', 1e-4, 'D_I', ref['grads_dis_i'])
    check_params(DV.export_reference_params(), dv, 'dis', 1e-4, 'D_V', ref['grads_dis_v'])
    check_params(G.export_reference_params(), gen, 'gen', 1e-4, 'G', ref['grads_gen'])


def _free_params_rel_l2(net, ref):
    """rel-L2 over all trained tensors of a network (running statistics and the exact-zero pre-BatchNorm biases left out)"""
    got = net.export_reference_params()
    num = den = 0.0
    for k, v in ref.items():
        if v.dtype.kind != 'f' or 'avg_' in k:
            continue
        if k.startswith('dc') and k.endswith('/b') and ('bn%s/gamma' % k[2]) in ref:
            continue
        a = np.asarray(got[k].cpu() if torch.is_tensor(got[k]) else got[k], F64)
        num += float(((a - v) ** 2).sum())
        den += float((v ** 2).sum())
    return (num / den) ** 0.5


@pytest.mark.parametrize("precision,nf,n", [('f32x3', 16, 3), ('f32', 16, 3)])
def test_free_running_iterations_with_the_devices_activation_decisions(pkg, precision, nf, n, monkeypatch):
    """Three FREE-RUNNING iterations of the headline's arithmetic ('f32x3': every launch that has a split form takes it; side streams)
    held to the fp32 tolerances in EVERY iteration: losses and the generated clip to 1e-5, all parameters to 1e-4 after the third
    (SURVEY 8c).  The device keeps its own parameters, Adam moments and BatchNorm statistics; the oracle keeps its own too and, per
    iteration, takes the device's ReLU / LeakyReLU decisions inside the band |pre-activation| < 1e-4 (oracle.net._decide) -- at
    n_filters >= 16 some pre-activation always lies within fp32 rounding of its kink, and without that a free-running comparison is
    tight in its first iteration only (tests/test_gpu_fullwidth.py holds its f32x3 schedule to 2e-3 for that reason).  Outside the
    band the two must agree (asserted).  Round 5's review, item 5b; model/updater.py:78-113.
    (Not at n_filters = 64: tried in round 6 with 2 clips -- the first iteration is tight, in the second 66 of ~2e7 decisions differ
    OUTSIDE the 1e-4 band: Adam's first updates are +-alpha whatever the gradient's size, so parameters whose gradient is rounding
    noise on both sides move 4e-4 apart, and with two samples per BatchNorm channel that reaches the pre-activations.  The
    full-width iteration is held tight by test_update_core_full_width_with_the_devices_activation_decisions, teacher-forced.)"""
    hl, lay, nets, step = pkg
    model, dim_zl = 'infogan', 6
    monkeypatch.setenv('MCG_SPLIT', 'always')
    rng = np.random.RandomState(4)
    gen = _f64(onet.init_generator(rng, dim_zl=dim_zl, n_filters=nf))
    di = _f64(onet.init_discriminator(rng, 2, 3, 7, nf))
    dv = _f64(onet.init_discriminator(rng, 3, 3, 7, nf))
    G, DI, DV = nets.GenNet(dim_zl=dim_zl, n_filters=nf), nets.DisNet(2, 3, 7, nf, use_noise=True), nets.DisNet(3, 3, 7, nf, use_noise=True)
    og, oi, ov = (oupd.new_adam_state(q) for q in (gen, di, dv))
    for net, p, st in ((G, gen, og), (DI, di, oi), (DV, dv, ov)):       # identical START only
        net.load_reference_params(p)
        net.load_adam_state(st)
    split_before, multi_before = hl.split_launches, hl.split_multi_launches
    ts = step.TrainStep(model, G, DI, DV, overlap=True, precision=precision)
    forced = []
    for it in range(3):
        x_real = rng.uniform(-1, 1, (n, 3, 16, 64, 64))
        t_real = rng.randint(0, 6, n)
        rnd = oupd.draw_step_randomness(rng, model, n, 3, nf, dim_zl=dim_zl, dtype=F64)
        inject = {'t': rnd['t'], 'gen': draw_to_dev(rnd['gen'])}
        for k in ('noise_i_real', 'noise_v_real', 'noise_i_fake', 'noise_v_fake'):
            inject[k] = noise_to_dev(lay, rnd[k])
        out = ts.run(dev(x_real), dev(t_real, torch.int32), inject)
        losses = ts.losses()
        kinks = device_decisions(out, G, DI, DV)
        kinks['eps'] = KINK_BAND
        ref = oupd.update_core(model, gen, di, dv, og, oi, ov, x_real, t_real, rnd, dim_zl=dim_zl, keep=True, kinks=kinks)
        assert ref['kink_disagree'] == 0, it
        forced.append(ref['kink_forced'])
        assert abs(losses['image_dis/loss'] - ref['loss_dis_i']) < 1e-5, it
        assert abs(losses['video_dis/loss'] - ref['loss_dis_v']) < 1e-5, it
        assert abs(losses['image_gen/loss'] - ref['loss_gen']) < 1e-5, it
        assert rel_l2(lay.act_from_dev(out['x_fake'], 3), ref['x_fake'][:, :3]) < 1e-5, it
        assert rel_l2(lay.act_from_dev(out['gx_fake'], 3), ref['gx_fake']) < 1e-4, it
    errs = {name: _free_params_rel_l2(net, p) for name, net, p in (('G', G, gen), ('D_I', DI, di), ('D_V', DV, dv))}
    print('free-running %s nf=%d: decisions taken from the device per iteration %s, parameters rel-L2 after 3 iterations %s' % (precision, nf, forced, errs))
    assert all(e < 1e-4 for e in errs.values()), errs
    assert G.t == DI.t == DV.t == 3
    if precision == 'f32x3':
        assert hl.split_launches - split_before >= 3 * 8, "the split form did not run"
        assert hl.split_multi_launches - multi_before >= 2 * 2, "the filters' split forms were not refreshed in one launch per Adam update"


@pytest.mark.parametrize("chains", [False, True])
def test_update_core_bf16_mfma_one_step(pkg, chains, monkeypatch):
    """(chains: the same iteration in the two-chain schedule on side streams, as the batch-256 bench line runs it.)
    BASELINE configs[2] arithmetic: conv operands rounded to bf16 inside the kernels, fp32 accumulation,
    parameters / BN statistics / Adam in fp32.  The oracle stays the float64 restatement; tolerances are the
    bf16 ones of SURVEY 8c (loss abs <= 5e-2, forward rel-L2 <= 2e-2).  Gradients pass through thousands of
    ReLU decisions that bf16 rounding moves, so they are held to direction (cosine) rather than to digits;
    the bf16 kernels themselves are pinned digit-for-digit in test_gpu_ops.py."""
    hl, lay, nets, step = pkg
    model, dim_zl, nf, n = 'normal', 6, 16, 4
    rng = np.random.RandomState(4242)
    gen = _f64(onet.init_generator(rng, dim_zl=dim_zl, n_filters=nf))
    di = _f64(onet.init_discriminator(rng, 2, 3, 1, nf))
    dv = _f64(onet.init_discriminator(rng, 3, 3, 1, nf))
    G = nets.GenNet(dim_zl=dim_zl, n_filters=nf)
    DI = nets.DisNet(2, 3, 1, nf, use_noise=True)
    DV = nets.DisNet(3, 3, 1, nf, use_noise=True)
    if chains:
        monkeypatch.setattr(step, 'CHAINS_MIN_N', 1)
    before = step.chain_iterations
    ts = step.TrainStep(model, G, DI, DV, precision='bf16', overlap=chains)
    assert G.precision == DI.precision == DV.precision == 'bf16'
    og, oi, ov = (oupd.new_adam_state(q) for q in (gen, di, dv))
    for net, p, st in ((G, gen, og), (DI, di, oi), (DV, dv, ov)):
        net.load_reference_params(p)
        net.load_adam_state(st)
    x_real = rng.uniform(-1, 1, (n, 3, 16, 64, 64))
    t_real = rng.randint(0, 6, n)
    rnd = oupd.draw_step_randomness(rng, model, n, 3, nf, dim_zl=dim_zl, dtype=F64)
    ref = oupd.update_core(model, gen, di, dv, og, oi, ov, x_real, t_real, rnd, dim_zl=dim_zl, keep=True)
    inject = {'t': rnd['t'], 'gen': draw_to_dev(rnd['gen'])}
    for k in ('noise_i_real', 'noise_v_real', 'noise_i_fake', 'noise_v_fake'):
        inject[k] = noise_to_dev(lay, rnd[k])
    out = ts.run(dev(x_real), dev(t_real, torch.int32), inject)
    assert step.chain_iterations - before == int(chains)
    losses = ts.losses()
    report = {}
    for k, r in (('image_dis/loss', 'loss_dis_i'), ('video_dis/loss', 'loss_dis_v'), ('image_gen/loss', 'loss_gen')):
        report[k] = abs(losses[k] - ref[r])
        assert report[k] < 5e-2, (k, losses[k], ref[r])
    report['x_fake'] = rel_l2(lay.act_from_dev(out['x_fake'], 3), ref['x_fake'][:, :3])
    assert report['x_fake'] < 2e-2, report
    assert report['x_fake'] > 1e-5, "bf16 rounding left no trace: the bf16 kernels did not run"

    def cosine(a, b):
        a = a.detach().cpu().double().numpy().ravel() if isinstance(a, torch.Tensor) else np.asarray(a, np.float64).ravel()
        b = np.asarray(b, np.float64).ravel()
        return float(a @ b / max(np.linalg.norm(a) * np.linalg.norm(b), 1e-300))

    for name, net, kind, refg in (('D_I', DI, 'dis', ref['grads_dis_i']), ('D_V', DV, 'dis', ref['grads_dis_v']),
                                  ('G', G, 'gen', ref['grads_gen'])):
        got = net.export_reference_grads()
        for k in refg:
            if k.endswith('/W'):
                report[name + '.' + k] = cosine(got[k], refg[k])
    print('bf16 step report:', {k: round(v, 5) for k, v in report.items()})
    for k, v in report.items():
        if k.endswith('/W'):
            assert v > 0.95, (k, v)
