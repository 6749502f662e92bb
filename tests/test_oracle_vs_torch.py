"""Pins the NumPy oracle against an independent torch-CPU/autograd statement.
CPU only.  float64 throughout, so agreement is to ~1e-10."""
import copy

import numpy as np
import pytest
import torch
import torch.nn.functional as TF

from oracle import functions as F
from oracle import net, updater
import torch_ref as R

F64 = np.float64


def _np(b):
    return b.detach().numpy() if isinstance(b, torch.Tensor) else np.asarray(b, F64)


def close(a, b, tol=1e-9, floor=1e-30):
    a, b = np.asarray(a, F64), _np(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    err = np.abs(a - b).max() / max(np.abs(b).max(), floor)
    assert err < tol, err


def grads_close(grads, ref, tol=1e-8):
    """ref: dict key -> torch grad.  The bias of a conv/deconv that feeds BatchNorm has an
    exactly-zero true gradient (BN removes the mean), so both sides hold rounding noise there:
    bias gradients are compared on the scale of the same layer's weight gradient."""
    for k in grads:
        floor = 1e-30
        if k.endswith('/b'):
            floor = np.abs(_np(ref[k[:-2] + '/W'])).max()
        close(grads[k], ref[k], tol, floor)


# ---------------------------------------------------------------------------- op level
@pytest.mark.parametrize("stride,pad,shape,k", [
    ((1, 2, 2), (0, 1, 1), (2, 3, 7, 12, 12), (4, 4, 4)),
    ((1, 3, 3), (0, 0, 0), (2, 5, 4, 4, 4), (4, 4, 4)),
    ((1, 2, 2), (0, 1, 1), (3, 4, 1, 8, 8), (1, 4, 4)),
])
def test_conv3d(stride, pad, shape, k):
    rng = np.random.RandomState(0)
    x = rng.randn(*shape)
    W = rng.randn(6, shape[1], *k)
    b = rng.randn(6)
    y = F.conv3d_fwd(x, W, b, stride, pad)
    xt, Wt, bt = R.tt(x, True), R.tt(W, True), R.tt(b, True)
    yt = TF.conv3d(xt, Wt, bt, stride=stride, padding=pad)
    close(y, yt)
    gy = rng.randn(*y.shape)
    yt.backward(R.tt(gy))
    gx, gW, gb = F.conv3d_bwd(x, W, gy, stride, pad)
    close(gx, xt.grad), close(gW, Wt.grad), close(gb, bt.grad)


@pytest.mark.parametrize("stride,pad,hw", [(1, 0, 1), (2, 1, 4), (2, 1, 8)])
def test_deconv2d(stride, pad, hw):
    rng = np.random.RandomState(1)
    x = rng.randn(3, 5, hw, hw)
    W = rng.randn(5, 7, 4, 4)
    b = rng.randn(7)
    y = F.deconv2d_fwd(x, W, b, stride, pad)
    xt, Wt, bt = R.tt(x, True), R.tt(W, True), R.tt(b, True)
    yt = TF.conv_transpose2d(xt, Wt, bt, stride=stride, padding=pad)
    close(y, yt)
    gy = rng.randn(*y.shape)
    yt.backward(R.tt(gy))
    gx, gW, gb = F.deconv2d_bwd(x, W, gy, stride, pad)
    close(gx, xt.grad), close(gW, Wt.grad), close(gb, bt.grad)


def test_batchnorm_train_and_running_stats():
    rng = np.random.RandomState(2)
    x = rng.randn(4, 6, 3, 5, 5) * 2 + 1
    gamma, beta = rng.randn(6), rng.randn(6)
    am, av = np.zeros(6), np.ones(6)
    y, cache = F.bn_train_fwd(x, gamma, beta, am, av)
    xt, gt, bt = R.tt(x, True), R.tt(gamma, True), R.tt(beta, True)
    rm, rv = torch.zeros(6, dtype=R.DT), torch.ones(6, dtype=R.DT)
    yt = TF.batch_norm(xt, rm, rv, gt, bt, training=True, momentum=0.1, eps=2e-5)
    close(y, yt)
    close(am, rm)
    # Chainer 3.1 folds eps into the running variance (quirk Q10); torch does not.
    m = x.size // 6
    close(av, rv + 0.1 * (m / (m - 1)) * 2e-5)
    gy = rng.randn(*y.shape)
    yt.backward(R.tt(gy))
    gx, gg, gb = F.bn_train_bwd(cache, gamma, gy)
    close(gx, xt.grad), close(gg, gt.grad), close(gb, bt.grad)


def test_softplus_sigmoid_ce():
    rng = np.random.RandomState(3)
    x = rng.randn(50) * 10
    close(F.softplus(x), TF.softplus(R.tt(x)))
    close(F.sigmoid(x), torch.sigmoid(R.tt(x)))
    logits, t = rng.randn(5, 6), rng.randint(0, 6, 5)
    lt = R.tt(logits, True)
    loss_t = TF.cross_entropy(lt, torch.as_tensor(t))
    loss_t.backward()
    loss, g = F.softmax_cross_entropy(logits, t)
    close(loss, loss_t), close(g, lt.grad)


def test_gru_is_chainer_not_torch():
    """Chainer's StatelessGRU: h' = (1-z) h + z h_bar with U applied to (r*h).  torch.nn.GRUCell
    computes r * (U h + b) and h' = (1-z) n + z h, so the two differ -- this test documents why
    the build ships its own GRU kernel."""
    rng = np.random.RandomState(4)
    p = {k[3:]: v.astype(F64) for k, v in net.init_generator(rng, dim_zl=6, n_filters=2).items() if k.startswith('g0/')}
    for k in p:
        if k.endswith('/b'):
            p[k] = rng.randn(*p[k].shape) * 0.1
    h, x = rng.randn(3, 10), rng.randn(3, 16)
    h1, _ = F.gru_step_fwd(p, h, x)
    pt = {'g0/' + k: R.tt(v) for k, v in p.items()}
    close(h1, R.gru_step(pt, R.tt(h), R.tt(x)))
    z = F.sigmoid(x @ p['W_z/W'].T + p['W_z/b'] + h @ p['U_z/W'].T + p['U_z/b'])
    r = F.sigmoid(x @ p['W_r/W'].T + p['W_r/b'] + h @ p['U_r/W'].T + p['U_r/b'])
    n_torch = np.tanh(x @ p['W/W'].T + p['W/b'] + r * (h @ p['U/W'].T + p['U/b']))
    assert np.abs(h1 - ((1 - z) * n_torch + z * h)).max() > 1e-3


# ---------------------------------------------------------------------------- net level
def _f64(p):
    return {k: (v.astype(F64) if v.dtype.kind == 'f' else v) for k, v in p.items()}


def _perturb(p, rng):
    """Move biases / BN affine parameters off their init so the test is not blind to them."""
    for k in p:
        if k.endswith(('/b', '/beta')):
            p[k] = rng.randn(*p[k].shape) * 0.1
        if k.endswith('/gamma'):
            p[k] = 1 + rng.randn(*p[k].shape) * 0.1
    return p


@pytest.mark.parametrize("ndim,out", [(2, 1), (3, 1), (3, 7)])
def test_discriminator_fwd_bwd(ndim, out):
    rng = np.random.RandomState(5)
    n, nf = 2, 4
    p = _perturb(_f64(net.init_discriminator(rng, ndim, 3, out, nf)), rng)
    shp = (n, 3, 64, 64) if ndim == 2 else (n, 3, 16, 64, 64)
    x = rng.uniform(-1, 1, shp)
    noise = [0.2 * rng.randn(*s) for s in net.dis_noise_shapes(ndim, n, 3, nf)]
    y, cache = net.dis_forward(copy.deepcopy(p), x, noise)
    pt = R.params_to_torch(p)
    xt = R.tt(x, True)
    yt = R.dis_forward(pt, xt, noise)
    close(y, yt)
    gy = rng.randn(*y.shape)
    yt.backward(R.tt(gy))
    grads = updater.zero_grads(p)
    gx = net.dis_backward(p, cache, gy, grads, need_gx=True)
    close(gx, xt.grad, 1e-8)
    grads_close(grads, {k: pt[k].grad for k in grads})


@pytest.mark.parametrize("dim_zl", [0, 6])
def test_generator_fwd_bwd(dim_zl):
    rng = np.random.RandomState(6)
    n, nf = 2, 4
    p = _perturb(_f64(net.init_generator(rng, dim_zl=dim_zl, n_filters=nf)), rng)
    draw = net.gen_draw(rng, n, dim_zl=dim_zl, dtype=F64)
    x, labels, cache = net.gen_forward(copy.deepcopy(p), draw)
    pt = R.params_to_torch(p)
    xt = R.gen_forward(pt, draw)
    close(x, xt)
    gx = rng.randn(*x.shape)
    xt.backward(R.tt(gx))
    grads = updater.zero_grads(p)
    net.gen_backward(p, cache, gx, grads)
    grads_close(grads, {k: pt[k].grad for k in grads})


@pytest.mark.parametrize("model", ["normal", "infogan"])
def test_losses(model):
    rng = np.random.RandomState(7)
    n, c = 4, (7 if model == 'infogan' else 1)
    t_real, t_fake = rng.randint(0, 6, n), rng.randint(0, 6, n)
    for is_video in (False, True):
        shp = (n, c, 1, 1, 1) if is_video else (n, c, 1, 1)
        yr, yf = rng.randn(*shp), rng.randn(*shp)
        yrt, yft = R.tt(yr, True), R.tt(yf, True)
        lt = R.loss_dis(model, is_video, yrt, yft, t_real, t_fake)
        lt.backward()
        l, gr, gf = updater.loss_dis(model, is_video, yr, yf, t_real, t_fake)
        close(l, lt), close(gr, yrt.grad), close(gf, yft.grad)
    yi, yv = rng.randn(n, c, 1, 1), rng.randn(n, c, 1, 1, 1)
    yit, yvt = R.tt(yi, True), R.tt(yv, True)
    lt = R.loss_gen(model, yit, yvt, t_fake)
    lt.backward()
    l, gi, gv = updater.loss_gen(model, yi, yv, t_fake)
    close(l, lt), close(gi, yit.grad), close(gv, yvt.grad)


def test_adam_wd_formula():
    """Chainer Adam (Q3) + WeightDecay on every parameter (Q4), two steps, against a literal
    scalar restatement."""
    p = {'a/W': np.array([0.5, -0.25]), 'a/b': np.array([0.1])}
    st = updater.new_adam_state(p)
    ref = {k: v.copy() for k, v in p.items()}
    m = {k: np.zeros_like(v) for k, v in p.items()}
    v_ = {k: np.zeros_like(v) for k, v in p.items()}
    for t in (1, 2):
        g = {'a/W': np.array([0.3, 1e-9]) * t, 'a/b': np.array([-2.0])}
        updater.adam_wd_update(p, g, st)
        lr = 2e-4 * np.sqrt(1 - 0.999 ** t) / (1 - 5e-5 ** t)
        for k in ref:
            gg = g[k] + 1e-5 * ref[k]
            m[k] = m[k] + (1 - 5e-5) * (gg - m[k])
            v_[k] = v_[k] + (1 - 0.999) * (gg * gg - v_[k])
            ref[k] = ref[k] - lr * m[k] / (np.sqrt(v_[k]) + 1e-8)
        for k in ref:
            close(p[k], ref[k], 1e-13)


# ---------------------------------------------------------------------------- step level
def _torch_dis_collect(p, x, noise):
    """torch forward that also returns, per layer, (conv input leaf value, conv output, post-BN)."""
    three_d = x.dim() == 5
    h, saved = x, {}
    for l in (1, 2, 3, 4):
        a = h + R.tt(noise[l - 1])
        if three_d:
            y = TF.conv3d(a, p['dc%d/W' % l], p['dc%d/b' % l], stride=(1, 2, 2), padding=(0, 1, 1))
        else:
            y = TF.conv2d(a, p['dc%d/W' % l], p['dc%d/b' % l], stride=2, padding=1)
        bn = R.bn_train(y, p, 'bn%d' % l) if l >= 2 else y
        h = TF.leaky_relu(bn, 0.2)
        saved[l] = (a.detach(), y.detach(), bn.detach())
    saved[5] = (h.detach(), None, None)
    if three_d:
        out = TF.conv3d(h, p['dc5/W'], p['dc5/b'], stride=(1, 3, 3), padding=0)
    else:
        out = TF.conv2d(h, p['dc5/W'], p['dc5/b'], stride=1, padding=0)
    return out, saved


def _torch_q5_input_grad(p_new, saved, gy, three_d):
    """Layer-by-layer vector-Jacobian products evaluated at the OLD saved points with the NEW
    parameters: exactly what Chainer's retained-array backward computes (quirk Q5)."""
    g = gy
    for l in (5, 4, 3, 2, 1):
        a = saved[l][0].clone().requires_grad_(True)
        W = p_new['dc%d/W' % l].detach()
        if three_d:
            stride, pad = ((1, 3, 3), 0) if l == 5 else ((1, 2, 2), (0, 1, 1))
            y = TF.conv3d(a, W, None, stride=stride, padding=pad)
        else:
            stride, pad = (1, 0) if l == 5 else (2, 1)
            y = TF.conv2d(a, W, None, stride=stride, padding=pad)
        (g,) = torch.autograd.grad(y, a, g)
        if l == 1:
            return g
        _, y_prev, bn_prev = saved[l - 1]
        b = bn_prev.clone().requires_grad_(True)
        (g,) = torch.autograd.grad(TF.leaky_relu(b, 0.2), b, g)
        if l - 1 >= 2:
            yl = y_prev.clone().requires_grad_(True)
            pp = {k: v.detach() for k, v in p_new.items()}
            (g,) = torch.autograd.grad(R.bn_train(yl, pp, 'bn%d' % (l - 1)), yl, g)


@pytest.mark.parametrize("model,dim_zl", [("normal", 0), ("normal", 6), ("infogan", 6)])
def test_update_core_against_autograd(model, dim_zl):
    rng = np.random.RandomState(8)
    n, nf = 2, 4
    out_c = 7 if model == 'infogan' else 1
    gen = _perturb(_f64(net.init_generator(rng, dim_zl=dim_zl, n_filters=nf)), rng)
    di = _perturb(_f64(net.init_discriminator(rng, 2, 3, out_c, nf)), rng)
    dv = _perturb(_f64(net.init_discriminator(rng, 3, 3, out_c, nf)), rng)
    x_real = rng.uniform(-1, 1, (n, 3, 16, 64, 64))
    t_real = rng.randint(0, 6, n)
    rnd = updater.draw_step_randomness(rng, model, n, 3, nf, dim_zl=dim_zl, dtype=F64)
    gen0, di0, dv0 = copy.deepcopy(gen), copy.deepcopy(di), copy.deepcopy(dv)
    og, oi, ov = (updater.new_adam_state(q) for q in (gen, di, dv))
    out = updater.update_core(model, gen, di, dv, og, oi, ov, x_real, t_real, rnd, dim_zl=dim_zl, keep=True)

    # ---- torch: forward everything with the OLD parameters
    pg, pi, pv = (R.params_to_torch(q) for q in (gen0, di0, dv0))
    t = rnd['t']
    xr = R.tt(x_real)
    y_real_i, _ = _torch_dis_collect(pi, xr[:, :, t], rnd['noise_i_real'])
    y_real_v, _ = _torch_dis_collect(pv, xr, rnd['noise_v_real'])
    x_fake_tn = R.gen_forward(pg, rnd['gen'])
    x_fake = x_fake_tn.permute(1, 2, 0, 3, 4)
    y_fake_i, s_i = _torch_dis_collect(pi, x_fake[:, :, t], rnd['noise_i_fake'])
    y_fake_v, s_v = _torch_dis_collect(pv, x_fake, rnd['noise_v_fake'])
    t_fake = rnd['gen']['labels']
    close(out['y_fake_v'], y_fake_v), close(out['y_real_i'], y_real_i)

    l_i = R.loss_dis(model, False, y_real_i, y_fake_i, t_real, t_fake)
    gi = torch.autograd.grad(l_i, [pi[k] for k in out['grads_dis_i']], retain_graph=True)
    grads_close(out['grads_dis_i'], dict(zip(out['grads_dis_i'], gi)), 1e-7)
    l_v = R.loss_dis(model, True, y_real_v, y_fake_v, t_real, t_fake)
    gv = torch.autograd.grad(l_v, [pv[k] for k in out['grads_dis_v']], retain_graph=True)
    grads_close(out['grads_dis_v'], dict(zip(out['grads_dis_v'], gv)), 1e-7)
    close(out['loss_dis_i'], l_i), close(out['loss_dis_v'], l_v)

    # ---- generator loss: value with old logits, gradient through the UPDATED discriminators (Q5)
    yi = y_fake_i.detach().requires_grad_(True)
    yv = y_fake_v.detach().requires_grad_(True)
    l_g = R.loss_gen(model, yi, yv, t_fake)
    close(out['loss_gen'], l_g)
    g_yi, g_yv = torch.autograd.grad(l_g, [yi, yv])
    gx_i = _torch_q5_input_grad(R.params_to_torch(di), s_i, g_yi, False)     # di/dv now hold the updated values
    gx_v = _torch_q5_input_grad(R.params_to_torch(dv), s_v, g_yv, True)
    gx = gx_v.clone()
    gx[:, :, t] += gx_i
    close(out['gx_fake'], gx, 1e-7)
    keys = list(out['grads_gen'])
    gg = torch.autograd.grad(x_fake, [pg[k] for k in keys], gx)
    grads_close(out['grads_gen'], dict(zip(keys, gg)), 1e-7)
    # Q5 is observable: the same gradient through the OLD discriminators differs.
    gx_old = _torch_q5_input_grad(pv, s_v, g_yv, True)
    assert (gx_old - gx_v).abs().max() > 1e-9 * gx_v.abs().max()

    # ---- parameters after the step = Chainer Adam on those gradients
    for p_new, p_old, grads in ((di, di0, out['grads_dis_i']), (dv, dv0, out['grads_dis_v']), (gen, gen0, out['grads_gen'])):
        st = updater.new_adam_state(p_old)
        updater.adam_wd_update(p_old, grads, st)
        for k in grads:
            close(p_new[k], p_old[k], 1e-12)
