"""CPU restatement of one MoCoGAN training iteration.

TEST INFRASTRUCTURE -- see ``oracle/__init__.py``.  Follows reference
``model/updater.py`` (``loss_dis`` :21-44, ``loss_gen`` :46-63, ``concat_label_video``
:65-76, ``update_core`` :78-113) and the optimiser set-up of ``train.py:93-101``
(Chainer-3.1 Adam with alpha=2e-4, beta1=5e-5, beta2=0.999, eps=1e-8 plus a
WeightDecay(1e-5) hook).
"""
import numpy as np

from . import functions as F
from . import net


# ----------------------------------------------------------------------------------------
# losses (value + gradient w.r.t. the logits)
# ----------------------------------------------------------------------------------------
def loss_dis(model, is_video, y_real, y_fake, t_real, t_fake, q1_rows=None):
    """model/updater.py:21-44.  The GAN term uses batch sample 0 only ([:1] slices the batch
    axis) yet divides by the full batch size (quirk Q1).  Returns (loss, gy_real, gy_fake).
    q1_rows (data-parallel emulation with synchronised BatchNorm only): the rows playing "sample 0", i.e. the
    first sample of every rank's shard -- the mean over ranks of the per-shard Q1 terms is their sum over the
    global batch size."""
    n = len(y_fake)
    rows = [0] if q1_rows is None else list(q1_rows)
    loss = F.softplus(-y_real[rows]).sum() / n + F.softplus(y_fake)[rows].sum() / n
    gr = np.zeros_like(y_real)
    gf = np.zeros_like(y_fake)
    gr[rows] = -F.sigmoid(-y_real[rows]) / n
    gf[rows] = F.sigmoid(y_fake[rows]) / n
    if model == 'infogan' and is_video:
        c = y_real.shape[1]
        lr, g1 = F.softmax_cross_entropy(y_real.reshape(n, c)[:, 1:], t_real)
        lf, g2 = F.softmax_cross_entropy(y_fake.reshape(n, c)[:, 1:], t_fake)
        loss = loss + lr + lf
        gr.reshape(n, c)[:, 1:] += g1
        gf.reshape(n, c)[:, 1:] += g2
    return loss, gr, gf


def loss_gen(model, y_fake_i, y_fake_v, t_fake):
    """model/updater.py:46-63.  Full batch, channel 0.  Returns (loss, gy_fake_i, gy_fake_v)."""
    n = len(y_fake_i)
    loss = F.softplus(-y_fake_i[:, 0]).sum() / n + F.softplus(-y_fake_v[:, 0]).sum() / n
    gi = np.zeros_like(y_fake_i)
    gv = np.zeros_like(y_fake_v)
    gi[:, 0] = -F.sigmoid(-y_fake_i[:, 0]) / n
    gv[:, 0] = -F.sigmoid(-y_fake_v[:, 0]) / n
    if model == 'infogan':
        li, g1 = F.softmax_cross_entropy(y_fake_i[:, 1:, 0, 0], t_fake)
        lv, g2 = F.softmax_cross_entropy(y_fake_v[:, 1:, 0, 0, 0], t_fake)
        loss = loss + li + lv
        gi[:, 1:, 0, 0] += g1
        gv[:, 1:, 0, 0, 0] += g2
    return loss, gi, gv


def concat_label_video(video, label, dim_zl):
    """model/updater.py:65-76 (cgan): append dim_zl planes of -1 with the label's plane +1."""
    n, c, t, h, w = video.shape
    lv = -np.ones((n, dim_zl, t, h, w), dtype=video.dtype)
    lv[np.arange(n), label] = 1.0
    return np.concatenate((video, lv), axis=1)


# ----------------------------------------------------------------------------------------
# Adam + WeightDecay (train.py:93-101; Chainer 3.1 optimizers/adam.py, optimizer.py)
# ----------------------------------------------------------------------------------------
ADAM_ALPHA = 2e-4
ADAM_BETA1 = 5e-5          # train.py:99-101 passes 5e-5 as beta1 (quirk Q2)
ADAM_BETA2 = 0.999         # the beta2 argument of make_optimizer is never forwarded
ADAM_EPS = 1e-8
WEIGHT_DECAY = 1e-5


def new_adam_state(p):
    return {'t': 0,
            'm': {k: np.zeros_like(p[k]) for k in net.trainable_keys(p)},
            'v': {k: np.zeros_like(p[k]) for k in net.trainable_keys(p)}}


def adam_wd_update(p, grads, state, alpha=ADAM_ALPHA, beta1=ADAM_BETA1, beta2=ADAM_BETA2,
                   eps=ADAM_EPS, wd=WEIGHT_DECAY):
    """GradientMethod.update order: hooks (g += wd * p for EVERY param, quirk Q4), t += 1,
    then per parameter  m += (1-b1)(g-m); v += (1-b2)(g*g-v); p -= lr * m / (sqrt(v)+eps)
    with lr = alpha * sqrt(1-b2^t) / (1-b1^t)  (quirk Q3).  In place."""
    state['t'] += 1
    t = state['t']
    lr = alpha * np.sqrt(1.0 - beta2 ** t) / (1.0 - beta1 ** t)
    for k in net.trainable_keys(p):
        dt = p[k].dtype.type
        g = grads[k] + dt(wd) * p[k]
        m, v = state['m'][k], state['v'][k]
        m += dt(1 - beta1) * (g - m)
        v += dt(1 - beta2) * (g * g - v)
        p[k] -= dt(lr) * m / (np.sqrt(v) + dt(eps))


def zero_grads(p):
    return {k: np.zeros_like(p[k]) for k in net.trainable_keys(p)}


# ----------------------------------------------------------------------------------------
# one iteration
# ----------------------------------------------------------------------------------------
def update_core(model, gen, dis_i, dis_v, opt_g, opt_i, opt_v, x_real, t_real, rnd,
                dim_zl=0, video_len=16, keep=False, reduce=None, q1_rows=None, kinks=None):
    """model/updater.py:78-113 with injected randomness.

    gen/dis_i/dis_v: parameter dicts (updated IN PLACE, as Chainer does).
    x_real (N,C,T,H,W); t_real (N,) int or None.
    rnd: dict with 't' (frame index, :96), 'noise_i_real', 'noise_v_real', 'noise_i_fake',
         'noise_v_fake' (lists of 4 pre-scaled addends or None) and 'gen' (net.gen_draw()).
    reduce: optional callable(name, grads_dict) applied in place to each network's gradients just
         before its Adam update -- the hook the data-parallel tests use to average gradients over
         ranks (the reference is single-device; SURVEY 8e defines DP as "mean of the per-shard
         gradients, per-shard BatchNorm statistics").
    kinks (tests only): {'eps': e, 'real_i' | 'real_v' | 'fake_i' | 'fake_v' | 'gen': {layer: boolean array}} -- another
         implementation's ReLU / LeakyReLU decisions, taken over inside the band |pre-activation| < e (net._decide);
         the result then carries 'kink_forced' and 'kink_disagree' counts.
    Returns dict(loss_dis_i, loss_dis_v, loss_gen [, intermediates when keep=True]).

    Ordering quirks reproduced: all four D forwards and the G forward run first with the OLD
    parameters; D_I is updated, then D_V, and only then G's loss is back-propagated through
    the D graphs -- whose W / gamma arrays have been mutated in place while the saved
    activations and BN statistics are the old ones (quirk Q5).  Gradients that Chainer
    computes into the *other* networks and then discards (quirk Q6) are skipped.
    """
    n = x_real.shape[0]
    if model == 'cgan':
        x_real = concat_label_video(x_real, t_real, dim_zl)
    t = rnd['t']

    def kk(name):
        return None if kinks is None else dict(kinks[name], eps=kinks['eps'])
    y_real_i, c_real_i = net.dis_forward(dis_i, x_real[:, :, t], rnd['noise_i_real'], kinks=kk('real_i'))
    y_real_v, c_real_v = net.dis_forward(dis_v, x_real, rnd['noise_v_real'], kinks=kk('real_v'))

    x_fake_tn, t_fake, c_gen = net.gen_forward(gen, rnd['gen'], video_len, kinks=kk('gen'))
    x_fake = x_fake_tn.transpose(1, 2, 0, 3, 4)                   # (T,N,C,H,W) -> (N,C,T,H,W), :102
    if model == 'cgan':
        x_fake = concat_label_video(x_fake, t_fake, dim_zl)
    y_fake_i, c_fake_i = net.dis_forward(dis_i, x_fake[:, :, t], rnd['noise_i_fake'], kinks=kk('fake_i'))
    y_fake_v, c_fake_v = net.dis_forward(dis_v, x_fake, rnd['noise_v_fake'], kinks=kk('fake_v'))

    out = {}
    # image_dis_optimizer.update(self.loss_dis, image_dis, ...)   :111
    l_i, gr, gf = loss_dis(model, False, y_real_i, y_fake_i, t_real, t_fake, q1_rows)
    g_i = zero_grads(dis_i)
    net.dis_backward(dis_i, c_real_i, gr, g_i)
    net.dis_backward(dis_i, c_fake_i, gf, g_i)
    if keep:
        out['grads_dis_i'] = {k: v.copy() for k, v in g_i.items()}
    if reduce is not None:
        reduce('image_dis', g_i)
    adam_wd_update(dis_i, g_i, opt_i)
    # video_dis_optimizer.update(self.loss_dis, video_dis, ...)   :112
    l_v, gr, gf = loss_dis(model, True, y_real_v, y_fake_v, t_real, t_fake, q1_rows)
    g_v = zero_grads(dis_v)
    net.dis_backward(dis_v, c_real_v, gr, g_v)
    net.dis_backward(dis_v, c_fake_v, gf, g_v)
    if keep:
        out['grads_dis_v'] = {k: v.copy() for k, v in g_v.items()}
    if reduce is not None:
        reduce('video_dis', g_v)
    adam_wd_update(dis_v, g_v, opt_v)
    # image_gen_optimizer.update(self.loss_gen, image_gen, ...)   :113
    l_g, gi, gv = loss_gen(model, y_fake_i, y_fake_v, t_fake)
    gx_i = net.dis_backward(dis_i, c_fake_i, gi, None, need_gx=True)      # updated D_I weights (Q5)
    gx_v = net.dis_backward(dis_v, c_fake_v, gv, None, need_gx=True)      # updated D_V weights (Q5)
    c_img = x_fake_tn.shape[2]
    gx = np.array(gx_v[:, :c_img])                                 # cgan label planes get no gradient path to G
    gx[:, :, t] += gx_i[:, :c_img]
    g_g = zero_grads(gen)
    net.gen_backward(gen, c_gen, gx.transpose(2, 0, 1, 3, 4), g_g)
    if keep:
        out['grads_gen'] = {k: v.copy() for k, v in g_g.items()}
        out.update(x_fake=x_fake, y_real_i=y_real_i, y_real_v=y_real_v, y_fake_i=y_fake_i,
                   y_fake_v=y_fake_v, gx_fake=gx)
    if reduce is not None:
        reduce('image_gen', g_g)
    adam_wd_update(gen, g_g, opt_g)
    out.update(loss_dis_i=float(l_i), loss_dis_v=float(l_v), loss_gen=float(l_g), t_fake=t_fake)
    # Distance of the closest pre-activation to the kink of its ReLU / LeakyReLU.  An fp32
    # implementation computes these values with ~1e-7 relative rounding error, so an element closer
    # than that to zero may take the other branch; with a handful of samples per BatchNorm channel a
    # single flipped element moves the gradients behind it by O(1e-2).  Parity tests use this number
    # to know whether a step is well conditioned for a tight comparison.
    out['min_margin'] = min(c['min_margin'] for c in (c_real_i, c_real_v, c_fake_i, c_fake_v, c_gen))
    if kinks is not None:
        out['kink_forced'] = sum(c.get('kink_forced', 0) for c in (c_real_i, c_real_v, c_fake_i, c_fake_v, c_gen))
        out['kink_disagree'] = sum(c.get('kink_disagree', 0) for c in (c_real_i, c_real_v, c_fake_i, c_fake_v, c_gen))
    return out


def draw_step_randomness(rng, model, n, in_channels=3, n_filters=64, dim_zc=50, dim_zm=10, dim_zl=0,
                         video_len=16, sigma=0.2, dtype=np.float32):
    """Draws everything one update_core consumes, in the reference's order:
    t; D_I real noise x4; D_V real noise x4; labels, h0, e_t, zc; D_I fake noise x4;
    D_V fake noise x4.  (Note: the reference draws ``t`` before the real-pass noise.)"""
    c_d = in_channels + (dim_zl if model == 'cgan' else 0)

    def noise(ndim):
        return [(sigma * rng.randn(*s)).astype(dtype)
                for s in net.dis_noise_shapes(ndim, n, c_d, n_filters, video_len)]
    rnd = {'t': int(rng.randint(0, video_len))}
    rnd['noise_i_real'] = noise(2)
    rnd['noise_v_real'] = noise(3)
    rnd['gen'] = net.gen_draw(rng, n, dim_zc, dim_zm, dim_zl, video_len, dtype)
    rnd['noise_i_fake'] = noise(2)
    rnd['noise_v_fake'] = noise(3)
    return rnd
