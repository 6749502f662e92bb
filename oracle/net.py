"""CPU restatement of the three MoCoGAN networks with hand-written backward passes.

TEST INFRASTRUCTURE -- see ``oracle/__init__.py``.  Follows reference ``model/net.py``:
``ImageGenerator`` (:17-117), ``ImageDiscriminator`` (:119-158), ``VideoDiscriminator``
(:160-199).  Parameters live in plain dicts keyed by Chainer's serialisation names
(``dc1/W``, ``bn2/gamma``, ``g0/W_r/W`` ...).  Randomness is injected: every routine takes
the arrays the reference would have drawn from ``np.random`` so device and oracle see the
same numbers.
"""
import numpy as np

from . import functions as F


# ----------------------------------------------------------------------------------------
# parameter construction
# ----------------------------------------------------------------------------------------
def _bn_params(p, name, c, dtype):
    p[name + '/gamma'] = np.ones(c, dtype)
    p[name + '/beta'] = np.zeros(c, dtype)
    p[name + '/avg_mean'] = np.zeros(c, dtype)
    p[name + '/avg_var'] = np.ones(c, dtype)
    p[name + '/N'] = np.zeros((), np.int64)


TRAINABLE_SUFFIXES = ('/W', '/b', '/gamma', '/beta')


def trainable_keys(p):
    return [k for k in p if k.endswith(TRAINABLE_SUFFIXES)]


def init_generator(rng, dim_zc=50, dim_zm=10, dim_zl=0, out_channels=3, n_filters=64, dtype=np.float32):
    """model/net.py:34-53.  GRU Linear links: LeCunNormal W, zero b (Chainer Linear default)."""
    p = {}
    nin = dim_zm + dim_zl
    for k in F.GRU_KEYS:
        cols = nin if k in ('W_r', 'W_z', 'W') else dim_zm
        p['g0/%s/W' % k] = F.lecun_normal(rng, (dim_zm, cols), dtype)
        p['g0/%s/b' % k] = np.zeros(dim_zm, dtype)
    nf = n_filters
    chans = [dim_zc + dim_zm, nf * 8, nf * 4, nf * 2, nf, out_channels]
    for i in range(5):
        p['dc%d/W' % (i + 1)] = F.glorot_normal(rng, (chans[i], chans[i + 1], 4, 4), dtype)
        p['dc%d/b' % (i + 1)] = np.zeros(chans[i + 1], dtype)
    for i in range(1, 5):
        _bn_params(p, 'bn%d' % i, chans[i], dtype)
    return p


def init_discriminator(rng, ndim, in_channels=3, out_channels=1, n_filters=64, dtype=np.float32):
    """ndim=2: ImageDiscriminator (model/net.py:130-141); ndim=3: VideoDiscriminator (:171-182)."""
    p = {}
    nf = n_filters
    chans = [in_channels, nf, nf * 2, nf * 4, nf * 8, out_channels]
    for i in range(5):
        p['dc%d/W' % (i + 1)] = F.glorot_normal(rng, (chans[i + 1], chans[i]) + (4,) * ndim, dtype)
        p['dc%d/b' % (i + 1)] = np.zeros(chans[i + 1], dtype)
    for i in (2, 3, 4):
        _bn_params(p, 'bn%d' % i, chans[i], dtype)
    return p


# ----------------------------------------------------------------------------------------
# Discriminators
# ----------------------------------------------------------------------------------------
def _dis_geometry(ndim, layer):
    """(stride, pad) of dc<layer>.  model/net.py:133-137 (2-D) and :174-178 (3-D)."""
    if ndim == 2:
        return ((1, 2, 2), (0, 1, 1)) if layer < 5 else ((1, 1, 1), (0, 0, 0))
    return ((1, 2, 2), (0, 1, 1)) if layer < 5 else ((1, 3, 3), (0, 0, 0))


def dis_noise_shapes(ndim, n, in_channels=3, n_filters=64, t=16, size=64):
    """Shapes of the four add_noise draws of one discriminator call (model/net.py:148-154,189-195)."""
    nf = n_filters
    if ndim == 2:
        return [(n, in_channels, size, size), (n, nf, size // 2, size // 2),
                (n, nf * 2, size // 4, size // 4), (n, nf * 4, size // 8, size // 8)]
    return [(n, in_channels, t, size, size), (n, nf, t - 3, size // 2, size // 2),
            (n, nf * 2, t - 6, size // 4, size // 4), (n, nf * 4, t - 9, size // 8, size // 8)]


def _decide(y, strict, kinks, l, cache):
    """The branch an activation takes per element: y >= 0 (leaky_relu) or y > 0 (relu, strict).  kinks (tests only):
    {'eps': e, l: boolean array of another implementation's decisions for layer l} -- inside the band |y| < e, where
    an fp32 implementation may legitimately land on the other side of the kink, THAT implementation's decision is
    taken over (and counted); outside the band a disagreement is counted as an error for the caller to assert on."""
    pos = (y > 0) if strict else (y >= 0)
    if kinks is not None and l in kinks:
        other = np.asarray(kinks[l], bool).reshape(y.shape)
        band = np.abs(y) < kinks['eps']
        differ = pos != other
        cache['kink_forced'] = cache.get('kink_forced', 0) + int(np.count_nonzero(differ & band))
        cache['kink_disagree'] = cache.get('kink_disagree', 0) + int(np.count_nonzero(differ & ~band))
        pos = np.where(band, other, pos)
    return pos


def dis_forward(p, x, noise=None, train=True, update_stats=True, kinks=None):
    """ImageDiscriminator.__call__ (model/net.py:143-158) / VideoDiscriminator.__call__ (:184-199).

    x: (N,C,H,W) or (N,C,T,H,W).  noise: list of 4 pre-scaled addends (or None) for the
    inputs of dc1..dc4.  Returns (y, cache); y has shape (N,out,1,1[,1]).  kinks: see _decide.
    """
    ndim = x.ndim - 2
    h = x[:, :, None] if ndim == 2 else x
    cache = {'ndim': ndim, 'a': {}, 'lrelu': {}, 'bn': {}, 'pos': {}}
    for l in (1, 2, 3, 4):
        add = None
        if train and noise is not None and noise[l - 1] is not None:
            add = noise[l - 1]
            add = add[:, :, None] if add.ndim == 4 else add
        a = F.add_noise(h, add)
        cache['a'][l] = a
        stride, pad = _dis_geometry(ndim, l)
        W = p['dc%d/W' % l]
        W = W[:, :, None] if W.ndim == 4 else W
        y = F.conv3d_fwd(a, W, p['dc%d/b' % l], stride, pad)
        if l >= 2:
            if train:
                y, bnc = F.bn_train_fwd(y, p['bn%d/gamma' % l], p['bn%d/beta' % l],
                                        p['bn%d/avg_mean' % l] if update_stats else None,
                                        p['bn%d/avg_var' % l] if update_stats else None)
                cache['bn'][l] = bnc
            else:
                y = F.bn_test_fwd(y, p['bn%d/gamma' % l], p['bn%d/beta' % l],
                                  p['bn%d/avg_mean' % l], p['bn%d/avg_var' % l])
        cache['min_margin'] = min(cache.get('min_margin', np.inf), float(np.abs(y).min()))
        pos = _decide(y, False, kinks, l, cache)
        h = np.where(pos, y, y * 0.2).astype(y.dtype, copy=False)           # F.leaky_relu_fwd with the decision made explicit
        cache['lrelu'][l] = h
        cache['pos'][l] = pos
    cache['a'][5] = h
    stride, pad = _dis_geometry(ndim, 5)
    W = p['dc5/W']
    W = W[:, :, None] if W.ndim == 4 else W
    y = F.conv3d_fwd(h, W, p['dc5/b'], stride, pad)
    if ndim == 2:
        y = y[:, :, 0]
    return y, cache


def dis_backward(p, cache, gy, grads=None, need_gx=False):
    """Backward of dis_forward.  ``p`` holds the CURRENT parameters (quirk Q5: when G's loss
    is back-propagated through D, W and gamma are the already-updated arrays while the cache
    holds the forward's activations).  Accumulates into ``grads`` (dict of zeros) when given.
    Returns gx (or None)."""
    ndim = cache['ndim']
    g = gy[:, :, None] if ndim == 2 else gy
    for l in (5, 4, 3, 2, 1):
        stride, pad = _dis_geometry(ndim, l)
        Wk = 'dc%d/W' % l
        W = p[Wk]
        W5 = W[:, :, None] if W.ndim == 4 else W
        want_gx = need_gx or l > 1
        gx, gW, gb = F.conv3d_bwd(cache['a'][l], W5, g, stride, pad, need_gx=want_gx)
        if grads is not None:
            grads[Wk] += gW[:, :, 0] if W.ndim == 4 else gW
            grads['dc%d/b' % l] += gb
        if l == 1:
            g = gx
            break
        # gx is the gradient w.r.t. a_l = lrelu_{l-1} + noise  ->  through lrelu, then BN of layer l-1
        g = np.where(cache['pos'][l - 1], gx, gx * 0.2).astype(gx.dtype, copy=False)   # F.leaky_relu_bwd on the forward's decisions
        if l - 1 >= 2:
            g, gg, gbeta = F.bn_train_bwd(cache['bn'][l - 1], p['bn%d/gamma' % (l - 1)], g)
            if grads is not None:
                grads['bn%d/gamma' % (l - 1)] += gg
                grads['bn%d/beta' % (l - 1)] += gbeta
    if g is None:
        return None
    return g[:, :, 0] if ndim == 2 else g


# ----------------------------------------------------------------------------------------
# Generator
# ----------------------------------------------------------------------------------------
GEN_DECONV = {1: (1, 0), 2: (2, 1), 3: (2, 1), 4: (2, 1), 5: (2, 1)}   # (stride, pad), model/net.py:44-48


def gen_draw(rng, batchsize, dim_zc=50, dim_zm=10, dim_zl=0, video_len=16, dtype=np.float32):
    """Draws, in the reference's order (model/net.py:91-92,66,71,102), what one
    ImageGenerator.__call__ consumes: labels, h0, e_0..e_{T-1}, zc."""
    d = {}
    d['labels'] = rng.randint(dim_zl, size=batchsize) if dim_zl else None
    d['h0'] = rng.normal(0, 0.33, size=[batchsize, dim_zm]).astype(dtype)
    d['e'] = np.stack([rng.normal(0, 0.33, size=[batchsize, dim_zm]).astype(dtype) for _ in range(video_len)])
    d['zc'] = rng.normal(0, 0.33, size=[batchsize, dim_zc]).astype(dtype)
    return d


def gen_forward(p, draw, video_len=16, train=True, update_stats=True, kinks=None):
    """ImageGenerator.__call__ (model/net.py:83-117) with make_zm (:61-81).

    Returns (x, labels, cache) with x of shape (T, N, C, 64, 64)."""
    h0, e, zc, labels = draw['h0'], draw['e'], draw['zc'], draw['labels']
    n, dim_zm = h0.shape
    dtype = h0.dtype
    gp = {k[3:]: v for k, v in p.items() if k.startswith('g0/')}
    dim_zl = gp['W/W'].shape[1] - dim_zm
    zl = None
    if dim_zl:
        zl = np.eye(dim_zl, dtype=dtype)[labels]                   # to_one_hot, :58-59
    h, steps, hs = h0, [], []
    for t in range(video_len):
        et = e[t] if zl is None else np.concatenate((zl, e[t]), axis=1)   # F.concat((zl, et)), :74
        h, c = F.gru_step_fwd(gp, h, et)
        steps.append(c)
        hs.append(h)
    zm = np.stack(hs)                                              # (T,N,dim_zm)
    zct = np.tile(zc, (video_len, 1, 1))                           # :103
    z = np.concatenate((zct, zm), axis=2).reshape(video_len * n, -1, 1, 1)   # :106-107
    cache = {'gru': steps, 'n': n, 'T': video_len, 'dim_zc': zc.shape[1], 'in': {}, 'bn': {}, 'act': {}, 'pos': {}}
    x = z
    for l in (1, 2, 3, 4, 5):
        s, pd = GEN_DECONV[l]
        cache['in'][l] = x
        y = F.deconv2d_fwd(x, p['dc%d/W' % l], p['dc%d/b' % l], s, pd)
        if l < 5:
            if train:
                y, bnc = F.bn_train_fwd(y, p['bn%d/gamma' % l], p['bn%d/beta' % l],
                                        p['bn%d/avg_mean' % l] if update_stats else None,
                                        p['bn%d/avg_var' % l] if update_stats else None)
                cache['bn'][l] = bnc
            else:
                y = F.bn_test_fwd(y, p['bn%d/gamma' % l], p['bn%d/beta' % l],
                                  p['bn%d/avg_mean' % l], p['bn%d/avg_var' % l])
            cache['min_margin'] = min(cache.get('min_margin', np.inf), float(np.abs(y).min()))
            cache['pos'][l] = _decide(y, True, kinks, l, cache)
            x = np.where(cache['pos'][l], y, 0).astype(y.dtype, copy=False)    # F.relu_fwd with the decision made explicit
        else:
            x = np.tanh(y)
        cache['act'][l] = x
    x = x.reshape(video_len, n, x.shape[1], 64, 64)                # :115
    return x, labels, cache


def gen_backward(p, cache, gx, grads):
    """Backward of gen_forward w.r.t. all generator parameters.  gx: (T,N,C,64,64)."""
    T, n = cache['T'], cache['n']
    g = gx.reshape((T * n,) + gx.shape[2:])
    for l in (5, 4, 3, 2, 1):
        s, pd = GEN_DECONV[l]
        if l == 5:
            g = F.tanh_bwd(cache['act'][5], g)
        else:
            g = np.where(cache['pos'][l], g, 0).astype(g.dtype, copy=False)    # F.relu_bwd on the forward's decisions
            g, gg, gbeta = F.bn_train_bwd(cache['bn'][l], p['bn%d/gamma' % l], g)
            grads['bn%d/gamma' % l] += gg
            grads['bn%d/beta' % l] += gbeta
        g, gW, gb = F.deconv2d_bwd(cache['in'][l], p['dc%d/W' % l], g, s, pd, need_gx=True)
        grads['dc%d/W' % l] += gW
        grads['dc%d/b' % l] += gb
    gz = g.reshape(T, n, -1)
    gzm = gz[:, :, cache['dim_zc']:]
    gp = {k[3:]: v for k, v in p.items() if k.startswith('g0/')}
    ggru = {k[3:]: grads[k] for k in grads if k.startswith('g0/')}
    gh = np.zeros_like(gzm[0])
    for t in reversed(range(T)):
        gh = gh + gzm[t]
        gh, _ = F.gru_step_bwd(gp, cache['gru'][t], gh, ggru)
    return grads
