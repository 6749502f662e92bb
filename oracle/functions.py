"""Op-level CPU restatement (NumPy) of the Chainer 3.1.0 functions the hot path calls.

TEST INFRASTRUCTURE -- see ``oracle/__init__.py``.  Layouts are the reference's:
activations (N, C, [T,] H, W); conv weights (Cout, Cin, [kt,] kh, kw); deconv weights
(Cin, Cout, kh, kw).  2-D ops are the 3-D ops with a unit T axis.
"""
import numpy as np


# ----------------------------------------------------------------------------------------
# helpers
# ----------------------------------------------------------------------------------------
def _triple(v):
    if isinstance(v, (tuple, list)):
        assert len(v) == 3
        return tuple(int(a) for a in v)
    return (int(v),) * 3


def conv_out_size(size, k, s, p):
    """Chainer ``get_conv_outsize`` (cover_all=False): floor((size + 2p - k) / s) + 1."""
    return (size + 2 * p - k) // s + 1


def deconv_out_size(size, k, s, p):
    """Chainer ``get_deconv_outsize``: s * (size - 1) + k - 2p."""
    return s * (size - 1) + k - 2 * p


# ----------------------------------------------------------------------------------------
# im2col / col2im  (Chainer ``im2col_nd_cpu`` / ``col2im_nd_cpu``)
# ----------------------------------------------------------------------------------------
def im2col_3d(x, ksize, stride, pad):
    """x (N,C,T,H,W) -> col (N,C,kt,kh,kw,To,Ho,Wo), zero padded."""
    kt, kh, kw = ksize
    st, sh, sw = stride
    pt, ph, pw = pad
    n, c, t, h, w = x.shape
    to, ho, wo = conv_out_size(t, kt, st, pt), conv_out_size(h, kh, sh, ph), conv_out_size(w, kw, sw, pw)
    xp = np.pad(x, ((0, 0), (0, 0), (pt, pt), (ph, ph), (pw, pw)))
    col = np.empty((n, c, kt, kh, kw, to, ho, wo), dtype=x.dtype)
    for a in range(kt):
        for b in range(kh):
            for d in range(kw):
                col[:, :, a, b, d] = xp[:, :, a:a + st * to:st, b:b + sh * ho:sh, d:d + sw * wo:sw]
    return col


def col2im_3d(col, stride, pad, dims):
    """col (N,C,kt,kh,kw,To,Ho,Wo) -> x (N,C,T,H,W): scatter-add, the adjoint of im2col_3d."""
    n, c, kt, kh, kw, to, ho, wo = col.shape
    st, sh, sw = stride
    pt, ph, pw = pad
    t, h, w = dims
    xp = np.zeros((n, c, t + 2 * pt + st - 1, h + 2 * ph + sh - 1, w + 2 * pw + sw - 1), dtype=col.dtype)
    for a in range(kt):
        for b in range(kh):
            for d in range(kw):
                xp[:, :, a:a + st * to:st, b:b + sh * ho:sh, d:d + sw * wo:sw] += col[:, :, a, b, d]
    return xp[:, :, pt:pt + t, ph:ph + h, pw:pw + w]


# ----------------------------------------------------------------------------------------
# Convolution (reference: L.ConvolutionND / L.Convolution2D, model/net.py:133-137,174-178)
# Chainer 3.1 CPU: y = tensordot(im2col(x), W) + b ; backward via tensordot + col2im.
# ----------------------------------------------------------------------------------------
def conv3d_fwd(x, W, b, stride, pad):
    stride, pad = _triple(stride), _triple(pad)
    col = im2col_3d(x, W.shape[2:], stride, pad)
    y = np.tensordot(col, W, ((1, 2, 3, 4), (1, 2, 3, 4)))        # (N,To,Ho,Wo,Co)
    y = np.moveaxis(y, 4, 1)
    if b is not None:
        y = y + b.reshape(1, -1, 1, 1, 1)
    return np.ascontiguousarray(y.astype(x.dtype, copy=False))


def conv3d_bwd(x, W, gy, stride, pad, need_gx=True):
    """Returns (gx, gW, gb)."""
    stride, pad = _triple(stride), _triple(pad)
    col = im2col_3d(x, W.shape[2:], stride, pad)
    gW = np.tensordot(gy, col, ((0, 2, 3, 4), (0, 5, 6, 7))).astype(W.dtype, copy=False)
    gb = gy.sum(axis=(0, 2, 3, 4))
    gx = None
    if need_gx:
        gcol = np.tensordot(W, gy, (0, 1))                         # (Ci,kt,kh,kw,N,To,Ho,Wo)
        gcol = np.moveaxis(gcol, 4, 0)                             # (N,Ci,kt,kh,kw,To,Ho,Wo)
        gx = col2im_3d(gcol, stride, pad, x.shape[2:])
    return gx, gW, gb


def conv2d_fwd(x, W, b, stride, pad):
    y = conv3d_fwd(x[:, :, None], W[:, :, None], b, (1, stride, stride), (0, pad, pad))
    return y[:, :, 0]


def conv2d_bwd(x, W, gy, stride, pad, need_gx=True):
    gx, gW, gb = conv3d_bwd(x[:, :, None], W[:, :, None], gy[:, :, None],
                            (1, stride, stride), (0, pad, pad), need_gx)
    return (None if gx is None else gx[:, :, 0]), gW[:, :, 0], gb


# ----------------------------------------------------------------------------------------
# Transposed convolution (reference: L.DeconvolutionND(2, ...), model/net.py:44-48)
# Chainer 3.1 CPU: gcol = tensordot(W, x); y = col2im(gcol) + b, outsize s(i-1)+k-2p.
# ----------------------------------------------------------------------------------------
def deconv2d_fwd(x, W, b, stride, pad):
    n, ci, hi, wi = x.shape
    _, co, kh, kw = W.shape
    ho, wo = deconv_out_size(hi, kh, stride, pad), deconv_out_size(wi, kw, stride, pad)
    gcol = np.tensordot(W, x, (0, 1))                              # (Co,kh,kw,N,Hi,Wi)
    gcol = np.moveaxis(gcol, 3, 0)[:, :, None, :, :, None]         # (N,Co,1,kh,kw,1,Hi,Wi)
    y = col2im_3d(gcol, (1, stride, stride), (0, pad, pad), (1, ho, wo))[:, :, 0]
    if b is not None:
        y = y + b.reshape(1, -1, 1, 1)
    return np.ascontiguousarray(y.astype(x.dtype, copy=False))


def deconv2d_bwd(x, W, gy, stride, pad, need_gx=True):
    """Returns (gx, gW, gb).  gx = conv2d(gy, W);  gW[ci,co,kh,kw] = sum x[ci] * im2col(gy)[co,kh,kw]."""
    _, co, kh, kw = W.shape
    col = im2col_3d(gy[:, :, None], (1, kh, kw), (1, stride, stride), (0, pad, pad))  # (N,Co,1,kh,kw,1,Hi,Wi)
    col = col[:, :, 0, :, :, 0]                                    # (N,Co,kh,kw,Hi,Wi)
    gW = np.tensordot(x, col, ((0, 2, 3), (0, 4, 5))).astype(W.dtype, copy=False)     # (Ci,Co,kh,kw)
    gb = gy.sum(axis=(0, 2, 3))
    gx = None
    if need_gx:
        gx = np.tensordot(col, W, ((1, 2, 3), (1, 2, 3)))          # (N,Hi,Wi,Ci)
        gx = np.ascontiguousarray(np.moveaxis(gx, 3, 1))
    return gx, gW, gb


# ----------------------------------------------------------------------------------------
# BatchNormalization, train mode (reference: L.BatchNormalization, model/net.py:50-53,
# 139-141,180-182; Chainer 3.1 defaults decay=0.9, eps=2e-5)
# ----------------------------------------------------------------------------------------
BN_EPS = 2e-5
BN_DECAY = 0.9


def _bn_axes(x):
    return (0,) + tuple(range(2, x.ndim))


def _ex(v, x):
    return v.reshape((1, -1) + (1,) * (x.ndim - 2))


def bn_train_fwd(x, gamma, beta, avg_mean=None, avg_var=None, eps=BN_EPS, decay=BN_DECAY):
    """Returns (y, cache).  Updates avg_mean/avg_var in place when given.

    Chainer 3.1 ``BatchNormalizationFunction.forward``: biased variance, ``var += eps``
    BEFORE both the normalisation and the running-variance update, unbiasing factor
    m / max(m - 1, 1).
    """
    axes = _bn_axes(x)
    mean = x.mean(axis=axes)
    var = x.var(axis=axes) + eps
    inv_std = 1.0 / np.sqrt(var)
    x_hat = (x - _ex(mean, x)) * _ex(inv_std, x)
    y = _ex(gamma, x) * x_hat + _ex(beta, x)
    if avg_mean is not None:
        m = x.size // gamma.size
        adjust = m / max(m - 1.0, 1.0)
        avg_mean *= decay
        avg_mean += (1 - decay) * mean
        avg_var *= decay
        avg_var += (1 - decay) * adjust * var
    cache = dict(x_hat=x_hat.astype(x.dtype, copy=False), inv_std=inv_std.astype(x.dtype, copy=False),
                 mean=mean.astype(x.dtype, copy=False))
    return y.astype(x.dtype, copy=False), cache


def bn_train_bwd(cache, gamma, gy):
    """Returns (gx, ggamma, gbeta).  ``gamma`` is the CURRENT parameter array (quirk Q5):
    Chainer's backward reads the retained input ndarray, which Adam may have mutated."""
    x_hat, inv_std = cache['x_hat'], cache['inv_std']
    axes = _bn_axes(gy)
    m = gy.size // gamma.size
    gbeta = gy.sum(axis=axes)
    ggamma = (gy * x_hat).sum(axis=axes)
    gx = _ex(gamma * inv_std, gy) * (gy - (x_hat * _ex(ggamma, gy) + _ex(gbeta, gy)) / m)
    return gx.astype(gy.dtype, copy=False), ggamma, gbeta


def bn_test_fwd(x, gamma, beta, avg_mean, avg_var, eps=BN_EPS):
    """Chainer ``fixed_batch_normalization`` (used only by util.py:92 in the reference)."""
    inv_std = 1.0 / np.sqrt(avg_var + eps)
    return (_ex(gamma * inv_std, x) * (x - _ex(avg_mean, x)) + _ex(beta, x)).astype(x.dtype, copy=False)


# ----------------------------------------------------------------------------------------
# Activations (model/net.py:110-114,149-155,190-196)
# ----------------------------------------------------------------------------------------
def leaky_relu_fwd(x, slope=0.2):
    return np.where(x >= 0, x, x * slope).astype(x.dtype, copy=False)


def leaky_relu_bwd(y, gy, slope=0.2):
    """Chainer 3.1 LeakyReLU.backward masks on the retained OUTPUT: gx = gy * (slope where y < 0)."""
    return np.where(y >= 0, gy, gy * slope).astype(gy.dtype, copy=False)


def relu_fwd(x):
    return np.maximum(x, 0)


def relu_bwd(y, gy):
    return np.where(y > 0, gy, 0).astype(gy.dtype, copy=False)


def tanh_bwd(y, gy):
    return gy * (1 - y * y)


def sigmoid(x):
    return 0.5 * np.tanh(0.5 * x) + 0.5            # Chainer's sigmoid: tanh(x * 0.5) * 0.5 + 0.5


def softplus(x):
    """Chainer F.softplus(beta=1): max(x, 0) + log1p(exp(-|x|))."""
    return np.maximum(x, 0) + np.log1p(np.exp(-np.abs(x)))


# ----------------------------------------------------------------------------------------
# add_noise (model/net.py:10-15).  ``addend`` is the already scaled fp32 tensor
# sigma * randn(shape) that Chainer adds (float64 draw, cast to the activation dtype).
# ----------------------------------------------------------------------------------------
def add_noise(x, addend):
    return x if addend is None else (x + addend.astype(x.dtype, copy=False))


# ----------------------------------------------------------------------------------------
# StatelessGRU (model/net.py:39-41,76; Chainer 3.1 links/connection/gru.py)
#   r = sigmoid(W_r x + U_r h); z = sigmoid(W_z x + U_z h)
#   h_bar = tanh(W x + U (r * h));  h' = (1 - z) * h + z * h_bar
# All six Linear links carry a bias.
# ----------------------------------------------------------------------------------------
GRU_KEYS = ('W_r', 'U_r', 'W_z', 'U_z', 'W', 'U')


def _lin(p, name, v):
    return v @ p[name + '/W'].T + p[name + '/b']


def gru_step_fwd(p, h, x):
    r = sigmoid(_lin(p, 'W_r', x) + _lin(p, 'U_r', h))
    z = sigmoid(_lin(p, 'W_z', x) + _lin(p, 'U_z', h))
    rh = r * h
    h_bar = np.tanh(_lin(p, 'W', x) + _lin(p, 'U', rh))
    h_new = (1 - z) * h + z * h_bar
    return h_new, dict(h=h, x=x, r=r, z=z, rh=rh, h_bar=h_bar)


def gru_step_bwd(p, c, gh_new, grads):
    """Accumulates parameter grads into ``grads`` (same keys as p); returns (gh, gx)."""
    h, x, r, z, rh, h_bar = c['h'], c['x'], c['r'], c['z'], c['rh'], c['h_bar']
    gh = gh_new * (1 - z)
    gz = gh_new * (h_bar - h)
    gh_bar = gh_new * z
    ga = gh_bar * (1 - h_bar * h_bar)                 # pre-tanh
    gaz = gz * z * (1 - z)                             # pre-sigmoid z
    grh = ga @ p['U/W']
    gh = gh + grh * r
    gr = grh * h
    gar = gr * r * (1 - r)                             # pre-sigmoid r
    gx = ga @ p['W/W'] + gaz @ p['W_z/W'] + gar @ p['W_r/W']
    gh = gh + gaz @ p['U_z/W'] + gar @ p['U_r/W']
    for name, gpre, inp in (('W', ga, x), ('U', ga, rh), ('W_z', gaz, x), ('U_z', gaz, h),
                            ('W_r', gar, x), ('U_r', gar, h)):
        grads[name + '/W'] += gpre.T @ inp
        grads[name + '/b'] += gpre.sum(axis=0)
    return gh, gx


# ----------------------------------------------------------------------------------------
# softmax cross entropy (model/updater.py:36-37,55-56; Chainer default normalize=True, mean)
# ----------------------------------------------------------------------------------------
def softmax_cross_entropy(x, t):
    """x (N,K), t (N,) int.  Returns (loss, gx)."""
    n = x.shape[0]
    xm = x - x.max(axis=1, keepdims=True)
    logp = xm - np.log(np.exp(xm).sum(axis=1, keepdims=True))
    loss = -logp[np.arange(n), t].mean()
    gx = np.exp(logp)
    gx[np.arange(n), t] -= 1
    return loss, gx / n


# ----------------------------------------------------------------------------------------
# Initialisers (model/net.py:35,131,172: GlorotNormal for conv/deconv; Chainer Linear default
# LeCunNormal for the GRU's six Linear links; biases zero; BN gamma=1 beta=0)
# ----------------------------------------------------------------------------------------
def _fans(shape):
    """Chainer ``initializer.get_fans``: fan_in = prod(shape[1:]), fan_out = shape[0] * prod(shape[2:])."""
    rf = int(np.prod(shape[2:])) if len(shape) > 2 else 1
    return shape[1] * rf, shape[0] * rf


def glorot_normal(rng, shape, dtype=np.float32):
    fan_in, fan_out = _fans(shape)
    return rng.normal(0.0, np.sqrt(2.0 / (fan_in + fan_out)), size=shape).astype(dtype)


def lecun_normal(rng, shape, dtype=np.float32):
    fan_in, _ = _fans(shape)
    return rng.normal(0.0, np.sqrt(1.0 / fan_in), size=shape).astype(dtype)
