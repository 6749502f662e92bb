"""Reference (Chainer) layouts <-> device layouts.  Pure tensor reshuffling (torch, any device);
used at the API boundary, by checkpoint IO and by the tests -- never inside the timed step.

Device layouts (include/mocogan_hip.h): activations [N][T][H][W][Cp] (Cp = C rounded up to a
multiple of 4, padded channels zero); every conv/deconv weight [Co][kt][kh][kw][Cip].
"""
import torch
import torch.nn.functional as TF


def pad4(c):
    return (c + 3) // 4 * 4


def act_to_dev(x):
    """(N,C,H,W) or (N,C,T,H,W) -> [N][T][H][W][Cp] (T = 1 for 4-D input)."""
    if x.dim() == 4:
        x = x.unsqueeze(2)
    n, c = x.shape[:2]
    x = x.permute(0, 2, 3, 4, 1)
    return TF.pad(x, (0, pad4(c) - c)).contiguous()


def act_from_dev(x, c, ndim=3):
    """[N][T][H][W][Cp] -> (N,C,T,H,W) (ndim=3) or (N,C,H,W) (ndim=2, T must be 1)."""
    x = x[..., :c].permute(0, 4, 1, 2, 3).contiguous()
    return x[:, :, 0] if ndim == 2 else x


def conv_w_to_dev(w):
    """Chainer Convolution weight (Co,Ci,[kt,]kh,kw) -> [Co][kt][kh][kw][Cip]."""
    if w.dim() == 4:
        w = w.unsqueeze(2)
    ci = w.shape[1]
    return TF.pad(w.permute(0, 2, 3, 4, 1), (0, pad4(ci) - ci)).contiguous()


def conv_w_from_dev(w, ci, ndim):
    w = w[..., :ci].permute(0, 4, 1, 2, 3).contiguous()
    return w[:, :, 0] if ndim == 2 else w


def deconv_w_to_dev(w):
    """Chainer Deconvolution weight (Cin,Cout,kh,kw) -> [Co=Cin][1][kh][kw][Cip=pad4(Cout)]: a
    deconvolution is the data-gradient of the convolution whose weight tensor is the same array."""
    return conv_w_to_dev(w)


def deconv_w_from_dev(w, cout):
    return conv_w_from_dev(w, cout, 2)


def vec_to_dev(v):
    """per-channel vector -> padded to a multiple of 4"""
    return TF.pad(v, (0, pad4(v.shape[0]) - v.shape[0])).contiguous()


GRU_LINKS = ('W_r', 'U_r', 'W_z', 'U_z', 'W', 'U')


def gru_to_dev(params, prefix='g0/'):
    """dict with Chainer keys g0/<link>/W, g0/<link>/b -> flat [W_r|b|U_r|b|W_z|b|U_z|b|W|b|U|b]."""
    parts = []
    for k in GRU_LINKS:
        parts += [params[prefix + k + '/W'].reshape(-1), params[prefix + k + '/b'].reshape(-1)]
    return torch.cat(parts).contiguous()


def gru_from_dev(flat, dim_zm, dim_zl, prefix='g0/'):
    out, p = {}, 0
    for k in GRU_LINKS:
        cols = dim_zm if k.startswith('U') else dim_zm + dim_zl
        out[prefix + k + '/W'] = flat[p:p + dim_zm * cols].reshape(dim_zm, cols)
        p += dim_zm * cols
        out[prefix + k + '/b'] = flat[p:p + dim_zm]
        p += dim_zm
    return out
