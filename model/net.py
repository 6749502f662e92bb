"""Reference-facing network classes: same module path, class names, constructor signatures,
attributes and call conventions as raahii/mocogan-chainer ``model/net.py`` (ImageGenerator :17-117,
ImageDiscriminator :119-158, VideoDiscriminator :160-199), running on the gfx950 kernel library
(``mocogan-chainer_amd``).  Arrays are torch tensors where the reference used chainer Variables.

Randomness follows the reference: latent codes and labels are drawn from NumPy's global generator on
the host (model/net.py:55-56,92) and add_noise from ``np.random.randn`` (model/net.py:13), in the
reference's order, so ``np.random.seed(s)`` selects the same draws a Chainer CPU run would consume.
(The training step itself uses the in-kernel Philox path by default: see model/updater.py.)
"""
import numpy as np
import torch

import mocogan_chainer_amd.nets as _nets
import mocogan_chainer_amd.hiplib as _hl
import mocogan_chainer_amd.layout as _lay
from mocogan_chainer_amd.nets import config       # config.train mirrors chainer.config.train


def _default_device():
    return 'cuda' if torch.cuda.is_available() else 'cpu'


def add_noise(x, use_noise, sigma):
    """model/net.py:10-15 -- x + sigma * randn(shape) in train mode (host generator, float64 draw cast
    to the activation dtype as Chainer's constant-add does)."""
    if config.train and use_noise:
        n = (sigma * np.random.randn(*x.shape)).astype(np.float32)
        return x + torch.as_tensor(n, device=x.device)
    return x


class _Link:
    """Minimal stand-in for chainer.Chain: parameter access, device moves, npz serialisation."""

    def to_gpu(self, device=None):
        self.impl.to('cuda' if device is None else 'cuda:%d' % device)
        return self

    def to_cpu(self):
        self.impl.to('cpu')
        return self

    def namedparams(self):
        for k, v in self.impl.export_reference_params().items():
            yield '/' + k, v

    def serialize_dict(self):
        return self.impl.export_reference_params()

    def load_dict(self, d):
        self.impl.load_reference_params(d)


class ImageGenerator(_Link):
    def __init__(self, dim_zc=50, dim_zm=10, dim_zl=0, out_channels=3, n_filters=64, video_len=16, device=None):
        self.dim_zc, self.dim_zm, self.dim_zl = dim_zc, dim_zm, dim_zl
        self.out_channels, self.n_filters, self.video_len = out_channels, n_filters, video_len
        self.n_hidden = dim_zc + dim_zm
        self.use_label = dim_zl != 0
        self.name = self.__class__.__name__
        self.impl = _nets.GenNet(dim_zc, dim_zm, dim_zl, out_channels, n_filters, video_len,
                                 device=device or _default_device())
        self.impl.init_weights(np.random)          # Chainer initialises at construction from np.random

    def make_hidden(self, batchsize, size):
        return np.random.normal(0, 0.33, size=[batchsize, size]).astype(np.float32)

    def to_one_hot(self, zl, xp=np):
        return np.eye(self.dim_zl, dtype=np.float32)[np.asarray(zl)]

    def _draw(self, batchsize, labels):
        d = {'labels': None if labels is None else torch.as_tensor(np.asarray(labels), dtype=torch.int32, device=self.impl.device)}
        h0 = self.make_hidden(batchsize, self.dim_zm)
        e = np.stack([self.make_hidden(batchsize, self.dim_zm) for _ in range(self.video_len)])
        d['h0'] = torch.as_tensor(h0, device=self.impl.device)
        d['e'] = torch.as_tensor(e, device=self.impl.device)
        return d

    def make_zm(self, batchsize, zl, xp=np):
        """(video_len, batchsize, dim_zm) motion codes from the fused GRU kernel.  zl: one-hot rows or None."""
        assert self.use_label == (zl is not None)
        labels = None if zl is None else np.argmax(np.asarray(zl), axis=1)
        d = self._draw(batchsize, labels)
        dev = self.impl.device
        z = torch.empty((self.video_len * batchsize, self.n_hidden), device=dev)
        saved = torch.empty((self.video_len, batchsize, 4 * self.dim_zm), device=dev)
        zc = torch.zeros((batchsize, self.dim_zc), device=dev)
        _hl.gru_seq_fwd(batchsize, self.video_len, self.dim_zm, self.dim_zl, self.dim_zc, self.impl.fp.param('g0'),
                        d['h0'], d['e'], d['labels'], zc, z, saved)
        return z.view(self.video_len, batchsize, self.n_hidden)[:, :, self.dim_zc:].contiguous()

    def __call__(self, batchsize, xp=np):
        """-> (x of shape (video_len, batchsize, channel, 64, 64), labels or None)   model/net.py:83-117"""
        labels = np.random.randint(self.dim_zl, size=batchsize) if self.use_label else None
        d = self._draw(batchsize, labels)
        d['zc'] = torch.as_tensor(self.make_hidden(batchsize, self.dim_zc), device=self.impl.device)
        x, self.last_saved = self.impl.forward(batchsize, d)
        # device layout [N][T][H][W][Cp] -> reference (T,N,C,H,W)
        x = x[..., :self.out_channels].permute(1, 0, 4, 2, 3).contiguous()
        return x, labels


class _Discriminator(_Link):
    NDIM = 2

    def __init__(self, in_channels=3, out_channels=1, n_filters=64, use_noise=False, noise_sigma=0.2, device=None):
        self.in_channels, self.out_channels, self.n_filters = in_channels, out_channels, n_filters
        self.use_noise, self.noise_sigma = use_noise, noise_sigma
        self.name = self.__class__.__name__
        self.impl = _nets.DisNet(self.NDIM, in_channels, out_channels, n_filters, use_noise, noise_sigma,
                                 device=device or _default_device())
        self.impl.init_weights(np.random)

    def __call__(self, x):
        x = torch.as_tensor(x, dtype=torch.float32, device=self.impl.device)
        n = x.shape[0]
        xd = _lay.act_to_dev(x)
        noise = None
        if config.train and self.use_noise:
            T = x.shape[2] if self.NDIM == 3 else 16
            shapes = _noise_shapes(self.NDIM, n, self.in_channels, self.n_filters, T)
            noise = [_lay.act_to_dev(torch.as_tensor((self.noise_sigma * np.random.randn(*s)).astype(np.float32),
                                                     device=self.impl.device)) for s in shapes]

        def first(out, na):
            _hl.bn_act_fwd(out.numel() // out.shape[-1], out.shape[-1], xd, None, _hl.ACT_NONE, out,
                           c_valid=self.in_channels, **na)
        logits, self.last_saved = self.impl.forward(n, first, noise)
        return logits.view((n, self.out_channels) + (1,) * self.NDIM)


def _noise_shapes(ndim, n, c, nf, t):
    if ndim == 2:
        return [(n, c, 64, 64), (n, nf, 32, 32), (n, nf * 2, 16, 16), (n, nf * 4, 8, 8)]
    return [(n, c, t, 64, 64), (n, nf, t - 3, 32, 32), (n, nf * 2, t - 6, 16, 16), (n, nf * 4, t - 9, 8, 8)]


class ImageDiscriminator(_Discriminator):
    """input (batchsize, C, 64, 64) -> (batchsize, out, 1, 1)   model/net.py:143-158"""
    NDIM = 2


class VideoDiscriminator(_Discriminator):
    """input (batchsize, C, 16, 64, 64) -> (batchsize, out, 1, 1, 1)   model/net.py:184-199"""
    NDIM = 3
