"""Independent torch-CPU (float64, autograd) statement of the same networks, used ONLY to
cross-check the NumPy oracle (the reference itself cannot run here).  It shares no
code with ``oracle/``: convolutions/BN/activations come from torch.nn.functional and every
gradient from autograd."""
import numpy as np
import torch
import torch.nn.functional as TF

DT = torch.float64


def tt(a, grad=False):
    t = torch.tensor(np.asarray(a), dtype=DT)
    return t.requires_grad_(grad)


def params_to_torch(p, grad=True):
    out = {}
    for k, v in p.items():
        if k.endswith(('/W', '/b', '/gamma', '/beta')):
            out[k] = tt(v, grad)
        elif k.endswith('/N'):
            continue
        else:
            out[k] = tt(v, False)
    return out


def bn_train(y, p, name):
    dims = [0] + list(range(2, y.dim()))
    return TF.batch_norm(y, None, None, p[name + '/gamma'], p[name + '/beta'], training=True, eps=2e-5)


def dis_forward(p, x, noise):
    """x (N,C,H,W) or (N,C,T,H,W)."""
    three_d = x.dim() == 5
    h = x
    for l in (1, 2, 3, 4):
        if noise is not None:
            h = h + tt(noise[l - 1])
        if three_d:
            h = TF.conv3d(h, p['dc%d/W' % l], p['dc%d/b' % l], stride=(1, 2, 2), padding=(0, 1, 1))
        else:
            h = TF.conv2d(h, p['dc%d/W' % l], p['dc%d/b' % l], stride=2, padding=1)
        if l >= 2:
            h = bn_train(h, p, 'bn%d' % l)
        h = TF.leaky_relu(h, 0.2)
    if three_d:
        return TF.conv3d(h, p['dc5/W'], p['dc5/b'], stride=(1, 3, 3), padding=0)
    return TF.conv2d(h, p['dc5/W'], p['dc5/b'], stride=1, padding=0)


def gru_step(p, h, x):
    def lin(name, v):
        return v @ p['g0/%s/W' % name].T + p['g0/%s/b' % name]
    r = torch.sigmoid(lin('W_r', x) + lin('U_r', h))
    z = torch.sigmoid(lin('W_z', x) + lin('U_z', h))
    hb = torch.tanh(lin('W', x) + lin('U', r * h))
    return (1 - z) * h + z * hb


def gen_forward(p, draw, video_len=16):
    h = tt(draw['h0'])
    n = h.shape[0]
    dim_zm = h.shape[1]
    dim_zl = p['g0/W/W'].shape[1] - dim_zm
    zl = None
    if dim_zl:
        zl = torch.eye(dim_zl, dtype=DT)[torch.as_tensor(draw['labels'])]
    hs = []
    for t in range(video_len):
        e = tt(draw['e'][t])
        if zl is not None:
            e = torch.cat((zl, e), dim=1)
        h = gru_step(p, h, e)
        hs.append(h)
    zm = torch.stack(hs)
    zc = tt(draw['zc']).unsqueeze(0).repeat(video_len, 1, 1)
    x = torch.cat((zc, zm), dim=2).reshape(video_len * n, -1, 1, 1)
    geo = {1: (1, 0), 2: (2, 1), 3: (2, 1), 4: (2, 1), 5: (2, 1)}
    for l in (1, 2, 3, 4, 5):
        s, pd = geo[l]
        x = TF.conv_transpose2d(x, p['dc%d/W' % l], p['dc%d/b' % l], stride=s, padding=pd)
        if l < 5:
            x = torch.relu(bn_train(x, p, 'bn%d' % l))
        else:
            x = torch.tanh(x)
    return x.reshape(video_len, n, x.shape[1], 64, 64)


def loss_dis(model, is_video, y_real, y_fake, t_real, t_fake):
    n = y_fake.shape[0]
    loss = TF.softplus(-y_real[:1]).sum() / n + TF.softplus(y_fake)[:1].sum() / n
    if model == 'infogan' and is_video:
        c = y_real.shape[1]
        loss = loss + TF.cross_entropy(y_real.reshape(n, c)[:, 1:], torch.as_tensor(t_real))
        loss = loss + TF.cross_entropy(y_fake.reshape(n, c)[:, 1:], torch.as_tensor(t_fake))
    return loss


def loss_gen(model, y_fake_i, y_fake_v, t_fake):
    n = y_fake_i.shape[0]
    loss = TF.softplus(-y_fake_i[:, 0]).sum() / n + TF.softplus(-y_fake_v[:, 0]).sum() / n
    if model == 'infogan':
        loss = loss + TF.cross_entropy(y_fake_i[:, 1:, 0, 0], torch.as_tensor(t_fake))
        loss = loss + TF.cross_entropy(y_fake_v[:, 1:, 0, 0, 0], torch.as_tensor(t_fake))
    return loss
