"""Size-independent properties at the FULL sizes of BASELINE configs[1] (batch 32, n_filters 64), where the
float64 oracle is too slow to run: the three conv passes of every layer must be mutually adjoint,
    <fprop(x, W), gy>  ==  <x, dgrad(gy, W)>  ==  <W, wgrad(x, gy)>,
and linear; with the tuned tiles (incl. split-K) and with the bf16 MFMA mode.  A Philox noise tensor of the
largest activation must have the moments of sigma * N(0,1)."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _layers(batch):
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import bench_layers
    return bench_layers.layers(batch)


def _dot(a, b):
    return float(torch.dot(a.reshape(-1).double(), b.reshape(-1).double()))


def _step_layers(batch):
    """the geometries ONE iteration at this per-GPU batch launches: D runs real and fake clips as one call of
    2 * batch (fprop / wgrad / dgrad for its own loss) and batch alone for G's loss; G runs 16 * batch frames"""
    seen, out = set(), []
    for lay_ in _layers(batch) + [l for l in _layers(2 * batch) if l[0].startswith('D_')]:
        if lay_[1:7] not in seen:
            seen.add(lay_[1:7])
            out.append(lay_)
    return out


@pytest.mark.parametrize("precision,tol,batch", [("f32", 2e-5, 32), ("bf16", 2e-2, 32), ("bf16", 2e-2, 256), ("f32x3", 2e-5, 32), ("f32x3", 2e-5, 128)],
                         ids=["f32-b32", "bf16-b32", "bf16-b256-configs2", "f32x3-b32-headline", "f32x3-b128-configs4-share"])
def test_conv_passes_are_adjoint_and_linear_at_full_size(precision, tol, batch):
    """batch 32 = BASELINE configs[1] (fp32 MFMA, and the headline's arithmetic: the same fp32 products formed on the bf16 pipe from
    three-term operands, held to the fp32 tolerance); bf16 at batch 256 = configs[2] (the shipped tile table holds its geometries:
    D at 512 and 256 clips, G at 4096 frames); batch 128 = one GPU's share of configs[4] (global batch 1024 on 8 GPUs: D at 256 and
    128 clips, G at 2048 frames -- the only part of that configuration a one-GPU box can run)."""
    import mocogan_chainer_amd.hiplib as hl
    hl.load()
    hl.set_autotune(True)
    gen = torch.Generator(device='cuda')
    gen.manual_seed(5)
    checked = refused = 0
    try:
        for name, N, T, H, Ci, Co, kt, ci_real in _step_layers(batch):
            split = precision == "f32x3" and Ci % 16 == 0 and Co % 16 == 0     # (the 4-channel layers of an f32x3 network run the fp32 kernels)
            # (ci_valid as the networks pass it: the shipped table's choices for the 4-channel layers -- e.g. the patch-in-LDS weight
            #  gradient at 256 clips -- were made, and are only valid, for the clip's three data channels)
            g = hl.make_geom(N, T, H, H, Ci, Co, kt, precision=precision if split else "f32" if precision == "f32x3" else precision,
                             ci_valid=ci_real)
            if split and not (hl.split_covers('fprop', g) and hl.split_covers('dgrad', g) and hl.split_covers('wgrad', g)):
                continue
            x = torch.randn((N, T, H, H, Ci), device='cuda', generator=gen)
            if Ci == 4:
                x[..., 3] = 0                                       # padded channel
            w = torch.randn((Co, kt, 4, 4, Ci), device='cuda', generator=gen) * 0.05
            if Ci == 4:
                w[..., 3] = 0
            gy = torch.randn((N, g.To, g.Ho, g.Wo, Co), device='cuda', generator=gen)
            y = torch.empty_like(gy)
            sp = (lambda t, **kw: hl.split_planes(t, **kw)) if split else (lambda t, **kw: t)
            wd = sp(w, run=16 * kt * 16 * Ci) if split else w              # the filter as the input gradient reads it
            gx = torch.empty_like(x)
            gw = torch.zeros_like(w)
            try:
                hl.conv_fprop(g, sp(x), sp(w), None, y)
                hl.conv_dgrad(g, sp(gy), wd, None, gx)
                hl.conv_wgrad(g, sp(x), sp(gy), gw)
            except hl.McgError:
                # a geometry whose split form the library refuses in one of the passes (the network then runs that pass on the fp32
                # kernels, nets._c*): nothing to check in the split form -- but most layers must have one
                assert split, (name, precision)
                refused += 1
                continue
            checked += 1
            a, b, c = _dot(y, gy), _dot(x, gx), _dot(w, gw)
            scale = float(torch.linalg.vector_norm(y.double()) * torch.linalg.vector_norm(gy.double()))
            assert abs(a - b) < tol * scale and abs(a - c) < tol * scale, (name, precision, a, b, c, scale)
            # linearity of fprop in x (same W): fprop(2 x1 - 3 x2) == 2 fprop(x1) - 3 fprop(x2)
            x2 = torch.randn_like(x)
            if Ci == 4:
                x2[..., 3] = 0
            y2, y3 = torch.empty_like(y), torch.empty_like(y)
            hl.conv_fprop(g, sp(x2), sp(w), None, y2)
            hl.conv_fprop(g, sp(2 * x - 3 * x2), sp(w), None, y3)
            err = float(torch.linalg.vector_norm((y3 - (2 * y - 3 * y2)).double()) / torch.linalg.vector_norm(y3.double()))
            assert err < (5e-6 if precision in ("f32", "f32x3") else 2e-2), (name, err)
            del x, x2, y, y2, y3, gy, gx, gw, w
        assert checked >= 8 and refused <= checked // 3, (checked, refused)
    finally:
        hl.set_autotune(False)


def test_philox_noise_moments_at_full_size():
    import mocogan_chainer_amd.hiplib as hl
    hl.load()
    n = 32 * 13 * 32 * 32 * 64                                      # D_V's largest noisy activation at batch 32
    out = torch.empty(n, device='cuda')
    hl.randn(out, 0.2, 1234, 77)
    d = out.double()
    m, v = float(d.mean()), float(d.var())
    k = float(((d - m) ** 4).mean() / v ** 2)
    assert abs(m) < 5e-5 and abs(v - 0.04) < 1e-4 and abs(k - 3.0) < 0.01, (m, v, k)
    out2 = torch.empty(n, device='cuda')
    hl.randn(out2, 0.2, 1234, 78)                                   # another stream: uncorrelated
    c = float((d * out2.double()).mean() / 0.04)
    assert abs(c) < 1e-3, c
