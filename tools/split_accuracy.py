#!/usr/bin/env python
"""Error of the three forms of a convolution GEMM against a float64 convolution (torch, CPU) on the same fp32 inputs:
fp32 MFMA (precision f32), fp32 values as three bf16 terms on the bf16 MFMA (f32x3), bf16-rounded operands (bf16).
usage: python tools/split_accuracy.py            (prints one line per layer and pass; small batch, the reference's channel widths)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mocogan_chainer_amd.hiplib as hl
import mocogan_chainer_amd.layout as lay


def rel(a, b):
    a = a.detach().cpu().double().numpy()
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def main():
    hl.load()
    rng = np.random.RandomState(1)
    print('%-28s %-6s %12s %12s %12s' % ('layer', 'pass', 'f32 MFMA', 'f32x3', 'bf16'))
    for name, N, Ti, H, Ci, Co, kt in (('D_V dc2 (64->128, 3-D)', 1, 7, 32, 64, 128, 4), ('D_V dc3 (128->256, 3-D)', 2, 6, 16, 128, 256, 4),
                                       ('D_V dc4 (256->512, 3-D)', 4, 5, 8, 256, 512, 4), ('G dc3 (256->128, 2-D)', 8, 1, 16, 128, 256, 1)):
        x = rng.uniform(-1, 1, (N, Ci, Ti, H, H)).astype(np.float32).astype(np.float64)
        W = (rng.randn(Co, Ci, kt, 4, 4) * 0.05).astype(np.float32).astype(np.float64)
        gy = rng.randn(N, Co, Ti - kt + 1, H // 2, H // 2).astype(np.float32).astype(np.float64)
        xt, wt = torch.tensor(x, requires_grad=True), torch.tensor(W, requires_grad=True)           # float64 on the CPU
        yt = torch.nn.functional.conv3d(xt, wt, None, (1, 2, 2), (0, 1, 1))
        yt.backward(torch.tensor(gy))
        y_ref, gx_ref, gW_ref = yt.detach().numpy(), xt.grad.numpy(), wt.grad.numpy()
        dev = lambda a: torch.tensor(a, dtype=torch.float32, device='cuda')
        xd, wd, gyd = lay.act_to_dev(dev(x)), lay.conv_w_to_dev(dev(W)), lay.act_to_dev(dev(gy))
        xs, ws, gys = hl.split_planes(xd), hl.split_planes(wd), hl.split_planes(gyd)
        wsd = hl.split_planes(wd, run=16 * kt * 16 * Ci)
        res = {}
        for prec in ('f32', 'f32x3', 'bf16'):
            g = hl.make_geom(N, Ti, H, H, Ci, Co, kt, precision=prec)
            sp = prec == 'f32x3'
            y = torch.empty((N, g.To, g.Ho, g.Wo, Co), device='cuda')
            hl.conv_fprop(g, xs if sp else xd, ws if sp else wd, None, y)
            gx = torch.empty((N, Ti, H, H, Ci), device='cuda')
            hl.conv_dgrad(g, gys if sp else gyd, wsd if sp else wd, None, gx)
            dw = torch.zeros_like(wd)
            hl.conv_wgrad(g, xs if sp else xd, gys if sp else gyd, dw)
            res[prec] = (rel(lay.act_from_dev(y, Co), y_ref), rel(lay.act_from_dev(gx, Ci), gx_ref), rel(lay.conv_w_from_dev(dw, Ci, 3), gW_ref))
        for i, p in enumerate(('fprop', 'dgrad', 'wgrad')):
            print('%-28s %-6s %12.2e %12.2e %12.2e' % (name, p, res['f32'][i], res['f32x3'][i], res['bf16'][i]))


if __name__ == '__main__':
    main()
