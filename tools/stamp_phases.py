#!/usr/bin/env python
"""Diagnostic: builds the conv library with -DMCG_STAMPS and prints, per D_V layer and pass, how the waves'
K-loop cycles split into  [LDS refill + barrier] / [address math + global load issue] / [LDS reads + MFMA] /
[end barrier].  Shares only -- the stamped build is slower than the real one."""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
so = '/tmp/libmocogan_stamps.so'
subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-fPIC', '-shared', '-std=c++17', '-DMCG_STAMPS',
                '-I' + os.path.join(ROOT, 'include'), os.path.join(ROOT, 'mocogan-chainer_amd/csrc/conv_gemm.hip'),
                os.path.join(ROOT, 'mocogan-chainer_amd/csrc/small_ops.hip'), '-o', so], check=True)
import mocogan_chainer_amd.build as B
B.lib_path = lambda: so
import mocogan_chainer_amd.hiplib as hl
hl.lib_path = lambda: so
import torch
sys.argv = [sys.argv[0]]
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import bench_layers as BL
lib = hl.load()
lib.mcg_debug_stamps.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
buf = (ctypes.c_ulonglong * 8)()
print('%-10s %-6s %8s %8s %8s %8s   cycles/wave' % ('layer', 'pass', 'refill', 'loads', 'compute', 'barrier'))
for name, N, T, H, Ci, Co, kt, ci_real in BL.layers(32):
    if not name.startswith('D_V'):
        continue
    g = hl.make_geom(N, T, H, H, Ci, Co, kt)
    x = torch.randn((N, T, H, H, Ci), device='cuda')
    y = torch.randn((N, g.To, g.Ho, g.Wo, Co), device='cuda')
    w = torch.randn((Co, kt, 4, 4, Ci), device='cuda') * 0.05
    dw = torch.zeros_like(w)
    for p, fn in (('fprop', lambda: hl.conv_fprop(g, x, w, None, y)), ('dgrad', lambda: hl.conv_dgrad(g, y, w, None, x)),
                  ('wgrad', lambda: hl.conv_wgrad(g, x, y, dw))):
        fn(); torch.cuda.synchronize()
        lib.mcg_debug_stamps(buf, 1)
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        lib.mcg_debug_stamps(buf, 1)
        tot = float(sum(buf[i] for i in range(4))) or 1.0
        waves = max(buf[4], 1)
        print('%-10s %-6s %7.1f%% %7.1f%% %7.1f%% %7.1f%%   %10.0f' % ((name, p) + tuple(100.0 * buf[i] / tot for i in range(4)) + (tot / waves,)))
