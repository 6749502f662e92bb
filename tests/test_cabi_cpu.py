"""CPU-only checks of the C-ABI boundary and the host logic around it: the shared library builds and
loads here (hipcc cross-compiles without a GPU), exports every symbol include/mocogan_hip.h declares,
validates its arguments before touching the device, and the layout / parameter plumbing round-trips.
No compute entry point is called with real work."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def hl():
    import mocogan_chainer_amd as pkg
    pkg.build()
    import mocogan_chainer_amd.hiplib as hiplib
    hiplib.load()
    return hiplib


def declared_symbols():
    src = open(os.path.join(ROOT, 'include', 'mocogan_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(mcg_[a-z0-9_]+)\s*\(', src)))


def test_library_exports_every_declared_symbol(hl):
    names = declared_symbols()
    assert len(names) >= 20
    lib = hl.load()
    for n in names:
        assert hasattr(lib, n), "libmocogan_hip.so does not export %s" % n
        assert n in hl.SIGNATURES, "hiplib.SIGNATURES has no ctypes prototype for %s" % n
    assert set(hl.SIGNATURES) == set(names)
    assert lib.mcg_version() == hl.ABI_VERSION
    assert 'define MCG_ABI_VERSION %d' % hl.ABI_VERSION in open(os.path.join(ROOT, 'include', 'mocogan_hip.h')).read()


def test_library_exports_nothing_the_header_does_not_declare(hl):
    """the dynamic symbol table holds exactly the declared mcg_* entry points (+ the diagnostic builds' mcg_debug_stamps, absent from
    the shipped build): helpers shared between the library's translation units have hidden visibility (round 5's review)."""
    import subprocess
    import mocogan_chainer_amd as build
    tool = '/opt/rocm/lib/llvm/bin/llvm-readelf'
    if not os.path.exists(tool):
        pytest.skip('llvm-readelf not found')
    out = subprocess.run([tool, '--dyn-syms', '-W', build.lib_path()], capture_output=True, text=True, check=True).stdout
    out = '\n'.join(l for l in out.splitlines() if ' UND ' not in l)                 # defined symbols only
    exported = sorted(set(re.findall(r'\b(mcg_[A-Za-z0-9_]+)\b', out)))
    assert exported == declared_symbols(), sorted(set(exported) ^ set(declared_symbols()))


def test_header_cites_reference_call_sites():
    src = open(os.path.join(ROOT, 'include', 'mocogan_hip.h')).read()
    for cite in ('model/net.py', 'model/updater.py', 'train.py:93-101'):
        assert cite in src


def test_argument_validation_happens_on_the_host(hl):
    lib = hl.load()
    g = hl.make_geom(1, 4, 12, 12, 4, 64, 4)                 # Ho = 6: not a power of two
    assert lib.mcg_conv_fprop(ctypes.byref(g), None, None, None, None, None) == -1
    g = hl.make_geom(1, 4, 16, 16, 3, 64, 4)                 # unpadded channels
    assert lib.mcg_conv_dgrad(ctypes.byref(g), None, None, None, None, 0, 0, None) == -1
    g = hl.make_geom(1, 2, 16, 16, 4, 64, 4)                 # To = Ti - kt + 1 <= 0
    assert lib.mcg_conv_wgrad(ctypes.byref(g), None, None, None, None) == -2
    g = hl.make_geom(1, 4, 16, 16, 4, 64, 4)                 # good geometry, null pointers
    assert lib.mcg_conv_fprop(ctypes.byref(g), None, None, None, None, None) == -1
    assert lib.mcg_fc_fprop(4, 30, 1, None, None, None, None, None) == -1
    assert lib.mcg_gru_seq_fwd(4, 16, 64, 0, 50, None, None, None, None, None, None, None, None) == -1
    assert lib.mcg_adam_wd(0, None, None, None, None, 0.0, 0.0, 0.0, 0.0, 0.0, 1.0, None, None) == -1
    assert int(lib.mcg_bn_workspace_bytes(0, 512)) == (512 * 2 * 512 + 3 * 512) * 4


def test_product_path_refuses_host_tensors(hl):
    g = hl.make_geom(1, 4, 16, 16, 4, 64, 4)
    x = torch.zeros(16)
    with pytest.raises(hl.McgError):
        hl.conv_fprop(g, x, x, None, x)
    with pytest.raises(hl.McgError):
        hl.adam_wd(x, x, x, x, 1e-3, 0.9, 0.999, 1e-8, 0.0)


def test_layout_round_trips():
    import mocogan_chainer_amd.layout as lay
    rng = np.random.RandomState(0)
    x = torch.tensor(rng.randn(2, 3, 5, 8, 8), dtype=torch.float32)
    xd = lay.act_to_dev(x)
    assert xd.shape == (2, 5, 8, 8, 4) and float(xd[..., 3].abs().max()) == 0
    assert torch.equal(lay.act_from_dev(xd, 3), x)
    x2 = torch.tensor(rng.randn(2, 3, 8, 8), dtype=torch.float32)
    assert torch.equal(lay.act_from_dev(lay.act_to_dev(x2), 3, 2), x2)
    w = torch.tensor(rng.randn(8, 3, 4, 4, 4), dtype=torch.float32)
    wd = lay.conv_w_to_dev(w)
    assert wd.shape == (8, 4, 4, 4, 4)
    assert torch.equal(lay.conv_w_from_dev(wd, 3, 3), w)
    # w_dev[co][kt][kh][kw][ci] == W[co][ci][kt][kh][kw]
    assert float(wd[5, 1, 2, 3, 2]) == float(w[5, 2, 1, 2, 3])
    dw = torch.tensor(rng.randn(60, 512, 4, 4), dtype=torch.float32)
    dd = lay.deconv_w_to_dev(dw)
    assert dd.shape == (60, 1, 4, 4, 512) and torch.equal(lay.deconv_w_from_dev(dd, 512), dw)
    params = {'g0/%s/%s' % (k, s): torch.tensor(rng.randn(*((10, 16 if k[0] == 'W' else 10) if s == 'W' else (10,))), dtype=torch.float32)
              for k in lay.GRU_LINKS for s in ('W', 'b')}
    flat = lay.gru_to_dev(params)
    assert flat.numel() == 3 * (160 + 10) + 3 * (100 + 10)
    back = lay.gru_from_dev(flat, 10, 6)
    for k in params:
        assert torch.equal(back[k], params[k])


def test_adam_hyper_matches_reference_settings():
    """train.py:93-101: Adam(alpha=2e-4, beta1=5e-5) (beta2 never forwarded) + WeightDecay(1e-5)."""
    import mocogan_chainer_amd.step as step
    from oracle import updater as oupd
    h = step.AdamHyper()
    assert (h.alpha, h.beta1, h.beta2, h.eps, h.weight_decay) == (2e-4, 5e-5, 0.999, 1e-8, 1e-5)
    for t in (1, 2, 10):
        assert np.isclose(h.lr(t), oupd.ADAM_ALPHA * np.sqrt(1 - oupd.ADAM_BETA2 ** t) / (1 - oupd.ADAM_BETA1 ** t), rtol=1e-15)
    with pytest.raises(ValueError):
        step.make_models('cgan', num_labels=0, device='cpu')


def test_missing_library_fails_loudly():
    """No CPU fallback: with the shared library absent the binding raises instead of computing something else."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r)\n"
            "import mocogan_chainer_amd.hiplib as hl\n"
            "try:\n    hl.load()\nexcept hl.McgError as e:\n    print('RAISED', 'no CPU fallback' in str(e))\n" % ROOT)
    env = dict(os.environ, MCG_LIB_PATH='/nonexistent/libmocogan_hip.so')
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, env=env, timeout=300)
    assert 'RAISED True' in r.stdout, r.stdout + r.stderr
