"""Permanent watch on round 3's unexplained failure (driver box only, never reproduced): the weight gradient of the 'bf16' launch
(fp32 tensors, rounded in the kernel) and of the 'bf16s' launch (bf16 tensors) at N2 T4 H8 Ci64 Co160 kt4 differed by 1.6e-4 in ONE
of 655 360 elements.  Here the launch pair is repeated 200 times per tile code with the allocator's neighbourhood changed every
repetition (throw-away tensors of varying size between the operands, side-stream traffic while the launches run), and EVERY result
is compared with the float64 oracle -- so a recurrence says which launch left the oracle, at which element, in which block tile of
which repetition, instead of "two launches disagree".  Collected last (tests/conftest.py)."""
import numpy as np
import pytest
import torch

from oracle import functions as F

pytestmark = pytest.mark.gpu

CASE = (2, 4, 8, 64, 160, 4)          # N, Ti, H, Ci, Co, kt -- the red test's geometry
REPS = 200


@pytest.fixture(scope="module")
def hl():
    import mocogan_chainer_amd.hiplib as hiplib
    hiplib.load()
    return hiplib


def _bf16_round(a):
    return torch.tensor(np.asarray(a, np.float32)).to(torch.bfloat16).double().numpy()


def _where(idx, Ci, kt):
    """element (co, a, kh, kw, ci) of dw[Co][kt][4][4][Ci] -> the GEMM coordinates a weight-gradient kernel computes it at:
    row = co, column = tap * Ci + ci; the 64x64 / 128x128 block tile and the 32x32 accumulator tile inside it"""
    co, a, kh, kw, ci = (int(v) for v in idx)
    col = ((a * 4 + kh) * 4 + kw) * Ci + ci
    return "dw[co %d][tap (%d,%d,%d)][ci %d] = GEMM (row %d, col %d): 64x64 tile (%d,%d), 128x128 tile (%d,%d), 32x32 block (%d,%d) lane col %d" % (
        co, a, kh, kw, ci, co, col, co // 64, col // 64, co // 128, col // 128, (co % 64) // 32, (col % 64) // 32, col % 32)


@pytest.mark.parametrize("tile", [0, 3])
def test_round3_weight_gradient_pair_repeated_against_the_oracle(hl, tile):
    import mocogan_chainer_amd.layout as lay
    N, Ti, H, Ci, Co, kt = CASE
    rng = np.random.RandomState(hash(CASE) % 2**31 + 5)            # the red test's inputs
    x, W = _bf16_round(rng.uniform(-1, 1, (N, Ci, Ti, H, H))), _bf16_round(rng.randn(Co, Ci, kt, 4, 4) * 0.1)
    rng.randn(Co)
    gy = _bf16_round(rng.randn(N, Co, Ti - kt + 1, H // 2, H // 2))
    _, gW_ref, _ = F.conv3d_bwd(x, W, gy, (1, 2, 2), (0, 1, 1))
    dev = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32, device="cuda")
    ref = lay.conv_w_to_dev(dev(gW_ref)).double()                  # [Co][kt][4][4][Ci] on the device, float64
    nref = float(ref.norm())
    side = torch.cuda.Stream()
    noise_src = torch.randn(1 << 20, device="cuda")
    worst = 0.0
    for rep in range(REPS):
        pads = [torch.empty(((rep * 7919 + k * 104729) % 4099 + 1) * 64, device="cuda") for k in range(3)]      # a different neighbourhood
        xd, wd, gyd = lay.act_to_dev(dev(x)), lay.conv_w_to_dev(dev(W)), lay.act_to_dev(dev(gy))
        del pads
        x16, gy16 = xd.to(torch.bfloat16), gyd.to(torch.bfloat16)
        with torch.cuda.stream(side):                               # traffic beside the launches
            for _ in range(2):
                noise_src.mul_(1.0001)
        out = {}
        for prec, (xa, ga) in (('bf16', (xd, gyd)), ('bf16s', (x16, gy16))):
            g = hl.make_geom(N, Ti, H, H, Ci, Co, kt, precision=prec)
            g.tile = tile
            dwd = torch.zeros_like(wd)
            hl.conv_wgrad(g, xa, ga, dwd)
            out[prec] = dwd
        for prec, dwd in out.items():
            err = float((dwd.double() - ref).norm()) / nref
            worst = max(worst, err)
            if not err < 1e-6:
                from guard import describe_diff
                d = (dwd.double() - ref).abs()
                bad = torch.nonzero(~(d <= 1e-5 * float(ref.abs().max())))[:6].cpu().numpy()
                other = out['bf16s' if prec == 'bf16' else 'bf16']
                raise AssertionError("repetition %d, tile code %d: the %r weight gradient left the oracle (rel-L2 %.3e). %s. Elements: %s"
                                     % (rep, tile, prec, err, describe_diff(dwd, other, ref.cpu().numpy(), names=(prec, "the other launch")),
                                        "; ".join(_where(i, Ci, kt) for i in bad)))
    torch.cuda.synchronize()
    assert worst < 1e-6
