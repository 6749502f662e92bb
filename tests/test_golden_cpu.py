"""The oracle against its committed golden vectors (tests/golden/, made by make_golden.py) and the
published Random123 known-answer vectors for Philox4x32-10.  CPU only."""
import glob
import os

import numpy as np
import pytest

from oracle import philox

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, 'golden')


def _load(name):
    return {k.replace('.', '/'): v for k, v in np.load(os.path.join(GOLD, name)).items()}


def test_philox_published_known_answers():
    # kat_vectors of Random123 (philox4x32 10): counter/key = 0; all-ones; digits of pi
    kat = np.array([[0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8],
                    [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd],
                    [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]], dtype=np.uint32)
    gold = _load('ops.npz')
    assert np.array_equal(gold['philox/kat'], kat)
    got = np.array([philox.philox4x32_10([c[0]], [c[1]], [c[2]], [c[3]], c[4], c[5]) for c in
                    ((0,) * 6, (0xffffffff,) * 6, (0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344, 0xa4093822, 0x299f31d0))],
                   dtype=np.uint32).reshape(3, 4)
    assert np.array_equal(got, kat)


def test_ops_match_golden():
    import importlib.util
    spec = importlib.util.spec_from_file_location('make_golden', os.path.join(GOLD, 'make_golden.py'))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    gold = _load('ops.npz')
    now = mg.op_cases()
    assert set(gold) == set(now)
    for k in gold:
        assert np.allclose(now[k], gold[k], rtol=1e-12, atol=1e-14), k


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, 'step_*.npz'))))
def test_update_core_matches_golden(path):
    import importlib.util
    spec = importlib.util.spec_from_file_location('make_golden', os.path.join(GOLD, 'make_golden.py'))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    name = os.path.basename(path)[:-4].split('_')
    model, dim_zl, seed = name[1], int(name[2][2:]), int(name[3][4:])
    assert (model, dim_zl, seed) in mg.STEP_CASES
    gold = _load(os.path.basename(path))
    now = mg.step_case(model, dim_zl, seed)
    assert set(gold) == set(now)
    for k in gold:
        if k.endswith('min_margin'):
            assert np.isclose(now[k], gold[k], rtol=1e-4), k     # a difference of tiny numbers
        else:
            assert np.allclose(now[k], gold[k], rtol=1e-9, atol=1e-12), k
