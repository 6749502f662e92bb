"""CPU oracle for the MoCoGAN training hot path -- TEST INFRASTRUCTURE ONLY.

This package restates, in plain NumPy, the arithmetic that the reference
(raahii/mocogan-chainer, ``model/net.py``, ``model/updater.py``, ``train.py:93-101``)
delegates to Chainer 3.1.0 (pinned at the reference's ``requirements.txt:1``; Chainer
itself is not vendored in the reference, not installed here and not installable
offline).  Every function cites the reference file:line whose behaviour it follows and,
where the arithmetic lives in Chainer, the published Chainer-v3 formula it restates.

PARITY UNPINNED: the reference ships no tests, golden vectors or fixtures for this path
and cannot be executed in this environment, so this oracle is pinned only by
(i) independent cross-checks against torch-CPU functional ops / autograd
(``tests/test_oracle_vs_torch.py``), (ii) finite-difference gradient checks, and
(iii) the golden fixtures under ``tests/golden/`` that it generated itself
(``tests/golden/make_golden.py``).

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py``
may import this package.  The product path (``mocogan-chainer_amd/``, ``model/``) never
does; it fails loudly when the HIP library is missing.

All arrays use the reference's own layouts (NCHW / NCDHW activations, Chainer weight
shapes and Chainer parameter names such as ``dc1/W`` or ``g0/W_r/W``).  Every routine is
dtype-generic: float64 is the parity oracle, float32 is the "port" CPU baseline (same
algorithm class as Chainer's CPU path: im2col + BLAS GEMM).
"""
