#!/usr/bin/env python
"""Diagnostic: builds the library with extra -D flags into /tmp and runs tools/bench_layers.py on it.
    python tools/probe_variant.py -DMCG_PROBE_SAMETILE -- --batch 32 --net D_V
MCG_PROBE_SAMETILE: every block loads tile 0 (operands always cache-resident): the gap to the real build is
the cost of the memory system, the rest is instruction issue / LDS / barriers."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
args = sys.argv[1:]
split = args.index('--') if '--' in args else len(args)
defs, rest = args[:split], args[split + 1:]
so = '/tmp/libmocogan_probe.so'
subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-fPIC', '-shared', '-std=c++17'] + defs +
               ['-I' + os.path.join(ROOT, 'include'), os.path.join(ROOT, 'mocogan-chainer_amd/csrc/conv_gemm.hip'),
                os.path.join(ROOT, 'mocogan-chainer_amd/csrc/small_ops.hip'), '-o', so], check=True)
import mocogan_chainer_amd.hiplib as hl
hl.lib_path = lambda: so
sys.argv = [sys.argv[0]] + rest
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import bench_layers
bench_layers.main()
