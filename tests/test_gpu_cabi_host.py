"""The C ABI is usable without Python or PyTorch: examples/cabi_host.cpp (hipMalloc'd buffers, a hipStream_t,
status codes) is compiled against include/mocogan_hip.h + libmocogan_hip.so and run; it checks conv3d fprop /
dgrad / wgrad, BatchNorm + LeakyReLU and Adam against plain CPU loops."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_abi_host_program(tmp_path):
    lib = os.path.join(ROOT, 'mocogan-chainer_amd', 'lib')
    assert os.path.exists(os.path.join(lib, 'libmocogan_hip.so')), 'run __graft_entry__.build() first'
    exe = str(tmp_path / 'cabi_host')
    subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O2', '-I' + os.path.join(ROOT, 'include'),
                    os.path.join(ROOT, 'examples', 'cabi_host.cpp'), '-L' + lib, '-lmocogan_hip', '-Wl,-rpath,' + lib, '-o', exe],
                   check=True, timeout=300)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    print(r.stdout, r.stderr)
    assert r.returncode == 0 and 'PASS' in r.stdout, r.stdout + r.stderr
