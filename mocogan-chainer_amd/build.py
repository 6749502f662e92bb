"""Builds lib/libmocogan_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SOURCES = ["csrc/conv_gemm.hip", "csrc/small_ops.hip"]


def lib_path():
    """In-tree library; MCG_LIB_PATH points the binding at another build of the same sources (A/B timing)."""
    return os.environ.get("MCG_LIB_PATH") or os.path.join(HERE, "lib", "libmocogan_hip.so")


def _stale():
    out = lib_path()
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    deps = [os.path.join(HERE, s) for s in SOURCES] + [os.path.join(HERE, "csrc", "mcg_common.h"),
                                                       os.path.join(ROOT, "include", "mocogan_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 -O3 -fPIC -shared -> lib/libmocogan_hip.so (in-tree)."""
    if not force and not _stale():
        return lib_path()
    os.makedirs(os.path.join(HERE, "lib"), exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-fPIC", "-shared", "-std=c++17",
           "-I" + os.path.join(ROOT, "include")] + [os.path.join(HERE, s) for s in SOURCES] + ["-o", lib_path()]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return lib_path()


if __name__ == "__main__":
    print(build(force=True, verbose=True))
