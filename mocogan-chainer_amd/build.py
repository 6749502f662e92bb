"""Builds lib/libmocogan_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SOURCES = ["csrc/conv_gemm.hip", "csrc/small_ops.hip"]


def lib_path():
    """In-tree library; MCG_LIB_PATH points the binding at another build of the same sources (A/B timing)."""
    return os.environ.get("MCG_LIB_PATH") or os.path.join(HERE, "lib", "libmocogan_hip.so")


def _flags():
    return os.environ.get("MCG_HIPCC_FLAGS", "").split()        # e.g. -DMCG_STAMPS for the diagnostic builds of tools/


def _stamp_path():
    return lib_path() + ".flags"


def _stale():
    out = lib_path()
    if not os.path.exists(out):
        return True
    # a library built with other compile flags (a diagnostic build such as -DMCG_STAMPS, or the reverse) is not this build
    try:
        if open(_stamp_path()).read() != " ".join(_flags()):
            return True
    except OSError:
        return True
    t = os.path.getmtime(out)
    deps = [os.path.join(HERE, s) for s in SOURCES] + [os.path.join(HERE, "csrc", "mcg_common.h"),
                                                       os.path.join(ROOT, "include", "mocogan_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 -O3 -fPIC -> lib/libmocogan_hip.so (in-tree).  The conv GEMM source is compiled as
    three translation units (one per pass, -DMCG_TU=1..3) next to small_ops.hip, all four in parallel, then linked."""
    if not force and not _stale():
        return lib_path()
    os.makedirs(os.path.join(HERE, "lib"), exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    extra = _flags()
    objdir = os.path.join(HERE, "lib", "obj")
    os.makedirs(objdir, exist_ok=True)
    common = [hipcc, "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-I" + os.path.join(ROOT, "include")] + extra
    units = [("conv_gemm_tu%d.o" % tu, ["-DMCG_TU=%d" % tu, os.path.join(HERE, "csrc/conv_gemm.hip")]) for tu in (1, 2, 3)]
    units.append(("small_ops.o", [os.path.join(HERE, "csrc/small_ops.hip")]))
    procs = []
    for obj, args in units:
        cmd = common + args + ["-c", "-o", os.path.join(objdir, obj)]
        if verbose:
            print(" ".join(cmd))
        procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, cmd)
    link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + [os.path.join(objdir, o) for o, _ in units] + ["-o", lib_path()]
    if verbose:
        print(" ".join(link))
    subprocess.run(link, check=True)
    import ctypes
    ctypes.CDLL(lib_path(), mode=os.RTLD_NOW)         # every kernel stub must resolve NOW (the HIP runtime binds them at load on the GPU box)
    with open(_stamp_path(), "w") as f:
        f.write(" ".join(extra))
    return lib_path()


if __name__ == "__main__":
    print(build(force=True, verbose=True))
