"""Build libptocr_hip.so (all HIP kernels + the C ABI of include/ptocr_hip.h) for gfx950, in-tree.

    python -m pytorchocr_amd.build [--force]

hipcc cross-compiles without a GPU.  The .so is git-ignored but travels with the repo snapshot.
"""
import glob
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libptocr_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"
# -ffp-contract=off: the post-process geometry must round op by op like the reference's x86 build.
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-Wno-unused-function"]
FLAGS += os.environ.get("PTOCR_EXTRA_HIPCC_FLAGS", "").split()        # kernel experiments, e.g. -DPTOCR_WINO_EXPERIMENT


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _flags_tag(flags=None):
    """Digest of the compile flags: an object built with other flags (a -D knock-out of tools/dbg) is stale."""
    return hashlib.sha256(" ".join([HIPCC] + (FLAGS if flags is None else flags)).encode()).hexdigest()[:16]


def _stale(obj, src):
    deps = [src] + glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(HERE, "..", "include", "ptocr_hip.h")]
    if (not os.path.exists(obj)) or any(os.path.getmtime(d) > os.path.getmtime(obj) for d in deps):
        return True
    try:
        with open(obj + ".flags") as f:
            return f.read().strip() != _flags_tag()
    except OSError:
        return True


def build(force=False, verbose=True):
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    objs, procs = [], []
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        objs.append(obj)
        if force or _stale(obj, src):
            # the library says which flags it was built with (ptocr_build_tag): a knock-out build left behind by tools/dbg is refused at load
            cmd = [HIPCC] + FLAGS + ['-DPTOCR_BUILD_TAG="%s"' % _flags_tag(), "-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            flagfile = obj + ".flags"
            if os.path.exists(flagfile):
                os.remove(flagfile)                            # stale until THIS compile has finished
            procs.append((src, subprocess.Popen(cmd)))
    failed = []
    for src, p in procs:
        if p.wait() != 0:
            failed.append(src)
            continue
        # written as each compile finishes, also when another one fails: an object never keeps a tag that is not its own
        with open(os.path.join(objdir, os.path.basename(src) + ".o.flags"), "w") as f:
            f.write(_flags_tag())
    if failed:
        raise RuntimeError("hipcc failed on " + ", ".join(failed))
    if procs or not os.path.exists(LIB):
        cmd = [HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print("built", LIB)
