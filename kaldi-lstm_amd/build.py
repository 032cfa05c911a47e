"""Build recipe for libklstm.so (hipcc, gfx950 only, in-tree so it travels with the repo)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRCS = ["csrc/klstm_kernels.hip", "csrc/klstm_persist.hip", "csrc/klstm_fold.hip", "csrc/klstm_engine.hip"]
HDRS = ["csrc/klstm_kernels.h", "csrc/klstm_math.h", "../include/klstm.h"]
LIB = os.path.join(HERE, "libklstm.so")


STAMP = LIB + ".srchash"


def source_hash():
    """Content hash of everything the library is built from (mtimes do not survive the copy to the GPU box)."""
    import hashlib
    h = hashlib.sha256()
    for f in SRCS + HDRS + ["build.py"]:
        with open(os.path.join(HERE, f), "rb") as fh:
            h.update(f.encode() + b"\0" + fh.read())
    return h.hexdigest()


def stale():
    if not os.path.exists(LIB) or not os.path.exists(STAMP):
        return True
    with open(STAMP) as fh:
        return fh.read().strip() != source_hash()


def build(force=False, verbose=False):
    if not force and not stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-Wall", "-Wno-unused-result", "-I/opt/rocm/include", "-o", LIB] + [os.path.join(HERE, s) for s in SRCS]
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
    cmd += ["-ldl"]
    subprocess.check_call(cmd)
    with open(STAMP, "w") as fh:
        fh.write(source_hash() + "\n")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose="--verbose" in sys.argv)
    print(LIB)
