"""Build recipe for libklstm.so (hipcc, gfx950 only, in-tree so it travels with the repo).

Every .hip source becomes its own object file (compiled in parallel, re-compiled only when that source or a header
changed: content hashes, because mtimes do not survive the copy to the GPU box); the link writes a temporary file that is
renamed over libklstm.so under a file lock, so concurrent ranks of a multi-process launch never dlopen a half-written
library."""
import concurrent.futures
import fcntl
import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRCS = ["csrc/klstm_kernels.hip", "csrc/klstm_persist.hip", "csrc/klstm_persist_bwd.hip", "csrc/klstm_persist_ms.hip", "csrc/klstm_persist_xl.hip", "csrc/klstm_fold.hip", "csrc/klstm_fold3.hip", "csrc/klstm_oneshot.hip", "csrc/klstm_outer.hip", "csrc/klstm_gemm16.hip",
        "csrc/klstm_engine.hip"]
HDRS = ["csrc/klstm_kernels.h", "csrc/klstm_math.h", "csrc/klstm_persist_dev.h", "../include/klstm.h"]
LIB = os.path.join(HERE, "libklstm.so")
OBJDIR = os.path.join(HERE, "build")
STAMP = LIB + ".srchash"
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-result", "-I/opt/rocm/include"]


def _hash(files):
    h = hashlib.sha256()
    for f in files:
        with open(os.path.join(HERE, f), "rb") as fh:
            h.update(f.encode() + b"\0" + fh.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def source_hash():
    """Content hash of everything the library is built from."""
    return _hash(SRCS + HDRS + ["build.py"])


def stale():
    if not os.path.exists(LIB) or not os.path.exists(STAMP):
        return True
    with open(STAMP) as fh:
        return fh.read().strip() != source_hash()


def _compile(hipcc, src, verbose):
    obj = os.path.join(OBJDIR, os.path.basename(src) + ".o")
    stamp = obj + ".hash"
    want = _hash([src] + HDRS)
    if os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read().strip() == want:
        return obj
    cmd = [hipcc] + FLAGS + ["-c", os.path.join(HERE, src), "-o", obj]
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
    subprocess.check_call(cmd)
    with open(stamp, "w") as fh:
        fh.write(want + "\n")
    return obj


def build(force=False, verbose=False):
    if not force and not stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not (os.path.exists(hipcc) or shutil.which(hipcc)):
        if os.path.exists(LIB):                      # a prebuilt library on a box without the compiler: use it
            sys.stderr.write("klstm build: hipcc not found, loading the existing %s as is\n" % LIB)
            return LIB
        raise RuntimeError("hipcc not found and %s does not exist" % LIB)
    os.makedirs(OBJDIR, exist_ok=True)
    with open(os.path.join(OBJDIR, ".lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)             # one builder at a time; the others find a fresh library afterwards
        if not force and not stale():
            return LIB
        if force:
            for f in os.listdir(OBJDIR):
                if f.endswith(".hash"):
                    os.remove(os.path.join(OBJDIR, f))
        with concurrent.futures.ThreadPoolExecutor(max_workers=len(SRCS)) as pool:
            objs = list(pool.map(lambda s: _compile(hipcc, s, verbose), SRCS))
        tmp = LIB + ".tmp.%d" % os.getpid()
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp] + objs + ["-ldl"])
        os.replace(tmp, LIB)
        with open(STAMP, "w") as fh:
            fh.write(source_hash() + "\n")
    return LIB


ADAPTER_SRC = os.path.join(HERE, "..", "tools", "kaldi_adapter_bench.cpp")
ADAPTER_EXE = os.path.join(HERE, "..", "tools", "kaldi_adapter_bench")


def build_adapter_bench():
    """tools/kaldi_adapter_bench: the C++ component of include/klstm_component.hpp as the Kaldi shim drives it, timed (bench.py's
    `kaldi_adapter` leg).  Plain g++ against the C-ABI: the mirror needs no HIP headers."""
    cxx = shutil.which("g++") or shutil.which("c++") or "/opt/rocm/bin/hipcc"
    tmp = ADAPTER_EXE + ".tmp.%d" % os.getpid()
    subprocess.check_call([cxx, "-std=c++17", "-O2", "-I" + os.path.join(HERE, "..", "include"), ADAPTER_SRC, "-L" + HERE, "-lklstm",
                           "-Wl,-rpath,$ORIGIN/../kaldi-lstm_amd", "-o", tmp])
    os.replace(tmp, ADAPTER_EXE)
    return ADAPTER_EXE


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose="--verbose" in sys.argv)
    print(LIB)
