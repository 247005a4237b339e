"""Build libpk_mi355.so (HIP kernels + C ABI) for gfx950, in-tree.

    python -m pocketkaldi_amd.build [--force]

hipcc cross-compiles without a GPU.  Device and host float code is compiled with
-ffp-contract=off: parity with the reference depends on a*b+c NOT being fused
where the reference's x86-64 build does not fuse (see csrc/frontend.hip).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libpk_mi355.so")
HIP_SOURCES = ["frontend.hip", "gemm.hip", "gemm_f16.hip", "tail.hip", "capi.hip"]
HOST_SOURCES = ["pk_tables.cc"]
HEADERS = ["pk_kernels.h", "pk_tables.h", "pk_logf.h", "pk_expf.h", "pk_dma.h", os.path.join("..", "..", "include", "pk_mi355.h")]
ARCH = "gfx950"


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in HIP_SOURCES + HOST_SOURCES + HEADERS]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    for s in HOST_SOURCES:
        o = os.path.join(CSRC, s + ".o")
        cmd = ["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-fPIC", "-c", os.path.join(CSRC, s), "-o", o]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        objs.append(o)
    for s in HIP_SOURCES:
        o = os.path.join(CSRC, s + ".o")
        cmd = [hipcc, "--offload-arch=" + ARCH, "-std=c++17", "-O3", "-ffp-contract=off", "-fPIC",
               "-Wall", "-Wno-unused-function", "-Wno-unused-value", "-Wno-unused-result", "-c", os.path.join(CSRC, s), "-o", o]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        objs.append(o)
    cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
