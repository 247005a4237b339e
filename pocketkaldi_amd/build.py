"""Build libpk_mi355.so (HIP kernels + C ABI) for gfx950, in-tree.

    python -m pocketkaldi_amd.build [--force]

hipcc cross-compiles without a GPU.  Device and host float code is compiled with
-ffp-contract=off: parity with the reference depends on a*b+c NOT being fused
where the reference's x86-64 build does not fuse (see csrc/frontend.hip).

Staleness is decided by CONTENT, not by mtime: the hash of every source and header is
written next to the library (libpk_mi355.buildhash) and compared on every load, so a
prebuilt library that travelled with the tree (the .so is git-ignored but ships to the GPU
box) can never be loaded against newer sources without a rebuild.
"""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libpk_mi355.so")
STAMP = os.path.join(HERE, "libpk_mi355.buildhash")
HIP_SOURCES = ["frontend.hip", "gemm.hip", "gemm_f16.hip", "tail.hip", "capi_model.hip", "capi_exec.hip", "capi_batch.hip",
               "capi_io.hip", "capi_collective.hip"]
HOST_SOURCES = ["pk_tables.cc"]
HEADERS = ["pk_kernels.h", "pk_tables.h", "pk_logf.h", "pk_expf.h", "pk_dma.h", "pk_tail_wave.h", "pk_wave.h", "pk_host.h", "libpk_mi355.map",
           os.path.join("..", "..", "include", "pk_mi355.h")]
ARCH = "gfx950"
HIP_FLAGS = ["--offload-arch=" + ARCH, "-std=c++17", "-O3", "-ffp-contract=off", "-fPIC", "-Wall",
             "-Wno-unused-function", "-Wno-unused-value", "-Wno-unused-result"]
HOST_FLAGS = ["-std=c++17", "-O2", "-ffp-contract=off", "-fPIC"]
# RCCL is bound at run time (dlopen in capi_collective.hip): the library has no link-time dependency on it.
# The version script exports the C entries of include/pk_mi355.h and nothing else.
LINK_LIBS = ["-ldl", "-Wl,--version-script=" + os.path.join(CSRC, "libpk_mi355.map")]
LINK_ID = ["-ldl", "-Wl,--version-script=libpk_mi355.map"]      # what the content hash sees: no absolute path (the tree moves)


def source_hash():
    h = hashlib.sha256()
    for s in HIP_SOURCES + HOST_SOURCES + HEADERS:
        with open(os.path.join(CSRC, s), "rb") as f:
            h.update(s.encode() + b"\0" + f.read() + b"\0")
    h.update(" ".join(HIP_FLAGS + HOST_FLAGS + LINK_ID).encode())
    return h.hexdigest()


def gemm_source_hash():
    """Hash of the sources of the dominant kernel alone (GemmKernel: gemm.hip + the headers it includes)
    and the device compile flags.  tools/profile_gpu.sh records it with every PMC section, so that
    bench.py reports roofline.traffic only from a measurement of THIS kernel."""
    h = hashlib.sha256()
    for s in ["gemm.hip", "pk_dma.h", "pk_kernels.h", "pk_tables.h", "pk_tail_wave.h", "pk_wave.h"]:   # gemm.hip and every header it includes
        with open(os.path.join(CSRC, s), "rb") as f:
            h.update(s.encode() + b"\0" + f.read() + b"\0")
    h.update(" ".join(HIP_FLAGS).encode())
    return h.hexdigest()[:16]


def _stale():
    if not os.path.exists(LIB) or not os.path.exists(STAMP):
        return True
    with open(STAMP) as f:
        return f.read().strip() != source_hash()


def have_compiler():
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    return os.path.exists(hipcc)


def build(force=False, verbose=False):
    """Serialised across processes by an flock next to the library: the ranks of a torchrun /
    mp.spawn launch on a fresh tree all find the stamp missing at once; the first one builds, the
    others wait and find the library fresh.  Objects and the temporary library carry the pid."""
    import fcntl
    if not force and not _stale():
        return LIB
    with open(LIB + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not _stale():      # another process built it while this one waited
                return LIB
            return _build_locked(verbose)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(verbose):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    digest = source_hash()
    pid = os.getpid()
    objs = []
    try:
        for s in HOST_SOURCES:
            o = os.path.join(CSRC, "%s.%d.o" % (s, pid))
            cmd = ["g++"] + HOST_FLAGS + ["-c", os.path.join(CSRC, s), "-o", o]
            if verbose:
                print(" ".join(cmd))
            objs.append(o)
            subprocess.check_call(cmd)
        procs = []
        for s in HIP_SOURCES:                       # the translation units are independent
            o = os.path.join(CSRC, "%s.%d.o" % (s, pid))
            cmd = [hipcc] + HIP_FLAGS + ["-c", os.path.join(CSRC, s), "-o", o]
            if verbose:
                print(" ".join(cmd))
            objs.append(o)
            procs.append((cmd, subprocess.Popen(cmd)))
        failed = None
        for cmd, p in procs:
            if p.wait() != 0 and failed is None:
                failed = subprocess.CalledProcessError(p.returncode, cmd)
        if failed:
            raise failed
        tmp = "%s.%d.tmp" % (LIB, pid)
        cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", tmp] + objs + LINK_LIBS
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        os.replace(tmp, LIB)
        with open(STAMP + ".%d" % pid, "w") as f:
            f.write(digest + "\n")
        os.replace(STAMP + ".%d" % pid, STAMP)
    finally:
        for o in objs:
            if os.path.exists(o):
                os.unlink(o)
    return LIB


if __name__ == "__main__":
    if "--hashes" in sys.argv:          # tools/profile_gpu.sh: what the profiled library was built from, and the
        # library FILE that will be loaded (tools/ab_so.sh swaps prebuilt libraries under a matching stamp: ADVICE round 3)
        with open(LIB, "rb") as f:
            file_sha = hashlib.sha256(f.read()).hexdigest()[:16]
        print("library_build_hash=%s gemm_source_hash=%s library_file_sha256=%s" % (source_hash()[:16], gemm_source_hash(), file_sha))
    else:
        print(build(force="--force" in sys.argv, verbose=True))
