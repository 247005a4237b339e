#!/usr/bin/env python3
"""Build an experiment variant of libpk_mi355.so next to the tree's own (gpurun_ab/<name>.so) for an A/B on one
GPU box (tools/gpu.sh ab): the product sources with extra -D flags.  The tree's library and stamp are not touched.

    python tools/build_variant.py nt -DPK_EXP_TAIL_NT
"""
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pocketkaldi_amd import build as B   # noqa: E402


def main():
    name, extra = sys.argv[1], sys.argv[2:]
    out_dir = os.path.join(REPO, "gpurun_ab")
    os.makedirs(out_dir, exist_ok=True)
    tmp = os.path.join(out_dir, "obj_" + name)
    os.makedirs(tmp, exist_ok=True)
    objs, procs = [], []
    for s in B.HOST_SOURCES:
        o = os.path.join(tmp, s + ".o")
        subprocess.check_call(["g++"] + B.HOST_FLAGS + extra + ["-c", os.path.join(B.CSRC, s), "-o", o])
        objs.append(o)
    for s in B.HIP_SOURCES:
        o = os.path.join(tmp, s + ".o")
        procs.append(subprocess.Popen(["/opt/rocm/bin/hipcc"] + B.HIP_FLAGS + extra + ["-c", os.path.join(B.CSRC, s), "-o", o]))
        objs.append(o)
    if any(p.wait() != 0 for p in procs):
        raise SystemExit("compile failed")
    lib = os.path.join(out_dir, name + ".so")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=" + B.ARCH, "-shared", "-fPIC", "-o", lib] + objs + B.LINK_LIBS)
    for o in objs:
        os.unlink(o)
    os.rmdir(tmp)
    print(lib)


if __name__ == "__main__":
    main()
