"""The C++ host mirror (include/pocketkaldi_amd.hpp): a caller shaped like the reference's own
test programs compiles against the headers and links to libpk_mi355.so (CPU), and passes on a GPU."""
import os
import subprocess

import pytest

import pocketkaldi_amd as pk

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(REPO, "tests", "cpp", "hot_path_test.cc")
BIN = os.path.join(REPO, "tests", "cpp", "hot_path_test.bin")


def build_binary():
    pk.lib()
    libdir = os.path.dirname(pk.lib_path())
    subprocess.check_call(["g++", "-std=c++11", "-O1", "-I", os.path.join(REPO, "include"), SRC, "-o", BIN,
                           "-L", libdir, "-l:libpk_mi355.so", "-Wl,-rpath," + libdir])
    return BIN


def test_cpp_caller_compiles_and_links():
    out = subprocess.check_output([build_binary(), "--link-only"], text=True)
    assert "pk_mi355" in out
    assert "host logic ok" in out          # frame counts and pocketkaldi::PartitionByFrames (no device needed)


@pytest.mark.gpu
def test_cpp_caller_runs_reference_style_tests():
    out = subprocess.check_output([build_binary(), os.path.join(REPO, "tests", "golden")], text=True)
    assert "hot_path_test ok" in out


def build_broadcast_binary():
    pk.lib()
    libdir = os.path.dirname(pk.lib_path())
    out = os.path.join(REPO, "tests", "cpp", "broadcast_test.bin")
    subprocess.check_call(["g++", "-std=c++11", "-O1", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(REPO, "include"),
                           "-I", "/opt/rocm/include", os.path.join(REPO, "tests", "cpp", "broadcast_test.cc"), "-o", out,
                           "-L", libdir, "-l:libpk_mi355.so", "-L", "/opt/rocm/lib", "-lrccl", "-lamdhip64",
                           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    return out


def test_cpp_multi_gpu_host_compiles_against_rccl():
    assert os.path.exists(build_broadcast_binary())


@pytest.mark.gpu
def test_cpp_host_broadcasts_the_model_over_rccl():
    """pk_mi355_am_broadcast from a C++ host linked to the system RCCL (one HIP runtime in the
    process, as in a real C++ deployment): tests/cpp/broadcast_test.cc."""
    out = subprocess.check_output([build_broadcast_binary()], text=True)
    assert "broadcast_test ok" in out


def test_device_logf_expf_restatements_match_libm_on_cpu():
    """csrc/pk_logf.h / pk_expf.h, the logf and expf of the kernels, compiled for the host and compared
    with the C library's on ~70 M floats (tests/cpp/libm_restated_test.cc; exhaustive sweeps:
    tools/logf_check.c, tools/expf_check.c)."""
    src = os.path.join(REPO, "tests", "cpp", "libm_restated_test.cc")
    binary = os.path.join(REPO, "tests", "cpp", "libm_restated_test.bin")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-ffp-contract=off", src, "-o", binary, "-lm"])
    out = subprocess.check_output([binary], text=True)
    assert out.strip().endswith(" 0 mismatches"), out
