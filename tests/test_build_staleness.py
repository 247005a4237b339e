"""pocketkaldi_amd.lib() never loads a library built from other sources (ADVICE r1 / VERDICT r1 #11):
staleness is decided by a content hash of every source and header, written next to the .so."""
import os

from pocketkaldi_amd import build as B


def test_fresh_library_is_not_stale_and_hash_covers_every_source(tmp_path, monkeypatch):
    B.build()                       # no-op when current
    assert not B._stale()
    digest = B.source_hash()
    assert open(B.STAMP).read().strip() == digest
    # any source or header edit changes the hash -> stale -> lib() rebuilds (or refuses without hipcc)
    for name in B.HIP_SOURCES + B.HOST_SOURCES + B.HEADERS:
        assert os.path.exists(os.path.join(B.CSRC, name))
    src = os.path.join(B.CSRC, B.HIP_SOURCES[0])
    fake = tmp_path / "csrc"
    os.makedirs(fake / ".." / ".." / "include", exist_ok=True)
    for name in B.HIP_SOURCES + B.HOST_SOURCES + B.HEADERS:
        data = open(os.path.join(B.CSRC, name), "rb").read()
        dst = os.path.normpath(os.path.join(fake, name))
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        open(dst, "wb").write(data + (b"\n// edited\n" if name == B.HEADERS[0] else b""))
    monkeypatch.setattr(B, "CSRC", str(fake))
    assert B.source_hash() != digest and B._stale()


def test_no_env_override_of_the_product_library(monkeypatch):
    import pocketkaldi_amd as pk
    monkeypatch.setenv("PK_MI355_LIB", "/tmp/other.so")
    assert pk.lib_path() == os.path.join(os.path.dirname(pk.__file__), "libpk_mi355.so")


def test_concurrent_builds_are_serialised():
    """on a fresh tree (stamp git-ignored, so missing) every rank of a torchrun / mp.spawn
    launch used to compile into the same object files at once.  Three processes that all find the stamp
    missing: exactly one compiles (flock next to the library), the others wait and find it fresh."""
    import subprocess
    import sys
    B.build()
    os.unlink(B.STAMP)
    code = ("import sys, time; sys.path.insert(0, %r); from pocketkaldi_amd import build as B; t = time.time(); "
            "B.build(); print('built' if time.time() - t > 0 else '', int(not B._stale()))" % os.path.dirname(os.path.dirname(B.HERE + "/")))
    procs = [subprocess.Popen([sys.executable, "-c", code], stdout=subprocess.PIPE, text=True) for _ in range(3)]
    outs = [p.communicate()[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert all(o.strip().endswith("1") for o in outs), outs
    assert not B._stale() and open(B.STAMP).read().strip() == B.source_hash()
    assert not [f for f in os.listdir(B.CSRC) if f.endswith(".o")]            # per-pid objects are removed
    assert not [f for f in os.listdir(B.HERE) if f.endswith(".tmp")]
