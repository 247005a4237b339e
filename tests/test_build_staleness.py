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
