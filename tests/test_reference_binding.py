"""the reference-side binding INTEGRATION.md section 2 prescribes really compiles.

A scratch VIEW of the reference tree (symbolic links under tmp_path -- nothing of the reference is
copied into the repository) is bound with include/reference_binding/apply_binding.py, then the
reference's own decoder.cc (unchanged) and pocketkaldi.cc (the two documented edits) are compiled
against include/reference_binding/decodable.h + include/pk_mi355.h, and every symbol those objects
now expect from outside is looked up in libpk_mi355.so's export table.

Needs /root/reference (this container only); skipped elsewhere.  No GPU.
"""
import ctypes
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
BINDING = os.path.join(ROOT, "include", "reference_binding")

pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "src")),
                                reason="reference tree not mounted")


def _view(tmp_path):
    """tmp/ref/{Makefile.am, src/*} as symbolic links to the reference."""
    root = tmp_path / "ref"
    (root / "src").mkdir(parents=True)
    os.symlink(os.path.join(REF, "Makefile.am"), root / "Makefile.am")
    for name in os.listdir(os.path.join(REF, "src")):
        os.symlink(os.path.join(REF, "src", name), root / "src" / name)
    return root


def _bind(root):
    r = subprocess.run([sys.executable, os.path.join(BINDING, "apply_binding.py"), str(root)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


def _gxx(src_dir, args):
    return subprocess.run(["g++", "-std=c++11", "-I", str(src_dir)] + args, cwd=src_dir,
                          capture_output=True, text=True)


def _undefined(obj):
    out = subprocess.run(["nm", "--undefined-only", str(obj)], capture_output=True, text=True, check=True).stdout
    return {line.split()[-1] for line in out.splitlines() if line.strip()}


def _exports(lib):
    out = subprocess.run(["nm", "-D", "--defined-only", lib], capture_output=True, text=True, check=True).stdout
    return {line.split()[-1] for line in out.splitlines() if line.strip()}


def test_decoder_cc_compiles_unchanged_against_the_binding(tmp_path):
    root = _view(tmp_path)
    _bind(root)
    src = root / "src"
    # decoder.cc / decoder.h are still the reference's own files (links), not edited copies
    assert os.path.islink(src / "decoder.cc") and os.path.islink(src / "decoder.h")
    r = _gxx(src, ["-fsyntax-only", "decoder.cc"])
    assert r.returncode == 0, r.stderr
    r = _gxx(src, ["-O1", "-c", "decoder.cc", "-o", str(tmp_path / "decoder.o")])
    assert r.returncode == 0, r.stderr
    und = _undefined(tmp_path / "decoder.o")
    # decoder.cc:49,252,276 -- unmangled C symbols, the ones libpk_mi355.so exports
    assert {"pk_decodable_islastframe", "pk_decodable_loglikelihood"} <= und
    assert not [s for s in und if "AcousticModel" in s]


def test_pocketkaldi_cc_compiles_with_the_two_documented_edits(tmp_path):
    root = _view(tmp_path)
    _bind(root)
    src = root / "src"
    text = (src / "pocketkaldi.cc").read_text()
    assert "pk_mi355_load(filename" in text and "pk_mi355_am_destroy(self->am)" in text
    assert "new AcousticModel" not in text
    # pocketkaldi.h (pk_t, pk_process's prototype) is untouched
    assert os.path.islink(src / "pocketkaldi.h")
    r = _gxx(src, ["-O1", "-c", "pocketkaldi.cc", "-o", str(tmp_path / "pocketkaldi.o")])
    assert r.returncode == 0, r.stderr
    und = _undefined(tmp_path / "pocketkaldi.o")
    assert {"pk_decodable_init", "pk_decodable_destroy", "pk_mi355_load", "pk_mi355_am_destroy",
            "pk_mi355_last_error"} <= und
    # nothing of the replaced classes is referenced any more (am.cc / nnet.cc leave the build)
    assert not [s for s in und if "AcousticModel" in s or "4Nnet" in s], und


def test_every_symbol_the_bound_objects_need_is_exported(tmp_path):
    from pocketkaldi_amd import build
    root = _view(tmp_path)
    _bind(root)
    src = root / "src"
    need = set()
    for cc in ("decoder.cc", "pocketkaldi.cc"):
        obj = tmp_path / (cc + ".o")
        assert _gxx(src, ["-O1", "-c", cc, "-o", str(obj)]).returncode == 0
        need |= {s for s in _undefined(obj) if s.startswith("pk_decodable_") or s.startswith("pk_mi355_")}
    assert need, "bound objects reference nothing of the library?"
    missing = need - _exports(build.LIB)
    assert not missing, missing


def test_makefile_am_drops_the_replaced_sources(tmp_path):
    root = _view(tmp_path)
    _bind(root)
    mk = (root / "Makefile.am").read_text()
    for gone in ("src/decodable.cc", "src/am.cc", "src/nnet.cc"):
        assert gone not in mk
    for kept in ("src/decoder.cc", "src/fst.cc", "src/pocketkaldi.cc", "src/symbol_table.cc"):
        assert kept in mk
    assert re.search(r"pocketkaldi_LDADD\s*=.*-lpk_mi355", mk)


def test_header_without_the_reference_guards_would_not_compile(tmp_path):
    """The failure VERDICT round 2 found: pk_mi355.h redefining pk_matrix_t / pk_vector_t inside the
    reference tree.  Strip the three guards from a scratch copy of the header and the same
    compile must fail with exactly that diagnosis -- i.e. the test above is able to fail."""
    root = _view(tmp_path)
    _bind(root)
    src = root / "src"
    hdr = (src / "pk_mi355.h").read_text()
    stripped = re.sub(r"#ifndef (POCKETKALDI_MATRIX_H_|POCKETKALDI_VECTOR_H_)[^\n]*\n(.*?)#endif\n", r"\2", hdr,
                      flags=re.S)
    assert stripped != hdr
    (src / "pk_mi355.h").write_text(stripped)
    r = _gxx(src, ["-fsyntax-only", "decoder.cc"])
    assert r.returncode != 0 and re.search(r"redefinition of .struct pk_matrix_t.", r.stderr), r.stderr


def test_apply_binding_refuses_a_tree_it_does_not_recognise(tmp_path):
    root = _view(tmp_path)
    cc = root / "src" / "pocketkaldi.cc"
    text = cc.read_text().replace("delete self->am;", "/* gone */")
    os.unlink(cc)
    cc.write_text(text)
    r = subprocess.run([sys.executable, os.path.join(BINDING, "apply_binding.py"), str(root)],
                       capture_output=True, text=True)
    assert r.returncode != 0 and "does not look like the reference" in (r.stdout + r.stderr)


def test_real_reference_decoder_library_binds_to_the_product_library():
    """oracle/_ref/libpkref_decoder.so = the reference's decoder.cc + fst.cc + util.cc + hashtable.cc +
    strlcpy.cc compiled against the binding (oracle/Makefile: ref_decoder).  Its only undefined pk_*
    symbols are the two decoder.cc calls, libpk_mi355.so defines both, and the struct layout the
    reference-side translation units see is the ABI's 24 bytes / offset 16."""
    from pocketkaldi_amd import build
    lib = os.path.join(ROOT, "oracle", "_ref", "libpkref_decoder.so")
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "ref_decoder"], check=True, capture_output=True)
    out = subprocess.run(["nm", "-D", "--undefined-only", lib], capture_output=True, text=True, check=True).stdout
    und = {l.split()[-1] for l in out.splitlines() if " pk_" in l}
    assert und == {"pk_decodable_islastframe", "pk_decodable_loglikelihood"}
    assert und <= _exports(build.LIB)
    ctypes.CDLL(build.LIB, mode=ctypes.RTLD_GLOBAL)
    dec = ctypes.CDLL(lib)
    assert dec.pkref_sizeof_decodable() == 24 and dec.pkref_offsetof_decodable_am() == 16
