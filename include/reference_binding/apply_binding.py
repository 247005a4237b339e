#!/usr/bin/env python3
"""apply_binding.py -- bind a pocketkaldi checkout to libpk_mi355.so.

    python include/reference_binding/apply_binding.py <pocketkaldi checkout> [--dry-run]

What it does to the checkout (nothing else is touched; decoder.cc, decoder.h, pocketkaldi.h, fst.*,
symbol_table.*, main.cc stay as they are):

 (a) src/decodable.h  <- include/reference_binding/decodable.h   (the handle type behind
     pk_decodable_t::am; same four functions, decodable.h:20-41)
     src/pk_mi355.h   <- include/pk_mi355.h
 (b) src/pocketkaldi.cc: two edits, both about OWNERSHIP of the model handle
       pk_destroy (pocketkaldi.cc:45): `delete self->am`            -> pk_mi355_am_destroy(self->am)
       pk_load    (pocketkaldi.cc:112-113): `new AcousticModel()` + `am->Read(conf)`
                                                                    -> pk_mi355_load(filename, ...)
     pk_process (pocketkaldi.cc:176-248) is unchanged: it keeps calling pk_decodable_init /
     Decoder::Decode / pk_decodable_destroy, which now score on the GPU.
 (c) Makefile.am: the sources the library replaces leave libpocketkaldi_a_SOURCES
     (Makefile.am:12-30) and -lpk_mi355 joins the link.

The edits are located by the statements they replace, so the script refuses (exit 2) a checkout
whose pocketkaldi.cc does not contain them exactly once.  tests/test_reference_binding.py runs this
script on a scratch view of /root/reference and compiles the result.
"""
import os
import re
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))

# sources whose work libpk_mi355.so does (Makefile.am:12-30).  gemm.cc / gemm_haswell.cc stay in the
# archive because matrix.cc's MatMat (matrix.cc:418-436) still names GEMM<float>; nothing on the path
# calls it any more.  fbank.cc / cmvn.cc / srfft.cc stay while pk_process keeps the CPU front-end
# (INTEGRATION.md 2(c) replaces that stage as well).
REPLACED_SOURCES = ["src/decodable.cc", "src/am.cc", "src/nnet.cc"]

DESTROY_OLD = re.compile(r"^(\s*)delete\s+self->am\s*;\s*$", re.M)
DESTROY_NEW = r"\1pk_mi355_am_destroy(self->am);  // libpk_mi355: the handle is not a C++ object"

LOAD_OLD = re.compile(
    r"^(\s*)self->am\s*=\s*new\s+AcousticModel\s*\(\s*\)\s*;\s*\n\s*status_vn\s*=\s*self->am->Read\s*\(\s*conf\s*\)\s*;\s*$",
    re.M)
LOAD_NEW = (
    r"\1{  // libpk_mi355: nnet / prior / tid2pdf / contexts / num_pdfs (am.cc:22-62) into HBM\n"
    r"\1  float cmvn41_unused[41];\n"
    r"\1  if (pk_mi355_load(filename, PK_MI355_PRECISION_F32, &self->am, cmvn41_unused) != 0)\n"
    r"\1    status_vn = pocketkaldi::Status::IOError(pk_mi355_last_error());\n"
    r"\1}")


def edit_pocketkaldi_cc(text):
    out, n1 = DESTROY_OLD.subn(DESTROY_NEW, text)
    out, n2 = LOAD_OLD.subn(LOAD_NEW, out)
    if n1 != 1 or n2 != 1:
        raise SystemExit("apply_binding: pocketkaldi.cc does not look like the reference's "
                         "(delete self->am: %d match(es), new AcousticModel + Read: %d)" % (n1, n2))
    return out


def edit_makefile_am(text):
    out = text
    for src in REPLACED_SOURCES:
        out, n = re.subn(r"^[ \t]*%s[ \t]*\\\n" % re.escape(src), "", out, flags=re.M)
        if n != 1:
            raise SystemExit("apply_binding: Makefile.am does not list %s exactly once" % src)
    out, n = re.subn(r"^(pocketkaldi_LDADD\s*=.*)$", r"\1 -lpk_mi355", out, flags=re.M)
    if n != 1:
        raise SystemExit("apply_binding: pocketkaldi_LDADD not found in Makefile.am")
    return out


def _rewrite(path, new_text, dry):
    """Replace `path` by a regular file holding new_text (a symlinked view stays a view elsewhere)."""
    if dry:
        print("would rewrite", path)
        return
    if os.path.islink(path):
        os.unlink(path)
    with open(path, "w") as f:
        f.write(new_text)


def main(argv):
    if len(argv) < 2:
        raise SystemExit(__doc__)
    root, dry = argv[1], "--dry-run" in argv[2:]
    src = os.path.join(root, "src")
    cc = os.path.join(src, "pocketkaldi.cc")
    new_cc = edit_pocketkaldi_cc(open(cc).read())
    mk = os.path.join(root, "Makefile.am")
    new_mk = edit_makefile_am(open(mk).read()) if os.path.exists(mk) else None
    _rewrite(cc, new_cc, dry)
    if new_mk is not None:
        _rewrite(mk, new_mk, dry)
    for name, origin in (("decodable.h", os.path.join(HERE, "decodable.h")),
                         ("pk_mi355.h", os.path.join(HERE, "..", "pk_mi355.h"))):
        dst = os.path.join(src, name)
        if dry:
            print("would install", dst)
            continue
        if os.path.islink(dst) or os.path.exists(dst):
            os.unlink(dst)
        shutil.copyfile(origin, dst)
    gone = os.path.join(src, "decodable.cc")
    if not dry and (os.path.islink(gone) or os.path.exists(gone)):
        os.unlink(gone)
    print("bound %s to libpk_mi355.so: decodable.h replaced, pocketkaldi.cc pk_load/pk_destroy edited%s"
          % (root, ", Makefile.am updated" if new_mk is not None else ""))


if __name__ == "__main__":
    main(sys.argv)
