#!/usr/bin/env python3
"""Model-file fixtures written by the REFERENCE's own converters.

Run in the build container (needs /root/reference; the scripts never travel, their outputs do):

    python3 tests/golden/make_model_fixtures.py

1. writes OUR inputs -- a tiny Kaldi-text nnet2 acoustic model, a transition-id -> pdf listing
   and a text CMVN-stats matrix, all seeded synthetic numbers in the formats Kaldi's
   nnet-am-copy / extract_id2pdf / copy-matrix print -- under tests/golden/refmodel/;
2. runs the reference's converters on them, unmodified, as the reference's users do:
       tool/convert_am.py          -> refmodel.nnet + refmodel.prior   (NNT0/LAY0/MAT0/VEC0)
       tool/convert_trans.py       -> refmodel_tid2pdf.bin             (VEC0 of int32)
       tool/convert_cmvn_stats.py  -> refmodel_cmvn.bin                (VEC0 of 41 floats)
3. writes the key = value model file the reference's pk_load reads (pocketkaldi.cc:72-144).

The tests rebuild the model in memory from the TEXT inputs and require the product's reader
(pk_mi355_load / pk_mi355_am_read) and the oracle's reader to agree with it bit for bit.
"""
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "refmodel")
REF_TOOLS = "/root/reference/tool"

FEAT, LEFT, RIGHT = 40, 2, 1          # splice 4 x 40 = 160 inputs
HIDDEN, PDFS = 24, 18
NUM_TIDS = 57


def fmt(v):
    return "%.7g" % v


def matrix_text(m):
    rows = ["  " + " ".join(fmt(x) for x in row) for row in m]
    return "[\n" + "\n".join(rows) + " ]"


def main():
    if not os.path.isdir(REF_TOOLS):
        raise SystemExit("needs the reference tree at /root/reference")
    os.makedirs(OUT, exist_ok=True)
    rng = np.random.default_rng(0xA11CE)
    din = FEAT * (LEFT + 1 + RIGHT)
    W1 = rng.standard_normal((HIDDEN, din)) * np.sqrt(2.0 / din)
    b1 = rng.standard_normal(HIDDEN) * 0.1
    W2 = rng.standard_normal((HIDDEN, HIDDEN)) * np.sqrt(2.0 / HIDDEN)
    b2 = rng.standard_normal(HIDDEN) * 0.1
    W3 = rng.standard_normal((PDFS, HIDDEN)) * np.sqrt(2.0 / HIDDEN)
    b3 = rng.standard_normal(PDFS) * 0.1
    prior = rng.uniform(0.5, 1.5, PDFS)
    prior /= prior.sum()

    def affine(W, b):
        return ("<AffineComponentPreconditionedOnline> <LearningRate> 0.001 <LinearParams> %s\n"
                "<BiasParams> [ %s ]\n<RankIn> 20 <RankOut> 80 </AffineComponentPreconditionedOnline>\n"
                % (matrix_text(W), " ".join(fmt(x) for x in b)))

    ctx = " ".join(str(c) for c in range(-LEFT, RIGHT + 1))
    am_txt = ("<TransitionModel> (not read by the converter) </TransitionModel>\n"
              "<Nnet> <NumComponents> 8 <Components>\n"
              "<SpliceComponent> <InputDim> %d <Context> [ %s ]\n <ConstComponentDim> 0 </SpliceComponent>\n" % (FEAT, ctx)
              + affine(W1, b1) +
              "<RectifiedLinearComponent> <Dim> %d </RectifiedLinearComponent>\n" % HIDDEN +
              "<NormalizeComponent> <Dim> %d </NormalizeComponent>\n" % HIDDEN
              + affine(W2, b2) +
              "<RectifiedLinearComponent> <Dim> %d </RectifiedLinearComponent>\n" % HIDDEN
              + affine(W3, b3) +
              "<SoftmaxComponent> <Dim> %d </SoftmaxComponent>\n" % PDFS +
              "</Components> </Nnet>\n[ %s ]\n" % " ".join(fmt(x) for x in prior))
    with open(os.path.join(OUT, "refmodel_am.txt"), "w") as f:
        f.write(am_txt)

    tid2pdf = rng.integers(0, PDFS, NUM_TIDS + 1)
    with open(os.path.join(OUT, "refmodel_id2pdf.txt"), "w") as f:
        f.write("%d\n%d\n" % (PDFS, NUM_TIDS))
        for tid in range(1, NUM_TIDS + 1):
            f.write("%d %d\n" % (tid, tid2pdf[tid]))

    count = 123456.0
    sums = count * (11.5 + 0.07 * np.arange(FEAT) + rng.standard_normal(FEAT) * 0.3)
    sq = count * (140.0 + rng.standard_normal(FEAT + 1))
    with open(os.path.join(OUT, "refmodel_cmvn.txt"), "w") as f:
        f.write(" [\n  " + " ".join("%.6f" % x for x in list(sums) + [count]) + "\n  "
                + " ".join("%.6f" % x for x in sq) + " ]\n")

    def run(tool, *args):
        subprocess.check_call([sys.executable, os.path.join(REF_TOOLS, tool)] + list(args), stdout=subprocess.DEVNULL)

    run("convert_am.py", os.path.join(OUT, "refmodel_am.txt"), os.path.join(OUT, "refmodel"))
    run("convert_trans.py", os.path.join(OUT, "refmodel_id2pdf.txt"), os.path.join(OUT, "refmodel_tid2pdf.bin"))
    run("convert_cmvn_stats.py", os.path.join(OUT, "refmodel_cmvn.txt"), os.path.join(OUT, "refmodel_cmvn.bin"))

    with open(os.path.join(OUT, "refmodel.conf"), "w") as f:
        f.write("# model file in the form pk_load reads (pocketkaldi.cc:72-144); the decoder's keys are\n"
                "# present but not used by acoustic scoring\n"
                "fst = HCLG.pfst\nsymbol_table = words.bin\n"
                "cmvn_stats = refmodel_cmvn.bin\nnnet = refmodel.nnet\nprior = refmodel.prior\n"
                "left_context = %d\nright_context = %d\nnum_pdfs = %d\ntid2pdf = refmodel_tid2pdf.bin\n"
                % (LEFT, RIGHT, PDFS))
    print("wrote", sorted(os.listdir(OUT)))


if __name__ == "__main__":
    main()
