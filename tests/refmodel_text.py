"""Parser for OUR text inputs of tests/golden/refmodel/ (written by make_model_fixtures.py in the
formats Kaldi's tools print): rebuilds the model in memory, so that the files the REFERENCE's
converters made from the same text can be checked against it.  Independent of convert_am.py:
a tokenizer over the <Tag> ... </Tag> stream, floats rounded to float32 exactly once."""
import os
import re

import numpy as np

DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "refmodel")


def _numbers(text):
    return [float(t) for t in text.replace("[", " ").replace("]", " ").split()]


def load_text_model():
    """-> (layers, prior, left, right, tid2pdf, cmvn41) from the text files."""
    txt = open(os.path.join(DIR, "refmodel_am.txt")).read()
    body = txt[txt.index("<Components>") + len("<Components>"):txt.index("</Components>")]
    layers, left, right = [], 0, 0
    for m in re.finditer(r"<(\w+)>(.*?)</\1>", body, re.S):
        tag, inner = m.group(1), m.group(2)
        if tag == "SpliceComponent":
            ctx = [int(v) for v in _numbers(inner[inner.index("<Context>") + 9:inner.index("]")])]
            left, right = -ctx[0], ctx[-1]
        elif tag == "AffineComponentPreconditionedOnline":
            lin = inner[inner.index("<LinearParams>") + 14:inner.index("<BiasParams>")]
            rows = [_numbers(r) for r in lin.strip().strip("[]").split("\n") if r.strip(" []")]
            bias = _numbers(inner[inner.index("<BiasParams>") + 12:inner.index("]", inner.index("<BiasParams>"))])
            layers.append(("linear", np.array(rows, dtype=np.float64).astype(np.float32),
                           np.array(bias, dtype=np.float64).astype(np.float32)))
        elif tag == "RectifiedLinearComponent":
            layers.append(("relu",))
        elif tag == "NormalizeComponent":
            layers.append(("normalize",))
        elif tag == "SoftmaxComponent":
            layers.append(("softmax",))
        else:
            raise ValueError(tag)
    prior = np.array(_numbers(txt[txt.index("</Nnet>") + 7:]), dtype=np.float64).astype(np.float32)
    lines = open(os.path.join(DIR, "refmodel_id2pdf.txt")).read().split("\n")
    tid2pdf = np.zeros(int(lines[1]) + 1, dtype=np.int32)
    for ln in lines[2:]:
        if ln.strip():
            t, p = ln.split()
            tid2pdf[int(t)] = int(p)
    cm = _numbers(open(os.path.join(DIR, "refmodel_cmvn.txt")).read())
    cmvn41 = np.array(cm[:41], dtype=np.float64).astype(np.float32)
    return layers, prior, left, right, tid2pdf, cmvn41
