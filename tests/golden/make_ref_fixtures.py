#!/usr/bin/env python3
"""Extract the reference's own known-answer DATA into fixtures under tests/golden/.

Run in the build container (needs /root/reference).  Only numbers are taken:
  * test/srfft_test.cc:11-271  -- the 128-point real-FFT input/output pair
  * test/nnet_test.cc:25-109   -- the four layer micro-tests (restated by hand below)
  * test/data/*.wav, *.txt, cmvn_stats.bin are copied verbatim (data files).
And outputs of the REAL reference code that builds here (oracle/_ref: srfft.cc,
gemm.cc, gemm_haswell.cc) on our own seeded inputs:
  * ref_srfft512.npz  -- 8 frames in, packed spectra out
  * ref_sgemm.npz     -- small A, B and the reference GEMM<float>::Gemm product
"""
import json
import os
import re
import shutil
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REPO)


def parse_float_array(text, name):
    m = re.search(r"float\s+%s\s*\[[^\]]*\]\s*=\s*\{([^}]*)\}" % name, text, re.S)
    return [float(tok.rstrip("f")) for tok in re.findall(r"[-+0-9.eE]+f?", m.group(1)) if tok.strip("f")]


def main():
    src = open(os.path.join(REF, "test/srfft_test.cc")).read()
    data = parse_float_array(src, "data")
    fft = parse_float_array(src, "fft_data")
    assert len(data) == 128 and len(fft) == 128
    known = {
        "srfft128": {"input": data, "output": fft, "tol_abs": 1e-6,
                     "source": "test/srfft_test.cc:11-284"},
        # test/nnet_test.cc:23-110 (numbers restated; W is [out][in] like the model file)
        "linear": {"W": [[0.1, 0.8, 0.9], [0.4, 0.2, 0.7], [0.2, 0.1, 0.1], [0.4, 0.3, 0.2]],
                   "b": [0.1, -0.1, 0.2, -0.2], "x": [0.3, -0.1, 0.9],
                   "y": [0.86, 0.63, 0.34, 0.07], "tol_abs": 1e-6,
                   "source": "test/nnet_test.cc:23-55"},
        "softmax": {"x": [0.3, -0.1, 0.9, 0.2],
                    "y": [0.2274135, 0.15243983, 0.41437442, 0.20577225], "tol_abs": 1e-6,
                    "source": "test/nnet_test.cc:57-73"},
        "relu": {"x": [0.3, -0.1, 0.9, 0.2], "y": [0.3, 0.0, 0.9, 0.2], "tol_abs": 1e-6,
                 "source": "test/nnet_test.cc:75-92"},
        "normalize": {"x": [0.3, -0.1, 0.9, 0.2], "sum_sq": 4.0, "tol_abs": 1e-4,
                      "source": "test/nnet_test.cc:94-110"},
        "wav_hello": {"num_samples": 7802, "first": [12, 38, -8], "num_frames": 47,
                      "source": "test/fbank_test.cc:15-56, test/data/en-us-hello.wav"},
    }
    with open(os.path.join(HERE, "ref_known_answers.json"), "w") as f:
        json.dump(known, f, indent=1)

    for name in ("en-us-hello.wav", "en-us-cat.wav", "cmvn_stats.bin",
                 "fbankmat_en-us-hello.wav.txt", "fbankcmvnmat_en-us-hello.wav.txt"):
        dst = os.path.join(HERE, name)
        if not os.path.exists(dst):
            shutil.copyfile(os.path.join(REF, "test/data", name), dst)

    from oracle import oracle as O
    assert O.have_ref()
    rng = np.random.default_rng(0xF17)
    frames = (rng.standard_normal((8, 512)) * np.array([1, 10, 100, 1e3, 1e4, 3e4, 1e-3, 0])[:, None]
              ).astype(np.float32)
    spectra = np.stack([O.ref_srfft(fr) for fr in frames])
    np.savez(os.path.join(HERE, "ref_srfft512.npz"), frames=frames, spectra=spectra)

    gem = {}
    for i, (m, n, k) in enumerate([(7, 16, 440), (19, 12, 1024), (6, 10, 2048), (13, 5, 513)]):
        A = rng.standard_normal((m, k)).astype(np.float32)
        B = (rng.standard_normal((k, n)) * 0.05).astype(np.float32)
        gem["A%d" % i] = A
        gem["B%d" % i] = B
        gem["C%d" % i] = O.ref_sgemm(A, B)
    np.savez(os.path.join(HERE, "ref_sgemm.npz"), **gem)
    print("fixtures written to", HERE)


if __name__ == "__main__":
    main()
