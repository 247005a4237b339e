"""Stress of the fused tail's hand-off at the benchmark's size (runs on the GPU box).

The fp32 last layer's launch turns its own logits into log-likelihoods (gemm.hip, TAIL variant): a row tile's
owner reads what up to 24 / 63 other workgroups -- on other XCDs, behind other L2s -- stored.  A visibility bug there
would be rare and size-dependent, and the seeded fuzz tests mostly sit under the 384-tile threshold of the fused form.
So: score the full batch with the tail as a launch of its own (same arithmetic; PK_MI355_FUSED_TAIL_MIN_TILES lifts
the threshold out of reach), keep that result on the device, then score N times with the fused form and require
every pass to be bit-identical over the whole [rows][pdfs] block.

    python3 tools/fused_tail_stress.py [passes_S] [passes_W]
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import pocketkaldi_amd as pk                    # noqa: E402
from pocketkaldi_amd import dist as pkdist      # noqa: E402
from pocketkaldi_amd import synth               # noqa: E402


def run(model, batch, seconds, passes):
    layers, prior, L, R = synth.model(model)
    waves = [synth.utterance(u, seconds) for u in range(batch)]
    ns = [len(w) for w in waves]
    dev = torch.device("cuda:0")
    pcm = torch.from_numpy(np.concatenate(waves)).to(dev)

    def scorer():
        am_ = pk.AcousticModel(layers, prior, L, R)       # (the PK_MI355_* switches are read when a model is made)
        bs_ = pk.BatchScorer(am_, synth.global_cmvn_stats(), batch, int(sum(ns)))
        bs_.set_waves_device(pcm.data_ptr(), ns, keep_alive=pcm)
        n_ = am_.num_pdfs()
        first = bs_.loglik_device(0)
        nbytes_ = (bs_.loglik_device(batch - 1) - first) + bs_.num_frames(batch - 1) * n_ * 4
        return am_, bs_, n_, nbytes_, pkdist.alias_device_bytes(first, nbytes_, dev)

    os.environ["PK_MI355_FUSED_TAIL_MIN_TILES"] = "2000000000"      # the tail as a launch of its own
    am0, bs0, n, nbytes, view0 = scorer()
    del os.environ["PK_MI355_FUSED_TAIL_MIN_TILES"]
    bs0.score(0.1, sync=True)
    want = view0.clone()
    bs0.score(0.1, sync=True)
    assert torch.equal(view0, want), "the stand-alone form is not even reproducible"
    bs0.enable_timing(True)
    bs0.score(0.1, sync=True)
    assert bs0.timing()["tail"][1] > 0, "the comparison run was meant to launch the tail on its own"
    del view0
    bs0.close()
    am0.close()
    am, bs, n, nbytes, view = scorer()

    bad = 0
    t0 = time.perf_counter()
    for p in range(passes):
        view.zero_()                             # a row nobody wrote must not pass on the last pass's bytes
        bs.score(0.1, sync=True)
        if not torch.equal(view, want):
            bad += 1
            diff = (view != want).nonzero()
            print("  pass %d: %d bytes differ, first at byte %d (row %d)" % (p, len(diff), int(diff[0]), int(diff[0]) // (4 * n)))
        if p % 50 == 49:
            print("  %s: %d passes, %d bad, %.1f s" % (model, p + 1, bad, time.perf_counter() - t0), flush=True)
    print("model %s, %d x %g s (%d frames x %d pdfs, %.2f GB compared per pass): %d passes of the fused tail, %d differ from "
          "the stand-alone tail" % (model, batch, seconds, bs.total_frames(), n, nbytes / 1e9, passes, bad))
    bs.close()
    am.close()
    return bad


if __name__ == "__main__":
    ps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    pw = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    batch = int(os.environ.get("PK_STRESS_BATCH", "256"))          # (ragged shapes: PK_STRESS_BATCH=251 PK_STRESS_SECONDS=9.73,
    seconds = float(os.environ.get("PK_STRESS_SECONDS", "10"))     #  several launches per pass: PK_MI355_CHUNK=65536)
    bad = run("S", batch, seconds, ps)
    bad += run("W", batch, seconds, pw)
    sys.exit(1 if bad else 0)
