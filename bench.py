#!/usr/bin/env python3
"""bench.py -- acoustic frames/s (fbank -> CMVN -> nnet log-likelihoods) on N MI355X.

    python bench.py --gpus N --steps K --warmup W           (N > 1: starts the line below itself, as a child)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the whole hot path over one batch of synthetic utterances
whose PCM is already resident in HBM.  Workload (BASELINE.json configs[2] per GPU,
configs[3] at N = 8): 256 utterances x 10 s of 16 kHz audio per GPU (255 488
frames), model S = 440 -> 4 x 1024 ReLU -> 3000 softmax, fp32 MFMA.  Utterances are
sharded over ranks (u -> rank u mod N), weights are broadcast once from rank 0 over
RCCL, there is no data-path collective: scaling is weak.

Rank 0 prints ONE JSON line with the contract keys plus "roofline" (the affine-GEMM
kernel, timed live with HIP events on the stream it runs on) and "cpu_baseline"
(the oracle port timed on this box's host cores, bounded sample, N = 1 only).
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

FP32_MFMA_PEAK_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
FP16_MFMA_PEAK_TFLOPS = 2500.0  # same guide, "Peak BF16/FP16 MFMA ~2.5 PF dense"
# what "bit-identical" is measured against: the cblas-dependent reference files
# cannot be built in this image, so the whole-path pin is the oracle, itself pinned as far as the reference allows
PARITY_PIN = ("bit-identical to the oracle restatement (oracle/pk_oracle.c; bitwise-pinned to the reference's srfft.cc and gemm.cc + gemm_haswell.cc compiled here, fbank/CMVN pinned to the reference's Kaldi dumps at 3e-5, layers to nnet_test.cc's answers)")


def measured_traffic():
    """HBM bytes per GEMM launch from the newest committed PMC passes (profiles/rNN_pmc_traffic.json):
    bench.py cannot run rocprofv3 around itself, so the number is the OFFLINE measurement of this
    same command (tools/profile_gpu.sh), and the bench line says so in roofline.traffic_source.
    The file records the hash of the GEMM kernel's sources it was measured on; when the library
    running now was built from other GEMM sources the figure is NOT reported (traffic = null and
    the reason instead) -- a stale number never travels silently."""
    import glob
    from pocketkaldi_amd import build as pkbuild
    now = pkbuild.gemm_source_hash()
    for path in sorted(glob.glob(os.path.join(REPO, "profiles", "r*_pmc_traffic.json")), reverse=True):
        try:
            with open(path) as f:
                d = json.load(f)
            value = d["gemm_avg_hbm_bytes_per_launch"]
        except Exception:
            continue
        rel = os.path.relpath(path, REPO)
        then = (d.get("measured_on") or {}).get("gemm_source_hash")
        if then != now:
            return None, ("%s was measured on GEMM sources %s, this library is built from %s: not reported "
                          "(re-run tools/profile_gpu.sh S + tools/make_profiles.py)" % (rel, then, now))
        return value, ("%s (offline rocprofv3 PMC passes of this command on the same GEMM sources, gemm_source_hash %s; "
                       "FETCH_SIZE x 2 + WRITE_SIZE per launch; not measured in this run)" % (rel, now))
    return None, "no profiles/r*_pmc_traffic.json"


def cpu_baseline(model_name, seconds, budget_s):
    """Oracle port (oracle/pk_oracle*.c: scalar fbank/CMVN + blocked AVX2 SGEMM of the
    reference's class), ONE thread, on a sample of the same workload bounded by TIME: utterances
    of the bench workload, in order, until budget_s seconds of CPU have been spent."""
    from oracle import oracle as O
    from pocketkaldi_amd import synth
    layers, prior, L, R = synth.model(model_name)
    nn = O.Nnet(layers)
    fb = O.Fbank()
    g = synth.global_cmvn_stats()
    frames, num_utts, dt = 0, 0, 0.0
    while dt < budget_s and num_utts < 4096:
        w = synth.utterance(num_utts, seconds)      # generation is not timed
        t0 = time.perf_counter()
        feats = O.cmvn(g, fb.compute(w))
        nn.am_compute(feats, prior, L, R, 0.1)
        dt += time.perf_counter() - t0
        frames += feats.shape[0]
        num_utts += 1
    out = {"value": frames / dt, "unit": "frames/s", "cores": 1, "kind": "port", **cpu_identity(),
           "sample": "%d utterances x %.0f s (%d frames), model %s, whole path, %.1f s of CPU (time-capped at %.0f s)"
                     % (num_utts, seconds, frames, model_name, dt, budget_s)}
    # The reference's own SGEMM (gemm.cc + gemm_haswell.cc, built into oracle/_ref/ where the
    # reference tree was available) beside the port's, on the layer shapes of one utterance:
    # the port is the thing timed above, this shows it runs at the reference kernel's speed.
    try:
        if O.have_ref():
            rng = np.random.default_rng(0)
            T = feats.shape[0]
            shapes = [(l[1].shape[1], l[1].shape[0]) for l in layers if l[0] == "linear"]
            flops, t_ref, t_port = 0.0, 0.0, 0.0
            for K, N in shapes:
                A = rng.standard_normal((T, K)).astype(np.float32)
                B = rng.standard_normal((K, N)).astype(np.float32)
                for fn, acc in ((O.ref_sgemm, "ref"), (O.sgemm, "port")):
                    fn(A, B)
                    t1 = time.perf_counter()
                    for _ in range(3):
                        fn(A, B)
                    d = (time.perf_counter() - t1) / 3
                    if acc == "ref":
                        t_ref += d
                    else:
                        t_port += d
                flops += 2.0 * T * K * N
            out["sgemm_gflops"] = {"reference_gemm_haswell": flops / t_ref / 1e9, "port": flops / t_port / 1e9,
                                   "shapes": "T = %d rows through the %d affine layers of model %s, 1 core"
                                             % (T, len(shapes), model_name)}
            # kind "reference" for the part of the path that IS the reference's own compiled code here: the affine layers
            # (> 99 % of the CPU path's time) through gemm.cc + gemm_haswell.cc, frames per second of one core
            out["nnet_affine_layers_reference_gemm"] = {"value": T / t_ref, "unit": "frames/s", "cores": 1, "kind": "reference",
                                                        "sample": "the %d affine layers of model %s on %d frames, the reference's own "
                                                                  "SGEMM (oracle/_ref), mean of 3 runs" % (len(shapes), model_name, T)}
    except Exception as e:          # the reference build is optional test infrastructure
        out["sgemm_gflops"] = {"error": str(e)}
    return out


def cpu_baseline_all_cores(model_name, seconds, budget_s):
    """the same port with utterance-level parallelism over the host cores
    of this box's share (one oracle instance per thread; the C calls release the GIL); every
    thread works until budget_s seconds of wall time have passed."""
    import threading
    from oracle import oracle as O
    from pocketkaldi_amd import synth
    # a 1-GPU box's CPU share is 16 cores whatever the affinity mask says
    cores = min(len(os.sched_getaffinity(0)), int(os.environ.get("PK_BENCH_CPU_THREADS", "16")))
    layers, prior, L, R = synth.model(model_name)
    g = synth.global_cmvn_stats()
    waves = [synth.utterance(u, seconds) for u in range(8)]
    frames = [0] * cores
    objs = [(O.Nnet(layers), O.Fbank()) for _ in range(cores)]
    deadline = [0.0]

    def work(i):
        nn, fb = objs[i]
        k = 0
        while time.perf_counter() < deadline[0]:
            feats = O.cmvn(g, fb.compute(waves[k % len(waves)]))
            nn.am_compute(feats, prior, L, R, 0.1)
            frames[i] += feats.shape[0]
            k += 1

    th = [threading.Thread(target=work, args=(i,)) for i in range(cores)]
    t0 = time.perf_counter()
    deadline[0] = t0 + budget_s
    for t in th:
        t.start()
    for t in th:
        t.join()
    dt = time.perf_counter() - t0
    return {"value": sum(frames) / dt, "unit": "frames/s", "cores": cores, "kind": "port", **cpu_identity(),
            "sample": "%d threads, %.0f s utterances (%d frames), model %s, whole path, %.1f s wall (time-capped at %.0f s)"
                      % (cores, seconds, sum(frames), model_name, dt, budget_s)}


def host_endpoints(pk, synth, am, batch, seconds, n_batches=8):
    """SURVEY.md section 8d's two other end-points of the same workload (never `value`):
      from_pinned_host : int16 PCM in page-locked host memory -> log-likelihoods complete in HBM;
      host_complete    : ... -> every utterance's log_prob on the host as decoder.cc consumes it
                         (pk_mi355_batch_fetch_all: one transfer per batch into a page-locked arena).
    Two batches in flight: the upload of batch k+1 and the result transfer of batch k-1 run on
    their own streams under the scoring of batch k."""
    g = synth.global_cmvn_stats()
    waves = [synth.utterance(u, seconds).astype(np.int16) for u in range(batch)]
    ns = [len(w) for w in waves]
    pin = pk.pinned_i16(int(sum(ns)))
    pin[:] = np.concatenate(waves)
    pair = [pk.BatchScorer(am, g, batch, int(sum(ns))) for _ in range(2)]
    out = {}
    for name, fetch in (("from_pinned_host", False), ("host_complete", True)):
        for s_ in pair:                              # untimed: first use allocates buffers / the arena
            s_.set_waves_i16_raw(pin, ns)
            s_.score(0.1, sync=True)
            if fetch:
                s_.fetch_all()
        t0 = time.perf_counter()
        done = 0
        busy = [False, False]
        for k in range(n_batches + 2):
            s_ = pair[k % 2]
            if busy[k % 2]:
                s_.synchronize()                     # scored (and, with fetch, on the host)
                done += s_.total_frames()
                busy[k % 2] = False
            if k < n_batches:
                s_.set_waves_i16_raw(pin, ns)        # H2D, 320 B/frame
                s_.score(0.1, sync=False)
                if fetch:
                    s_.fetch_all(sync=False)         # D2H, 4 * num_pdfs B/frame
                busy[k % 2] = True
        dt = time.perf_counter() - t0
        out[name] = {"value": done / dt, "unit": "frames/s", "ms_per_batch": dt / n_batches * 1e3,
                     "batches": n_batches, "in_flight": 2}
    out["from_pinned_host"]["start"] = "int16 PCM in page-locked host memory"
    out["from_pinned_host"]["end"] = "log-likelihoods complete in HBM"
    out["host_complete"]["start"] = "int16 PCM in page-locked host memory"
    out["host_complete"]["end"] = "pk_decodable_t.log_prob of every utterance on the host (page-locked arena views)"
    out["host_complete"]["d2h_gb_per_s"] = out["host_complete"]["value"] * 4.0 * am.num_pdfs() / 1e9
    for s_ in pair:
        s_.close()
    return out


def other_configs(pk, synth, torch, steps):
    """BASELINE.json configs[1] (one utterance, model S) and configs[4] (wide model on the fp16
    matrix cores), device-complete like the headline; supplementary lines, N = 1 only."""
    out = {}
    g = synth.global_cmvn_stats()
    # configs[1]: latency of ONE 10 s utterance through the whole path
    layers, prior, L, R = synth.model("S")
    am = pk.AcousticModel(layers, prior, L, R)
    w = synth.utterance(0, 10.0)
    pcm = torch.from_numpy(w).cuda()
    bs = pk.BatchScorer(am, g, 1, len(w))
    bs.set_waves_device(pcm.data_ptr(), [len(w)], keep_alive=pcm)
    for _ in range(10):
        bs.score(0.1, sync=True)
    n = 200
    t0 = time.perf_counter()
    for _ in range(n):
        bs.score(0.1, sync=True)          # synchronous: this is a latency, not a pipelined rate
    dt = (time.perf_counter() - t0) / n
    bs.enable_timing(True)
    bs.score(0.1, sync=True)
    stage_us = {k: v[0] * 1e3 for k, v in bs.timing().items()}      # HIP events around every launch (adds ~1 us each)
    launches = sum(v[1] for v in bs.timing().values())
    bs.enable_timing(False)
    # throughput with several single-utterance pipelines in flight (one stream each): the launch
    # ramps of one utterance's kernels run under the other utterances' MFMAs
    more = [pk.BatchScorer(am, g, 1, len(w)) for _ in range(2)]
    for k, b in enumerate(more):
        b.set_waves([synth.utterance(1 + k, 10.0)])
    pipelined = {}
    for nflight in (2, 3):
        group = [bs] + more[:nflight - 1]
        for _ in range(5):
            for b in group:
                b.score(0.1, sync=False)
        for b in group:
            b.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            for b in group:
                b.score(0.1, sync=False)
        for b in group:
            b.synchronize()
        pipelined["%d_in_flight_ms_per_utterance" % nflight] = (time.perf_counter() - t0) / (n * nflight) * 1e3
    out["configs[1] one utterance, model S, f32"] = {
        "ms_per_utterance": dt * 1e3, "frames_per_s": bs.total_frames() / dt, "frames": bs.total_frames(),
        "gemm_tflops": am.flops_per_frame() * bs.total_frames() / dt / 1e12,
        "mfma_bound_ms": am.flops_per_frame() * bs.total_frames() / (FP32_MFMA_PEAK_TFLOPS * 1e12) * 1e3,
        "launches_per_utterance": launches, "stage_us": stage_us,
        "pipelined": pipelined,
        "note": "ms_per_utterance is the synchronous latency of ONE utterance (score + stream sync per utterance); it is "
                "launch-bound: 8 kernels whose fixed cost (dispatch, first DMA, epilogue: ~4.6 us for a K = 16 GEMM, "
                "tools/ramp_probe.py) and 1.3 us boundaries are not covered by other work; 'pipelined' is the rate with "
                "2-3 independent single-utterance pipelines in flight"}
    for b in more:
        b.close()
    bs.close()
    # configs[4]: wide model, fp16 matrix cores (split-fp16 operands), 256 utterances
    layers, prior, L, R = synth.model("W")
    amw = pk.AcousticModel(layers, prior, L, R, precision="f16x3")
    B = 256
    waves = [synth.utterance(u, 10.0) for u in range(B)]
    ns = [len(x) for x in waves]
    pcm = torch.from_numpy(np.concatenate(waves)).cuda()
    bw = pk.BatchScorer(amw, g, B, int(sum(ns)))
    bw.set_waves_device(pcm.data_ptr(), ns, keep_alive=pcm)
    bw.calibrate()                               # the deployment flow of the fp16 modes
    bw.score(0.1, sync=True)
    bw.enable_timing(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        bw.score(0.1, sync=False)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    gm = bw.timing()["gemm"][0]
    alg = amw.flops_per_frame() * bw.total_frames() / (gm * 1e-3) / 1e12
    out[W_F16X3_KEY] = {
        "frames_per_s": bw.total_frames() / dt, "ms_per_step": dt * 1e3,
        "gemm_tflops_algorithmic": alg, "gemm_tflops_mfma_issued": 3.0 * alg,
        "fp16_mfma_peak": FP16_MFMA_PEAK_TFLOPS, "frac_of_fp16_peak_issued": 3.0 * alg / FP16_MFMA_PEAK_TFLOPS,
        "operand_exponents": {"w_exp": [int(e) for e in amw.exponents()[0]], "x_exp_calibrated": [int(e) for e in amw.exponents()[1]]},
        "stage_ms_per_step": {k: bw.timing()[k][0] for k in pk.KINDS}}
    ll_x3 = bw.fetch(0).log_prob()
    bw.close()
    # the same workload with plain fp16 operands (one MFMA per product): the stated-tolerance ceiling
    amp = pk.AcousticModel(layers, prior, L, R, precision="f16")
    bw = pk.BatchScorer(amp, g, B, int(sum(ns)))
    bw.set_waves_device(pcm.data_ptr(), ns, keep_alive=pcm)
    bw.calibrate()
    bw.score(0.1, sync=True)
    bw.enable_timing(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        bw.score(0.1, sync=False)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    gm = bw.timing()["gemm"][0]
    alg = amp.flops_per_frame() * bw.total_frames() / (gm * 1e-3) / 1e12
    ll_p = bw.fetch(0).log_prob()
    err = float(np.max(np.abs(ll_p.astype(np.float64) - ll_x3) / np.maximum(np.abs(ll_x3), 1.0)))
    out[PLAIN_F16_KEY] = {
        "frames_per_s": bw.total_frames() / dt, "ms_per_step": dt * 1e3,
        "gemm_tflops_algorithmic": alg, "frac_of_fp16_peak_issued": alg / FP16_MFMA_PEAK_TFLOPS,
        "measured_max_err_vs_f16x3": err,
        "err_definition": "max |ll - ll_f16x3| / max(|ll_f16x3|, 1) over utterance 0: a comparison of two product modes; "
                          "the comparison with the ORACLE is measured_max_err_vs_oracle, added by the cpu_baseline leg",
        "stage_ms_per_step": {k: bw.timing()[k][0] for k in pk.KINDS}}
    bw.close()
    # utterance 0 of both modes, for the cpu_baseline leg's oracle comparison (not part of the JSON line)
    return out, {"wave": waves[0], "f16x3": ll_x3, "f16": ll_p}


W_F16X3_KEY = "configs[4] wide model 6 x 2048 -> 8000, f16x3, 256 utterances"
PLAIN_F16_KEY = "configs[4] wide model, plain f16 (1 MFMA per product; OUTSIDE the 1e-4 contract)"


def oracle_errors_wide(sample):
    """cpu_baseline leg only: utterance 0 of BASELINE configs[4] through the ORACLE (model W, one thread, ~1 s)
    against what the two fp16 modes produced for it -- max |ll - ref| / max(|ref|, 1)."""
    from oracle import oracle as O
    from pocketkaldi_amd import synth
    layers, prior, L, R = synth.model("W")
    feats = O.cmvn(synth.global_cmvn_stats(), O.Fbank().compute(sample["wave"]))
    ref = O.Nnet(layers).am_compute(feats, prior, L, R, 0.1).astype(np.float64)
    den = np.maximum(np.abs(ref), 1.0)
    return {m: float(np.max(np.abs(sample[m].astype(np.float64) - ref) / den)) for m in ("f16x3", "f16")}


def f16_kernel_label(precision):
    """Which kernel a precision mode runs (gemm_f16.hip: LaunchGemmF16; PK_MI355_F16_SHAPE forces a form for A/B runs)."""
    if precision == "f32":
        return "GemmKernel (fp32 MFMA affine layers, %d launches/step)"
    forced = os.environ.get("PK_MI355_F16_SHAPE", "")
    k32 = forced == "16" or (forced != "32" and precision == "f16x3")
    name = "GemmF16K32Kernel (v_mfma_f32_16x16x32_f16" if k32 else "GemmF16Kernel (v_mfma_f32_32x32x16_f16"
    terms = "3 MFMA per algorithmic product" if precision == "f16x3" else "1 MFMA per product, plain fp16 operands"
    return name + ", " + terms + ", %d launches/step)"


def cpu_identity():
    """CPU model and logical core count of the box the baseline was timed on (every table
    states core count and CPU model)."""
    model = None
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return {"cpu_model": model or "unknown", "nproc": os.cpu_count() or 0,
            "affinity_cores": len(os.sched_getaffinity(0))}


def free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launcher_command(gpus, argv, port):
    """The driver's own N > 1 command shape, built from this process's arguments."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus),
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def self_launch(gpus, argv):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the ranks as a CHILD
    torch.distributed.run (one process per GPU) and relay rank 0's single JSON line and the child's exit
    code.  This process has made no GPU call (torch is not even imported yet): on this pool a process
    that initialised the GPU must never exec another, and a parent holding the device would take a
    slot of the card away from the ranks."""
    import subprocess
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = launcher_command(gpus, argv, free_port())
    print("bench.py: no launcher around --gpus %d, starting: %s" % (gpus, " ".join(cmd)), file=sys.stderr, flush=True)
    child = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env, cwd=REPO)
    for line in child.stdout:                # rank 0 prints the one contract line; anything else goes to stderr
        if line.lstrip().startswith("{"):
            sys.stdout.write(line)
            sys.stdout.flush()
        else:
            sys.stderr.write(line)
    return child.wait()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=256, help="utterances per GPU")
    ap.add_argument("--seconds", type=float, default=10.0, help="audio per utterance")
    ap.add_argument("--model", default="S", choices=["S", "W", "tiny"])
    ap.add_argument("--precision", default="f32", choices=["f32", "f16x3", "f16"],
                    help="f32: bit-exact fp32 MFMA (default); f16x3: split-fp16 on the fp16 matrix cores (inside the 1e-4 "
                         "contract); f16: plain fp16 operands, one MFMA per product, OUTSIDE the contract (~1e-3)")
    ap.add_argument("--softmax", default="stable", choices=["stable", "reference"],
                    help="stable: overflow-safe log-softmax tail (default); reference: the reference's softmax "
                         "operations one by one -- with f32 the whole path is then bit-identical to the oracle restatement of the CPU path")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (default); gloo = rehearsal of the multi-rank flow, e.g. two ranks on ONE GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-precision", action="store_true",
                    help="skip the supplementary run in the other precision mode")
    ap.add_argument("--cpu-seconds", type=float, default=12.0,
                    help="time cap of each CPU-baseline leg (1 thread, then all host cores), seconds")
    ap.add_argument("--no-host-endpoints", action="store_true",
                    help="skip the from_pinned_host / host_complete end-points (N = 1 only)")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the supplementary BASELINE configs[1] / configs[4] lines (N = 1 only)")
    ap.add_argument("--ragged", action="store_true",
                    help="utterance lengths U[2 s, 20 s] (seeded) instead of equal lengths; ranks take the shards "
                         "pkdist.partition_by_frames gives them (longest-first greedy on the frame count)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args.gpus, sys.argv[1:]))

    import torch
    import pocketkaldi_amd as pk
    from pocketkaldi_amd import dist as pkdist
    from pocketkaldi_amd import synth

    rank, local_rank, world = pkdist.env_world()
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    ndev = torch.cuda.device_count()
    dev_index = local_rank % max(ndev, 1)        # rehearsal: several ranks may share one GPU
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    pk.set_device(dev_index)
    pkdist.init(args.backend)
    cdev = dev if args.backend == "nccl" else "cpu"   # where the small scalar collectives live

    # ---- model: every rank builds the same structure; only rank 0 has real values
    layers, prior, L, R = synth.model(args.model)
    if rank != 0:
        layers = [(l[0], np.zeros_like(l[1]), np.zeros_like(l[2])) if l[0] == "linear" else l
                  for l in layers]
        prior = np.full_like(prior, 1.0)
    am = pk.AcousticModel(layers, prior, L, R, precision=args.precision).set_softmax(args.softmax)
    import torch.distributed as tdist

    def calibrate_on_root(model, precision):
        """fp16 modes, as INTEGRATION.md 3b has a deployment do it right behind pk_load: the root places every
        layer's input operand inside the fp16 window on one utterance (pk_mi355_batch_calibrate).  The exponents
        live in the weight blob, so the ONE broadcast below carries them -- a rank left at the default exponents would
        fail the replica check."""
        if precision == "f32" or rank != 0 or os.environ.get("PK_BENCH_NO_CALIBRATE", "") == "1":   # (A/B switch)
            return None
        w0 = synth.utterance(0, args.seconds)
        tmp = pk.BatchScorer(model, synth.global_cmvn_stats(), 1, len(w0))
        tmp.set_waves([w0])
        tmp.calibrate()
        tmp.close()
        return [int(e) for e in model.exponents()[1]]

    calibrated = calibrate_on_root(am, args.precision)
    pkdist.broadcast_model(am, dev, src=0)          # the one RCCL collective of the path (no-op at world 1)

    # ---- synthetic PCM, resident in HBM before any timed region
    sharding = None
    if args.ragged:
        # the reference's real workload is a ragged list (main.cc:34-46): batch x world utterances of U[2 s, 20 s],
        # sharded by frame count (longest-first greedy); every rank computes the same map, nothing is exchanged
        secs = [synth.ragged_seconds(u) for u in range(args.batch * world)]
        frames_all = [pkdist.frames_of(round(s * synth.SAMPLE_RATE)) for s in secs]
        shards = pkdist.partition_by_frames(frames_all, world)
        ids = shards[rank]
        waves = [synth.utterance(u, secs[u]) for u in ids]
        loads = [sum(frames_all[u] for u in s) for s in shards]
        sharding = {"policy": "pkdist.partition_by_frames (longest-first greedy on frame counts)",
                    "utterances": len(secs), "seconds": "U[2, 20] per utterance, seeded by utterance id",
                    "per_rank_utterances": [len(s) for s in shards], "per_rank_frames": loads,
                    "imbalance": pkdist.imbalance(frames_all, shards),
                    "imbalance_u_mod_N": pkdist.imbalance(frames_all, pkdist.partition_round_robin(frames_all, world)),
                    "imbalance_definition": "max over ranks of the shard's frames / mean - 1"}
    else:
        ids = pkdist.utterance_ids(rank, world, args.batch)
        waves = [synth.utterance(u, args.seconds) for u in ids]
    ns = [len(w) for w in waves]
    nutts = len(ids)
    pcm = torch.from_numpy(np.concatenate(waves)).to(dev)
    bs = pk.BatchScorer(am, synth.global_cmvn_stats(), nutts, int(sum(ns)))
    bs.set_waves_device(pcm.data_ptr(), ns, keep_alive=pcm)
    frames_per_step = bs.total_frames()

    # ---- broadcast check: every rank scores utterance 0 with its replica
    replica_check = "not run (no process group)"
    if world > 1 or tdist.is_initialized():
        chk = pk.BatchScorer(am, synth.global_cmvn_stats(), 1, 16000)
        chk.set_waves([synth.utterance(0, 1.0)])
        chk.score(0.1)
        s = float(np.sum(chk.fetch(0).log_prob().astype(np.float64)))
        if not pkdist.all_ranks_agree(s, cdev):
            raise SystemExit("weight broadcast mismatch across ranks")
        chk.close()
        replica_check = "passed"

    # ---- the C entry of the same collective: with a real RCCL group, an ncclComm_t of our
    # own over all ranks (ctypes on torch's RCCL; the id shared by one small broadcast), pk_mi355_am_broadcast_from
    # into a second, zero-initialised model on every rank, and that model's scores compared across ranks and with
    # the root's real model.  Before the timed region; no new process.
    c_abi_broadcast = "not run (no process group)"
    binding = comm = None
    if tdist.is_initialized():
        if args.backend != "nccl":
            c_abi_broadcast = "skipped: gloo rehearsal (two RCCL ranks cannot share one device; the C entry runs over a real communicator at nranks = 1 in tests/)"
        elif os.environ.get("PK_BENCH_SKIP_C_ABI", "") == "1":
            c_abi_broadcast = "skipped: PK_BENCH_SKIP_C_ABI=1"
        else:
            # A communicator that cannot be made (no librccl symbol, a refused bootstrap) is the environment's
            # failure, not the path's: every rank learns of it through the process group, the line says so, and the
            # measurement goes on.  A failed broadcast or a score mismatch is the product's and stops the run.
            binding, comm, why = pkdist.try_make_rccl_comm(rank, world, cdev)
            if not comm:
                c_abi_broadcast = "unavailable: no second communicator (%s)" % why
        if comm:
            zl = [(l[0], np.zeros_like(l[1]), np.zeros_like(l[2])) if l[0] == "linear" else l for l in layers]
            am_z = pk.AcousticModel(zl, np.full_like(prior, 1.0), L, R, precision=args.precision).set_softmax(args.softmax)
            try:
                am_z.broadcast(comm, root=0, src=am)          # on the root the bytes sent are `am`'s; everyone receives into am_z
            finally:
                binding.comm_destroy(comm)

            def probe(model):
                c = pk.BatchScorer(model, synth.global_cmvn_stats(), 1, 16000)
                c.set_waves([synth.utterance(0, 1.0)])
                c.score(0.1)
                v = float(np.sum(c.fetch(0).log_prob().astype(np.float64)))
                c.close()
                return v
            sz, sa = probe(am_z), probe(am)
            # both agreements are collectives EVERY rank enters, whatever its own result (a rank that left before
            # the all_reduce would leave the others waiting in it): only then do all ranks stop together
            same_everywhere = pkdist.all_ranks_agree(sz, cdev)
            all_local_ok = pkdist.all_ranks_agree(0.0 if sz == sa else 1.0 + rank, cdev)
            if not (same_everywhere and all_local_ok):
                raise SystemExit("pk_mi355_am_broadcast (C ABI) mismatch: rank %d scores %r, expected %r" % (rank, sz, sa))
            am_z.close()
            c_abi_broadcast = "passed"

    def timed_steps(scorer):
        """W untimed + K timed passes bracketed by barrier + device sync.  The per-kernel HIP events
        (two per launch, ~0.13 ms per step of dispatch gaps) are recorded on the LAST timed pass
        only -- the one scorer.timing() reports -- so the other K - 1 run as a caller's would."""
        for _ in range(args.warmup):
            scorer.score(0.1, sync=False)
        scorer.synchronize()
        pkdist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(args.steps):
            if k == args.steps - 1:
                scorer.enable_timing(True)
            scorer.score(0.1, sync=False)
        torch.cuda.synchronize()
        pkdist.barrier()
        elapsed = time.perf_counter() - t0
        tm_ = scorer.timing()
        scorer.enable_timing(False)
        timed_steps.own_elapsed = elapsed
        return pkdist.max_over_ranks(elapsed, cdev), tm_

    dt, tm = timed_steps(bs)
    # every rank's own rate over ITS barrier-to-barrier time (stragglers show as a low minimum)
    rank_lo, rank_hi = pkdist.min_max_over_ranks(frames_per_step * args.steps / timed_steps.own_elapsed, cdev)
    total_frames = pkdist.sum_over_ranks(frames_per_step, cdev)

    # ---- supplementary: the same workload in the other precision mode (same K, W)
    other = None
    if not args.no_other_precision:
        other_prec = "f16x3" if args.precision == "f32" else "f32"
        bs.close()
        am2 = pk.AcousticModel(layers, prior, L, R, precision=other_prec).set_softmax(args.softmax)
        calibrated2 = calibrate_on_root(am2, other_prec)
        pkdist.broadcast_model(am2, dev, src=0)
        bs2 = pk.BatchScorer(am2, synth.global_cmvn_stats(), nutts, int(sum(ns)))
        bs2.set_waves_device(pcm.data_ptr(), ns, keep_alive=pcm)
        dt2, tm2 = timed_steps(bs2)
        g2 = tm2["gemm"][0]
        other = {"precision": other_prec, "value": total_frames * args.steps / dt2, "unit": "frames/s",
                 "ms_per_step": dt2 / args.steps * 1e3,
                 "gemm_tflops_algorithmic": (am2.flops_per_frame() * frames_per_step / (g2 * 1e-3) / 1e12) if g2 > 0 else 0.0,
                 "stage_ms_per_step": {k: tm2[k][0] for k in pk.KINDS},
                 "operand_exponents_calibrated_on_root": calibrated2,
                 "parity": "f32: affine layers bit-identical to committed outputs of the reference's own SGEMM (gemm.cc + gemm_haswell.cc built here); f16x3: split-fp16 operands on the "
                           "fp16 matrix cores, log-likelihoods within 1e-4*max(|ref|,1) (measured ~1e-6), tests/test_gpu_parity.py"}
        bs2.close()

    # ---- supplementary: the same workload with the reference's softmax arithmetic (bit-identical path)
    ref_softmax = None
    if not args.no_other_precision and args.softmax == "stable" and args.precision == "f32":
        am.set_softmax("reference")
        bs3 = pk.BatchScorer(am, synth.global_cmvn_stats(), nutts, int(sum(ns)))
        bs3.set_waves_device(pcm.data_ptr(), ns, keep_alive=pcm)
        dt3, tm3 = timed_steps(bs3)
        ref_softmax = {"softmax": "reference", "value": total_frames * args.steps / dt3, "unit": "frames/s",
                       "ms_per_step": dt3 / args.steps * 1e3, "stage_ms_per_step": {k: tm3[k][0] for k in pk.KINDS},
                       "parity": "every stage %s (tests/test_gpu_parity.py::" % PARITY_PIN +
                                 "test_reference_softmax_whole_path_is_bit_identical_to_the_reference)"}
        bs3.close()
        am.set_softmax("stable")

    if rank == 0:
        value = total_frames * args.steps / dt
        gemm_ms, gemm_launches = tm["gemm"]
        flops_per_step = am.flops_per_frame() * frames_per_step
        achieved = flops_per_step / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
        hidden, nh, pdfs = synth.MODELS[args.model]
        peak = FP32_MFMA_PEAK_TFLOPS if args.precision == "f32" else FP16_MFMA_PEAK_TFLOPS
        out = {
            "metric": "acoustic frames/sec (fbank->nnet log-likelihoods)",
            "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"f32": "f32", "f16x3": "f16x3 (fp16 hi+lo operands, 3 MFMA per product, f32 accumulate)",
                      "f16": "f16 (plain fp16 operands, f32 accumulate; outside the 1e-4 contract)"}[args.precision],
            "data": "synthetic",
            "config": {"workload": ("%d utterances x %s 16 kHz per GPU (BASELINE configs[2]; x8 = configs[3]), "
                                    "440 -> %d x %d ReLU -> %d softmax, fbank+CMVN+nnet, PCM resident in HBM"
                                    % (args.batch, "U[2, 20] s (ragged, length-balanced shards)" if args.ragged
                                       else "%.0f s" % args.seconds, nh, hidden, pdfs)),
                       "acoustic_model": args.model, "softmax": args.softmax, "utterances_per_gpu": args.batch,
                       "operand_exponents_calibrated_on_root": calibrated,
                       "frames_per_gpu_per_step": int(frames_per_step),
                       "parallelism": "utterance-sharded x%d, weights broadcast once (%s)" % (
                           world, "RCCL" if args.backend == "nccl" else "gloo rehearsal"),
                       "collective": {"backend": args.backend, "process_group": bool(tdist.is_initialized()),
                                      "replica_check": replica_check, "c_abi_broadcast": c_abi_broadcast},
                       "per_rank_frames_per_s": {"min": rank_lo, "max": rank_hi,
                                                 "note": "each rank's own frames over its own barrier-to-barrier time; "
                                                         "value = all ranks' frames over the slowest rank's time"}},
            "roofline": {"bound": "mfma",
                         "kernel": f16_kernel_label(args.precision) % gemm_launches,
                         "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                         "frac": achieved / peak,
                         "flop_per_frame": am.flops_per_frame(),
                         "kernel_ms_per_step": gemm_ms,
                         "note": ("fp32 default: the last affine layer's launch also turns its logits into log-likelihoods "
                                  "(fused tail, DESIGN.md 3.1), so kernel_ms_per_step and `achieved` include the tail's work; "
                                  "PK_MI355_FUSED_TAIL32=0 separates them again") if args.precision == "f32" and args.softmax == "stable" else None,
                         "algorithmic_bytes_per_launch": None, "traffic": None, "traffic_source": None},
            "stage_ms_per_step": {k: tm[k][0] for k in pk.KINDS},
        }
        if sharding is not None:
            out["config"]["sharding"] = sharding
        if other is not None:
            out["other_precision"] = other
        if ref_softmax is not None:
            out["reference_softmax"] = ref_softmax
        # the HBM-bound stages against the 8 TB/s peak (algorithmic
        # bytes per frame = 800 fbank, 320 CMVN, 2 * 4 * num_pdfs log-softmax tail)
        hbm_peak = 8000.0
        per_frame = {"fbank": 800.0, "cmvn": 320.0, "tail": 8.0 * pdfs}
        out["stage_roofline"] = {}
        for k, bpf in per_frame.items():
            ms = tm[k][0]
            gbs = bpf * frames_per_step / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
            out["stage_roofline"][k] = {"bound": "hbm", "bytes_per_frame": bpf, "achieved": gbs,
                                        "peak": hbm_peak, "unit": "GB/s", "frac": gbs / hbm_peak}
        # SURVEY 8(d) prices the front-end against HBM by contract; what limits it in practice (PMC):
        out["stage_roofline"]["fbank"]["limited_by"] = "vector-instruction issue: 482 vector + 104 LDS instructions per frame (profiles/r05_fbank_counters.txt, DESIGN.md 3.3)"
        out["stage_roofline"]["cmvn"]["limited_by"] = "the serial window-sum recurrence, one rounding per frame (DESIGN.md 3.4)"
        if tm["tail"][1] == 0:
            out["stage_roofline"]["tail"]["limited_by"] = ("no launch of its own: fused into the last affine layer's launch "
                                                           "(its time is inside the gemm stage, DESIGN.md 3.1 / 3.5)")
        if args.model == "S" and args.batch == 256 and not args.ragged and gemm_launches and args.precision == "f32":
            traffic, source = measured_traffic()
            out["roofline"]["traffic"] = traffic
            out["roofline"]["traffic_source"] = source
            # operands read once + output written once, averaged over the launches of a step
            # (layer 1 reads the 40-dim features, the splice is a view)
            lay = [(40, 440, 1024)] + [(1024, 1024, 1024)] * 3 + [(1024, 1024, 3000)]
            rows = frames_per_step                      # (the layer stack's rows are compact: no rows for the context pads)
            alg = sum(4.0 * (rows * kin + k * n + rows * n) for kin, k, n in lay)
            if tm["tail"][1] == 0:
                # fused tail: the last layer's launch also writes the log-likelihoods (12 000 B/frame when
                # fused).  Its logits still make a round trip through memory inside the launch (sc1 stores, read back by the
                # workgroup that completes the row tile): that round trip is in `traffic`, not in the algorithmic bytes.
                alg += 4.0 * rows * 3000
                out["roofline"]["traffic_note"] = ("the last layer's launch moves its logits through memory once more "
                                                   "(written sc1, read back by the row tile's owner) and writes the log-likelihoods: "
                                                   "~9.6 GB of the step's traffic is the fused tail, not operand re-reads")
            out["roofline"]["algorithmic_bytes_per_launch"] = alg / gemm_launches
        if world == 1 and not args.no_host_endpoints and args.precision == "f32":
            bs.close()
            out["endpoints"] = {"device_complete": {"value": value, "unit": "frames/s",
                                                    "start": "float PCM resident in HBM", "end": "log-likelihoods complete in HBM",
                                                    "note": "= value, the headline"}}
            out["endpoints"].update(host_endpoints(pk, synth, am, args.batch, args.seconds))
            # SURVEY 8(d) defines the metric from PCM in pinned host memory; `value` (the contract's start point: inputs
            # resident in HBM) stays the headline, the 8(d) start point travels beside it under its own name
            out["value_from_pinned_host"] = out["endpoints"]["from_pinned_host"]["value"]
            out["value_from_pinned_host_note"] = ("same workload and unit as `value`, timed from int16 PCM in page-locked host "
                                                  "memory (upload included, two batches in flight) to log-likelihoods complete in HBM")
        if world == 1 and not args.no_other_configs:
            bs.close()
            out["other_configs"], wide_sample = other_configs(pk, synth, torch, max(2, min(args.steps, 3)))
        else:
            wide_sample = None
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.model, args.seconds, args.cpu_seconds)
            if wide_sample is not None:      # the fp16 modes' error against the oracle
                errs = oracle_errors_wide(wide_sample)
                out["other_configs"][W_F16X3_KEY]["measured_max_err_vs_oracle"] = errs["f16x3"]
                out["other_configs"][PLAIN_F16_KEY]["measured_max_err_vs_oracle"] = errs["f16"]
            out["cpu_baseline_all_cores"] = cpu_baseline_all_cores(args.model, args.seconds, min(args.cpu_seconds, 8.0))
        print(json.dumps(out), flush=True)
    pkdist.barrier()
    bs.close()
    pkdist.shutdown()


if __name__ == "__main__":
    main()
