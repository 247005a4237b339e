"""bench.py prints ONE JSON line carrying the driver's contract keys plus roofline / cpu_baseline."""
import json
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_line_has_the_contract_keys():
    out = subprocess.check_output(
        [sys.executable, os.path.join(REPO, "bench.py"), "--steps", "2", "--warmup", "1", "--batch", "8",
         "--seconds", "2", "--cpu-seconds", "1", "--no-other-configs"], text=True, cwd=REPO)
    lines = [l for l in out.splitlines() if l.strip()]
    assert len(lines) == 1, "exactly one line on stdout"
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
              "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1
    assert d["unit"] == "frames/s" and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "acoustic_model" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source"):
        assert k in r, k
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] == "port" and c["cores"] == 1 and c["value"] > 0
    # SURVEY section 8d's three end-points travel in the same line; `value` stays device-complete
    e = d["endpoints"]
    assert e["device_complete"]["value"] == d["value"]
    for k in ("from_pinned_host", "host_complete"):
        assert e[k]["unit"] == "frames/s" and e[k]["value"] > 0
    assert "time-capped" in c["sample"] and "time-capped" in d["cpu_baseline_all_cores"]["sample"]
    # value = frames of all steps / wall time
    assert abs(d["value"] - d["config"]["frames_per_gpu_per_step"] / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    # round 5: CPU model and core count beside every CPU number; SURVEY 8(d)'s start
    # point (pinned host memory) as a named top-level number next to `value`
    for leg in (c, d["cpu_baseline_all_cores"]):
        assert isinstance(leg["cpu_model"], str) and leg["cpu_model"] and leg["nproc"] >= 1 and leg["affinity_cores"] >= 1
    # the reference's own compiled code (gemm.cc + gemm_haswell.cc) timed in the same run, where oracle/_ref travelled
    if "nnet_affine_layers_reference_gemm" in c:
        rg = c["nnet_affine_layers_reference_gemm"]
        assert rg["kind"] == "reference" and rg["cores"] == 1 and rg["value"] > 0 and rg["unit"] == "frames/s"
    assert d["value_from_pinned_host"] == e["from_pinned_host"]["value"]
    assert "page-locked" in d["value_from_pinned_host_note"]


def _one_json_line(stdout):
    lines = [l for l in stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.gpu
def test_driver_launch_shape_two_ranks_on_one_gpu():
    """BASELINE configs[3]'s launch path: the driver's own command shape for N > 1 -- fresh processes from
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P
    bench.py --gpus N ...` -- rehearsed with two ranks sharing the one GPU (gloo: two RCCL ranks cannot share
    a device).  Rank 1 builds zero weights and receives rank 0's through bench.py's own broadcast; bench.py's
    replica check (every rank scores a common utterance) must pass; rank 0 prints exactly one line whose value
    is the frames of BOTH ranks over the slowest rank's time."""
    env = dict(os.environ, PYTHONPATH=REPO)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "PK_DIST_FORCE"):
        env.pop(k, None)
    r = subprocess.run(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
         "--master-port", str(_free_port()), os.path.join(REPO, "bench.py"), "--gpus", "2", "--backend", "gloo",
         "--batch", "8", "--seconds", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
        capture_output=True, text=True, cwd=REPO, env=env, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    d = _one_json_line(r.stdout)
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert "x2" in d["config"]["parallelism"] and "gloo rehearsal" in d["config"]["parallelism"]
    per_rank = d["config"]["frames_per_gpu_per_step"]
    assert per_rank == 8 * 198                         # 2 s -> 198 frames (fbank.cc:35-42), 8 utterances per rank
    assert abs(d["value"] - 2 * per_rank / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    assert "cpu_baseline" not in d and "endpoints" not in d and "other_configs" not in d   # N = 1 only
    assert d["other_precision"]["precision"] == "f16x3" and d["other_precision"]["value"] > 0
    # round 4: the C-ABI leg of the collective needs real RCCL ranks and says so; every rank's own rate is in the line
    c = d["config"]["collective"]
    assert c["replica_check"] == "passed" and c["c_abi_broadcast"].startswith("skipped: gloo rehearsal")
    pr = d["config"]["per_rank_frames_per_s"]
    assert 0 < pr["min"] <= pr["max"] and pr["min"] >= d["value"] / 2 * 0.999     # value = 2 ranks' frames over the slowest


@pytest.mark.gpu
def test_forced_process_group_runs_the_real_rccl_broadcast_inside_bench():
    """PK_DIST_FORCE=1 at world 1: bench.py calls init_process_group("nccl") -- RCCL -- and
    torch.distributed.broadcast on the zero-copy alias of the model's device blob, then its replica check.
    One rank, so the collective moves nothing between GPUs (RCCL has never seen two ranks in this repository's
    build loop: DESIGN.md section 6), but every call of the N-rank flow is the real one."""
    env = dict(os.environ, PYTHONPATH=REPO, PK_DIST_FORCE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    r = subprocess.run(
        [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "1", "--backend", "nccl", "--batch", "8", "--seconds", "2",
         "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-other-configs", "--no-host-endpoints",
         "--no-other-precision"],
        capture_output=True, text=True, cwd=REPO, env=env, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    d = _one_json_line(r.stdout)
    assert d["n_gpus"] == 1 and "RCCL" in d["config"]["parallelism"]
    # round 4: an ncclComm_t made through ctypes on torch's RCCL, pk_mi355_am_broadcast_from into a zero-initialised
    # second model, its scores equal to the real model's -- the C entry over a real communicator, inside bench.py
    assert d["config"]["collective"] == {"backend": "nccl", "process_group": True, "replica_check": "passed",
                                         "c_abi_broadcast": "passed"}
    pr = d["config"]["per_rank_frames_per_s"]
    assert abs(pr["min"] - pr["max"]) < 1e-9 * pr["max"] and pr["max"] >= d["value"] * 0.999


def _launch_env():
    env = dict(os.environ, PYTHONPATH=REPO)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "PK_DIST_FORCE"):
        env.pop(k, None)
    return env


@pytest.mark.gpu
def test_bare_gpus_2_launches_its_own_ranks_and_prints_one_line():
    """`python bench.py --gpus N` with NO launcher around it (the N = 1 command shape
    with another N) starts torch.distributed.run itself, as a child, and relays rank 0's one line and the exit code.
    Two ranks share the one GPU here (gloo); ragged utterances exercise the length-balanced sharding."""
    r = subprocess.run(
        [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--backend", "gloo", "--ragged",
         "--batch", "6", "--steps", "2", "--warmup", "1", "--no-other-precision"],
        capture_output=True, text=True, cwd=REPO, env=_launch_env(), timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout                       # nothing but the contract line reaches stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert d["config"]["collective"]["replica_check"] == "passed" and d["config"]["collective"]["process_group"] is True
    sh = d["config"]["sharding"]
    assert sh["utterances"] == 12 and sum(sh["per_rank_utterances"]) == 12 and len(sh["per_rank_frames"]) == 2
    assert sh["per_rank_frames"][0] == d["config"]["frames_per_gpu_per_step"]          # rank 0's shard
    assert 0.0 <= sh["imbalance"] <= sh["imbalance_u_mod_N"] + 1e-12
    total = sum(sh["per_rank_frames"])
    assert abs(d["value"] - total / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    assert "torch.distributed.run" in r.stderr             # the launcher line is on stderr


def test_bare_gpus_2_starts_two_ranks_without_touching_the_gpu_in_the_parent():
    """The same entry on a box WITHOUT a GPU: the parent must get as far as starting the child launcher (it makes no
    GPU call and does not import torch), both ranks then refuse to run without an MI355X, and the parent relays the
    launcher's non-zero exit code with nothing on stdout."""
    sys.path.insert(0, REPO)
    import bench
    cmd = bench.launcher_command(8, ["--gpus", "8", "--steps", "3"], 29123)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "8" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[cmd.index("--master-port") + 1] == "29123" and cmd[-4:] == ["--gpus", "8", "--steps", "3"]
    assert os.path.samefile(cmd[-5], os.path.join(REPO, "bench.py"))
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the launch itself is test_bare_gpus_2_launches_its_own_ranks_and_prints_one_line")
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--backend", "gloo",
                        "--batch", "2", "--seconds", "1", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, cwd=REPO, env=_launch_env(), timeout=600)
    assert r.returncode != 0 and r.stdout.strip() == ""
    assert "no launcher around --gpus 2" in r.stderr and r.stderr.count("needs an MI355X") >= 2, r.stderr[-3000:]


def test_offline_traffic_figure_is_tied_to_the_gemm_sources(tmp_path, monkeypatch):
    """roofline.traffic comes from committed PMC passes; it is reported only when that file says it was
    measured on the GEMM sources the running library is built from."""
    sys.path.insert(0, REPO)
    import bench
    from pocketkaldi_amd import build as B
    (tmp_path / "profiles").mkdir()
    monkeypatch.setattr(bench, "REPO", str(tmp_path))
    assert bench.measured_traffic()[0] is None
    f = tmp_path / "profiles" / "r09_pmc_traffic.json"
    f.write_text(json.dumps({"gemm_avg_hbm_bytes_per_launch": 2.5e9, "measured_on": {"gemm_source_hash": "0" * 16}}))
    value, why = bench.measured_traffic()
    assert value is None and "not reported" in why
    f.write_text(json.dumps({"gemm_avg_hbm_bytes_per_launch": 2.5e9, "measured_on": {"gemm_source_hash": B.gemm_source_hash()}}))
    value, why = bench.measured_traffic()
    assert value == 2.5e9 and "same GEMM sources" in why
    # the dominant launch runs the tail too: its header is part of what the figure is tied to
    import inspect
    assert "pk_tail_wave.h" in inspect.getsource(B.gemm_source_hash)


def test_roofline_kernel_label_follows_the_precision_and_the_forced_shape(monkeypatch):
    """roofline.kernel named GemmF16Kernel / 3 MFMA for every fp16 run.  f16x3 runs GemmF16K32Kernel
    (16x16x32, 3 MFMA per product), plain f16 GemmF16Kernel (32x32x16, 1 MFMA); PK_MI355_F16_SHAPE forces either."""
    sys.path.insert(0, REPO)
    import bench
    monkeypatch.delenv("PK_MI355_F16_SHAPE", raising=False)
    assert bench.f16_kernel_label("f32") % 10 == "GemmKernel (fp32 MFMA affine layers, 10 launches/step)"
    x3, plain = bench.f16_kernel_label("f16x3") % 12, bench.f16_kernel_label("f16") % 12
    assert x3.startswith("GemmF16K32Kernel (v_mfma_f32_16x16x32_f16, 3 MFMA") and "12 launches" in x3
    assert plain.startswith("GemmF16Kernel (v_mfma_f32_32x32x16_f16, 1 MFMA")
    monkeypatch.setenv("PK_MI355_F16_SHAPE", "32")
    assert (bench.f16_kernel_label("f16x3") % 1).startswith("GemmF16Kernel (v_mfma_f32_32x32x16_f16, 3 MFMA")
    monkeypatch.setenv("PK_MI355_F16_SHAPE", "16")
    assert (bench.f16_kernel_label("f16") % 1).startswith("GemmF16K32Kernel (v_mfma_f32_16x16x32_f16, 1 MFMA")
