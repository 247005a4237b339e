"""BASELINE configs[3] (utterance-sharded over N GPUs, weights broadcast once) rehearsed on ONE GPU.

  * pk_mi355_am_broadcast -- the C-ABI form of the one collective -- over a real RCCL communicator
    at nranks = 1 (two RCCL ranks cannot share a device);
  * two fresh processes (gloo; both on GPU 0) run bench.py's multi-rank flow -- rank 1 starts from
    zero weights, pkdist.broadcast_model aliases the device blob and broadcasts into it, each rank
    scores its utterance shard (u -> rank u mod N) -- and every utterance's log-likelihood CRC must
    equal a single-process run over the same utterance ids.
"""
import ctypes as C
import json
import os
import socket
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import pocketkaldi_amd as pk
from pocketkaldi_amd import synth

PER_RANK = 5
SECONDS = [1.0, 0.4, 2.5, 0.03, 7.0]          # ragged shards, one utterance shorter than a frame, one past the CMVN window


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _score_shard(am, ids):
    waves = [synth.utterance(u, SECONDS[i % len(SECONDS)]) for i, u in enumerate(ids)]
    bs = pk.BatchScorer(am, synth.global_cmvn_stats(), len(waves), sum(len(w) for w in waves))
    bs.set_waves(waves)
    bs.score(0.1)
    out = {int(u): zlib.crc32(bs.fetch(i).log_prob().tobytes()) for i, u in enumerate(ids)}
    bs.close()
    return out


def _rank_main(rank, world, port, out_dir, precision):
    os.environ.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world),
                       "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    import torch
    from pocketkaldi_amd import dist as pkdist
    torch.cuda.set_device(0)                      # rehearsal: both ranks share the one GPU
    pk.set_device(0)
    pkdist.init("gloo")
    layers, prior, L, R = synth.model("S")
    if rank != 0:                                 # as bench.py: only rank 0 holds real values
        layers = [(l[0], np.zeros_like(l[1]), np.zeros_like(l[2])) if l[0] == "linear" else l for l in layers]
        prior = np.full_like(prior, 1.0)
    am = pk.AcousticModel(layers, prior, L, R, precision=precision)
    before = _score_shard(am, [0])
    pkdist.broadcast_model(am, torch.device("cuda", 0), src=0)
    ids = pkdist.utterance_ids(rank, world, PER_RANK)
    res = {"rank": rank, "ids": ids, "crc": _score_shard(am, ids), "probe_before": before[0],
           "probe_after": _score_shard(am, [0])[0]}
    # the replica check bench.py runs after the broadcast
    res["agree"] = pkdist.all_ranks_agree(float(res["probe_after"]))
    pkdist.barrier()
    with open(os.path.join(out_dir, "rank%d.json" % rank), "w") as f:
        json.dump(res, f)
    pkdist.shutdown()


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
def test_two_ranks_on_one_gpu_broadcast_and_score(tmp_path, precision):
    import torch.multiprocessing as mp
    world = 2
    pk.lib()            # build (if stale) in the parent, once, before any rank exists
    mp.spawn(_rank_main, args=(world, _free_port(), str(tmp_path), precision), nprocs=world, join=True)
    res = [json.load(open(tmp_path / ("rank%d.json" % r))) for r in range(world)]
    # single-process run over the same utterance ids with the real weights
    layers, prior, L, R = synth.model("S")
    am = pk.AcousticModel(layers, prior, L, R, precision=precision)
    for r in res:
        assert r["agree"] is True
        assert sorted(r["ids"]) == [r["rank"] + world * i for i in range(PER_RANK)]
        want = _score_shard(am, r["ids"])
        assert {int(k): v for k, v in r["crc"].items()} == want, "rank %d scores differ from the single-process run" % r["rank"]
    # rank 1 really started from other weights and really received rank 0's
    assert res[1]["probe_before"] != res[0]["probe_before"]
    assert res[1]["probe_after"] == res[0]["probe_after"] == res[0]["probe_before"]
    assert set(res[0]["ids"]).isdisjoint(res[1]["ids"])


class _UniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]


def test_c_abi_broadcast_over_a_real_rccl_communicator_at_one_rank():
    """pk_mi355_am_broadcast(am, ncclComm_t, root, stream): what a C++ pk_load calls on every rank.
    nranks = 1 here (one GPU); the blob must come through intact, on a caller stream and on the
    call's own."""
    try:
        rccl = C.CDLL("librccl.so.1", mode=C.RTLD_GLOBAL)
    except OSError:
        rccl = C.CDLL("/opt/rocm/lib/librccl.so.1", mode=C.RTLD_GLOBAL)
    rccl.ncclGetUniqueId.argtypes = [C.POINTER(_UniqueId)]
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, _UniqueId, C.c_int]
    rccl.ncclCommDestroy.argtypes = [C.c_void_p]
    pk.set_device(0)
    layers, prior, L, R = synth.model("tiny")
    am = pk.AcousticModel(layers, prior, L, R)
    feats = np.random.default_rng(1).standard_normal((30, 40)).astype(np.float32)
    want = pk.Decodable(am, 0.1, feats).log_prob()
    uid = _UniqueId()
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    comm = C.c_void_p()
    assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
    try:
        am.broadcast(comm.value, root=0)                       # own stream, synchronous
        assert np.array_equal(pk.Decodable(am, 0.1, feats).log_prob(), want)
        bs = pk.BatchScorer(am, synth.global_cmvn_stats(), 1, 16000)
        am.broadcast(comm.value, root=0, stream=bs.stream())   # caller's stream, asynchronous
        bs.synchronize()
        assert np.array_equal(pk.Decodable(am, 0.1, feats).log_prob(), want)
        with pytest.raises(pk.PkError):
            am.broadcast(None, root=0)
        # RCCL writes a DIFFERENT model's blob: `other` shares nothing with `am` and starts from zero
        # weights; the out-of-place form sends am's blob and receives into other's
        zl = [(l[0], np.zeros_like(l[1]), np.zeros_like(l[2])) if l[0] == "linear" else l for l in layers]
        other = pk.AcousticModel(zl, np.full_like(prior, 1.0), L, R)
        assert not np.array_equal(pk.Decodable(other, 0.1, feats).log_prob(), want)
        other.broadcast(comm.value, root=0, src=am)
        assert np.array_equal(pk.Decodable(other, 0.1, feats).log_prob(), want)
        zero = pk.AcousticModel(zl, np.full_like(prior, 1.0), L, R, precision="f16x3")
        with pytest.raises(pk.PkError, match="layout"):         # another precision = another blob layout
            other.broadcast(comm.value, root=0, src=zero)
    finally:
        rccl.ncclCommDestroy(comm)
