"""The N > 1 path on CPU: world_size-2 gloo run of the same helpers bench.py uses with
RCCL (utterance sharding, the one weight-blob broadcast, max-over-ranks timing)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from pocketkaldi_amd import dist as pkdist


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out_dir):
    os.environ.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world),
                       "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    r, lr, w = pkdist.init("gloo")
    assert (r, w) == (rank, world)
    # weight blob: rank 0 holds the real bytes, the others zeros of the same size
    rng = np.random.default_rng(0x5EED)
    real = rng.integers(0, 256, 4096, dtype=np.uint8)
    blob = torch.from_numpy(real.copy() if rank == 0 else np.zeros_like(real))
    pkdist.broadcast_blob(blob, src=0)
    assert np.array_equal(blob.numpy(), real)
    # the replica check bench.py runs after the broadcast
    assert pkdist.all_ranks_agree(float(blob.numpy().astype(np.float64).sum()))
    assert not pkdist.all_ranks_agree(float(rank))
    # sharding: u -> rank u mod world, per_rank each (weak scaling)
    ids = pkdist.utterance_ids(rank, world, 5)
    assert all(u % world == rank for u in ids) and len(ids) == 5
    gathered = [None] * world
    torch.distributed.all_gather_object(gathered, ids)
    assert sorted(sum(gathered, [])) == list(range(5 * world))
    # timing protocol: max over ranks of the elapsed time, sum over ranks of the frames
    assert pkdist.max_over_ranks(1.0 + rank) == float(world)
    assert pkdist.sum_over_ranks(998 * 5) == 998 * 5 * world
    assert pkdist.min_max_over_ranks(10.0 * (rank + 1)) == (10.0, 10.0 * world)      # per-rank rates in the bench line
    # the ncclUniqueId hand-shake of the C-ABI broadcast check: rank 0's 128 bytes reach every rank
    uid = bytes(range(128)) if rank == 0 else None
    assert pkdist.share_bytes(uid, 128, src=0) == bytes(range(128))
    # ... and the communicator is made from them on every rank, through the ctypes binding (a fake RCCL here)
    fake = os.path.join(out_dir, "libfakerccl.so")
    binding, comm = pkdist.make_rccl_comm(rank, world, "cpu", pkdist.RcclBinding(fake))
    assert comm == 0x5000 + 16 * world + rank
    binding.comm_destroy(comm)
    # the form bench.py uses, which must leave no rank waiting when one of them cannot join: all succeed ...
    b2, c2, why = pkdist.try_make_rccl_comm(rank, world, "cpu", lambda: pkdist.RcclBinding(fake))
    assert c2 == 0x5000 + 16 * world + rank and why == ""
    b2.comm_destroy(c2)
    # ... rank 1 has no library to bind: EVERY rank comes back empty-handed, with a reason ...

    def broken():
        if rank == 1:
            raise RuntimeError("no RCCL on this rank")
        return pkdist.RcclBinding(fake)
    b3, c3, why = pkdist.try_make_rccl_comm(rank, world, "cpu", broken)
    assert b3 is None and c3 is None and why
    assert ("no RCCL on this rank" in why) == (rank == 1)
    # ... rank 1 is refused by ncclCommInitRank (a rank outside the communicator): the joined rank lets go again
    class Misnumbered(pkdist.RcclBinding):
        def comm_init_rank(self, nranks, uid_bytes, r):
            return super().comm_init_rank(nranks, uid_bytes, r + 7 * (r == 1))
    b4, c4, why = pkdist.try_make_rccl_comm(rank, world, "cpu", lambda: Misnumbered(fake))
    assert b4 is None and c4 is None and why
    assert ("invalid argument" in why) == (rank == 1)
    pkdist.barrier()
    open(os.path.join(out_dir, "ok%d" % rank), "w").write("ok")
    pkdist.shutdown()


def test_length_balanced_sharding_of_a_ragged_utterance_list():
    """length-balanced bin-packing when T varies.  2 048 utterances of U[2 s, 20 s] (bench.py --ragged
    at N = 8): longest-first greedy leaves the slowest rank within 2 % of the mean (measured 1e-5), u mod N does not."""
    from pocketkaldi_amd import synth
    secs = [synth.ragged_seconds(u) for u in range(2048)]
    assert min(secs) >= 2.0 and max(secs) <= 20.0 and len(set(secs)) > 2000
    frames = [pkdist.frames_of(round(s * synth.SAMPLE_RATE)) for s in secs]
    assert pkdist.frames_of(399) == 0 and pkdist.frames_of(400) == 1 and pkdist.frames_of(160000) == 998   # fbank.cc:35-42
    for world in (1, 2, 4, 8):
        shards = pkdist.partition_by_frames(frames, world)
        assert sorted(sum(shards, [])) == list(range(2048))            # every utterance exactly once
        assert all(s == sorted(s) for s in shards)
        bal, rr = pkdist.imbalance(frames, shards), pkdist.imbalance(frames, pkdist.partition_round_robin(frames, world))
        assert bal <= 0.02 and bal <= rr + 1e-12
        assert shards == pkdist.partition_by_frames(list(frames), world)   # a pure function of the list
    assert pkdist.imbalance(frames, pkdist.partition_round_robin(frames, 8)) > 0.02    # what the greedy removes
    # degenerate lists
    assert pkdist.partition_by_frames([], 4) == [[], [], [], []]
    assert pkdist.partition_by_frames([5], 2) == [[0], []]
    eq = pkdist.partition_by_frames([998] * 16, 8)
    assert all(len(s) == 2 for s in eq) and pkdist.imbalance([998] * 16, eq) == 0.0


FAKE_RCCL = r"""
/* a stand-in for librccl's four entry points, to test the ctypes binding (argument types, the 128-byte
   ncclUniqueId passed BY VALUE) without a GPU */
#include <string.h>
typedef struct { char internal[128]; } ncclUniqueId;
static int id_ok(const ncclUniqueId *u) { for (int i = 0; i < 128; ++i) if ((unsigned char)u->internal[i] != (unsigned char)(i * 7 + 3)) return 0; return 1; }
int ncclGetUniqueId(ncclUniqueId *u) { for (int i = 0; i < 128; ++i) u->internal[i] = (char)(i * 7 + 3); return 0; }
int ncclCommInitRank(void **comm, int nranks, ncclUniqueId id, int rank) {
  if (!id_ok(&id)) return 4;                       /* ncclInvalidArgument */
  if (rank < 0 || rank >= nranks) return 4;
  *comm = (void *)(long)(0x5000 + 16 * nranks + rank);
  return 0;
}
int ncclCommDestroy(void *comm) { return ((long)comm & ~0xffL) == 0x5000 ? 0 : 5; }
const char *ncclGetErrorString(int r) { return r == 4 ? "invalid argument (fake)" : r == 5 ? "invalid usage (fake)" : "?"; }
"""


def _build_fake_rccl(dirname):
    import subprocess
    src = os.path.join(dirname, "fakerccl.c")
    open(src, "w").write(FAKE_RCCL)
    lib = os.path.join(dirname, "libfakerccl.so")
    subprocess.check_call(["gcc", "-shared", "-fPIC", "-O1", src, "-o", lib])
    return lib


def test_rccl_ctypes_binding_against_a_fake_library(tmp_path):
    """pocketkaldi_amd.dist.RcclBinding is what bench.py uses at N > 1 to make the ncclComm_t it hands to
    pk_mi355_am_broadcast: the binding -- signatures, struct by value, error strings --
    is checked here against a library with RCCL's symbol table; the real library is used on the GPU box."""
    b = pkdist.RcclBinding(_build_fake_rccl(str(tmp_path)))
    uid = b.unique_id()
    assert uid == bytes((i * 7 + 3) & 0xFF for i in range(128))
    assert b.comm_init_rank(8, uid, 5) == 0x5000 + 16 * 8 + 5
    with pytest.raises(RuntimeError, match="invalid argument"):
        b.comm_init_rank(8, bytes(128), 5)                 # another id: the fake checks what arrived by value
    with pytest.raises(RuntimeError, match="invalid argument"):
        b.comm_init_rank(2, uid, 2)
    with pytest.raises(ValueError):
        b.comm_init_rank(2, uid[:100], 0)
    b.comm_destroy(0x5000 + 16 * 8 + 5)
    with pytest.raises(RuntimeError, match="invalid usage"):
        b.comm_destroy(0x77)
    # the copy torch maps at import (its own, bundled) is the one a real run binds to -- found without loading anything
    found = pkdist.loaded_rccl_path()
    assert found and os.path.exists(found) and "librccl" in os.path.basename(found)


def test_world2_gloo(tmp_path):
    world = 2
    _build_fake_rccl(str(tmp_path))
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(tmp_path / ("ok%d" % r)) for r in range(world))


def test_single_process_helpers_are_noops():
    assert pkdist.utterance_ids(0, 1, 3) == [0, 1, 2]
    assert pkdist.max_over_ranks(2.5) == 2.5
    assert pkdist.all_ranks_agree(1.0)
    t = torch.arange(4, dtype=torch.uint8)
    assert pkdist.broadcast_blob(t) is t
