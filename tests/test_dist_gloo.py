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
    pkdist.barrier()
    open(os.path.join(out_dir, "ok%d" % rank), "w").write("ok")
    pkdist.shutdown()


def test_world2_gloo(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(tmp_path / ("ok%d" % r)) for r in range(world))


def test_single_process_helpers_are_noops():
    assert pkdist.utterance_ids(0, 1, 3) == [0, 1, 2]
    assert pkdist.max_over_ranks(2.5) == 2.5
    assert pkdist.all_ranks_agree(1.0)
    t = torch.arange(4, dtype=torch.uint8)
    assert pkdist.broadcast_blob(t) is t
