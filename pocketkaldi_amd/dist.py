"""Multi-GPU plumbing: one process per GPU, utterance-sharded, no data-path collective.

The path shards by utterance (the only cross-frame state -- CMVN window, +-context
splice -- lives inside one utterance), every rank holds a full weight replica, and
the only exchange is ONE broadcast of the packed weight blob at load (RCCL over xGMI
on GPUs; ``backend="nccl"`` is RCCL on ROCm).  These helpers are backend-agnostic
so the same code is exercised with ``gloo`` on CPU in tests/test_dist_gloo.py.
"""
import ctypes as C
import os

import torch
import torch.distributed as dist


def env_world():
    """(rank, local_rank, world_size) from the torch.distributed.run environment."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init(backend):
    """Join the job described by RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT."""
    rank, local_rank, world = env_world()
    force = os.environ.get("PK_DIST_FORCE", "") == "1"    # exercise the collective path at world 1
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def shutdown():
    if dist.is_initialized():
        dist.destroy_process_group()


def utterance_ids(rank, world, per_rank):
    """Utterance u goes to rank u mod world; weak scaling: per_rank each."""
    return [rank + world * i for i in range(per_rank)]


def frames_of(num_samples):
    """Fbank::CalcNumFrames (fbank.cc:35-42): 25 ms windows, 10 ms hop, snip-edges."""
    n = int(num_samples)
    return 0 if n < 400 else 1 + (n - 400) // 160


def partition_by_frames(frames, world):
    """Length-balanced sharding for ragged utterance lists (the reference's own
    workload is a ragged list: main.cc:34-46 scores one WAV after another, each T from fbank.cc:35-42).
    Longest-first greedy: utterances in order of decreasing frame count, each to the rank with the
    fewest frames so far (ties: the lower rank; equal lengths keep their input order, so the map is
    a pure function of `frames` and every rank computes the same one without talking).
    -> list of `world` lists of utterance indices, each in increasing order."""
    import heapq
    order = sorted(range(len(frames)), key=lambda u: (-int(frames[u]), u))
    heap = [(0, r) for r in range(world)]
    shards = [[] for _ in range(world)]
    for u in order:
        load, r = heapq.heappop(heap)
        shards[r].append(u)
        heapq.heappush(heap, (load + int(frames[u]), r))
    return [sorted(s) for s in shards]


def partition_round_robin(frames, world):
    """u -> rank u mod world over the same list (what utterance_ids does for equal lengths)."""
    return [list(range(r, len(frames), world)) for r in range(world)]


def imbalance(frames, shards):
    """max over ranks of the shard's frames / mean - 1: the fraction of the slowest rank's time the
    average rank would sit idle."""
    loads = [sum(int(frames[u]) for u in s) for s in shards]
    mean = sum(loads) / float(len(loads))
    return (max(loads) / mean - 1.0) if mean > 0 else 0.0


def broadcast_blob(blob_u8, src=0):
    """The one collective of the path: rank `src`'s weight blob to every rank, in place.
    (With the gloo rehearsal backend a device tensor is staged through the host.)"""
    if dist.is_initialized():
        if blob_u8.is_cuda and dist.get_backend() == "gloo":
            host = blob_u8.cpu()
            dist.broadcast(host, src=src)
            blob_u8.copy_(host)
        else:
            dist.broadcast(blob_u8, src=src)
    return blob_u8


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def max_over_ranks(value, device="cpu"):
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, device="cpu"):
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def all_ranks_agree(value, device="cpu"):
    """True iff every rank holds the same float (used to verify the weight broadcast)."""
    hi = torch.tensor([float(value)], dtype=torch.float64, device=device)
    lo = hi.clone()
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    return float(hi.item()) == float(lo.item())


class DeviceBytes:
    """A raw device allocation exposed through __cuda_array_interface__ so that
    torch can alias it (zero-copy) for the broadcast."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1",
                                         "data": (int(ptr), False), "version": 2}


def alias_device_bytes(ptr, nbytes, device):
    return torch.as_tensor(DeviceBytes(ptr, nbytes), device=device)


def broadcast_model(am, device, src=0):
    """Replicate rank `src`'s weights into this rank's model, in place: the model's packed device
    blob is aliased as a torch tensor (zero-copy) and broadcast -- the ONE collective of the path
    (RCCL over xGMI with backend "nccl"; staged through the host under the gloo rehearsal backend).
    The C/C++ form of the same step is pk_mi355_am_broadcast (include/pk_mi355.h)."""
    if not dist.is_initialized():
        return
    ptr, nbytes = am.blob()
    broadcast_blob(alias_device_bytes(ptr, nbytes, device), src=src)
    torch.cuda.synchronize(device)


# ---- an ncclComm_t of our own, through ctypes on the RCCL copy ALREADY LOADED in the process (torch's), so that a
# multi-rank run exercises the C entry pk_mi355_am_broadcast as well as the torch.distributed path (VERDICT round 3,
# next #4).  rccl.h: ncclGetUniqueId(ncclUniqueId *), ncclCommInitRank(ncclComm_t *, int nranks, ncclUniqueId id
# BY VALUE, int rank), ncclCommDestroy(ncclComm_t), ncclGetErrorString(ncclResult_t); NCCL_UNIQUE_ID_BYTES = 128.

NCCL_UNIQUE_ID_BYTES = 128


class NcclUniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * NCCL_UNIQUE_ID_BYTES)]


def loaded_rccl_path():
    """Path of the librccl mapped into this process (torch loads its own copy for backend "nccl"), or None."""
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                path = line.rsplit(" ", 1)[-1].strip()
                if "librccl" in os.path.basename(path):
                    return path
    except OSError:
        pass
    return None


class RcclBinding:
    """The four RCCL entry points the C-ABI check needs, bound with ctypes.  `path`: the library file (dlopen of a
    file that is already mapped returns that copy -- one RCCL per process, as pk_mi355_am_broadcast requires)."""

    def __init__(self, path=None):
        path = path or os.environ.get("PK_MI355_RCCL_LIB") or loaded_rccl_path()
        if not path:
            raise RuntimeError("no RCCL is loaded in this process (init a torch.distributed nccl group first, "
                               "or name the library in PK_MI355_RCCL_LIB)")
        self.path = path
        self.lib = C.CDLL(path)
        self.lib.ncclGetUniqueId.argtypes = [C.POINTER(NcclUniqueId)]
        self.lib.ncclGetUniqueId.restype = C.c_int
        self.lib.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, NcclUniqueId, C.c_int]
        self.lib.ncclCommInitRank.restype = C.c_int
        self.lib.ncclCommDestroy.argtypes = [C.c_void_p]
        self.lib.ncclCommDestroy.restype = C.c_int
        self.lib.ncclGetErrorString.argtypes = [C.c_int]
        self.lib.ncclGetErrorString.restype = C.c_char_p

    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError("%s failed: %s" % (what, self.lib.ncclGetErrorString(rc).decode()))

    def unique_id(self):
        uid = NcclUniqueId()
        self._check(self.lib.ncclGetUniqueId(C.byref(uid)), "ncclGetUniqueId")
        return C.string_at(C.addressof(uid), NCCL_UNIQUE_ID_BYTES)

    def comm_init_rank(self, nranks, uid_bytes, rank):
        if len(uid_bytes) != NCCL_UNIQUE_ID_BYTES:
            raise ValueError("ncclUniqueId is %d bytes" % NCCL_UNIQUE_ID_BYTES)
        uid = NcclUniqueId()
        C.memmove(C.addressof(uid), uid_bytes, NCCL_UNIQUE_ID_BYTES)
        comm = C.c_void_p()
        self._check(self.lib.ncclCommInitRank(C.byref(comm), int(nranks), uid, int(rank)), "ncclCommInitRank")
        return comm.value

    def comm_destroy(self, comm):
        self._check(self.lib.ncclCommDestroy(C.c_void_p(comm)), "ncclCommDestroy")


def share_bytes(payload, n, src=0, device="cpu"):
    """`n` bytes from rank `src` to every rank through the process group (the ncclUniqueId hand-shake)."""
    t = torch.zeros(n, dtype=torch.uint8, device=device)
    if payload is not None:
        t.copy_(torch.frombuffer(bytearray(payload), dtype=torch.uint8))
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(t, src=src)
    return bytes(t.cpu().numpy().tobytes())


def make_rccl_comm(rank, world, device="cpu", binding=None):
    """(binding, ncclComm_t) over all ranks of the job: rank 0 draws the id, one small broadcast shares it,
    every rank joins.  No new process, nothing re-executed."""
    b = binding or RcclBinding()
    os.environ.setdefault("PK_MI355_RCCL_LIB", b.path)    # libpk_mi355 binds ncclBroadcast in the same copy (capi_collective.hip: BindRccl)
    uid = share_bytes(b.unique_id() if rank == 0 else None, NCCL_UNIQUE_ID_BYTES, 0, device)
    return b, b.comm_init_rank(world, uid, rank)


def try_make_rccl_comm(rank, world, device="cpu", binding_factory=RcclBinding):
    """make_rccl_comm for a caller that must go on without it (bench.py): (binding, comm, "") when EVERY rank joined,
    else (None, None, reason) on every rank.  Each step that can fail on one rank alone is followed by an agreement
    over the process group, so no rank is left waiting in a collective the others never enter."""
    binding, why = None, ""
    try:
        binding = binding_factory()
        os.environ.setdefault("PK_MI355_RCCL_LIB", binding.path)
    except Exception as e:      # noqa: BLE001 -- the reason travels to the caller
        why = "%s: %s" % (type(e).__name__, e)
    if not all_ranks_agree(0.0 if binding else 1.0 + rank, device) or binding is None:
        return None, None, why or "another rank could not bind RCCL"
    uid = None
    if rank == 0:
        try:
            uid = binding.unique_id()
        except Exception as e:      # noqa: BLE001
            why = "%s: %s" % (type(e).__name__, e)
    uid = share_bytes(uid, NCCL_UNIQUE_ID_BYTES, 0, device)       # all zero when the root could not draw one
    if not any(uid):
        return None, None, why or "the root could not draw an ncclUniqueId"
    comm = None
    try:
        comm = binding.comm_init_rank(world, uid, rank)
    except Exception as e:      # noqa: BLE001
        why = "%s: %s" % (type(e).__name__, e)
    if not all_ranks_agree(0.0 if comm else 1.0 + rank, device) or not comm:
        if comm:
            binding.comm_destroy(comm)
        return None, None, why or "another rank could not join the communicator"
    return binding, comm, ""


def min_max_over_ranks(value, device="cpu"):
    lo = torch.tensor([float(value)], dtype=torch.float64, device=device)
    hi = lo.clone()
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    return float(lo.item()), float(hi.item())
