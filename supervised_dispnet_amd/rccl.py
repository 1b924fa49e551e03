"""The gradient all-reduce on a library-owned HIP stream, straight on RCCL (SURVEY.md 8b "RCCL surface").

Replaces the reference's `torch.nn.DataParallel(disp_net)` gradient reduce (train.py:316-317) together with distributed.GradReducer.
torch.distributed is used for BOOTSTRAP only (rank / world from the launcher, the 128-byte ncclUniqueId travels through its store with
broadcast_object_list); the data path is

    ncclCommInitRank(comm, world, id, rank)                                once
    hipEventRecord(ev, producer stream) ; hipStreamWaitEvent(comm stream, ev)   per bucket, for BOTH compute streams (main + the weight-
    ncclAllReduce(buf, buf, n, ncclFloat, ncclSum, comm, comm stream)            gradient side stream, engine.WGRAD_STREAM)
    hipStreamWaitEvent(main, comm stream's last event)                     once, before the optimizer

so the exchange of bucket i overlaps the backward kernels of the layers below it, no host synchronisation anywhere, and the two compute
streams of the backward pass stay enabled under data parallelism (every bucket is fenced against both of them explicitly).
xGMI is point-to-point (7 links x ~153 GB/s per GPU): the 79.5 MB arena goes as a few ~20 MB buckets so that the ring's per-link
latency term is paid a handful of times, not per parameter.

The RCCL library is the one PyTorch already mapped (torch/lib/librccl.so): one copy of the runtime per process.
"""
import ctypes as C
import os

import torch

NCCL_UNIQUE_ID_BYTES = 128
ncclFloat32, ncclSum = 7, 0


class _UniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * NCCL_UNIQUE_ID_BYTES)]


_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    cands = [os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"), "librccl.so", "/opt/rocm/lib/librccl.so"]
    err = None
    for c in cands:
        try:
            lib = C.CDLL(c)
            break
        except OSError as e:
            err = e
    else:
        raise RuntimeError("librccl.so not found (%s)" % err)
    lib.ncclGetUniqueId.argtypes, lib.ncclGetUniqueId.restype = [C.POINTER(_UniqueId)], C.c_int
    lib.ncclCommInitRank.argtypes, lib.ncclCommInitRank.restype = [C.POINTER(C.c_void_p), C.c_int, _UniqueId, C.c_int], C.c_int
    lib.ncclAllReduce.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    lib.ncclAllReduce.restype = C.c_int
    lib.ncclCommDestroy.argtypes, lib.ncclCommDestroy.restype = [C.c_void_p], C.c_int
    lib.ncclGetErrorString.argtypes, lib.ncclGetErrorString.restype = [C.c_int], C.c_char_p
    _lib = lib
    return lib


def _check(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed: %s" % (what, load().ncclGetErrorString(rc).decode(errors="replace")))


class Communicator(object):
    """One RCCL communicator + one HIP stream owned by this library.  `world` / `rank` default to torch.distributed's (which must be
    initialised when world > 1: its store carries the unique id); a single-rank communicator needs no torch.distributed at all."""

    def __init__(self, device=None, rank=None, world=None):
        import torch.distributed as dist
        lib = load()
        initialised = dist.is_available() and dist.is_initialized()
        self.world = world if world is not None else (dist.get_world_size() if initialised else 1)
        self.rank = rank if rank is not None else (dist.get_rank() if initialised else 0)
        self.device = torch.device(device if device is not None else ("cuda", torch.cuda.current_device()))
        uid = _UniqueId()
        if self.rank == 0:
            _check(lib.ncclGetUniqueId(C.byref(uid)), "ncclGetUniqueId")
        if self.world > 1:
            box = [bytes(uid.internal) if self.rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            C.memmove(C.byref(uid), box[0], NCCL_UNIQUE_ID_BYTES)
        self.comm = C.c_void_p()
        with torch.cuda.device(self.device):
            _check(lib.ncclCommInitRank(C.byref(self.comm), self.world, uid, self.rank), "ncclCommInitRank")
            self.stream = torch.cuda.Stream(device=self.device)
        self._events = []
        # bench.py --comm-standin (a PROJECTION of the HBM side of a ring all-reduce on the one-GPU pool): round trips of every reduced
        # range through a scratch buffer on the communicator's stream, behind its ncclAllReduce.  0 = off (always, outside that flag).
        self.standin_round_trips = 0
        self._standin_scratch = None

    def all_reduce_sum_(self, tensor, producers):
        """In-place fp32 sum over the ranks on the communicator's stream, after everything enqueued so far on each stream of
        `producers` (event fences, no host involvement)."""
        if tensor.dtype != torch.float32 or not tensor.is_contiguous() or tensor.device != self.device:
            raise ValueError("all_reduce_sum_ takes a contiguous fp32 tensor on %s" % (self.device,))
        for s in producers:
            ev = torch.cuda.Event()
            ev.record(s)
            self.stream.wait_event(ev)
        _check(load().ncclAllReduce(tensor.data_ptr(), tensor.data_ptr(), tensor.numel(), ncclFloat32, ncclSum, self.comm,
                                    self.stream.cuda_stream), "ncclAllReduce")
        if self.standin_round_trips > 0 and tensor.numel() % 4 == 0 and tensor.data_ptr() % 16 == 0:
            from . import _lib as dn
            if self._standin_scratch is None or self._standin_scratch.numel() < tensor.numel():
                self._standin_scratch = torch.empty(tensor.numel(), dtype=torch.float32, device=self.device)
            for _ in range(self.standin_round_trips):
                dn.call("dn_ubench_copy", tensor.data_ptr(), self._standin_scratch.data_ptr(), tensor.numel(), self.stream.cuda_stream)
                dn.call("dn_ubench_copy", self._standin_scratch.data_ptr(), tensor.data_ptr(), tensor.numel(), self.stream.cuda_stream)

    def join(self, stream=None):
        """Make `stream` (default: the current one) wait for every collective enqueued so far."""
        (stream or torch.cuda.current_stream(self.device)).wait_stream(self.stream)

    def destroy(self):
        if self.comm:
            self.stream.synchronize()
            load().ncclCommDestroy(self.comm)
            self.comm = C.c_void_p()
