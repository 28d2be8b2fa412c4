"""Several GPUs behind the C ABI (include/biogpu.h, csrc/comm.hip): shard ranges and the one collective of a sharded
batch — the all-gather of fixed-size result records — over RCCL (`Comm.rccl`) or, for the ranks of one node that RCCL
cannot serve (several ranks on one GPU; no GPU at all), staged through POSIX shared memory (`Comm.host`).
shard.py does the same through torch.distributed for bench.py; this is what a host program in any language binds."""
import ctypes as C

import numpy as np

from . import _lib

ID_BYTES = 128


def shard_range(n_units, rank, world):
    lo, hi = C.c_uint64(0), C.c_uint64(0)
    _lib.check(_lib.lib().bg_shard_range(n_units, rank, world, C.byref(lo), C.byref(hi)), "bg_shard_range")
    return int(lo.value), int(hi.value)


def shard_balanced(costs, world):
    c = np.ascontiguousarray(costs, dtype=np.uint64)
    b = np.zeros(world + 1, dtype=np.uint64)
    _lib.check(_lib.lib().bg_shard_balanced(c.ctypes.data, len(c), world, b.ctypes.data), "bg_shard_balanced")
    return [int(v) for v in b]


def unique_id():
    buf = np.zeros(ID_BYTES, dtype=np.uint8)
    _lib.check(_lib.lib().bg_comm_unique_id(buf.ctypes.data), "bg_comm_unique_id")
    return buf.tobytes()


class Comm:
    def __init__(self, h, rank, world, ctx):
        self.h, self.rank, self.world, self.ctx = h, rank, world, ctx

    @classmethod
    def rccl(cls, ctx, rank, world, uid):
        h = C.c_void_p()
        buf = np.frombuffer(uid, dtype=np.uint8).copy()
        _lib.check(_lib.lib().bg_comm_init(ctx.h, rank, world, buf.ctypes.data, C.byref(h)), "bg_comm_init")
        return cls(h, rank, world, ctx)

    @classmethod
    def host(cls, ctx, rank, world, name):
        h = C.c_void_p()
        _lib.check(_lib.lib().bg_comm_init_host(ctx.h if ctx is not None else None, rank, world, name.encode(), C.byref(h)),
                   "bg_comm_init_host")
        return cls(h, rank, world, ctx)

    def gather_ptr(self, local_ptr, n_local, rec_bytes, all_ptr, stream=0, all_cap=None):
        """device (or, for a ctx-less host communicator, host) pointers; returns the per-rank record counts.
        all_cap: records `all_ptr` can hold — checked on every rank before anything moves (bg_gather_records_cap)"""
        counts = np.zeros(self.world, dtype=np.uint64)
        if all_cap is None:
            _lib.check(_lib.lib().bg_gather_records(self.h, local_ptr, n_local, rec_bytes, all_ptr, counts.ctypes.data, stream),
                       "bg_gather_records")
        else:
            _lib.check(_lib.lib().bg_gather_records_cap(self.h, local_ptr, n_local, rec_bytes, all_ptr, all_cap, counts.ctypes.data, stream),
                       "bg_gather_records_cap")
        return counts

    def gather_host(self, local, total_records):
        """numpy records [n_local, ...] in host memory -> all records in rank order (bg_gather_records_host: through the shared
        segment for a host communicator, staged through device scratch for an RCCL one); total_records = the room of the result"""
        local = np.ascontiguousarray(local)
        rec_bytes = local.dtype.itemsize * int(np.prod(local.shape[1:], dtype=np.int64))
        out = np.zeros((total_records,) + local.shape[1:], dtype=local.dtype)
        counts = np.zeros(self.world, dtype=np.uint64)
        _lib.check(_lib.lib().bg_gather_records_host(self.h, local.ctypes.data, local.shape[0], rec_bytes, out.ctypes.data, total_records,
                                                     counts.ctypes.data), "bg_gather_records_host")
        return out[:int(counts.sum())], counts

    def world_info(self):
        """bg_comm_world: what the communicator is, read back from RCCL itself — {"world", "rccl_ranks" (ncclCommCount; 0 for a
        host-staged communicator), "rccl_rank", "last_path" ("all_gather" | "grouped_broadcast" (ragged) | "host_shm" | None),
        "gathers"}"""
        info = np.zeros(5, dtype=np.int64)
        _lib.check(_lib.lib().bg_comm_world(self.h, info.ctypes.data), "bg_comm_world")
        return {"world": int(info[0]), "rccl_ranks": int(info[1]), "rccl_rank": int(info[2]),
                "last_path": [None, "all_gather", "grouped_broadcast", "host_shm"][int(info[3])], "gathers": int(info[4])}

    def free(self):
        if self.h:
            _lib.lib().bg_comm_free(self.h)
            self.h = None
