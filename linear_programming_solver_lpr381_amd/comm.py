"""X1 -- the incumbent exchange of the sharded searches on the library's own RCCL communicator (include/lpx.h, lpx_comm_*).

One process per GPU: `lpx_init(local_rank)`, then every rank calls `init(rank, world, id)` with the 128 bytes rank 0 got from
`unique_id()` (the host ships them: here any callable `bcast(bytes_or_None) -> bytes`), or `init_tcp(...)` when no side channel
exists.  While the communicator lives, `BranchAndBound(..., rank=r, world=w)` / `BranchAndBoundKnapsack(...)` without an
`allreduce_max` callback exchange their bound through it: ncclAllReduce(ncclMax, ncclDouble), Models/Branch&Bound.cs:182,191.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib

ID_BYTES = 128


def unique_id() -> bytes:
    buf = (C.c_uint8 * ID_BYTES)()
    _lib.check(_lib.lib().lpx_comm_unique_id(buf))
    return bytes(buf)


def init(rank: int, world: int, uid: bytes) -> None:
    if len(uid) != ID_BYTES:
        raise ValueError(f"the communicator id has {ID_BYTES} bytes, got {len(uid)}")
    buf = (C.c_uint8 * ID_BYTES).from_buffer_copy(uid)
    _lib.check(_lib.lib().lpx_comm_init(rank, world, buf))


def init_tcp(rank: int, world: int, host: str = "127.0.0.1", port: int = 29641) -> None:
    _lib.check(_lib.lib().lpx_comm_init_tcp(rank, world, host.encode(), port))


def allreduce_max(vals) -> np.ndarray:
    a = np.ascontiguousarray(vals, dtype=np.float64).copy()
    _lib.check(_lib.lib().lpx_comm_allreduce_max(a.ctypes.data_as(_lib.dp), a.size))
    return a


def info() -> dict:
    r, w, n, ms, v = C.c_int(), C.c_int(), C.c_int64(), C.c_double(), C.c_int()
    _lib.check(_lib.lib().lpx_comm_info(C.byref(r), C.byref(w), C.byref(n), C.byref(ms), C.byref(v)))
    return {"rank": r.value, "world": w.value, "allreduces": n.value, "allreduce_ms": ms.value, "rccl_version": v.value}


def destroy() -> None:
    _lib.check(_lib.lib().lpx_comm_destroy())
