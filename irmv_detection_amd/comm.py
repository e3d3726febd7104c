"""ctypes binding of include/irmv_comm.h (libirmv_comm.so) + the launcher glue of the one-process-per-GPU form.

No torch anywhere: the ranks of `bench.py --gpus N` run the same torch-free hand-off as N = 1 (a torch HIP context in
the process puts libirmv_hip.so on torch's bundled ROCm 7.0 runtime instead of the 7.2 it was built against -- see
DESIGN.md section 6a).  The only thing the ranks exchange outside RCCL is the 128-byte communicator id, which rank 0
drops into a file named after the launcher's pid (single node, like the bench contract); a multi-node launcher would
carry the same 128 bytes through its own store.

Reference: single-device (test/yolo_test.cpp:16); sharding and the one broadcast are SURVEY.md section 8e.
"""
from __future__ import annotations

import ctypes as C
import os
import tempfile
import time
from typing import List, Optional, Tuple

from . import _build

ID_BYTES = 128
_lib = None


class CommError(RuntimeError):
    pass


def load():
    global _lib
    if _lib is None:
        path = os.environ.get("IRMV_COMM_LIB_PATH") or _build.build_comm()
        lib = C.CDLL(path)
        lib.irmv_comm_last_error.restype = C.c_char_p
        lib.irmv_comm_init_all.argtypes = [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_void_p)]
        lib.irmv_comm_unique_id.argtypes = [C.c_char_p]
        lib.irmv_comm_init_rank.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        lib.irmv_comm_nranks.argtypes = [C.c_void_p]
        lib.irmv_comm_local_ranks.argtypes = [C.c_void_p]
        lib.irmv_comm_broadcast_blob.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
        lib.irmv_comm_allreduce_f64.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_int]
        lib.irmv_comm_destroy.argtypes = [C.c_void_p]
        lib.irmv_comm_destroy.restype = None
        _lib = lib
    return _lib


def _check(rc: int) -> None:
    if rc != 0:
        raise CommError(f"irmv_comm error {rc}: {load().irmv_comm_last_error().decode(errors='replace')}")


def env_rank_world() -> Tuple[int, int, int]:
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")))


def device_index(local_rank: int) -> int:
    """HIP device of this rank: LOCAL_RANK, unless IRMV_FORCE_DEVICE pins every rank to one card (rehearsal)."""
    forced = os.environ.get("IRMV_FORCE_DEVICE")
    return int(forced) if forced is not None else local_rank


def shard_frames(total_frames: int, rank: int, world: int) -> List[int]:
    """Round-robin ownership: frame i -> rank i mod world."""
    return list(range(rank, total_frames, world))


def id_file() -> str:
    """Where rank 0 leaves the communicator id: keyed by the launcher (parent) pid, the rendezvous port and -- under an
    elastic launcher, whose agent (and with it the parent pid and the port) survives a worker restart -- the run id and the
    restart count, so that neither two launches on one node nor two generations of one launch ever see each other's file."""
    gen = "_".join(os.environ.get(k, "") for k in ("TORCHELASTIC_RUN_ID", "TORCHELASTIC_RESTART_COUNT"))
    gen = "".join(c if c.isalnum() else "-" for c in gen)[:48]
    return os.path.join(tempfile.gettempdir(), f"irmv_comm_{os.getppid()}_{os.environ.get('MASTER_PORT', '0')}_{gen}.id")


_T_START = time.time()      # this process's start (import) time: id files much older than it belong to an earlier launch
NONCE_BYTES = 16
STALE_S = 120.0


def exchange_id(rank: int, make_id, path: Optional[str] = None, timeout_s: float = 180.0, with_nonce: bool = False):
    """Rank 0 calls make_id() and publishes the bytes (write + atomic rename) followed by a per-launch nonce; the others
    wait for the file.  A file left behind by an earlier launch that happened to share the launcher pid and port is never
    accepted: rank 0 replaces it, and the others ignore a file written more than STALE_S before they started.
    -> the id bytes, or (id bytes, nonce string) with with_nonce."""
    path = path or id_file()
    if rank == 0:
        data = make_id()
        nonce = f"{int(_T_START * 1e3) & 0xffffffffff:010x}{os.getpid() & 0xffffff:06x}".encode()
        assert len(nonce) == NONCE_BYTES
        tmp = f"{path}.{os.getpid()}.tmp"
        with open(tmp, "wb") as f:
            f.write(data + nonce)
        os.replace(tmp, path)
        return (data, nonce.decode()) if with_nonce else data
    t0 = time.time()
    while True:
        try:
            if os.path.getmtime(path) >= _T_START - STALE_S:
                with open(path, "rb") as f:
                    data = f.read()
                if len(data) == ID_BYTES + NONCE_BYTES:
                    return (data[:ID_BYTES], data[ID_BYTES:].decode()) if with_nonce else data[:ID_BYTES]
        except (FileNotFoundError, OSError):
            pass
        if time.time() - t0 > timeout_s:
            raise CommError(f"rank {rank}: no (fresh) communicator id at {path} after {timeout_s:.0f} s")
        time.sleep(0.02)


class FileReduce:
    """Reductions of one float per rank through small files (single node): the side channel that carries the communicator
    id also carries the ranks' agreement on whether RCCL came up, and -- if it did not -- the bench's barriers and clocks,
    so that a node where RCCL cannot initialise still yields per-GPU numbers (reported as such; the weight broadcast is
    then replaced by every rank generating the same seeded blob)."""

    def __init__(self, rank: int, world: int, base: str):
        # `base` carries the launch's nonce (Comm.__init__): rounds of another launch can never be read as this one's
        self.rank, self.world, self.base, self.seq = rank, world, base, 0

    def gather(self, x: float, timeout_s: float = 600.0) -> List[float]:
        self.seq += 1
        mine = f"{self.base}.r{self.seq}.{self.rank}"
        tmp = mine + ".tmp"
        with open(tmp, "w") as f:
            f.write(repr(float(x)))
        os.replace(tmp, mine)
        vals, t0 = [], time.time()
        for r in range(self.world):
            path = f"{self.base}.r{self.seq}.{r}"
            while True:
                try:
                    with open(path) as f:
                        vals.append(float(f.read()))
                    break
                except (FileNotFoundError, ValueError):
                    if time.time() - t0 > timeout_s:
                        raise CommError(f"rank {self.rank}: rank {r} never reached reduction {self.seq}")
                    time.sleep(0.002)
        if self.seq > 2:        # everybody has read round seq - 2 by now (they wrote seq - 1 after reading it)
            old = f"{self.base}.r{self.seq - 2}.{self.rank}"
            if os.path.exists(old):
                os.remove(old)
        return vals

    def close(self, ack_timeout_s: float = 30.0) -> None:
        """Two closing rounds: once the second is complete on a rank, every rank has READ the first, so each rank removes
        its own files of every round up to the first and then ACKNOWLEDGES -- a `done.<rank>` file written after its second
        gather returned, i.e. after it has read every file of the last round.  Rank 0 sweeps the launch's prefix (unique to
        the launch) only once every acknowledgement is there; past `ack_timeout_s` it leaves the last round's few bytes to
        the temp directory instead of pulling a file from under a descheduled peer.  The job has succeeded by the time
        close() runs: a peer that disappears during it costs nothing but those files (no CommError out of here)."""
        import glob
        try:
            self.gather(0.0, timeout_s=ack_timeout_s)
            self.gather(0.0, timeout_s=ack_timeout_s)
        except CommError:
            return
        for q in range(1, self.seq):
            path = f"{self.base}.r{q}.{self.rank}"
            if os.path.exists(path):
                os.remove(path)
        with open(f"{self.base}.done.{self.rank}", "w"):
            pass
        if self.rank != 0:
            return
        t0 = time.time()
        while not all(os.path.exists(f"{self.base}.done.{r}") for r in range(self.world)):
            if time.time() - t0 > ack_timeout_s:
                return
            time.sleep(0.002)
        for path in glob.glob(glob.escape(self.base) + ".r*") + glob.glob(glob.escape(self.base) + ".done.*"):
            try:
                os.remove(path)
            except OSError:
                pass


class Comm:
    """One process per GPU (RANK / LOCAL_RANK / WORLD_SIZE from the launcher).  With one rank nothing is loaded and every
    call is the identity.  `native` tells whether the RCCL communicator is up on EVERY rank (the ranks agree on it through
    the file channel); if not, reductions fall back to that channel and broadcast_blob() returns None."""

    def __init__(self):
        self.rank, self.local_rank, self.world = env_rank_world()
        self.device = device_index(self.local_rank)
        self._h = C.c_void_p()
        self._id_path = None
        self._files = None
        self.native = self.world > 1
        self.native_error = None
        if self.world > 1:
            self._id_path = id_file()
            if self.rank == 0 and os.path.exists(self._id_path):
                os.remove(self._id_path)            # a leftover of an earlier launch with this launcher pid and port
            # Stage 1, BEFORE anyone enters ncclCommInitRank (which blocks until every rank has joined): each rank reports
            # whether its library loaded, its device exists and the id is a real one.  One failing rank then takes everybody
            # to the fallback together, instead of leaving the healthy ranks inside RCCL waiting for it.
            L, uid, pre_ok = None, bytes(ID_BYTES), 1.0
            try:
                L = load()
            except (CommError, OSError) as e:
                pre_ok, self.native_error = 0.0, f"libirmv_comm.so: {e}"

            def make():
                if L is None:
                    return bytes(ID_BYTES)          # the others are waiting for a file: a dummy id, refused below
                buf = C.create_string_buffer(ID_BYTES)
                if L.irmv_comm_unique_id(buf) != 0:
                    return bytes(ID_BYTES)
                return buf.raw
            uid, nonce = exchange_id(self.rank, make, self._id_path, with_nonce=True)
            self._files = FileReduce(self.rank, self.world, f"{self._id_path}.{nonce}")
            if uid == bytes(ID_BYTES):
                pre_ok, self.native_error = 0.0, self.native_error or "rank 0 could not create a communicator id"
            try:
                from . import capi
                if self.device < 0 or self.device >= capi.device_count():
                    pre_ok, self.native_error = 0.0, self.native_error or f"no HIP device {self.device} on this rank"
            except Exception as e:                  # noqa: BLE001 -- whatever keeps this rank from counting devices keeps it out of RCCL
                pre_ok, self.native_error = 0.0, self.native_error or f"device count: {e}"
            ok = 1.0 if min(self._files.gather(pre_ok)) > 0.5 else 0.0
            if ok > 0.5:
                # Stage 2: everybody goes in.  A rank that fails in here reports it and then waits only briefly: its peers
                # may be blocked inside ncclCommInitRank for good, and a prompt non-zero exit lets the launcher end the job.
                try:
                    _check(L.irmv_comm_init_rank(uid, self.world, self.rank, self.device, C.byref(self._h)))
                except (CommError, OSError) as e:
                    ok, self.native_error = 0.0, str(e)
            stage2_wait = float(os.environ.get("IRMV_COMM_STAGE2_WAIT_S", "60"))
            self.native = min(self._files.gather(ok, timeout_s=600.0 if ok > 0.5 else stage2_wait)) > 0.5
            if not self.native:
                import sys
                print(f"[irmv_comm] rank {self.rank}: RCCL communicator not available on every rank"
                      f"{' (' + self.native_error + ')' if self.native_error else ''}: file-based reductions, no weight broadcast", file=sys.stderr, flush=True)
                if self._h:
                    load().irmv_comm_destroy(self._h)
                    self._h = C.c_void_p()

    def broadcast_blob(self, blob: Optional[bytes], root: int = 0) -> Optional[Tuple[int, int]]:
        """-> (device pointer, bytes) of the blob on this rank's GPU (communicator-owned).  Only `root` passes bytes.
        None when the RCCL communicator is not up (every rank then builds the blob itself)."""
        assert self.world > 1
        if not self.native:
            return None
        L = load()
        ptr, n = C.c_void_p(), C.c_uint64()
        if self.rank == root:
            assert blob is not None
            buf = (C.c_ubyte * len(blob)).from_buffer_copy(blob)
            _check(L.irmv_comm_broadcast_blob(self._h, buf, len(blob), root, C.byref(ptr), C.byref(n)))
        else:
            _check(L.irmv_comm_broadcast_blob(self._h, None, 0, root, C.byref(ptr), C.byref(n)))
        return int(ptr.value), int(n.value)

    def _reduce(self, x: float, op: int) -> float:
        if self.world == 1:
            return float(x)
        if not self.native:
            vals = self._files.gather(x)
            return float(max(vals) if op == 1 else sum(vals))
        v = C.c_double(x)
        _check(load().irmv_comm_allreduce_f64(self._h, C.byref(v), op))
        return float(v.value)

    def sum_over_ranks(self, x: float) -> float:
        return self._reduce(x, 0)

    def max_over_ranks(self, x: float) -> float:
        return self._reduce(x, 1)

    def barrier(self) -> None:
        self._reduce(0.0, 0)

    def close(self) -> None:
        if self._h:
            load().irmv_comm_destroy(self._h)
            self._h = C.c_void_p()
        if self._files:
            self._files.close()
        if self.rank == 0 and self._id_path and os.path.exists(self._id_path):
            os.remove(self._id_path)
