"""libirmv_comm.so (include/irmv_comm.h): the RCCL weight broadcast of the multi-GPU path, without torch.

CPU: the library loads and exports what the header declares; frames shard round-robin; the 128-byte communicator id
travels from rank 0 to the other rank processes; a one-rank Comm is the identity and loads nothing.
GPU (one MI355X): ncclCommInitAll(1) and ncclCommInitRank(1 of 1) come up, the broadcast blob arrives in device memory and
an engine built from that device pointer gives the detections of an engine built from the host blob; the C++ runner
(tools/irmv_multi_gpu.cpp: one thread + one engine per GPU) runs end to end.  More than one GPU is the driver's to
measure (SCALE_rNN.json); the reference is single-device (test/yolo_test.cpp:16).
"""
import ctypes as C
import json
import multiprocessing as mp
import os
import time
import re
import subprocess

import numpy as np
import pytest

from conftest import ROOT
from irmv_detection_amd import _build, comm

HEADER = os.path.join(ROOT, "include", "irmv_comm.h")
BIN = os.path.join(ROOT, "tests", "cpp", "_bin")


def _runner():
    os.makedirs(BIN, exist_ok=True)
    _build.build(); _build.build_comm()
    exe = os.path.join(BIN, "irmv_multi_gpu")
    src = os.path.join(ROOT, "tools", "irmv_multi_gpu.cpp")
    if not os.path.exists(exe) or os.path.getmtime(exe) < os.path.getmtime(src):
        subprocess.check_call([_build.hipcc(), "-O2", "-std=c++17", "-I", os.path.join(ROOT, "include"), src, "-L", _build.LIB_DIR,
                               "-lirmv_hip", "-lirmv_comm", "-lpthread", f"-Wl,-rpath,{_build.LIB_DIR}", "-o", exe])
    return exe


def test_library_exports_every_declared_symbol():
    lib = comm.load()
    declared = set(re.findall(r"\b(irmv_comm_[a-z0-9_]+)\s*\(", open(HEADER).read()))
    assert len(declared) == 9, declared
    for name in declared:
        assert hasattr(lib, name), name
    needed = subprocess.check_output(["readelf", "-d", _build.COMM_PATH], text=True)
    assert "librccl.so" in needed and "torch" not in needed


def test_runner_builds_on_the_cpu_host():
    assert os.path.exists(_runner())   # built here so the binary travels to the GPU box with the tree


def test_frames_shard_round_robin():
    assert comm.shard_frames(10, 1, 4) == [1, 5, 9]
    owners = sorted(i for r in range(8) for i in comm.shard_frames(64, r, 8))
    assert owners == list(range(64))


def _reader(path, q):
    q.put(comm.exchange_id(1, lambda: b"", path, timeout_s=30))


def test_communicator_id_reaches_the_other_rank(tmp_path):
    path = str(tmp_path / "id")
    q = mp.get_context("spawn").Queue()
    p = mp.get_context("spawn").Process(target=_reader, args=(path, q))
    p.start()                                    # the reader polls before the file exists
    uid = bytes(range(128))
    assert comm.exchange_id(0, lambda: uid, path) == uid
    assert q.get(timeout=60) == uid
    p.join(30)
    assert p.exitcode == 0


def _reducer(rank, base, q):
    fr = comm.FileReduce(rank, 2, base)
    out = [fr.gather(float(rank + 1)), fr.gather(10.0 * (rank + 1)), fr.gather(0.5), fr.gather(float(rank))]
    fr.close()
    q.put((rank, out))


def test_file_channel_reductions_between_two_processes(tmp_path):
    """The fallback the bench's ranks agree on when RCCL cannot come up on a node: every rank sees every rank's value."""
    base = str(tmp_path / "chan")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_reducer, args=(r, base, q)) for r in range(2)]
    for p in ps:
        p.start()
    got = dict(q.get(timeout=60) for _ in range(2))
    for p in ps:
        p.join(30)
        assert p.exitcode == 0
    assert got[0] == got[1] == [[1.0, 2.0], [10.0, 20.0], [0.5, 0.5], [0.0, 1.0]]
    assert [f for f in os.listdir(tmp_path) if f.startswith("chan.r")] == []   # two closing rounds, then rank 0 sweeps the launch's prefix: nothing stays


def test_a_stale_id_file_is_not_accepted(tmp_path):
    """A file left by an earlier launch that shared the launcher pid and port (crash before cleanup): a non-root rank ignores
    it -- it is older than the rank itself by more than STALE_S -- and takes the fresh one; the nonce rank 0 appends keys the
    reduction files of THIS launch."""
    path = str(tmp_path / "id")
    with open(path, "wb") as f:
        f.write(bytes([7]) * comm.ID_BYTES + b"0" * comm.NONCE_BYTES)
    old = time.time() - comm.STALE_S - 3600
    os.utime(path, (old, old))
    with pytest.raises(comm.CommError):
        comm.exchange_id(1, lambda: b"", path, timeout_s=0.3)
    uid = bytes(range(128))
    got, nonce = comm.exchange_id(0, lambda: uid, path, with_nonce=True)
    assert got == uid and len(nonce) == comm.NONCE_BYTES
    assert comm.exchange_id(1, lambda: b"", path, timeout_s=5, with_nonce=True) == (uid, nonce)


def test_one_rank_is_the_identity(monkeypatch):
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        monkeypatch.delenv(k, raising=False)
    c = comm.Comm()
    assert (c.rank, c.world, c.device) == (0, 1, 0)
    assert c.max_over_ranks(1.5) == 1.5 and c.sum_over_ranks(2.5) == 2.5
    c.barrier(); c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("how", ["init_all", "init_rank"])
def test_broadcast_blob_feeds_an_engine(blob, frame0, how):
    from irmv_detection_amd.engine import YoloEngine
    L = comm.load()
    h = C.c_void_p()
    if how == "init_all":
        assert L.irmv_comm_init_all(1, None, C.byref(h)) == 0, L.irmv_comm_last_error()
    else:
        uid = C.create_string_buffer(comm.ID_BYTES)
        assert L.irmv_comm_unique_id(uid) == 0, L.irmv_comm_last_error()
        assert L.irmv_comm_init_rank(uid.raw, 1, 0, 0, C.byref(h)) == 0, L.irmv_comm_last_error()
    try:
        assert L.irmv_comm_nranks(h) == 1 and L.irmv_comm_local_ranks(h) == 1
        ptr, n = C.c_void_p(), C.c_uint64()
        buf = (C.c_ubyte * len(blob)).from_buffer_copy(blob)
        assert L.irmv_comm_broadcast_blob(h, buf, len(blob), 0, C.byref(ptr), C.byref(n)) == 0, L.irmv_comm_last_error()
        assert n.value == len(blob) and ptr.value
        v = C.c_double(3.25)
        assert L.irmv_comm_allreduce_f64(h, C.byref(v), 1) == 0 and v.value == 3.25
        assert L.irmv_comm_allreduce_f64(h, C.byref(v), 0) == 0 and v.value == 3.25
        with YoloEngine(None, (1280, 1024), weights_device_ptr=ptr.value, weights_bytes=n.value, num_slots=1) as a, \
                YoloEngine(None, (1280, 1024), weights_blob=blob, num_slots=1) as b:
            for e in (a, b):
                e.get_src_image_buffer(0)[:] = frame0
            da, db = a.detect(0), b.detect(0)
            assert len(da) == len(db) > 0
            assert np.array_equal(a.read_head(0), b.read_head(0))
    finally:
        L.irmv_comm_destroy(h)


@pytest.mark.gpu
def test_cpp_runner_one_thread_and_engine_per_gpu(tmp_path, blob):
    exe = _runner()
    (tmp_path / "model.irmw").write_bytes(blob)
    out = subprocess.run([exe, "--weights", str(tmp_path / "model.irmw"), "--gpus", "1", "--slots", "16", "--steps", "4", "--group", "8"],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    r = json.loads(out.stdout.strip().splitlines()[-1])
    assert r["n_gpus"] == 1 and r["weights_bytes"] == len(blob)
    assert r["fps_hbm_resident"] > 1000 and r["fps_host_inclusive"] > 500
    assert r["detections_last_step"][0] > 0
