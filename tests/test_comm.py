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
    assert [f for f in os.listdir(tmp_path) if f.startswith("chan.")] == []   # two closing rounds, every rank acknowledges, then rank 0 sweeps the launch's prefix: nothing stays


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


def test_close_does_not_wait_for_a_vanished_peer(tmp_path):
    """the job has succeeded when close() runs.  A peer that never reaches the closing rounds (descheduled, gone)
    costs rank 0 the acknowledgement window, not a CommError after 600 s, and nothing is pulled from under anybody."""
    fr = comm.FileReduce(0, 2, str(tmp_path / "chan"))
    t0 = time.time()
    fr.close(ack_timeout_s=0.3)
    assert time.time() - t0 < 5.0


def test_id_file_is_keyed_per_launch_generation(tmp_path, monkeypatch):
    """an elastic agent keeps its pid and port across a worker restart, so a crashed generation's id file -- only
    SECONDS old, well inside STALE_S -- would be read by a non-root rank of the next one before rank 0 replaces it.  The
    file name carries TORCHELASTIC_RUN_ID and TORCHELASTIC_RESTART_COUNT: the next generation looks somewhere else."""
    monkeypatch.setenv("MASTER_PORT", "29517")
    monkeypatch.setenv("TORCHELASTIC_RUN_ID", "job/7")
    monkeypatch.setenv("TORCHELASTIC_RESTART_COUNT", "0")
    gen0 = comm.id_file()
    with open(gen0, "wb") as f:                                   # what the crashed generation left: seconds old
        f.write(bytes([7]) * comm.ID_BYTES + b"0" * comm.NONCE_BYTES)
    try:
        monkeypatch.setenv("TORCHELASTIC_RESTART_COUNT", "1")
        gen1 = comm.id_file()
        assert gen1 != gen0 and os.path.dirname(gen1) == os.path.dirname(gen0)
        with pytest.raises(comm.CommError):                       # nothing there for the new generation until ITS rank 0 writes
            comm.exchange_id(1, lambda: b"", gen1, timeout_s=0.3)
        monkeypatch.delenv("TORCHELASTIC_RUN_ID"); monkeypatch.delenv("TORCHELASTIC_RESTART_COUNT")
        assert comm.id_file() not in (gen0, gen1)                 # a plain launcher: its own key again
    finally:
        os.remove(gen0)


FAKE_COMM_C = r"""
/* stand-in for libirmv_comm.so (tests only): rank 1's ncclCommInitRank FAILS, rank 0's blocks as RCCL does while a peer is missing */
#include <stdint.h>
#include <string.h>
#include <unistd.h>
const char *irmv_comm_last_error(void) { return "fake: ncclCommInitRank failed on this rank"; }
int irmv_comm_init_all(int n, const int *d, void **o) { (void)n; (void)d; (void)o; return -1; }
int irmv_comm_unique_id(uint8_t *id) { memset(id, 0x5a, 128); return 0; }
int irmv_comm_init_rank(const uint8_t *id, int nranks, int rank, int device, void **out)
{ (void)id; (void)nranks; (void)device; (void)out; if (rank == 0) { sleep(600); } return -2; }
int irmv_comm_nranks(const void *c) { (void)c; return 0; }
int irmv_comm_local_ranks(const void *c) { (void)c; return 0; }
int irmv_comm_broadcast_blob(void *c, const void *b, uint64_t n, int r, void **p, uint64_t *o) { (void)c; (void)b; (void)n; (void)r; (void)p; (void)o; return -1; }
int irmv_comm_allreduce_f64(void *c, double *v, int op) { (void)c; (void)v; (void)op; return -1; }
void irmv_comm_destroy(void *c) { (void)c; }
"""

RANK_SCRIPT = """
import os, sys, time
sys.path.insert(0, {root!r})
from irmv_detection_amd import capi, comm
if os.environ.get("FAKE_DEVICE_COUNT"):
    capi.device_count = lambda: int(os.environ["FAKE_DEVICE_COUNT"])
t0 = time.time()
c = comm.Comm()
print("RESULT", c.rank, int(c.native), round(time.time() - t0, 2), flush=True)
s = c.sum_over_ranks(float(c.rank + 1))
c.barrier()
c.close()
print("SUM", s, flush=True)
"""


def _spawn_ranks(tmp_path, env_extra, world=2):
    script = tmp_path / "rank.py"
    script.write_text(RANK_SCRIPT.format(root=ROOT))
    ps = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_PORT="29533", TMPDIR=str(tmp_path), **env_extra)
        env.pop("IRMV_FORCE_DEVICE", None)
        ps.append(subprocess.Popen([os.sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    return ps


def test_a_rank_failing_inside_init_rank_exits_nonzero_within_the_window(tmp_path):
    """Stage 2 of the agreement (comm.py): a rank whose ncclCommInitRank FAILS reports it and then waits only
    IRMV_COMM_STAGE2_WAIT_S for its peers -- they may be blocked inside RCCL for good -- before it raises, so that the launcher
    can end the job.  The library is a stand-in (IRMV_COMM_LIB_PATH) whose rank 0 blocks the way RCCL does."""
    src = tmp_path / "fake_comm.c"
    src.write_text(FAKE_COMM_C)
    lib = tmp_path / "libfake_comm.so"
    subprocess.check_call(["gcc", "-shared", "-fPIC", "-O1", str(src), "-o", str(lib)])
    ps = _spawn_ranks(tmp_path, dict(IRMV_COMM_LIB_PATH=str(lib), FAKE_DEVICE_COUNT="8", IRMV_COMM_STAGE2_WAIT_S="2"))
    t0 = time.time()
    try:
        out1, err1 = ps[1].communicate(timeout=60)
        assert ps[1].returncode != 0, out1 + err1                # the failing rank: CommError, non-zero exit ...
        assert time.time() - t0 < 30                              # ... inside the window (2 s here, 60 s by default), not after 600 s
        assert "never reached reduction" in err1
        assert ps[0].poll() is None                               # its peer is still inside "RCCL": the launcher's to end
    finally:
        for p in ps:
            if p.poll() is None:
                p.kill()
            p.wait(10)


@pytest.mark.gpu
def test_one_rank_without_a_device_takes_both_ranks_to_the_fallback(tmp_path):
    """Stage 1 of the agreement, the asymmetric case it was written for, staged on a one-GPU box: rank 1's device (LOCAL_RANK 1)
    does not exist.  BOTH ranks must come out of Comm() within seconds with native == False -- rank 0 never enters
    ncclCommInitRank to wait for a peer that cannot join --, their file-channel reductions must work, and both exit 0."""
    from irmv_detection_amd import capi
    if capi.device_count() != 1:
        pytest.skip("needs exactly one visible GPU (rank 1 -> device 1 must not exist)")
    ps = _spawn_ranks(tmp_path, {})
    outs = [p.communicate(timeout=120) for p in ps]
    for r, (p, (out, err)) in enumerate(zip(ps, outs)):
        assert p.returncode == 0, (r, out, err)
        res = [l.split() for l in out.splitlines() if l.startswith("RESULT")][0]
        assert int(res[1]) == r and int(res[2]) == 0, out         # native == False on BOTH ranks ("rccl": false in the bench line)
        assert float(res[3]) < 30.0, out                          # seconds, not RCCL's rendezvous timeout
        assert "SUM 3.0" in out                                   # 1 + 2 through the file channel
        assert "RCCL communicator not available on every rank" in err
    assert "no HIP device 1" in outs[1][1]


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
