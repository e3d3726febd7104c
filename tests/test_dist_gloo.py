"""N > 1 path on CPU: two gloo ranks share the weight blob by one broadcast, own
disjoint frames, and agree on the max-over-ranks timing the bench reports."""
import hashlib
import os
import socket
import sys

import pytest
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    from irmv_detection_amd import dist as D
    from irmv_detection_amd import weights
    r, lr, w = D.init("gloo")
    dev = torch.device("cpu")
    blob = weights.synthetic_blob(0) if r == 0 else None          # only rank 0 builds / reads the blob
    t = D.broadcast_blob(blob, dev)
    digest = hashlib.sha256(t.numpy().tobytes()).hexdigest()
    frames = D.shard_frames(37, r, w)
    D.barrier()
    tmax = D.max_over_ranks(1.0 + r, dev)
    tot = D.sum_over_ranks(len(frames), dev)
    q.put((r, digest, frames, tmax, tot))
    torch.distributed.destroy_process_group()


def test_two_rank_broadcast_and_sharding(blob):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=180) for _ in ps)
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    want = hashlib.sha256(blob).hexdigest()
    assert res[0][1] == res[1][1] == want                          # replica is byte-identical
    f0, f1 = res[0][2], res[1][2]
    assert sorted(f0 + f1) == list(range(37)) and not set(f0) & set(f1)
    assert res[0][3] == res[1][3] == 2.0                           # MAX over ranks
    assert res[0][4] == res[1][4] == 37.0


def test_shard_frames_properties():
    from irmv_detection_amd.dist import shard_frames
    for world in (1, 2, 4, 8):
        owned = [shard_frames(100, r, world) for r in range(world)]
        assert sorted(sum(owned, [])) == list(range(100))
        assert max(map(len, owned)) - min(map(len, owned)) <= 1
