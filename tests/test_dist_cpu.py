"""CPU tests of the N>1 path: two gloo ranks, sequences sharded round-robin, model bytes
broadcast from rank 0, ids all-gathered -- with a deterministic stand-in for the per-rank decode (the real
decode needs a GPU; the collectives and the partition logic are what is covered here)."""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np

from conftest import ROOT
from nano_amd import dist as nd


def test_shard_indices_partition():
    for n_seq in (1, 7, 64):
        for world in (1, 2, 4, 8):
            parts = [nd.shard_indices(n_seq, r, world) for r in range(world)]
            flat = sorted(i for p in parts for i in p)
            assert flat == list(range(n_seq))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


WORKER = textwrap.dedent("""
    import os, sys, numpy as np
    sys.path.insert(0, %(root)r)
    from nano_amd import dist as nd, modelfile as mf
    dist = nd.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    path = %(path)r
    buf = nd.broadcast_file_bytes(path if rank == 0 else None, src=0)
    raw = buf.numpy()
    spec, off = nd.split_model_bytes(bytes(raw[:260]))
    ref = np.fromfile(path, dtype=np.uint8)
    assert raw.size == ref.size and np.array_equal(raw, ref)
    assert spec == mf.read_header(path) and off == mf.param_layout(spec).params_offset
    n_seq, T = 7, 5
    owned = nd.shard_indices(n_seq, rank, world)
    local = np.array([[1000 * i + t for t in range(T)] for i in owned], np.int64).reshape(len(owned), T)
    allids = nd.gather_ids(local, owned, n_seq)
    want = np.array([[1000 * i + t for t in range(T)] for i in range(n_seq)], np.int64)
    assert np.array_equal(allids, want), allids
    dist.barrier()
    print("RANK_OK", rank)
""")


def test_two_rank_gloo_broadcast_and_gather(tmp_path):
    from nano_amd import modelfile as mf
    path = str(tmp_path / "tiny.bin")
    mf.write_model(path, mf.preset("tiny-nano", "q80", group_size=32), seed=1)
    script = str(tmp_path / "worker.py")
    open(script, "w").write(WORKER % {"root": ROOT, "path": path})
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, script], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=240)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"RANK_OK {r}" in o, o[-2000:]
