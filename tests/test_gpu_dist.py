"""The N > 1 path of bench.py on the ONE GPU of the test box (round-4 review, missing #3): two real ranks started by
bench.py's own launcher (torch.distributed.run on 127.0.0.1), sequence i on rank i mod 2, the model file's bytes broadcast from rank 0,
every rank decoding its sequences on the device, the per-window times all-reduced (MAX) and the timed ids all-gathered -- over gloo,
because RCCL refuses two ranks on one device (NANO_BENCH_BACKEND=gloo; the collectives' payloads live on the host, everything else is
the nccl path's code).  The gathered ids must be those of ONE rank decoding the same six prompts."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def run_bench(extra, env_extra, timeout=900):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--total-seqs", "6", "--steps", "8", "--warmup", "2",
                        "--no-cpu-baseline", "--no-kernel-table"] + extra, env=env, capture_output=True, text=True, timeout=timeout)
    # (torch.distributed.run --tee prefixes every line of a rank with "[default0]:")
    lines = [ln[ln.index('{"metric"'):] for ln in r.stdout.strip().splitlines() if '{"metric"' in ln]
    assert r.returncode == 0 and lines, (r.returncode, r.stderr[-3000:])
    return json.loads(lines[-1])


def test_two_ranks_share_the_gpu_over_gloo_and_agree_with_one_rank():
    one = run_bench(["--gpus", "1"], {"NANO_BENCH_NO_TRAFFIC": "1"})
    two = run_bench(["--gpus", "2"], {"NANO_BENCH_BACKEND": "gloo", "NANO_BENCH_NO_TRAFFIC": "1"})
    assert two["n_gpus"] == 2 and two["config"]["world_size_seen"] == 2
    assert two["config"]["collectives"] == "gloo" and two["config"]["windows_all_reduced"] is True
    assert two["config"]["sequences"] == 6 and two["config"]["sequences_per_gpu"] == 3 and two["scaling"] == "strong"
    assert one["config"]["world_size_seen"] == 1 and one["config"]["sequences_per_gpu"] == 6
    # six prompts, eight timed steps each: the all-gathered ids of the sharded run == the ids of the unsharded run
    assert two["config"]["timed_ids_crc32"] == one["config"]["timed_ids_crc32"], (one["config"], two["config"])
    assert two["value"] > 0 and two["steps"] == 8


def test_one_rank_through_rccl_is_the_unsharded_run():
    """The RCCL (backend "nccl") path itself with the one rank a 1-GPU box allows: started the way the driver starts ranks
    (python -m torch.distributed.run --nproc-per-node 1 ... bench.py --gpus 1), NANO_BENCH_FORCE_DIST=1 makes the single rank take the
    distributed code -- init_process_group("nccl"), the device-side broadcast of the model file's bytes, the barriers, the all-reduce of the
    window times, the all-gather of the ids.  Same ids as the plain run."""
    one = run_bench(["--gpus", "1"], {"NANO_BENCH_NO_TRAFFIC": "1"})
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update({"NANO_BENCH_NO_TRAFFIC": "1", "NANO_BENCH_FORCE_DIST": "1"})
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", "29577",
                        os.path.join(ROOT, "bench.py"), "--gpus", "1", "--total-seqs", "6", "--steps", "8", "--warmup", "2", "--no-cpu-baseline", "--no-kernel-table"],
                       env=env, capture_output=True, text=True, timeout=900)
    lines = [ln[ln.index('{"metric"'):] for ln in r.stdout.strip().splitlines() if '{"metric"' in ln]
    assert r.returncode == 0 and lines, (r.returncode, r.stderr[-3000:])
    d = json.loads(lines[-1])
    assert d["config"]["collectives"] == "nccl" and d["config"]["windows_all_reduced"] is True and d["n_gpus"] == 1
    assert d["config"]["timed_ids_crc32"] == one["config"]["timed_ids_crc32"]
