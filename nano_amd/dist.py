"""Batched-prompt data parallelism over the GPUs of one node.

Independent sequences partition trivially: sequence i -> rank i mod G; every rank holds a full weight
replica and decodes its sequences as one batch, so there is NO collective on the data path.  The only
exchanges are
  * one broadcast of the model file's bytes from rank 0 at load (RCCL over xGMI when the process group
    is "nccl": the parameter blob is broadcast straight into device memory and handed to the backend
    as a device pointer; "gloo" on CPU for tests), and
  * one all-gather of the generated token ids at the end.
torch.distributed is plumbing here (rendezvous + the two collectives); the decode itself never touches it.
"""
from __future__ import annotations

import os
from typing import List, Optional, Sequence, Tuple

import numpy as np


def env_world() -> Tuple[int, int, int]:
    """(rank, world_size, local_rank) from the torchrun environment (1-process defaults)."""
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def shard_indices(n_seq: int, rank: int, world: int) -> List[int]:
    """Sequences owned by `rank`: i mod world == rank (round-robin keeps ragged tails balanced)."""
    return [i for i in range(n_seq) if i % world == rank]


def init_process_group(backend: Optional[str] = None):
    """Initialise torch.distributed from the environment.  backend: "nccl" (= RCCL on ROCm) or "gloo"."""
    import torch
    import torch.distributed as dist
    rank, world, local = env_world()
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(local)
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return dist


def broadcast_file_bytes(path: Optional[str], *, src: int = 0, device: Optional[str] = None):
    """Broadcast a model file from `src` to every rank.  Returns a uint8 torch tensor (on `device` if
    given, i.e. the bytes travel GPU-to-GPU over xGMI) holding the whole file on every rank."""
    import torch
    import torch.distributed as dist
    rank = dist.get_rank()
    dev = torch.device(device) if device else torch.device("cpu")
    n = torch.zeros(1, dtype=torch.int64, device=dev)
    if rank == src:
        raw = np.fromfile(path, dtype=np.uint8)
        n[0] = raw.size
    dist.broadcast(n, src=src)
    buf = torch.empty(int(n.item()), dtype=torch.uint8, device=dev)
    if rank == src:
        buf.copy_(torch.from_numpy(raw))
    dist.broadcast(buf, src=src)
    return buf


def split_model_bytes(head: bytes) -> Tuple["object", int]:
    """(ModelSpec, byte offset of the parameter blob) from the first >=260 bytes of a model file."""
    from . import modelfile as mf
    spec = mf.read_header(head[:256])
    tok_bytes = int(np.frombuffer(head[256:260], "<u4")[0])
    return spec, 256 + tok_bytes


def gather_ids(local_ids: np.ndarray, owned: Sequence[int], n_seq: int) -> np.ndarray:
    """All-gather per-rank id matrices [len(owned), T] into the global [n_seq, T] (same on every rank)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size()
    T = local_ids.shape[1] if local_ids.ndim == 2 and local_ids.size else 0
    tmax = torch.tensor([T], dtype=torch.int64)
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    tmax = tmax.to(dev)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    T = int(tmax.item())
    per = (n_seq + world - 1) // world
    mine = torch.full((per, T), -1, dtype=torch.int64, device=dev)
    if local_ids.size:
        mine[:local_ids.shape[0], :local_ids.shape[1]] = torch.from_numpy(local_ids.astype(np.int64)).to(dev)
    parts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine)
    out = np.full((n_seq, T), -1, np.int64)
    for r in range(world):
        idx = shard_indices(n_seq, r, world)
        out[idx] = parts[r].cpu().numpy()[:len(idx)]
    return out
