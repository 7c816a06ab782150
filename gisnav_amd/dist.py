"""One process per GPU; frame<->tile pairs are independent, so the path shards with no data-path
collective.  The process group (RCCL over xGMI on the GPU box, gloo in the CPU
tests) is used for exactly three things: one weight broadcast at start-up, barriers around the timed
region, and a gather of fixed-size per-pair result records.

The reference itself is single-process (one pose per ROS message,
ros/gisnav/gisnav/core/pose_node.py:178-184,497); batching and sharding are this build's additions.
"""
from __future__ import annotations

import os
import socket
import time
from typing import Dict, Tuple

import numpy as np
import torch
import torch.distributed as dist

RECORD_F64 = 16  # pair_index, ok, n_match, n_inliers, R (9), t (3)


def env_world() -> Tuple[int, int, int]:
    """(rank, local_rank, world_size) from the torchrun environment (1-process defaults)."""
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def init(backend: str) -> Tuple[int, int, int]:
    rank, local_rank, world = env_world()
    if backend == "nccl" and torch.cuda.is_available():
        torch.cuda.set_device(local_rank % max(torch.cuda.device_count(), 1))   # RCCL binds its communicator to the current device
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def shard_range(total_pairs: int, rank: int, world: int) -> range:
    """Contiguous shard: pair p -> rank p // ceil(total / world) (the last rank may be short)."""
    per = -(-total_pairs // world)
    return range(min(rank * per, total_pairs), min((rank + 1) * per, total_pairs))


def _comm_device(device: torch.device) -> torch.device:
    """gloo moves host tensors; nccl (RCCL) moves device tensors."""
    return torch.device("cpu") if dist.get_backend() == "gloo" else device


def _flatten(sd: Dict[str, np.ndarray]):
    keys = sorted(sd)
    return keys, np.concatenate([np.asarray(sd[k], np.float32).reshape(-1) for k in keys])


def broadcast_state_dict(sd: Dict[str, np.ndarray], device: torch.device, src: int = 0) -> Dict[str, np.ndarray]:
    """Rank `src` holds the weights (47.5 MB f32); every other rank receives them in ONE broadcast.
    All ranks must pass a state dict with the same keys/shapes (values are overwritten off-src)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return sd
    keys, flat = _flatten(sd)
    buf = torch.from_numpy(flat).to(_comm_device(device))
    dist.broadcast(buf, src=src)
    flat = buf.cpu().numpy()
    out, off = {}, 0
    for k in keys:
        n = int(np.prod(np.shape(sd[k]))) if np.ndim(sd[k]) else 1
        out[k] = flat[off: off + n].reshape(np.shape(sd[k])).copy()
        off += n
    return out


def barrier() -> None:
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        if dist.get_backend() == "nccl":
            dist.barrier(device_ids=[torch.cuda.current_device()])
        else:
            dist.barrier()


def max_over_ranks(value: float, device: torch.device) -> float:
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=_comm_device(device))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value: float, device: torch.device) -> float:
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=_comm_device(device))
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def pack_records(first_pair: int, out: dict) -> torch.Tensor:
    """[B, 16] f64 result records from a PoseEngine.estimate output (stays on the device)."""
    B = out["R"].shape[0]
    rec = torch.zeros((B, RECORD_F64), dtype=torch.float64, device=out["R"].device)
    rec[:, 0] = torch.arange(first_pair, first_pair + B, dtype=torch.float64, device=rec.device)
    rec[:, 1] = out["ok"].to(torch.float64)
    rec[:, 2] = out["n_match"].to(torch.float64)
    rec[:, 3] = out["n_inliers"].to(torch.float64)
    rec[:, 4:13] = out["R"].reshape(B, 9)
    rec[:, 13:16] = out["t"].reshape(B, 3)
    return rec


def gather_records(rec: torch.Tensor) -> torch.Tensor:
    """all_gather of equally-sized per-rank record blocks -> [world * B, 16] on every rank."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return rec
    src = rec.contiguous().to(_comm_device(rec.device))
    parts = [torch.empty_like(src) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, src)
    return torch.cat(parts, 0).to(rec.device)


def free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return int(sk.getsockname()[1])


def device_identity(index: int) -> str:
    """A string that differs between physical GPUs (uuid when torch exposes it, else PCI location): N ranks must report N of them."""
    p = torch.cuda.get_device_properties(index)
    for attr in ("uuid", "pci_bus_id"):
        v = getattr(p, attr, None)
        if v is not None:
            return f"{attr}:{v}|dev{getattr(p, 'pci_device_id', '')}|dom{getattr(p, 'pci_domain_id', '')}"
    return f"index:{index}"


def gather_strings(text: str) -> list:
    """all_gather of one short string per rank (rank order); [text] at world 1."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [text]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, text)
    return out


def rccl_selfcheck(device_index: int = 0, weight_mb: float = 47.5) -> dict:
    """The collectives the N > 1 path uses, on a WORLD-1 `nccl` (= RCCL) group bound to cuda:device_index: one broadcast of a
    weight-sized f32 buffer, barrier, all_reduce MAX / SUM, all_gather of [32, 16] f64 result records.  A 1-GPU lease can run it, so
    API / device-placement / IPC-mode errors of the RCCL path show up without an 8-GPU node.  Creates and destroys its own process
    group; raises if one is already initialised."""
    if dist.is_initialized():
        raise RuntimeError("rccl_selfcheck creates its own process group; call it before dist.init()")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(device_index)
    dev = torch.device("cuda", device_index)
    t0 = time.perf_counter()
    dist.init_process_group(backend="nccl", init_method=f"tcp://127.0.0.1:{free_port()}", rank=0, world_size=1)
    try:
        init_ms = (time.perf_counter() - t0) * 1e3
        n = int(weight_mb * 1e6 / 4)
        buf = torch.arange(n, dtype=torch.float32, device=dev)
        dist.broadcast(buf, src=0)                       # first call builds the communicator
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        dist.broadcast(buf, src=0)
        torch.cuda.synchronize()
        bcast_ms = (time.perf_counter() - t1) * 1e3
        dist.barrier(device_ids=[device_index])
        mx = torch.tensor([3.5], dtype=torch.float64, device=dev); dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = torch.tensor([32.0], dtype=torch.float64, device=dev); dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        rec = torch.arange(32 * RECORD_F64, dtype=torch.float64, device=dev).reshape(32, RECORD_F64)
        parts = [torch.empty_like(rec)]
        dist.all_gather(parts, rec)
        torch.cuda.synchronize()
        ok = bool(float(mx.item()) == 3.5 and float(sm.item()) == 32.0 and torch.equal(parts[0], rec) and float(buf[-1].item()) == float(n - 1))
        return {"ok": ok, "backend": dist.get_backend(), "world": dist.get_world_size(), "device": device_identity(device_index),
                "init_ms": round(init_ms, 1), "broadcast_mb": round(n * 4 / 1e6, 1), "broadcast_ms": round(bcast_ms, 3)}
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    import json
    import sys
    if "--selfcheck" in sys.argv:
        res = rccl_selfcheck(0)
        print(json.dumps(res), flush=True)
        sys.exit(0 if res["ok"] else 1)
