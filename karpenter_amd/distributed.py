"""Multi-GPU consolidation: one process per GPU, candidates (single-node) or prefixes (multi-node) dealt round-robin,
no collective on the data path, then ONE all-gather of the fixed-size kg_candidate_record rows (RCCL over xGMI when the
process group is "nccl"; gloo in the CPU tests) and the reference's selection replayed on the gathered records
(kg_consolidate_select).  Solve itself does not shard.

The gather moves at most a few hundred KB, so what it costs is latency, not bandwidth: the records go through buffers that
are allocated once per (size, world, device) -- a pinned host staging pair and two device tensors on CUDA -- as one
all_gather_into_tensor with one stream synchronisation, instead of fresh tensors, a list all_gather and a .cpu() per rank."""
import time

import numpy as np
import torch
import torch.distributed as dist

from . import abi

_BUFFERS = {}


def _buffers(nbytes, world, device):
    key = (nbytes, world, str(device))
    b = _BUFFERS.get(key)
    if b is None:
        cuda = str(device).startswith("cuda")
        host_in = torch.zeros(nbytes, dtype=torch.uint8)
        host_out = torch.zeros(nbytes * world, dtype=torch.uint8)
        if cuda:
            host_in, host_out = host_in.pin_memory(), host_out.pin_memory()
            b = (host_in, host_out, torch.zeros(nbytes, dtype=torch.uint8, device=device), torch.zeros(nbytes * world, dtype=torch.uint8, device=device))
        else:
            b = (host_in, host_out, None, None)
        _BUFFERS[key] = b
    return b


def gather_records(local, n_units_total, device):
    """all-gather of the ranks' record arrays (variable length: padded to the largest possible shard, the count in front)."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    if world == 1:
        return local
    rec_size = abi.CANDIDATE_RECORD.itemsize
    cap = (n_units_total + world - 1) // world
    nbytes = (cap * rec_size + 8 + 15) & ~15
    host_in, host_out, dev_in, dev_out = _buffers(nbytes, world, device)
    hin = host_in.numpy()
    hin[:8] = np.frombuffer(np.int64(len(local)).tobytes(), dtype=np.uint8)
    raw = local.view(np.uint8).reshape(-1)
    hin[8:8 + raw.shape[0]] = raw
    if dev_in is not None:
        dev_in.copy_(host_in, non_blocking=True)
        dist.all_gather_into_tensor(dev_out, dev_in)
        host_out.copy_(dev_out, non_blocking=True)
        torch.cuda.current_stream().synchronize()
    else:
        parts_t = [torch.empty_like(host_in) for _ in range(world)]
        dist.all_gather(parts_t, host_in)
        for r, t in enumerate(parts_t):
            host_out[r * nbytes:(r + 1) * nbytes] = t
    hout = host_out.numpy()
    parts = []
    for r in range(world):
        a = hout[r * nbytes:(r + 1) * nbytes]
        k = int(np.frombuffer(a[:8].tobytes(), dtype=np.int64)[0])
        parts.append(np.frombuffer(a[8:8 + k * rec_size].tobytes(), dtype=abi.CANDIDATE_RECORD))
    return np.concatenate(parts)


def consolidate(L, hcat, hcl, pending, nodepools, order, multi, spot_to_spot=False, device="cpu", want_info=False, cands=None, out=None, want_hist=True):
    """-> (decision dict on every rank, records evaluated by this rank, device_ms of this rank); with want_info:
    (decision dict, info) where info holds this rank's wall time per part (records_ms = the kg_consolidate_* call,
    gather_ms, select_ms), device_ms, n_local and the verdict histogram of the gathered records [no-op, delete, replace]."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    # (`cands`: this rank's candidate descriptor, `out`: its record buffer -- built once by a caller that polls, else per call)
    cd = cands if cands is not None else abi.build_candidates(order, spot_to_spot=spot_to_spot, shard=rank, n_shards=world)
    t0 = time.perf_counter()
    local, dms = L.consolidate_records([hcat], hcl, pending, nodepools, cd, multi, out=out)
    t1 = time.perf_counter()
    n_units = (min(len(order), 101) - 1 if multi else len(order)) if len(order) else 0
    allrec = gather_records(local, max(n_units, 1), device)
    t2 = time.perf_counter()
    dec = L.consolidate_select(hcat, allrec, len(order), multi)
    t3 = time.perf_counter()
    if not want_info:
        return dec, len(local), dms
    hist = ([int((allrec["decision"] == k).sum()) for k in (0, 1, 2)] if len(allrec) else [0, 0, 0]) if want_hist else None
    return dec, {"records_ms": (t1 - t0) * 1e3, "gather_ms": (t2 - t1) * 1e3, "select_ms": (t3 - t2) * 1e3, "device_ms": dms, "n_local": len(local),
                 "decisions": hist}
