"""Multi-GPU consolidation: one process per GPU, candidates (single-node) or prefixes (multi-node) dealt round-robin,
no collective on the data path, then ONE all-gather of the fixed-size kg_candidate_record rows (RCCL over xGMI when the
process group is "nccl"; gloo in the CPU tests) and the reference's selection replayed on the gathered records
(kg_consolidate_select).  Solve itself does not shard."""
import numpy as np
import torch
import torch.distributed as dist

from . import abi


def gather_records(local, n_units_total, device):
    """all_gather of variable-length record arrays (padded to the largest shard)."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    if world == 1:
        return local
    rec_size = abi.CANDIDATE_RECORD.itemsize
    cap = (n_units_total + world - 1) // world
    buf = np.zeros(cap * rec_size + 8, dtype=np.uint8)
    buf[:8] = np.frombuffer(np.int64(len(local)).tobytes(), dtype=np.uint8)
    raw = local.view(np.uint8).reshape(-1)
    buf[8:8 + raw.shape[0]] = raw
    t = torch.from_numpy(buf).to(device)
    out = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(out, t)
    parts = []
    for o in out:
        a = o.cpu().numpy()
        k = int(np.frombuffer(a[:8].tobytes(), dtype=np.int64)[0])
        parts.append(np.frombuffer(a[8:8 + k * rec_size].tobytes(), dtype=abi.CANDIDATE_RECORD))
    return np.concatenate(parts)


def consolidate(L, hcat, hcl, pending, nodepools, order, multi, spot_to_spot=False, device="cpu"):
    """-> (decision dict on every rank, records evaluated by this rank, device_ms of this rank)"""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    cd = abi.build_candidates(order, spot_to_spot=spot_to_spot, shard=rank, n_shards=world)
    local, dms = L.consolidate_records([hcat], hcl, pending, nodepools, cd, multi)
    n_units = (min(len(order), 101) - 1 if multi else len(order)) if len(order) else 0
    allrec = gather_records(local, max(n_units, 1), device)
    return L.consolidate_select(hcat, allrec, len(order), multi), len(local), dms
