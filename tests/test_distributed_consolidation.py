"""world_size-2 and -3 consolidation on CPU: gloo all-gather of the shards' records + selection == the oracle's sequential decision.
Each rank drives the TEST-ONLY emulation build (the GPU ranks run the same host code over libkarpgpu.so and RCCL)."""
import os
import sys

import pytest
import torch.multiprocessing as mp

from tests.helpers import EMU_PATH

pytestmark = pytest.mark.skipif(not os.path.exists(EMU_PATH), reason="emulation build missing: run __graft_entry__.build()")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q, n_nodes=90):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from karpenter_amd import abi, distributed, lib, workloads as wl
    from oracle import loader
    from tests.cons_common import cluster
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        L = lib.Library(EMU_PATH)
        types, nodes, bound, pending = cluster("dense", n_nodes)
        order = wl.disruption_order(nodes, bound)
        cat, cl = abi.build_catalog(types), abi.build_cluster(nodes, bound)
        pend, nps = abi.build_pods(pending), abi.build_nodepools([wl.default_nodepool()], wl.WELL_KNOWN_LABELS)
        hcat, hcl = L.catalog_upload(cat), L.cluster_upload(cl)
        out = {}
        for multi in (False, True):
            dec, info = distributed.consolidate(L, hcat, hcl, pend, nps, order, multi, spot_to_spot=True, want_info=True)
            n_local = info["n_local"]
            assert sum(info["decisions"]) == (len(order) - 1 if multi else len(order))  # the gathered records cover every unit
            st, ref = loader.consolidate_decide([cat], cl, pend, nps, abi.build_candidates(order, spot_to_spot=True), multi)
            assert st == 0
            keys = ("decision", "n_candidates", "first_candidate", "nodepool", "types", "spot_only", "candidate_price")
            out[multi] = (all(dec[k] == ref[k] for k in keys), n_local)
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


def test_two_ranks_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    n_single = sum(out[False][1] for _, out in res)
    n_multi = sum(out[True][1] for _, out in res)
    assert n_single == 90 and n_multi == 89   # every unit evaluated exactly once across the shards
    for _, out in res:
        assert out[False][0] and out[True][0]


def test_three_ranks_uneven_units():
    """world_size 3 over 91 candidates (91 and 90 units: neither divides by three -- the shards differ in length and the
    padded gather carries the counts)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31700 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 3, port, q, 91)) for r in range(3)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(out[False][1] for _, out in res) == [30, 30, 31] and sorted(out[True][1] for _, out in res) == [30, 30, 30]
    for _, out in res:
        assert out[False][0] and out[True][0]
