#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on MI355X: pods/sec in Solve (10k mixed pods x 776 instance types, config C2).

    python bench.py [--gpus N] [--steps K] [--warmup W]

A "step" is one kg_solve over one batch of C2 pods: the catalog is already resident in HBM (built by the
library itself, kg_catalog_build_ec2, before the timed region); the step covers requirement compile, the device kernels
and the result download with the price-ordered truncation, i.e. everything between handing over the pod batch and having
the NodeClaims.  Solve does not shard (FFD is one chain of dependent placements): with N ranks each rank runs an
independent replica of the batch ("replicas only", weak scaling, no data-path collective).

The same JSON line carries, each driver-timed the same way:
  * `config3`, `config5` (N = 1 only): BASELINE configs 3 and 5 at full size -- 50 000 pods with zone / hostname spread and
    anti-affinity; 100 000 pods x 4 weighted NodePools with limits x 2 000 existing nodes x 6 zones;
  * `consolidation`: config C4 (5 000-node cluster, single-node = one SimulateScheduling per candidate, multi-node = one
    per prefix of the first 100), units dealt round-robin over the ranks, one all-gather of the fixed-size records (RCCL),
    the reference's selection replayed.  Per leg a WARM figure (`value`: the disruption controller polls an unchanged
    snapshot, its plan is resident) and a COLD one (`cold`: a new cluster snapshot every round -- upload, plan, round);
  * `consolidation.dense`: the same on a cluster at 90-100 % utilisation, whose records hold REPLACE and no-op verdicts
    (the C4 cluster of SURVEY.md 8(d), at 35 %, deletes every candidate).

Nothing on the GPU legs imports tests/ or oracle/: the workloads come from karpenter_amd.workloads / karpenter_amd.problem.
`cpu_baseline` (rank 0, N = 1) times the CPU oracle on the same inputs -- the checker as a reported baseline, never as
the thing measured.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "roofline.json")) as _f:
    ROOFLINE = json.load(_f)   # frozen row sizes + peaks
HBM_PEAK_GBS = float(ROOFLINE["peak"]["hbm_GBps"])  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E ~8 TB/s
ROWS_NOTE = "%d B/existing node, %d B/NodeClaim, %d B/instance type (roofline.json)" % (
    ROOFLINE["row_bytes"]["existing_node"], ROOFLINE["row_bytes"]["node_claim"], ROOFLINE["row_bytes"]["instance_type"])
PMC_FILE = os.path.join("profiles", "r03_pmc.json")


def pmc_traffic(kernel, default_workload):
    """HBM bytes per launch of `kernel` on the default workload, from the committed rocprofv3 PMC passes of this same
    command (FETCH_SIZE, WRITE_SIZE, each its own run; gfx950 correction of MI355X_MICROARCH.md applied; made by
    tools/gpu_round.sh + tools/pmc_summary.py) -- hardware counters cannot be read from inside the run, so this describes
    the builder's profiling run, not this one.  -> (bytes or None, source file or None)."""
    path = os.path.join(ROOT, PMC_FILE)
    if not default_workload or not os.path.exists(path):
        return None, None
    with open(path) as f:
        k = json.load(f).get("kernels", {})
    hit = [v for name, v in k.items() if name.split("<")[0].split("::")[-1] == kernel]
    return (max(v["hbm_bytes_per_launch_corrected"] for v in hit), PMC_FILE) if hit else (None, None)


def cpu_model():
    """Everything /proc/cpuinfo says about the model (some hosts report only a family name)."""
    want = ("model name", "cpu family", "model", "stepping", "cpu MHz", "cache size")
    got = {}
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                k, _, v = line.partition(":")
                k = k.strip()
                if k in want and k not in got:
                    got[k] = v.strip()
                if not line.strip():
                    break
    except OSError:
        pass
    return "%s (family %s model %s stepping %s, %s MHz, cache %s)" % tuple(got.get(k, "?") for k in want)


def measured_copy_gbs(torch):
    """Device-to-device stream copy on this box (the second roofline denominator): 1 GiB, read + write."""
    n = 1 << 28
    a = torch.empty(n, dtype=torch.float32, device="cuda")
    b = torch.empty_like(a)
    a.fill_(1.0)
    for _ in range(2):
        b.copy_(a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    del a, b
    return 2.0 * n * 4 / (ms * 1e-3) / 1e9


def solve_kernel_name(L, before):
    st = L.stats()
    if st.spec_solves > before[1]:
        return "k_solve_spec"
    return "k_solve_wave" if st.wave_solves > before[0] else "k_solve"


def solve_leg(L, hcat, prob, steps, warmup, barrier=None, want_alg=True):
    """`steps` timed kg_solve calls of one problem -> dict (value = pods/s of this process)."""
    from karpenter_amd import abi
    alg = 0.0
    if want_alg:  # rows the reference visits: one untimed solve with the visit counters on (the exact, slower scan)
        L.set_collect_stats(True)
        alg = prob.run_library(L, hcat).res.algorithmic_bytes
    L.set_collect_stats(False)
    out = abi.ResultBuffers(prob.n_pods)  # the caller's output buffers, allocated once as a controller would
    before = (L.stats().wave_solves, L.stats().spec_solves)
    for _ in range(warmup):
        prob.run_library(L, hcat, out=out)
    if barrier:
        barrier()
    t0 = time.perf_counter()
    ms = {"host_compile": [], "prepass_kernels": [], "k_solve": [], "host_decode_truncate": []}
    rb = None
    for _ in range(steps):
        rb = prob.run_library(L, hcat, out=out)
        ms["host_compile"].append(rb.res.host_compile_ms)
        ms["prepass_kernels"].append(rb.res.prepass_ms)
        ms["k_solve"].append(rb.res.solve_ms)
        ms["host_decode_truncate"].append(rb.res.host_decode_ms)
    if barrier:
        barrier()
    dt = time.perf_counter() - t0
    k_ms = sum(ms["k_solve"]) / steps
    achieved = alg / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
    return {"dt": dt, "value": prob.n_pods * steps / dt, "unit": "pods/s", "steps": steps, "warmup": warmup, "ms_per_step": dt * 1e3 / steps,
            "n_claims": int(rb.res.n_claims), "breakdown_ms": {k: sum(v) / steps for k, v in ms.items()},
            "roofline": {"bound": "hbm", "kernel": solve_kernel_name(L, before), "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None, "algorithmic_bytes_per_launch": alg,
                         "note": "rows the reference algorithm visits up to its first feasible index (SURVEY.md 8d): " + ROWS_NOTE}}


def cons_leg(args, L, hcat, cl, order, multi, world, device, barrier, dist, torch, default_workload):
    """One consolidation leg (single- or multi-node) on one cluster descriptor: warm rounds on a resident snapshot, then
    cold rounds (a new snapshot each: kg_cluster_upload + plan build + round)."""
    import numpy as np
    from karpenter_amd import abi, distributed, workloads as wl
    pend, nps = abi.build_pods([]), abi.build_nodepools([wl.default_nodepool()], wl.WELL_KNOWN_LABELS)
    hcl = L.cluster_upload(cl)
    # the candidate descriptor of this rank's shard and its record buffer: built once, as a controller that polls keeps them
    rank = dist.get_rank() if world > 1 else 0
    cd = abi.build_candidates(order, shard=rank, n_shards=world)
    recs = np.zeros(max(len(order), 1), abi.CANDIDATE_RECORD)
    alg = 0.0
    if not args.no_alg_bytes:  # (one untimed round with the visit counters on)
        L.set_collect_stats(True)
        distributed.consolidate(L, hcat, hcl, pend, nps, order, multi, device=device)
        alg = L.stats().cons_algorithmic_bytes
        L.set_collect_stats(False)
        L.dll.kg_cluster_free(hcl)   # (its plan was built with the counters on: a fresh snapshot for the timed rounds)
        hcl = L.cluster_upload(cl)
    for _ in range(max(args.warmup, 1)):
        distributed.consolidate(L, hcat, hcl, pend, nps, order, multi, device=device)
    barrier()
    t0 = time.perf_counter()
    dms, kms, ims, tim = [], [], [], []
    dec = info = None
    for _ in range(args.steps):
        dec, info = distributed.consolidate(L, hcat, hcl, pend, nps, order, multi, device=device, want_info=True, cands=cd, out=recs, want_hist=False)
        st = L.stats()
        dms.append(info["device_ms"])
        kms.append(st.cons_simulate_ms)
        ims.append(st.cons_sim_init_ms)
        tim.append(info)
    barrier()
    dt = time.perf_counter() - t0
    info["decisions"] = distributed.consolidate(L, hcat, hcl, pend, nps, order, multi, device=device, want_info=True, cands=cd, out=recs)[1]["decisions"]  # (untimed: the verdict histogram)
    L.dll.kg_cluster_free(hcl)
    # cold rounds: what a controller pays when the cluster changed since its last poll
    cold_steps = 0 if args.no_cold else max(1, min(args.steps, 5))
    cdt, cup = 0.0, []
    if cold_steps:
        barrier()
        c0 = time.perf_counter()
        for _ in range(cold_steps):
            u0 = time.perf_counter()
            h2 = L.cluster_upload(cl)
            cup.append((time.perf_counter() - u0) * 1e3)
            distributed.consolidate(L, hcat, h2, pend, nps, order, multi, device=device, cands=cd, out=recs)
            L.dll.kg_cluster_free(h2)
        barrier()
        cdt = time.perf_counter() - c0
    if world > 1:
        t = torch.tensor([dt, cdt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt, cdt = float(t[0].item()), float(t[1].item())
    units = (min(len(order), 101) - 1) if multi else len(order)
    k_ms = sum(kms) / len(kms)
    achieved = alg / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
    traffic, src = pmc_traffic("k_simulate", default_workload and not multi and world == 1)
    hist = info["decisions"] if info else None
    out = {"value": units * args.steps / dt, "unit": "prefixes/s" if multi else "candidates/s", "units_per_round": units,
           "ms_per_round": dt * 1e3 / args.steps, "device_ms_per_round_rank0": sum(dms) / len(dms),
           "k_simulate_ms_rank0": k_ms, "k_sim_init_ms_rank0": sum(ims) / len(ims),
           "host_ms_per_round_rank0": sum(i["records_ms"] - i["device_ms"] for i in tim) / len(tim),
           "gather_ms_per_round_rank0": sum(i["gather_ms"] for i in tim) / len(tim), "select_ms_per_round_rank0": sum(i["select_ms"] for i in tim) / len(tim),
           "decision": {k: dec[k] for k in ("decision", "n_candidates", "first_candidate")},
           "record_verdicts": {"noop": hist[0], "delete": hist[1], "replace": hist[2]} if hist else None,
           "roofline": {"bound": "hbm", "kernel": "k_simulate", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                        "traffic": traffic, "traffic_source": src, "algorithmic_bytes_per_launch": alg,
                        "note": "this rank's simulations of one round: sum over simulations of the rows the reference visits, " + ROWS_NOTE +
                                " (SURVEY.md 8d) / HIP-event time of k_simulate"}}
    if achieved > HBM_PEAK_GBS:
        out["roofline"]["note"] += (" -- above the HBM peak: the count is the REFERENCE algorithm's row visits (a no-op candidate walks every "
                                    "existing node); the kernel answers most of those visits from the compact prefilter record of a node and reads "
                                    "the full row only where the prefilter passes, so this figure is work avoided, not bytes moved")
    if cold_steps:
        out["cold"] = {"value": units * cold_steps / cdt, "unit": out["unit"], "ms_per_round": cdt * 1e3 / cold_steps, "rounds": cold_steps,
                       "cluster_upload_ms_rank0": sum(cup) / len(cup),
                       "what": "a new cluster snapshot every round: kg_cluster_upload + consolidation plan + the round"}
    if world > 1:  # what every rank spent where (the scaling curve's explanation)
        mine = {"rank": dist.get_rank(), "k_simulate_ms": k_ms, "device_ms": sum(dms) / len(dms), "host_ms": out["host_ms_per_round_rank0"],
                "gather_ms": out["gather_ms_per_round_rank0"], "units": info["n_local"]}
        allr = [None] * world
        dist.all_gather_object(allr, mine)
        out["per_rank"] = allr
    return out


def bench_consolidation(args, L, hcat, types, zones, world, rank, barrier, dist, torch):
    """C4: consolidation on the synthetic clusters, units sharded over the ranks."""
    from karpenter_amd import abi, workloads as wl
    device = "cuda" if world > 1 else "cpu"
    nodes, bound = wl.existing_cluster(args.nodes, 11, types, zones)
    order = wl.disruption_order(nodes, bound)
    cl = abi.build_cluster(nodes, bound)
    out = {"workload": "C4: %d-node cluster (seed 11, ~12 pods/node, 35%% utilisation), %d bound pods, 0 pending" % (len(nodes), len(bound)),
           "scaling": "strong", "collective": "all_gather of %d-byte kg_candidate_record rows" % abi.CANDIDATE_RECORD.itemsize}
    for multi in [m for m in (False, True) if args.cons in ("both", "multi" if m else "single")]:
        out["multi_node" if multi else "single_node"] = cons_leg(args, L, hcat, cl, order, multi, world, device, barrier, dist, torch, args.nodes == 5000)
    if not args.no_dense:
        dn, db = wl.dense_cluster(args.nodes, 17, types, zones, lo=0.9, hi=1.0)
        dorder = wl.disruption_order(dn, db)
        dcl = abi.build_cluster(dn, db)
        dense = {"workload": "%d-node cluster (seed 17) filled to 90-100%% of each node's cpu, %d bound pods, 0 pending" % (len(dn), len(db))}
        for multi in [m for m in (False, True) if args.cons in ("both", "multi" if m else "single")]:
            dense["multi_node" if multi else "single_node"] = cons_leg(args, L, hcat, dcl, dorder, multi, world, device, barrier, dist, torch, False)
        out["dense"] = dense
    if rank == 0 and not args.no_cpu_baseline:
        # the CPU oracle on the first candidates of the same cluster (test infrastructure, used here as the reported baseline only)
        from oracle import loader
        from tests.helpers import catalog776
        k = min(30, len(order))
        cat = abi.build_catalog(catalog776(len(zones))[1])
        pend, nps = abi.build_pods([]), abi.build_nodepools([wl.default_nodepool()], wl.WELL_KNOWN_LABELS)
        c0 = time.perf_counter()
        loader.consolidate_records([cat], cl, pend, nps, abi.build_candidates(order[:k]), False)
        ct = time.perf_counter() - c0
        out["cpu_baseline"] = {"value": k / ct, "unit": "candidates/s", "cores": 1, "kind": "port", "cpu_model": cpu_model(),
                               "sample": "the first %d candidates of the same cluster, one oracle SimulateScheduling + computeConsolidation each "
                                         "(a scheduler rebuilt per candidate, as the reference does)" % k}
    return out


def bench_group(args, types, zones, n_devices):
    """C4 through kg_group: ONE process drives n_devices GPUs (context + worker thread per device, records gathered in this
    process, selection replayed) -- how a Go controller, which is one process without torch or RCCL, would use the node.
    Reported next to the contract's one-process-per-GPU numbers, never as `value`."""
    from karpenter_amd import abi, lib, workloads as wl
    nodes, bound = wl.existing_cluster(args.nodes, 11, types, zones)
    order = wl.disruption_order(nodes, bound)
    infos, ctx = wl.ec2_facts_776(), wl.ec2_context(zones)
    ec2 = abi.build_ec2_catalog(infos, ctx, ctx["limits"], ctx["bandwidth"], {i["InstanceType"]: zones for i in infos})
    cl = abi.build_cluster(nodes, bound)
    pend, nps = abi.build_pods([]), abi.build_nodepools([wl.default_nodepool()], wl.WELL_KNOWN_LABELS)
    G = lib.Group(n_devices=n_devices)
    out = {"n_devices": G.n, "how": "kg_group_consolidate: one host thread and one HIP stream per device, no collective"}
    try:
        G.set_collect_stats(False)
        hc, hl = G.catalog_build_ec2(ec2), G.cluster_upload(cl)
        for multi in [m for m in (False, True) if args.cons in ("both", "multi" if m else "single")]:
            cd = abi.build_candidates(order)
            for _ in range(max(args.warmup, 1)):
                G.consolidate([hc], hl, pend, nps, cd, multi, want_records=False)
            t0 = time.perf_counter()
            dms = []
            for _ in range(args.steps):
                dec, _, ms = G.consolidate([hc], hl, pend, nps, cd, multi, want_records=False)
                dms.append(ms)
            dt = time.perf_counter() - t0
            units = (min(len(order), 101) - 1) if multi else len(order)
            out["multi_node" if multi else "single_node"] = {
                "value": units * args.steps / dt, "unit": "prefixes/s" if multi else "candidates/s", "ms_per_round": dt * 1e3 / args.steps,
                "device_ms_per_round_slowest_device": sum(dms) / len(dms), "decision": {k: dec[k] for k in ("decision", "n_candidates", "first_candidate")}}
        G.cluster_free(hl)
        G.catalog_free(hc)
    finally:
        G.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--pods", type=int, default=10000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--nodes", type=int, default=5000, help="C4 cluster size")
    ap.add_argument("--no-consolidation", action="store_true")
    ap.add_argument("--cons", choices=("both", "single", "multi"), default="both", help="consolidation legs to run (profiling passes pick one)")
    ap.add_argument("--no-alg-bytes", action="store_true", help="skip the untimed visit-counter rounds (profiling passes: every launch of a kernel is then the same work)")
    ap.add_argument("--no-cold", action="store_true", help="skip the cold consolidation rounds")
    ap.add_argument("--no-dense", action="store_true", help="skip the dense-cluster consolidation legs")
    ap.add_argument("--no-c3c5", action="store_true", help="skip the full-size config 3 / config 5 solve legs")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))

    import torch
    import torch.distributed as dist
    from karpenter_amd import lib, problem as pb, workloads as wl

    torch.cuda.set_device(local_rank)
    L = lib.Library(device=local_rank)
    # catalog-776 encoded by the library from the raw EC2 facts (the List() miss path), resident before any timed region
    zones, hcat, types = pb.product_catalog(L, 3)
    group = None
    if world > 1:
        # the in-process variant over the same `world` GPUs, by rank 0 alone, BEFORE the ranks rendezvous (the others wait in
        # init_process_group on the host: their GPUs are idle, and the timed regions below are not touched)
        if rank == 0 and not args.no_consolidation:
            try:
                group = bench_group(args, types, zones, world)
            except Exception as e:  # noqa: BLE001  (a secondary number must never cost the contract's line)
                group = {"error": repr(e)[:300]}
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # every rank solves its own replica of the C2 batch (seed 42 on rank 0, the committed seed)
    prob = pb.Problem(None, wl.mixed_pods(args.pods, 42 + rank, zones, pb.families(types)), [wl.default_nodepool()])
    c2 = solve_leg(L, hcat, prob, args.steps, args.warmup, barrier, want_alg=not args.no_alg_bytes)
    dt = c2.pop("dt")
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    c3 = c5 = None
    if world == 1 and not args.no_c3c5:
        p3 = pb.Problem(None, wl.spread_pods(200, 250, 7), [wl.default_nodepool()])
        c3 = solve_leg(L, hcat, p3, max(1, min(args.steps, 3)), 1, barrier, want_alg=not args.no_alg_bytes)
        c3.pop("dt")
        c3["workload"] = "C3: 50 000 pods = 200 deployments x 250 replicas, zone spread (maxSkew 1) on all, hostname spread on half, hostname anti-affinity on a tenth, seed 7"
        del p3
        z6, hcat6, t6 = pb.product_catalog(L, 6)
        nodes, bound = wl.existing_cluster(2000, 13, t6, z6, nodepool="np-a")
        p5 = pb.Problem(None, wl.mixed_pods(100000, 99, z6, pb.families(t6)), pb.c5_nodepools(), nodes=nodes, bound=bound)
        c5 = solve_leg(L, hcat6, p5, max(1, min(args.steps, 2)), 1, barrier, want_alg=False)
        c5.pop("dt")
        # (the rows the reference visits on C5: counted once with the exact scan -- 38 s, not repeated here -- and committed)
        try:
            with open(os.path.join(ROOT, "profiles", "r03_alg_bytes.json")) as f:
                ab = json.load(f)["config5"]
            k5 = c5["breakdown_ms"]["k_solve"]
            c5["roofline"].update({"algorithmic_bytes_per_launch": ab["algorithmic_bytes_per_launch"], "achieved": ab["algorithmic_bytes_per_launch"] / (k5 * 1e-3) / 1e9,
                                   "frac": ab["algorithmic_bytes_per_launch"] / (k5 * 1e-3) / 1e9 / HBM_PEAK_GBS})
            c5["roofline"]["note"] += " -- algorithmic bytes from profiles/r03_alg_bytes.json (" + ab["source"].split(";")[0] + ")"
        except (OSError, KeyError, ValueError):
            c5["roofline"]["note"] += " -- not collected for this leg (the exact-scan run with the visit counters takes longer than the leg)"
        c5["workload"] = "C5: 100 000 mixed pods (seed 99) x 776 types x 6 zones x spot+on-demand, 4 NodePools weights 100/50/10/0 with cpu limits 2 000 / 10 000 cores, 2 000 existing nodes, truncation to 60"
        del p5
        L.dll.kg_catalog_free(hcat6)

    cons = None
    if not args.no_consolidation:
        cons = bench_consolidation(args, L, hcat, types, zones, world, rank, barrier, dist, torch)

    if rank == 0:
        copy_gbs = measured_copy_gbs(torch)
        value = args.pods * world * args.steps / dt
        roof = c2["roofline"]
        traffic, tsrc = pmc_traffic(roof["kernel"], args.pods == 10000 and world == 1)
        roof.update({"traffic": traffic, "traffic_source": tsrc, "peak_measured_copy": copy_gbs, "frac_of_measured_copy": roof["achieved"] / copy_gbs})
        out = {
            "metric": "pods/sec in Solve (10k pods x 750 types) + consolidation candidates/sec", "value": value, "unit": "pods/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt * 1000.0 / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "int64", "data": "synthetic",
            "config": {"workload": "C2: %d mixed cpu/mem/gpu pods with arch/zone/instance-family/generation constraints x %d types (the documented us-east-1 catalog-776 minus "
                                   "the types without a row in the reference's ENI-limits table) x 3 zones x spot+on-demand, 1 NodePool, seed 42"
                                   % (args.pods, len(types)), "replicas": world, "n_claims": c2["n_claims"],
                       "catalog": "built by the library (kg_catalog_build_ec2) from the reference's generated EC2 facts"},
            "breakdown_ms": c2["breakdown_ms"], "roofline": roof,
        }
        if c3 is not None:
            out["config3"], out["config5"] = c3, c5
        if cons is not None:
            if group is None:
                try:
                    group = bench_group(args, types, zones, 0)  # every visible device (1 on a one-GPU box)
                except Exception as e:  # noqa: BLE001
                    group = {"error": repr(e)[:300]}
            cons["in_process_group"] = group
            out["consolidation"] = cons
        if not args.no_cpu_baseline:
            # CPU oracle (oracle/oracle.cpp, -O2, one thread) on the same batch: 7 full Solves of the workload (~1 s each).
            # The oracle's own catalog (oracle/catalog_oracle.py) describes the same 776 types (tests/catalog_common.py).
            from tests.helpers import Problem as OracleProblem, catalog776
            oz, otypes = catalog776(3)
            oprob = OracleProblem(otypes, wl.mixed_pods(args.pods, 42, oz, pb.families(otypes)), [wl.default_nodepool()])
            times = []
            for _ in range(7):
                c0 = time.perf_counter()
                oprob.run_oracle()
                times.append(time.perf_counter() - c0)
            best = sorted(times)[len(times) // 2]
            out["cpu_baseline"] = {"value": args.pods / best, "unit": "pods/s", "cores": 1, "kind": "port",
                                   "cpu_model": cpu_model(), "host_cores_visible": os.cpu_count(), "runs": len(times),
                                   "sample": "7 full Solves of the same %d-pod batch, median wall of ko_solve (descriptor parse included), one thread" % args.pods,
                                   "go_reference": "not run: no Go toolchain and no sigs.k8s.io/karpenter module on this box (BASELINE.md section 2; source under bench/go_ref/)"}
            out["speedup_vs_cpu_baseline"] = value / out["cpu_baseline"]["value"]
        print(json.dumps(out))
    L.dll.kg_catalog_free(hcat)
    L.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
