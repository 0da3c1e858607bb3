#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on MI355X: pods/sec in Solve (10k mixed pods x 776 instance types, config C2).

    python bench.py [--gpus N] [--steps K] [--warmup W]

A "step" is one kg_solve over one batch of C2 pods: the catalog is already resident in HBM
(kg_catalog_upload before the timed region); the step covers requirement compile, the device kernels and the
result download / price-ordered truncation, i.e. everything between handing over the pod batch and having the
NodeClaims.  Solve does not shard (FFD is one chain of dependent placements): with N ranks each rank runs an
independent replica of the batch ("replicas only", weak scaling, no data-path collective).

The second half of the metric, consolidation candidates/sec (config C4: 5000-node cluster, single-node = one
SimulateScheduling per candidate, multi-node = one per prefix of the first 100), DOES shard: units are dealt round-robin
over the ranks, each rank runs its simulations with no data-path collective, and one all-gather of the fixed-size
records (RCCL) feeds the reference's selection.  It is reported in the `consolidation` object of the same line
(strong scaling: the whole job is the same 5000 candidates at every N).

Prints ONE JSON line on rank 0 with `roofline` (k_solve: algorithmic bytes of SURVEY.md 8(d) / HIP-event kernel
time) and `cpu_baseline` (the CPU oracle, one thread, same batch).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E ~8 TB/s


PMC_FILE = os.path.join("profiles", "r02f_pmc.json")


def pmc_traffic(kernel, default_workload):
    """HBM bytes per launch of `kernel` on the default workload, from the committed rocprofv3 PMC passes of this same
    command (FETCH_SIZE, WRITE_SIZE, each its own run; gfx950 correction of MI355X_MICROARCH.md applied; made by
    tools/gpu_round.sh + tools/pmc_summary.py) -- hardware counters cannot be read from inside the run.  -> (bytes or None,
    source file or None); None for any workload other than the one profiled."""
    path = os.path.join(ROOT, PMC_FILE)
    if not default_workload or not os.path.exists(path):
        return None, None
    with open(path) as f:
        k = json.load(f).get("kernels", {})
    hit = [v for name, v in k.items() if name.split("<")[0].split("::")[-1] == kernel]
    return (max(v["hbm_bytes_per_launch_corrected"] for v in hit), PMC_FILE) if hit else (None, None)


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


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


def bench_consolidation(args, L, hcat, types, zones, world, rank, barrier, dist, torch):
    """C4: consolidation on a synthetic cluster (seed 11), single- and multi-node, units sharded over the ranks."""
    from karpenter_amd import abi, distributed, workloads as wl
    nodes, bound = wl.existing_cluster(args.nodes, 11, types, zones)
    order = wl.disruption_order(nodes, bound)
    cl = abi.build_cluster(nodes, bound)
    pend, nps = abi.build_pods([]), abi.build_nodepools([wl.default_nodepool()], wl.WELL_KNOWN_LABELS)
    hcl = L.cluster_upload(cl)
    device = "cuda" if world > 1 else "cpu"
    out = {"workload": "C4: %d-node cluster (seed 11, ~12 pods/node, 35%% utilisation), %d bound pods, 0 pending" % (len(nodes), len(bound)),
           "scaling": "strong", "collective": "all_gather of %d-byte kg_candidate_record rows" % abi.CANDIDATE_RECORD.itemsize}
    for multi in [m for m in (False, True) if args.cons in ("both", "multi" if m else "single")]:
        # algorithmic bytes of one round (rows the reference visits, summed over this rank's simulations): one untimed round
        # with the visit counters on
        alg = 0.0
        if not args.no_alg_bytes:
            L.set_collect_stats(True)
            distributed.consolidate(L, hcat, hcl, pend, nps, order, multi, device=device)
            alg = L.stats().cons_algorithmic_bytes
            L.set_collect_stats(False)
        for _ in range(args.warmup):
            distributed.consolidate(L, hcat, hcl, pend, nps, order, multi, device=device)
        barrier()
        t0 = time.perf_counter()
        dms, kms, ims = [], [], []
        for _ in range(args.steps):
            dec, n_local, ms = distributed.consolidate(L, hcat, hcl, pend, nps, order, multi, device=device)
            st = L.stats()
            dms.append(ms)
            kms.append(st.cons_simulate_ms)
            ims.append(st.cons_sim_init_ms)
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        units = (min(len(order), 101) - 1) if multi else len(order)
        k_ms = sum(kms) / len(kms)
        achieved = alg / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
        traffic, src = pmc_traffic("k_simulate", args.nodes == 5000 and not multi and world == 1)
        out["multi_node" if multi else "single_node"] = {
            "value": units * args.steps / dt, "unit": "prefixes/s" if multi else "candidates/s", "units_per_round": units,
            "ms_per_round": dt * 1e3 / args.steps, "device_ms_per_round_rank0": sum(dms) / len(dms),
            "k_simulate_ms_rank0": k_ms, "k_sim_init_ms_rank0": sum(ims) / len(ims),
            "decision": {k: dec[k] for k in ("decision", "n_candidates", "first_candidate")},
            "roofline": {"bound": "hbm", "kernel": "k_simulate", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": src, "algorithmic_bytes_per_launch": alg,
                         "note": "rank 0's simulations of one round: sum over simulations of 212 B x existing nodes visited + 400 B x "
                                 "NodeClaims visited + 336 B x instance types filtered (SURVEY.md 8d) / HIP-event time of k_simulate"}}
    if rank == 0 and not args.no_cpu_baseline:
        from oracle import loader
        k = min(30, len(order))
        cat = abi.build_catalog(types)
        c0 = time.perf_counter()
        st, ro, _ = loader.consolidate_records([cat], cl, pend, nps, abi.build_candidates(order[:k]), False)
        ct = time.perf_counter() - c0
        out["cpu_baseline"] = {"value": k / ct, "unit": "candidates/s", "cores": 1, "kind": "port", "cpu_model": cpu_model(),
                               "sample": "the first %d candidates of the same cluster, one oracle SimulateScheduling + computeConsolidation each" % k}
    L.dll.kg_cluster_free(hcl)
    return out


def bench_group(args, types, zones, n_devices):
    """C4 through kg_group: ONE process drives n_devices GPUs (context + worker thread per device, records gathered in this
    process, selection replayed) -- how a Go controller, which is one process without torch or RCCL, would use the node.
    Reported next to the contract's one-process-per-GPU numbers, never as `value`."""
    from karpenter_amd import abi, lib, workloads as wl
    nodes, bound = wl.existing_cluster(args.nodes, 11, types, zones)
    order = wl.disruption_order(nodes, bound)
    cl, cat = abi.build_cluster(nodes, bound), abi.build_catalog(types)
    pend, nps = abi.build_pods([]), abi.build_nodepools([wl.default_nodepool()], wl.WELL_KNOWN_LABELS)
    G = lib.Group(n_devices=n_devices)
    out = {"n_devices": G.n, "how": "kg_group_consolidate: one host thread and one HIP stream per device, no collective"}
    try:
        G.set_collect_stats(False)
        hc, hl = G.catalog_upload(cat), G.cluster_upload(cl)
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
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))

    import torch
    import torch.distributed as dist
    from karpenter_amd import lib, workloads as wl
    from tests.helpers import Problem, catalog776, families

    zones, types = catalog776()
    torch.cuda.set_device(local_rank)
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

    # every rank solves its own replica of the C2 batch (seed 42 on rank 0, the committed seed)
    pods = wl.mixed_pods(args.pods, 42 + rank, zones, families(types))
    prob = Problem(types, pods, [wl.default_nodepool()])
    L = lib.Library(device=local_rank)
    hcat = L.catalog_upload(prob.cat)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # algorithmic bytes (rows the reference visits) come from one untimed solve with the visit counters on;
    # the timed steps run with them off, as a production caller would
    alg_bytes = 0.0
    if not args.no_alg_bytes:
        L.set_collect_stats(True)
        alg_bytes = prob.run_library(L, hcat).res.algorithmic_bytes
    L.set_collect_stats(False)
    from karpenter_amd import abi
    out_buffers = abi.ResultBuffers(args.pods)  # the caller's output buffers, allocated once as a controller would
    for _ in range(args.warmup):
        prob.run_library(L, hcat, out=out_buffers)
    barrier()
    t0 = time.perf_counter()
    solve_ms, pre_ms, compile_ms, decode_ms = [], [], [], []
    for _ in range(args.steps):
        rb = prob.run_library(L, hcat, out=out_buffers)
        solve_ms.append(rb.res.solve_ms)
        pre_ms.append(rb.res.prepass_ms)
        compile_ms.append(rb.res.host_compile_ms)
        decode_ms.append(rb.res.host_decode_ms)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    cons = None
    if not args.no_consolidation:
        cons = bench_consolidation(args, L, hcat, types, zones, world, rank, barrier, dist, torch)

    if rank == 0:
        copy_gbs = measured_copy_gbs(torch)
        ms_per_step = dt * 1000.0 / args.steps
        value = args.pods * world * args.steps / dt
        k_ms = sum(solve_ms) / len(solve_ms)
        achieved = alg_bytes / (k_ms * 1e-3) / 1e9
        kname = "k_solve_wave" if L.stats().wave_solves else "k_solve"
        traffic, tsrc = pmc_traffic(kname, args.pods == 10000 and world == 1)
        out = {
            "metric": "pods/sec in Solve (10k pods x 750 types) + consolidation candidates/sec", "value": value, "unit": "pods/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "int64", "data": "synthetic",
            "config": {"workload": "C2: %d mixed cpu/mem/gpu pods with arch/zone/instance-family/generation constraints x 776 types x 3 zones x spot+on-demand, 1 NodePool, seed 42"
                                   % args.pods, "replicas": world, "n_claims": int(rb.res.n_claims)},
            "breakdown_ms": {"host_compile": sum(compile_ms) / len(compile_ms), "prepass_kernels": sum(pre_ms) / len(pre_ms),
                             "k_solve": k_ms, "host_decode_truncate": sum(decode_ms) / len(decode_ms)},
            "roofline": {"bound": "hbm", "kernel": kname, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": tsrc,
                         "peak_measured_copy": copy_gbs, "frac_of_measured_copy": achieved / copy_gbs,
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "note": "rows the reference algorithm visits up to its first feasible index (SURVEY.md 8d): "
                                 "212 B/existing node, 400 B/NodeClaim, 336 B/instance type"},
        }
        if cons is not None:
            if group is None:
                try:
                    group = bench_group(args, types, zones, 0)  # every visible device (1 on a one-GPU box)
                except Exception as e:  # noqa: BLE001
                    group = {"error": repr(e)[:300]}
            cons["in_process_group"] = group
            out["consolidation"] = cons
        if not args.no_cpu_baseline:
            # CPU oracle (oracle/oracle.cpp, -O2, one thread) on the same batch: 7 full Solves of the workload (~1 s each)
            times = []
            for _ in range(7):
                c0 = time.perf_counter()
                prob.run_oracle()
                times.append(time.perf_counter() - c0)
            best = sorted(times)[len(times) // 2]
            out["cpu_baseline"] = {"value": args.pods / best, "unit": "pods/s", "cores": 1, "kind": "port",
                                   "cpu_model": cpu_model(), "host_cores_visible": os.cpu_count(), "runs": len(times),
                                   "sample": "7 full Solves of the same %d-pod batch, median wall of ko_solve (descriptor parse included), one thread" % args.pods,
                                   "go_reference": "not run: no Go toolchain and no sigs.k8s.io/karpenter module on this box (BASELINE.md section 2)"}
            out["speedup_vs_cpu_baseline"] = value / out["cpu_baseline"]["value"]
        print(json.dumps(out))
    L.dll.kg_catalog_free(hcat)
    L.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
