#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on MI355X: pods/sec in Solve (10k mixed pods x 776 instance types, config C2).

    python bench.py [--gpus N] [--steps K] [--warmup W]

A "step" is one kg_solve over one batch of C2 pods: the catalog is already resident in HBM
(kg_catalog_upload before the timed region); the step covers requirement compile, the device kernels and the
result download / price-ordered truncation, i.e. everything between handing over the pod batch and having the
NodeClaims.  Solve does not shard (FFD is one chain of dependent placements): with N ranks each rank runs an
independent replica of the batch ("replicas only", weak scaling, no data-path collective).

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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--pods", type=int, default=10000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))

    import torch
    import torch.distributed as dist
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from karpenter_amd import lib, workloads as wl
    from tests.helpers import Problem, catalog776, families

    zones, types = catalog776()
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
    L.set_collect_stats(True)
    alg_bytes = prob.run_library(L, hcat).res.algorithmic_bytes
    L.set_collect_stats(False)
    for _ in range(args.warmup):
        prob.run_library(L, hcat)
    barrier()
    t0 = time.perf_counter()
    solve_ms, pre_ms, compile_ms, decode_ms = [], [], [], []
    for _ in range(args.steps):
        rb = prob.run_library(L, hcat)
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

    if rank == 0:
        ms_per_step = dt * 1000.0 / args.steps
        value = args.pods * world * args.steps / dt
        k_ms = sum(solve_ms) / len(solve_ms)
        achieved = alg_bytes / (k_ms * 1e-3) / 1e9
        out = {
            "metric": "pods/sec in Solve (10k pods x 750 types)", "value": value, "unit": "pods/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "int64", "data": "synthetic",
            "config": {"workload": "C2: %d mixed cpu/mem/gpu pods with arch/zone/instance-family/generation constraints x 776 types x 3 zones x spot+on-demand, 1 NodePool, seed 42"
                                   % args.pods, "replicas": world, "n_claims": int(rb.res.n_claims)},
            "breakdown_ms": {"host_compile": sum(compile_ms) / len(compile_ms), "prepass_kernels": sum(pre_ms) / len(pre_ms),
                             "k_solve": k_ms, "host_decode_truncate": sum(decode_ms) / len(decode_ms)},
            "roofline": {"bound": "hbm", "kernel": "k_solve", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "note": "rows the reference algorithm visits up to its first feasible index (SURVEY.md 8d): "
                                 "212 B/existing node, 400 B/NodeClaim, 336 B/instance type"},
        }
        if not args.no_cpu_baseline:
            # CPU oracle (oracle/oracle.cpp, -O2, one thread) on the same batch: 3 full Solves of the workload
            times = []
            for _ in range(3):
                c0 = time.perf_counter()
                prob.run_oracle()
                times.append(time.perf_counter() - c0)
            best = sorted(times)[len(times) // 2]
            out["cpu_baseline"] = {"value": args.pods / best, "unit": "pods/s", "cores": 1, "kind": "port",
                                   "sample": "3 full Solves of the same %d-pod batch, median wall of ko_solve (descriptor parse included)" % args.pods}
        print(json.dumps(out))
    L.dll.kg_catalog_free(hcat)
    L.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
