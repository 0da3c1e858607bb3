"""Shared test plumbing: catalogs, descriptor building, oracle-vs-library comparison."""
import functools
import os

from karpenter_amd import abi, lib, workloads as wl
from oracle import loader

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU_PATH = os.path.join(ROOT, "tests", "emu", "libkarpgpu_emu.so")


@functools.lru_cache(maxsize=None)
def catalog776(n_zones=3):
    zones = wl.zones_n(n_zones)
    types = loader.catalog_types(wl.ec2_facts_776(), wl.ec2_context(zones))
    return zones, types


def families(types):
    return sorted({t["name"].split(".")[0] for t in types})


def diff_results(a, b, limit=5):
    """Human-readable differences between two decoded results ('' when identical)."""
    out = []
    if a["pods"] != b["pods"]:
        d = [(i, x, y) for i, (x, y) in enumerate(zip(a["pods"], b["pods"])) if x != y]
        out.append("pod assignment differs at %d pods, first %s" % (len(d), d[:limit]))
    if len(a["claims"]) != len(b["claims"]):
        out.append("claim count %d vs %d" % (len(a["claims"]), len(b["claims"])))
    for i, (x, y) in enumerate(zip(a["claims"], b["claims"])):
        if x != y:
            for k in x:
                if x[k] != y[k]:
                    out.append("claim %d field %s: %s vs %s" % (i, k, str(x[k])[:200], str(y[k])[:200]))
            break
    return "\n".join(out)


def result_digest(decoded):
    """sha256 over the canonical JSON of a decoded result (pod assignments and every claim field diff_results compares)."""
    import hashlib
    import json
    return hashlib.sha256(json.dumps(decoded, sort_keys=True, separators=(",", ":")).encode()).hexdigest()


def full_size_problem(name):
    """BASELINE.json configs 3 and 5 at their full sizes."""
    if name == "c3_full":
        zones, types = catalog776()
        return Problem(types, wl.spread_pods(200, 250, 7), [wl.default_nodepool()])
    if name == "c5_full":
        z6, t6 = catalog776(6)
        nodes, bound = wl.existing_cluster(2000, 13, t6, z6, nodepool="np-a")
        return Problem(t6, wl.mixed_pods(100000, 99, z6, families(t6)), c5_nodepools(), nodes=nodes, bound=bound)
    raise KeyError(name)


def _with_host_ports(pods, seed, every=3):
    """Gives every `every`-th pod one or two host ports out of a small set (wildcard and specific ips, tcp / udp), so
    that conflicts (same port + protocol, equal ips or either 0.0.0.0) decide placements."""
    r = wl.SplitMix64(seed)
    ips = [None, "10.0.0.1", "10.0.0.2"]
    for i, p in enumerate(pods):
        if i % every:
            continue
        hp = [(ips[r.below(3)], 8000 + r.below(4), 6 if r.below(4) else 17)]
        if r.below(3) == 0:
            hp.append((ips[r.below(3)], 9000 + r.below(2), 6))
        p["host_ports"] = hp
    return pods


class Problem:
    """One scheduling problem in descriptor form, usable by both the oracle and the library."""

    def __init__(self, types, pods, nodepools, nodes=None, bound=None, max_instance_types=60):
        self.types = types
        self.raw_pods, self.raw_nodes = pods, nodes  # (kept for property checks)
        self.n_pods = len(pods)
        self.cat = abi.build_catalog(types)
        self.pods = abi.build_pods(pods)
        self.nodepools = abi.build_nodepools(nodepools, wl.WELL_KNOWN_LABELS, max_instance_types)
        self.cluster = abi.build_cluster(nodes, bound or []) if nodes is not None else None

    def run_oracle(self):
        st, rb = loader.solve([self.cat], self.cluster, self.pods, self.nodepools, self.n_pods)
        assert st == 0, "oracle status %d" % st
        return rb

    def run_library(self, L, hcat=None):
        own = hcat is None
        if own:
            hcat = L.catalog_upload(self.cat)
        hcl = L.cluster_upload(self.cluster) if self.cluster is not None else None
        try:
            return L.solve([hcat], hcl, self.pods, self.nodepools, self.n_pods)
        finally:
            if hcl is not None:
                L.dll.kg_cluster_free(hcl)
            if own:
                L.dll.kg_catalog_free(hcat)


def c5_nodepools():
    """C5: four weighted NodePools, the two heaviest with cpu limits (scaled)."""
    return [wl.default_nodepool("np-a", weight=100, limits={"cpu": 200000}), wl.default_nodepool("np-b", weight=50, limits={"cpu": 1000000}),
            wl.default_nodepool("np-c", weight=10), wl.default_nodepool("np-d", weight=0)]


def scenario(name):
    """Named parity scenarios shared by the emulation (CPU) and GPU test files."""
    zones, types = catalog776()
    if name == "c3_small":
        return Problem(types, wl.spread_pods(6, 20, 7), [wl.default_nodepool()])
    if name == "c3_mid":
        return Problem(types, wl.spread_pods(20, 50, 7), [wl.default_nodepool()])
    if name == "c3_large":
        return Problem(types, wl.spread_pods(200, 25, 7), [wl.default_nodepool()])
    if name == "existing_small":
        nodes, bound = wl.existing_cluster(60, 11, types, zones)
        return Problem(types, wl.mixed_pods(200, 3, zones, families(types)), [wl.default_nodepool()], nodes=nodes, bound=bound)
    if name == "existing_mid":
        nodes, bound = wl.existing_cluster(300, 12, types, zones)
        return Problem(types, wl.mixed_pods(1000, 4, zones, families(types)), [wl.default_nodepool()], nodes=nodes, bound=bound)
    if name == "existing_large":
        nodes, bound = wl.existing_cluster(2000, 12, types, zones)
        return Problem(types, wl.mixed_pods(4000, 4, zones, families(types)), [wl.default_nodepool()], nodes=nodes, bound=bound)
    if name in ("c5_small", "c5_mid"):
        z6, t6 = catalog776(6)
        n_nodes, n_pods = (100, 1500) if name == "c5_small" else (600, 12000)
        nodes, bound = wl.existing_cluster(n_nodes, 13, t6, z6, nodepool="np-a")
        return Problem(t6, wl.mixed_pods(n_pods, 99, z6, families(t6)), c5_nodepools(), nodes=nodes, bound=bound)
    if name == "host_ports":
        # host ports on a third of the pods: HostPortUsage conflicts split pods over NodeClaims that resources would share
        return Problem(types, _with_host_ports(wl.mixed_pods(600, 21, zones, families(types)), 5), [wl.default_nodepool()])
    if name == "host_ports_existing":
        # ... and on existing nodes whose bound pods already hold some of the ports
        nodes, bound = wl.existing_cluster(120, 14, types, zones)
        _with_host_ports(bound, 6, every=5)
        return Problem(types, _with_host_ports(wl.mixed_pods(500, 22, zones, families(types)), 7, every=2), [wl.default_nodepool()],
                       nodes=nodes, bound=bound)
    if name == "many_claims":
        # more NodeClaims in flight than the scan order the kernel keeps in LDS (KG_ORD = 2048): 2200 pods with hostname
        # self-anti-affinity (one per node), then small pods that land on those claims and reorder them
        sel = {"match_labels": {"app": "solo"}}
        big = [{"uid": "solo-%05d" % i, "ns": "default", "creation_ts": 1700000000, "labels": {"app": "solo"},
                "requests": {"cpu": 2000, "memory": 4 * wl.GI * 1000},
                "anti_affinity_required": [{"key": wl.L_HOSTNAME, "selector": sel}]} for i in range(2200)]
        small = wl.mixed_pods(500, 8, zones, families(types), prefix="small")
        return Problem(types, big + small, [wl.default_nodepool()])
    raise KeyError(name)
