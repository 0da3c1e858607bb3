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


class Problem:
    """One scheduling problem in descriptor form, usable by both the oracle and the library."""

    def __init__(self, types, pods, nodepools, nodes=None, bound=None, max_instance_types=60):
        self.types = types
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
