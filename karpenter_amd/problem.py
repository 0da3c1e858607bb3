"""One scheduling problem in descriptor form, and the synthetic catalogs of BASELINE.json's configs built by the product
itself (kg_catalog_build_ec2, the List() miss path of pkg/providers/instancetype/instancetype.go:100-175): what bench.py's
GPU legs and smoke() use, with no import of tests/ or oracle/.  tests/helpers.py adds the oracle side."""
from karpenter_amd import abi, workloads as wl

CATALOG_RESOURCES = ["cpu", "memory", "ephemeral-storage", "pods", "vpc.amazonaws.com/pod-eni", "nvidia.com/gpu", "amd.com/gpu",
                     "aws.amazon.com/neuron", "habana.ai/gaudi", "vpc.amazonaws.com/efa", "vpc.amazonaws.com/PrivateIPv4Address"]


class Problem:
    """Pods + NodePools (+ existing nodes and their bound pods) over a catalog.  `types` = list of catalog type dicts
    (name, requirements, capacity, overhead, offerings) when the catalog is uploaded from a descriptor, None when every run
    names a resident catalog handle."""

    def __init__(self, types, pods, nodepools, nodes=None, bound=None, max_instance_types=60):
        self.types = types
        self.raw_pods, self.raw_nodes, self.raw_bound, self.raw_nodepools = pods, nodes, bound, nodepools  # (kept for property checks / variations)
        self.n_pods = len(pods)
        self.cat = abi.build_catalog(types) if types is not None and types and "offerings" in types[0] else None
        self.pods = abi.build_pods(pods)
        self.nodepools = abi.build_nodepools(nodepools, wl.WELL_KNOWN_LABELS, max_instance_types)
        self.cluster = abi.build_cluster(nodes, bound or []) if nodes is not None else None

    def run_library(self, L, hcat=None, out=None):
        own = hcat is None
        if own:
            hcat = L.catalog_upload(self.cat)
        hcl = L.cluster_upload(self.cluster) if self.cluster is not None else None
        try:
            return L.solve([hcat], hcl, self.pods, self.nodepools, self.n_pods, out=out)
        finally:
            if hcl is not None:
                L.dll.kg_cluster_free(hcl)
            if own:
                L.dll.kg_catalog_free(hcat)


def families(types):
    return sorted({t["name"].split(".")[0] for t in types})


def c5_nodepools():
    """C5: four weighted NodePools {100, 50, 10, 0}, the two heaviest with cpu limits of 2 000 and 10 000 cores."""
    return [wl.default_nodepool("np-a", weight=100, limits={"cpu": 2000000}), wl.default_nodepool("np-b", weight=50, limits={"cpu": 10000000}),
            wl.default_nodepool("np-c", weight=10), wl.default_nodepool("np-d", weight=0)]


def product_catalog(L, n_zones=3):
    """catalog-776 encoded by the library from the raw EC2 facts -> (zones, resident catalog handle, type
    views).  A type view carries what the synthetic clusters need of a type -- name, single-valued labels, capacity,
    overhead -- read back from the encoder (kg_catalog_query / kg_catalog_query_requirements)."""
    zones = wl.zones_n(n_zones)
    infos = wl.ec2_facts_776()
    ctx = wl.ec2_context(zones)
    oz = {i["InstanceType"]: zones for i in infos}
    h = L.catalog_build_ec2(abi.build_ec2_catalog(infos, ctx, ctx["limits"], ctx["bandwidth"], oz))
    types = []
    for i in range(L.catalog_size(h)[0]):
        name, cap, alloc = L.catalog_query(h, i, CATALOG_RESOURCES)
        reqs = [(k, "In" if v else "DoesNotExist", v) for k, v in L.catalog_requirements(h, i).items()]
        # (computeCapacity lists its ten resources whatever their value, PrivateIPv4Address on Windows only: types.go:193-210, 67-69)
        types.append({"name": name, "requirements": reqs, "capacity": {k: v for k, v in cap.items() if v or k != CATALOG_RESOURCES[-1]},
                      "overhead": {k: cap[k] - alloc[k] for k in cap if cap[k] - alloc[k]}})
    return zones, h, types
