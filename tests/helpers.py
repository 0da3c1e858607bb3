"""Shared test plumbing: catalogs, descriptor building, oracle-vs-library comparison."""
import functools
import os

from karpenter_amd import abi, lib, problem as _problem, workloads as wl
from oracle import loader

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU_PATH = os.path.join(ROOT, "tests", "emu", "libkarpgpu_emu.so")


@functools.lru_cache(maxsize=None)
def catalog776(n_zones=3):
    zones = wl.zones_n(n_zones)
    types = loader.catalog_types(wl.ec2_facts_776(), wl.ec2_context(zones))
    return zones, types


families = _problem.families


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


def c4_full(dense=False):
    """BASELINE.json config 4 at full size: 5 000 existing nodes (seed 11), ~12 bound pods each at ~35 %
    utilisation, 0 pending pods, 1 NodePool -> (types, nodes, bound, candidate order).  At 35 % every candidate's pods fit
    elsewhere (5 000 DELETE verdicts); dense=True is the bench's second leg: the same 5 000 nodes filled to 90-100 % of
    their cpu (seed 17), whose single-node records split into no-op / DELETE / REPLACE about evenly."""
    zones, types = catalog776()
    nodes, bound = wl.dense_cluster(5000, 17, types, zones, lo=0.9, hi=1.0) if dense else wl.existing_cluster(5000, 11, types, zones)
    return types, nodes, bound, wl.disruption_order(nodes, bound)


def records_digest(records):
    """sha256 over the comparable part of kg_candidate_record rows (abi.record_tuple), in unit order."""
    import hashlib
    import json
    return hashlib.sha256(json.dumps([abi.record_tuple(r) for r in records], separators=(",", ":")).encode()).hexdigest()


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


class Problem(_problem.Problem):
    """One scheduling problem in descriptor form, usable by both the oracle and the library."""

    def run_oracle(self):
        st, rb = loader.solve([self.cat], self.cluster, self.pods, self.nodepools, self.n_pods)
        assert st == 0, "oracle status %d" % st
        return rb


c5_nodepools = _problem.c5_nodepools


# parity scenarios for the scheduler features beyond the BASELINE shapes (Preferences.Relax steps, minDomains, daemon
# overhead, node taints / tolerations, capacity-type spread, pod affinity, OR-ed node-affinity terms, negative operators)
FEATURE_SCENARIOS = ["schedule_anyway", "prefer_no_schedule", "preferred_pod_affinity", "min_domains", "daemon_overhead", "node_taints",
                     "ct_spread", "pod_affinity", "two_required_terms", "negative_selectors", "relax_everything", "volume_limits",
                     "nodes_without_topology_key"]


def _pod(uid, cpu, mem_gi, app, ts=1700000000, **kw):
    p = {"uid": uid, "ns": "default", "creation_ts": ts, "labels": {"app": app}, "requests": {"cpu": cpu, "memory": mem_gi * wl.GI * 1000}}
    p.update(kw)
    return p


def feature_scenario(name):
    zones, types = catalog776()
    fams = families(types)
    G = wl.G
    if name == "schedule_anyway":
        # NodePool "a-only" serves zone a; "rest" serves b / c but is tainted.  A ScheduleAnyway zone spread (maxSkew 1) lets the
        # first replica of each app into zone a; the second cannot keep the skew there, cannot reach b / c, and drops the
        # constraint (Preferences.Relax step 5) to land in zone a after all.  A DoNotSchedule twin stays unschedulable.
        a_only = wl.default_nodepool("a-only", weight=10, extra_reqs=[(wl.L_ZONE, "In", [zones[0]])])
        rest = wl.default_nodepool("rest", extra_reqs=[(wl.L_ZONE, "In", zones[1:])])
        rest["taints"] = [("team", "other", "NoSchedule")]
        pods = []
        for d in range(6):
            app = "any-%d" % d
            sel = {"match_labels": {"app": app}}
            when = "ScheduleAnyway" if d % 3 else "DoNotSchedule"
            spreads = [{"key": wl.L_ZONE, "max_skew": 1, "when": when, "selector": sel}]
            if d % 2:
                spreads.append({"key": wl.L_HOSTNAME, "max_skew": 2, "when": "ScheduleAnyway", "selector": sel})
            for k in range(7):
                pods.append(_pod("%s-%02d" % (app, k), [250, 500, 1000][d % 3], 1 + d % 2, app, ts=1700000000 + d, spreads=spreads))
        return Problem(types, pods + wl.mixed_pods(60, 17, zones, fams), [rest, a_only])
    if name == "prefer_no_schedule":
        # the heavier NodePool carries a PreferNoSchedule taint: pods that do not tolerate it go to the lighter pool while it
        # has room under its limit, then fail, gain the toleration (Relax step 6) and use the tainted pool
        soft = wl.default_nodepool("soft", weight=50)
        soft["taints"] = [("maintenance", "soon", "PreferNoSchedule")]
        plain = wl.default_nodepool("plain", weight=10, limits={"cpu": 40000})
        pods = wl.mixed_pods(300, 23, zones, fams)
        for i, p in enumerate(pods):
            if i % 5 == 0:
                p["tolerations"] = [("maintenance", "Equal", "soon", "PreferNoSchedule")]
            elif i % 7 == 0:
                p["tolerations"] = [(None, "Exists", None, None)]
            elif i % 11 == 0:
                p["tolerations"] = [("maintenance", "Exists", None, "NoSchedule")]  # wrong effect: does not help
        return Problem(types, pods, [soft, plain])
    if name == "preferred_pod_affinity":
        # preferred pod affinity towards an app that does not exist (dropped first), preferred hostname anti-affinity among the
        # replicas that a cpu limit cannot honour for all of them (dropped once the NodePool runs out), weights decide the order
        pool = wl.default_nodepool(limits={"cpu": 2000000})
        pods = []
        for d in range(4):
            app = "pref-%d" % d
            sel = {"match_labels": {"app": app}}
            ghost = {"match_labels": {"app": "ghost-%d" % (d % 2)}}
            for k in range(12):
                p = _pod("%s-%02d" % (app, k), 500, 1, app, ts=1700000000 + d)
                p["affinity_preferred"] = [{"key": wl.L_ZONE, "selector": ghost, "weight": 10 + d}]
                if d % 2 == 0:
                    p["affinity_preferred"].append({"key": wl.L_HOSTNAME, "selector": ghost, "weight": 90})
                p["anti_affinity_preferred"] = [{"key": wl.L_HOSTNAME, "selector": sel, "weight": 50}]
                if d == 3:
                    p["anti_affinity_preferred"].append({"key": wl.L_ZONE, "selector": sel, "weight": 70})
                pods.append(p)
        return Problem(types, pods + wl.mixed_pods(40, 29, zones, fams), [pool])
    if name == "min_domains":
        # minDomains above the number of zones: the global minimum counts as 0, so every zone takes maxSkew replicas and the
        # rest stay pending; minDomains the zones do meet behaves like a plain spread
        pods = []
        for d, (md, skew) in enumerate(((4, 1), (3, 1), (5, 2), (2, 1), (None, 1))):
            app = "md-%d" % d
            sel = {"match_labels": {"app": app}}
            for k in range(9):
                pods.append(_pod("%s-%02d" % (app, k), 500, 2, app, ts=1700000000 + d,
                                 spreads=[{"key": wl.L_ZONE, "max_skew": skew, "min_domains": md, "when": "DoNotSchedule", "selector": sel}]))
        return Problem(types, pods, [wl.default_nodepool()])
    if name == "daemon_overhead":
        # daemonset requests: NodePool daemon overhead on every new NodeClaim (cpu, memory, pods and a resource no pod asks
        # for), remaining daemon requests on existing nodes
        heavy = wl.default_nodepool("heavy", weight=10)
        heavy["daemon_overhead"] = {"cpu": 1500, "memory": 3 * wl.GI * 1000, "pods": 4000, "ephemeral-storage": 2 * wl.GI * 1000}
        light = wl.default_nodepool("light", extra_reqs=[(G + "/instance-category", "In", ["c", "m"])])
        light["daemon_overhead"] = {"cpu": 200, "memory": 256 * wl.MI * 1000, "pods": 2000}
        nodes, bound = wl.existing_cluster(80, 31, types, zones, nodepool="heavy")
        for i, nd in enumerate(nodes):
            if i % 2 == 0:
                nd["daemon_requests"] = {"cpu": 300 + 100 * (i % 5), "memory": 512 * wl.MI * 1000, "pods": 1000}
        pods = wl.mixed_pods(500, 37, zones, fams)
        for i, p in enumerate(pods):
            if i % 4 == 0:
                p["node_selector"] = dict(p.get("node_selector") or {}, **{wl.L_NODEPOOL: "light" if i % 8 else "heavy"})
        return Problem(types, pods, [heavy, light], nodes=nodes, bound=bound)
    if name == "node_taints":
        # tainted existing nodes (NoSchedule / NoExecute / PreferNoSchedule, with and without values) against Equal / Exists
        # tolerations with and without key and effect; a tainted NodePool next to a plain one
        nodes, bound = wl.existing_cluster(150, 41, types, zones)
        taints = [[("dedicated", "gpu", "NoSchedule")], [("dedicated", "batch", "NoSchedule")], [("evict", "", "NoExecute")],
                  [("dedicated", "gpu", "NoSchedule"), ("evict", "", "NoExecute")], [("soft", "x", "PreferNoSchedule")], []]
        for i, nd in enumerate(nodes):
            nd["taints"] = taints[i % len(taints)]
        tols = [[("dedicated", "Equal", "gpu", "NoSchedule")], [("dedicated", "Exists", None, None)], [(None, "Exists", None, None)],
                [("evict", "Exists", None, "NoExecute")], [("dedicated", "Equal", "gpu", None), ("evict", "Equal", "", "NoExecute")],
                [("dedicated", "Equal", "batch", "NoExecute")], [(None, "Exists", None, "NoSchedule")], []]
        pods = wl.mixed_pods(600, 43, zones, fams)
        for i, p in enumerate(pods):
            p["tolerations"] = tols[i % len(tols)]
        tainted = wl.default_nodepool("tainted", weight=10)
        tainted["taints"] = [("dedicated", "batch", "NoSchedule")]
        return Problem(types, pods, [tainted, wl.default_nodepool()], nodes=nodes, bound=bound)
    if name == "volume_limits":
        # ExistingNode.Add's volume limits (Appendix A.5): nodes with a CSINode count per driver, bound pods already holding
        # volumes (some nodes full, one over its limit), pending pods with 0..3 volumes of two drivers; nodes without a limit
        # for a driver take any number; NodeClaims have no limit
        nodes, bound = wl.existing_cluster(120, 51, types, zones, util=0.2)
        EBS, EFS = "ebs.csi.aws.com", "efs.csi.aws.com"
        for i, nd in enumerate(nodes):
            if i % 4 != 3:
                nd["volume_limits"] = {EBS: [1, 2, 4, 25][i % 4]}
            if i % 5 == 0:
                nd["volume_limits"] = dict(nd.get("volume_limits") or {}, **{EFS: 1})
        for i, p in enumerate(bound):
            if i % 3 == 0:
                p["volumes"] = [(EBS, "vol-bound-%05d" % i)] + ([(EBS, "vol-bound-%05d-b" % i)] if i % 9 == 0 else [])
        pods = wl.mixed_pods(500, 53, zones, fams)
        for i, p in enumerate(pods):
            k = i % 4
            p["volumes"] = [(EBS, "vol-%04d-%d" % (i, j)) for j in range(k)] + ([(EFS, "fs-%04d" % i)] if i % 7 == 0 else [])
        return Problem(types, pods, [wl.default_nodepool()], nodes=nodes, bound=bound)
    if name == "ct_spread":
        # topology spread over capacity types (scheduling.md:370-373), alone and together with a zone spread
        pods = []
        for d in range(5):
            app = "ct-%d" % d
            sel = {"match_labels": {"app": app}}
            spreads = [{"key": wl.L_CT, "max_skew": 1 + d % 2, "when": "DoNotSchedule", "selector": sel}]
            if d >= 3:
                spreads.append({"key": wl.L_ZONE, "max_skew": 1, "when": "DoNotSchedule", "selector": sel})
            for k in range(10):
                p = _pod("%s-%02d" % (app, k), [500, 1000, 2000][d % 3], 2, app, ts=1700000000 + d, spreads=spreads)
                if d == 2 and k % 2:
                    p["node_selector"] = {wl.L_CT: "on-demand"}
                pods.append(p)
        return Problem(types, pods + wl.mixed_pods(50, 47, zones, fams), [wl.default_nodepool()])
    if name == "pod_affinity":
        # required pod affinity in a provisioning Solve: self-selecting replicas bootstrap one zone / one node and follow it;
        # followers of another app land where that app went; followers of nobody stay pending; anti-affinity to a peer app
        # on the zone key keeps two apps apart
        pods = []
        for d in range(3):
            app = "herd-%d" % d
            sel = {"match_labels": {"app": app}}
            key = wl.L_ZONE if d != 1 else wl.L_HOSTNAME
            for k in range(8):
                pods.append(_pod("%s-%02d" % (app, k), 500 if d != 1 else 250, 1, app, ts=1700000000 + d, affinity_required=[{"key": key, "selector": sel}]))
        for k in range(6):
            pods.append(_pod("follow-%02d" % k, 250, 1, "follow", ts=1700000100, affinity_required=[{"key": wl.L_ZONE, "selector": {"match_labels": {"app": "herd-0"}}}]))
        for k in range(3):
            pods.append(_pod("lost-%02d" % k, 250, 1, "lost", ts=1700000200, affinity_required=[{"key": wl.L_ZONE, "selector": {"match_labels": {"app": "nobody"}}}]))
        for k in range(6):
            pods.append(_pod("apart-%02d" % k, 2000, 4, "apart", ts=1700000300,
                             anti_affinity_required=[{"key": wl.L_ZONE, "selector": {"match_labels": {"app": "herd-2"}}}]))
        return Problem(types, pods + wl.mixed_pods(40, 53, zones, fams), [wl.default_nodepool()])
    if name == "two_required_terms":
        # nodeSelectorTerms are OR-ed: the first term is tried, Relax drops it when nothing satisfies it (step 1); three
        # terms relax twice; a satisfiable first term never looks at the second
        pods = wl.mixed_pods(240, 59, zones, fams)
        for i, p in enumerate(pods):
            p.pop("required_terms", None)
            if i % 3 == 0:
                p["required_terms"] = [[(G + "/instance-family", "In", ["nosuch"])], [(G + "/instance-family", "In", ["m5", "c5", "r5"])]]
            elif i % 3 == 1:
                p["required_terms"] = [[(G + "/instance-category", "In", ["c"]), (G + "/instance-generation", "Gt", ["5"])],
                                       [(G + "/instance-category", "In", ["zz"])]]
            elif i % 9 == 2:
                p["required_terms"] = [[(wl.L_ZONE, "In", ["mars-1a"])], [(G + "/instance-cpu", "In", ["3"])], [(wl.L_ARCH, "In", ["arm64"])]]
        return Problem(types, pods, [wl.default_nodepool()])
    if name == "nodes_without_topology_key":
        # a third of the existing nodes carry no zone label (and some no capacity-type label): they are outside every domain
        # of a zone (capacity-type) topology -- pods that consult such a group cannot land there, their bound pods are not
        # counted -- while pods without topology still fill them
        nodes, bound = wl.existing_cluster(45, 23, types, zones, util=0.2)
        for i, nd in enumerate(nodes):
            if i % 3 == 0:
                del nd["labels"][wl.L_ZONE]
            if i % 5 == 0:
                del nd["labels"][wl.L_CT]
        for i, p in enumerate(bound):
            p["labels"] = {"app": "web-%d" % (i % 3)}
        pods = []
        for d in range(3):
            app = "web-%d" % d
            sel = {"match_labels": {"app": app}}
            for k in range(14):
                kw = {}
                if d == 0:
                    kw["spreads"] = [{"key": wl.L_ZONE, "max_skew": 1, "when": "DoNotSchedule", "selector": sel}]
                elif d == 1:
                    kw["anti_affinity_required"] = [{"key": wl.L_ZONE, "selector": {"match_labels": {"app": "lonely"}}}]
                    kw["spreads"] = [{"key": wl.L_CT, "max_skew": 2, "when": "DoNotSchedule", "selector": sel}]
                else:
                    kw["affinity_required"] = [{"key": wl.L_ZONE, "selector": sel}]
                pods.append(_pod("%s-%02d" % (app, k), [250, 500][k % 2], 1, app, ts=1700000000 + d, **kw))
        pods += [_pod("lonely-%d" % k, 500, 1, "lonely", anti_affinity_required=[{"key": wl.L_ZONE, "selector": {"match_labels": {"app": "lonely"}}}]) for k in range(4)]
        return Problem(types, pods + wl.mixed_pods(40, 71, zones, fams), [wl.default_nodepool()], nodes=nodes, bound=bound)
    if name == "negative_selectors_wk":
        # the same without the custom label key: negative operators on well-known keys only, which the speculative
        # multi-wave kernel serves (a NodeClaim never lacks a well-known key in the sense of Requirements.Compatible)
        prob = feature_scenario("negative_selectors")
        pods = [p for p in prob.raw_pods if not any(k == "example.com/team" for t in p.get("required_terms", []) for k, _, _ in t)]
        return Problem(types, pods, [wl.default_nodepool()])
    if name == "negative_selectors":
        # NotIn / DoesNotExist / Exists / Lt (and Gt) in node-affinity terms, on keys every type defines, keys only some
        # types define (gpu / local nvme) and a key no type defines
        pods = wl.mixed_pods(360, 61, zones, fams)
        terms = [[(G + "/instance-family", "NotIn", ["m5", "c5", "t3", "t3a"])],
                 [(G + "/instance-gpu-name", "DoesNotExist", [])],
                 [(G + "/instance-local-nvme", "Exists", [])],
                 [(G + "/instance-cpu", "Lt", ["9"]), (G + "/instance-cpu", "Gt", ["1"])],
                 [(wl.L_ZONE, "NotIn", [zones[1]])],
                 [(G + "/instance-generation", "Lt", ["6"]), (G + "/instance-category", "NotIn", ["t"])],
                 [("example.com/team", "DoesNotExist", [])],
                 [(wl.L_CT, "NotIn", ["spot"])],
                 [(G + "/instance-local-nvme", "Gt", ["500"])],
                 [(G + "/instance-memory", "Lt", ["16385"]), (wl.L_ARCH, "NotIn", ["arm64"])]]
        for i, p in enumerate(pods):
            p.pop("required_terms", None)
            if i % 2 == 0:
                p["required_terms"] = [terms[(i // 2) % len(terms)]]
        return Problem(types, pods, [wl.default_nodepool()])
    if name == "relax_everything":
        # one pod class walks the whole Preferences.Relax ladder: two required terms, preferred pod affinity and
        # anti-affinity, two preferred node-affinity terms, a ScheduleAnyway spread, a PreferNoSchedule NodePool
        soft = wl.default_nodepool("soft")
        soft["taints"] = [("soft", "yes", "PreferNoSchedule")]
        pods = []
        for k in range(10):
            app = "ladder"
            sel = {"match_labels": {"app": app}}
            p = _pod("ladder-%02d" % k, 1000, 2, app)
            p["required_terms"] = [[(G + "/instance-family", "In", ["nosuch"])], [(G + "/instance-category", "In", ["m", "c"])]]
            p["affinity_preferred"] = [{"key": wl.L_ZONE, "selector": {"match_labels": {"app": "ghost"}}, "weight": 5}]
            p["anti_affinity_preferred"] = [{"key": wl.L_ZONE, "selector": sel, "weight": 5}]
            p["preferred_terms"] = [(20, [(G + "/instance-family", "In", ["nosuch2"])]), (10, [(wl.L_ARCH, "In", ["arm64"])])]
            p["spreads"] = [{"key": wl.L_CT, "max_skew": 1, "when": "ScheduleAnyway", "selector": sel}]
            pods.append(p)
        return Problem(types, pods + wl.mixed_pods(30, 67, zones, fams), [soft])
    raise KeyError(name)


def scenario(name):
    """Named parity scenarios shared by the emulation (CPU) and GPU test files."""
    if name in FEATURE_SCENARIOS or name == "negative_selectors_wk":
        return feature_scenario(name)
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


def check_context_age(L):
    """Kernel choice and limits depend on the batch and the catalog, not on what a long-lived context has seen: the context
    first learns 80 zone names, 70 capacity-type names and 300 instance-family names from unrelated clusters and batches
    (more than the 64 values a 1-word key holds, more than the 16 the single-wave kernel stages), then solves C2 on 3 zones:
    same results as the oracle, served by the single-wave kernel."""
    zones, types = catalog776()
    fams = families(types)
    t0 = [t for t in types if t["name"] == "m5.large"][0]
    nodes = []
    for i in range(300):
        labels = {wl.L_ZONE: "zz-%02d" % (i % 80), wl.L_CT: "ct-%02d" % (i % 70), wl.L_HOSTNAME: "old-%03d" % i, wl.L_NODEPOOL: "default",
                  wl.L_IT: "m5.large", wl.G + "/instance-family": "fam%03d" % i, wl.L_ARCH: "arch%d" % (i % 20)}
        nodes.append({"name": "old-%03d" % i, "labels": labels, "taints": [], "available": {"cpu": 1000, "memory": wl.GI * 1000, "pods": 10000},
                      "capacity": dict(t0["capacity"]), "daemon_requests": {}, "initialized": True, "managed": True, "deleting": False})
    hcl = L.cluster_upload(abi.build_cluster(nodes, []))
    L.dll.kg_cluster_free(hcl)
    elsewhere = wl.mixed_pods(200, 77, zones, fams)
    for i, p in enumerate(elsewhere):
        p["node_selector"] = {wl.L_ZONE: "elsewhere-%02d" % (i % 40)}
    prob = Problem(types, elsewhere, [wl.default_nodepool()])
    assert diff_results(prob.run_oracle().decode(), prob.run_library(L).decode()) == ""
    prob = Problem(types, wl.mixed_pods(2000, 42, zones, fams), [wl.default_nodepool()])
    ro = prob.run_oracle()
    L.set_collect_stats(False)
    try:
        w0 = L.stats().wave_solves
        rg = prob.run_library(L)
        assert L.stats().wave_solves == w0 + 1, "an aged context lost the single-wave kernel"
    finally:
        L.set_collect_stats(True)
    d = diff_results(ro.decode(), rg.decode())
    assert d == "", d
    # topology over zones on the aged context (1-word key: at most 64 values IN THE BATCH)
    p3 = scenario("c3_small")
    assert diff_results(p3.run_oracle().decode(), p3.run_library(L).decode()) == ""
