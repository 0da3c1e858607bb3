"""Volume limits of ExistingNode.Add beyond the parity scenarios: the limits do change outcomes,
and a volume shared between pods -- a set union the device path does not model -- is refused, so the caller can fall back."""
import pytest

from karpenter_amd import abi, workloads as wl
from tests.helpers import Problem, diff_results, feature_scenario


def check_volume_limits(L):
    prob = feature_scenario("volume_limits")
    want = prob.run_oracle().decode()
    assert diff_results(prob.run_library(L).decode(), want) == ""
    # the same batch without limits lands differently: the limits are what kept pods off the nodes
    free_nodes = [dict(nd, volume_limits={}) for nd in prob.raw_nodes]
    free = Problem(prob.types, prob.raw_pods, prob.raw_nodepools, nodes=free_nodes, bound=prob.raw_bound).run_oracle().decode()
    on_nodes = lambda r: sum(1 for k, _ in r["pods"] if k == abi.POD_EXISTING_NODE)
    assert on_nodes(free) > on_nodes(want) > 0
    # a node already over a driver's limit takes no pod at all, with or without volumes (ExceedsLimits walks the union of the
    # node's and the pod's volumes over every driver -- PARITY UNPINNED, oracle/oracle.cpp)
    held = {}
    for p in prob.raw_bound:
        for drv, vol in p.get("volumes", []):
            held.setdefault((p["node"], drv), set()).add(vol)
    over = {n for (n, drv), vols in held.items() if drv in (prob.raw_nodes[n].get("volume_limits") or {}) and len(vols) > prob.raw_nodes[n]["volume_limits"][drv]}
    assert over, "the scenario holds nodes over their limit"
    assert not any(k == abi.POD_EXISTING_NODE and i in over for k, i in want["pods"])
    # internal bookkeeping stays inside: no claim reports a volume resource
    assert all(not any("volumes" in k for k in c["requests"]) for c in want["claims"])
    assert all(not any("volumes" in k for k in c["requests"]) for c in prob.run_library(L).decode()["claims"])
    # two pending pods mounting one volume; a pending pod mounting a volume that a bound pod holds
    pods = [dict(p) for p in prob.raw_pods]
    pods[5] = dict(pods[5], volumes=[("ebs.csi.aws.com", "vol-shared")])
    pods[9] = dict(pods[9], volumes=[("ebs.csi.aws.com", "vol-shared")])
    with pytest.raises(Exception, match="EUNSUPPORTED"):
        Problem(prob.types, pods, prob.raw_nodepools, nodes=prob.raw_nodes, bound=prob.raw_bound).run_library(L)
    held = next(p["volumes"][0] for p in prob.raw_bound if p.get("volumes"))
    pods = [dict(p) for p in prob.raw_pods]
    pods[3] = dict(pods[3], volumes=[held])
    with pytest.raises(Exception, match="EUNSUPPORTED"):
        Problem(prob.types, pods, prob.raw_nodepools, nodes=prob.raw_nodes, bound=prob.raw_bound).run_library(L)
    # (the oracle models the union: it answers both)
    assert Problem(prob.types, pods, prob.raw_nodepools, nodes=prob.raw_nodes, bound=prob.raw_bound).run_oracle().decode()["pods"]
