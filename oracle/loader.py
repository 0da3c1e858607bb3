"""TEST INFRASTRUCTURE ONLY -- loads oracle/_build/liboracle.so and exposes ko_* entry points.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import ctypes as C
import os
import subprocess

from karpenter_amd import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "_build", "liboracle.so")
        if not os.path.exists(path):
            build()
        _LIB = C.CDLL(path)
        _LIB.ko_solve.restype = C.c_int32
        _LIB.ko_solve.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        _LIB.ko_consolidate_records.restype = C.c_int32
        _LIB.ko_consolidate_records.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                                C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_double)]
        _LIB.ko_consolidate_decide.restype = C.c_int32
        _LIB.ko_consolidate_decide.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
    return _LIB


def solve(catalogs, cluster, pods, nodepools, n_pods, cap_claims=None):
    """catalogs: list of abi.Built(CatalogDesc); cluster: Built or None; returns (status, ResultBuffers)."""
    arr = (C.c_void_p * len(catalogs))(*[C.addressof(c.desc) for c in catalogs])
    rb = abi.ResultBuffers(n_pods, cap_claims)
    st = lib().ko_solve(arr, len(catalogs), C.addressof(cluster.desc) if cluster is not None else None,
                        C.addressof(pods.desc), C.addressof(nodepools.desc), C.addressof(rb.res))
    return st, rb


def catalog_types(infos, ctx):
    """Instance types for the given EC2 facts and provider context, via the catalog oracle."""
    from oracle import catalog_oracle as co
    zones = [s["zone"] for s in ctx["subnets"]]
    oz = ctx.get("offering_zones") or {i["InstanceType"]: zones for i in infos}
    return co.list_instance_types(infos, ctx, ctx["limits"], ctx["bandwidth"], oz)


def consolidate_records(catalogs, cluster, pending, nodepools, cands, multi):
    """computeConsolidation per unit of the shard -> (status, records array, ms)."""
    import numpy as np
    arr = (C.c_void_p * len(catalogs))(*[C.addressof(c.desc) for c in catalogs])
    n = cands.desc.n_candidates
    recs = np.zeros(max(n, 1), abi.CANDIDATE_RECORD)
    k, ms = C.c_uint32(0), C.c_double(0)
    st = lib().ko_consolidate_records(arr, len(catalogs), C.addressof(cluster.desc), C.addressof(pending.desc), C.addressof(nodepools.desc),
                                      C.addressof(cands.desc), 1 if multi else 0, recs.ctypes.data, recs.shape[0], C.byref(k), C.byref(ms))
    return st, recs[:k.value], ms.value


def consolidate_decide(catalogs, cluster, pending, nodepools, cands, multi, cap_types=1024):
    """The reference's sequential selection -> (status, decision dict)."""
    arr = (C.c_void_p * len(catalogs))(*[C.addressof(c.desc) for c in catalogs])
    db = abi.DecisionBuffer(cap_types)
    st = lib().ko_consolidate_decide(arr, len(catalogs), C.addressof(cluster.desc), C.addressof(pending.desc), C.addressof(nodepools.desc),
                                     C.addressof(cands.desc), 1 if multi else 0, C.addressof(db.dec))
    return st, db.decode()
