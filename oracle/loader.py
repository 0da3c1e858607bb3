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
        _LIB.ko_launch_overrides.restype = C.c_int32
        _LIB.ko_launch_overrides.argtypes = [C.c_void_p, C.c_void_p, abi.Range, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32,
                                             C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
        _LIB.ko_launch_filter.restype = C.c_int32
        _LIB.ko_launch_filter.argtypes = [C.c_void_p, C.c_void_p, abi.Range, abi.Range, C.c_int32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32,
                                          C.POINTER(C.c_uint32), C.POINTER(C.c_uint8)]
        _LIB.ko_truncate.restype = C.c_int32
        _LIB.ko_truncate.argtypes = [C.c_void_p, C.c_void_p, abi.Range, C.c_void_p, C.c_uint32, C.c_int32, C.c_void_p, C.POINTER(C.c_uint32),
                                     C.POINTER(C.c_uint8)]
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


def candidates_order(catalog, cluster, n_nodes, lifetime=None, blocked=None):
    """-> (status, node indices in candidate order, costs)"""
    import numpy as np
    f = lib().ko_candidates_order
    f.restype = C.c_int32
    f.argtypes = [C.c_void_p] * 6 + [C.c_uint32, C.c_void_p]
    out, costs, n = np.zeros(max(n_nodes, 1), "<i4"), np.zeros(max(n_nodes, 1), "<f8"), C.c_uint32(0)
    lt = None if lifetime is None else np.ascontiguousarray(lifetime, "<f8")
    bl = None if blocked is None else np.ascontiguousarray(blocked, "u1")
    st = f(C.addressof(catalog.desc) if catalog is not None else None, C.addressof(cluster.desc), abi._ptr(lt), abi._ptr(bl), out.ctypes.data,
           costs.ctypes.data, n_nodes, C.byref(n))
    return st, out[:n.value].tolist(), costs[:n.value].tolist()


def consolidate_validate(catalogs, cluster, pending, nodepools, candidates, decision, type_ids):
    """-> (status, valid)"""
    import numpy as np
    f = lib().ko_consolidate_validate
    f.restype = C.c_int32
    f.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int32, C.c_void_p, C.c_uint32, C.c_void_p]
    arr = (C.c_void_p * len(catalogs))(*[C.addressof(c.desc) for c in catalogs])
    ca, ta, v = np.array(list(candidates), "<i4"), np.array(list(type_ids), "<u4"), C.c_uint8(0)
    st = f(arr, len(catalogs), C.addressof(cluster.desc), C.addressof(pending.desc), C.addressof(nodepools.desc), ca.ctypes.data, len(ca),
           decision, abi._ptr(ta), len(ta), C.byref(v))
    return st, bool(v.value)


def consolidate_decide(catalogs, cluster, pending, nodepools, cands, multi, cap_types=1024):
    """The reference's sequential selection -> (status, decision dict)."""
    arr = (C.c_void_p * len(catalogs))(*[C.addressof(c.desc) for c in catalogs])
    db = abi.DecisionBuffer(cap_types)
    st = lib().ko_consolidate_decide(arr, len(catalogs), C.addressof(cluster.desc), C.addressof(pending.desc), C.addressof(nodepools.desc),
                                     C.addressof(cands.desc), 1 if multi else 0, C.addressof(db.dec))
    return st, db.decode()


def launch_filter(catalog, requirements, requests, well_known, max_types=60):
    import numpy as np
    pools, rr, qq = abi.build_launch(requirements, requests)
    wk = (C.c_char_p * len(well_known))(*[w.encode() for w in well_known])
    out = np.zeros(4096, "<u4")
    n, mv = C.c_uint32(0), C.c_uint8(0)
    st = lib().ko_launch_filter(C.addressof(catalog.desc), C.addressof(pools.desc), rr, qq, max_types, wk, len(well_known), out.ctypes.data,
                                out.shape[0], C.byref(n), C.byref(mv))
    assert st == 0
    return out[:n.value].tolist(), bool(mv.value)


def launch_overrides(catalog, requirements, type_ids, subnet_zones, well_known):
    import numpy as np
    zones = list(subnet_zones)
    pools, rr, _, zid = abi.build_launch(requirements, {}, zones or ["-"])
    ids, zs = np.array(list(type_ids), dtype="<u4"), np.array(zid[:len(zones)], dtype="<u4")
    wk = (C.c_char_p * len(well_known))(*[w.encode() for w in well_known])
    out = np.zeros(max(len(ids), 1) * max(len(zones), 1) * 2, abi.OVERRIDE)
    n, ct = C.c_uint32(0), C.c_int32(-1)
    st = lib().ko_launch_overrides(C.addressof(catalog.desc), C.addressof(pools.desc), rr, ids.ctypes.data if len(ids) else None, len(ids),
                                   zs.ctypes.data if len(zs) else None, len(zs), wk, len(well_known), C.byref(ct), out.ctypes.data, out.shape[0], C.byref(n))
    assert st == 0
    return ("spot" if ct.value == 1 else "on-demand"), [(int(o["type_id"]), zones[int(o["subnet"])]) for o in out[:n.value]]


def truncate(catalog, requirements, type_ids, max_types=60):
    import numpy as np
    pools, rr, _ = abi.build_launch(requirements, {})
    ids = np.array(list(type_ids), dtype="<u4")
    out = np.zeros(max(len(ids), 1), "<u4")
    n, mv = C.c_uint32(0), C.c_uint8(0)
    st = lib().ko_truncate(C.addressof(catalog.desc), C.addressof(pools.desc), rr, ids.ctypes.data if len(ids) else None, len(ids), max_types,
                           out.ctypes.data, C.byref(n), C.byref(mv))
    assert st == 0
    return out[:n.value].tolist(), bool(mv.value)
