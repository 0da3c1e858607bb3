"""Python binding of libkarpgpu's C ABI (include/karpgpu.h) -- the same calls a Go shim makes.

`Library()` loads karpenter_amd/libkarpgpu.so (hipcc, gfx950) and fails loudly when it is
missing or when no GPU is present: there is no CPU fallback in the product.  Tests that need to
debug host logic without a GPU pass the path of the TEST-ONLY emulation build explicitly.
"""
import ctypes as C
import os

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_PATH = os.path.join(_HERE, "libkarpgpu.so")

EXPORTS = ["kg_init", "kg_destroy", "kg_last_error", "kg_stats_get", "kg_catalog_upload", "kg_catalog_build_ec2", "kg_catalog_update_offerings", "kg_candidates_order", "kg_consolidate_validate",
           "kg_catalog_query", "kg_catalog_size", "kg_catalog_free", "kg_cluster_upload", "kg_cluster_free", "kg_solve",
           "kg_truncate", "kg_launch_filter", "kg_consolidate_single", "kg_consolidate_multi", "kg_consolidate_select", "kg_solve_noalloc", "kg_catalog_query_requirements", "kg_launch_overrides",
           "kg_device_count", "kg_group_init", "kg_group_destroy", "kg_group_size", "kg_group_context", "kg_group_last_error", "kg_group_catalog_upload",
           "kg_group_catalog_update_offerings", "kg_group_catalog_free", "kg_group_cluster_upload", "kg_group_cluster_free", "kg_group_consolidate"]


class KgError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__("%s: %s" % (abi.STATUS.get(status, status), msg))
        self.status = status


class Library:
    def __init__(self, path=None, device=-1):
        path = path or DEFAULT_PATH
        if not os.path.exists(path):
            raise RuntimeError("libkarpgpu.so not built (%s): run __graft_entry__.build()" % path)
        self.dll = C.CDLL(path)
        d = self.dll
        d.kg_init.argtypes = [C.c_int32, C.POINTER(C.c_void_p)]
        d.kg_last_error.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
        d.kg_destroy.argtypes = [C.c_void_p]
        d.kg_catalog_upload.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_void_p)]
        d.kg_catalog_free.argtypes = [C.c_void_p]
        d.kg_catalog_update_offerings.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64]
        d.kg_catalog_build_ec2.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_void_p)]
        d.kg_catalog_query.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_char_p, C.c_size_t]
        d.kg_catalog_size.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]
        d.kg_cluster_upload.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]
        d.kg_cluster_free.argtypes = [C.c_void_p]
        d.kg_solve.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        d.kg_solve_noalloc.argtypes = d.kg_solve.argtypes
        d.kg_backend.restype = C.c_char_p
        for fn in (d.kg_consolidate_single, d.kg_consolidate_multi):
            fn.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32,
                           C.POINTER(C.c_uint32), C.POINTER(C.c_double)]
        d.kg_launch_filter.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, abi.Range, abi.Range, C.c_int32, C.c_void_p, C.c_uint32,
                                       C.POINTER(C.c_uint32), C.POINTER(C.c_uint8)]
        d.kg_launch_overrides.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, abi.Range, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32,
                                          C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
        d.kg_truncate.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, abi.Range, C.c_void_p, C.c_uint32, C.c_int32, C.c_void_p,
                                  C.POINTER(C.c_uint32), C.POINTER(C.c_uint8)]
        d.kg_consolidate_select.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int32, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
        d.kg_set_collect_stats.argtypes = [C.c_void_p, C.c_int32]
        self.ctx = C.c_void_p()
        st = d.kg_init(device, C.byref(self.ctx))
        if st != 0:
            raise KgError(st, "kg_init failed (no GPU?)")

    def backend(self):
        return self.dll.kg_backend().decode()

    def set_collect_stats(self, enabled):
        """Reference-visit counters (kg_results.visited_*, algorithmic_bytes): exact but one extra pass per placement."""
        self._check(self.dll.kg_set_collect_stats(self.ctx, int(enabled)))

    def _check(self, st):
        if st != 0:
            buf = C.create_string_buffer(1024)
            self.dll.kg_last_error(self.ctx, buf, 1024)
            raise KgError(st, buf.value.decode())

    def catalog_upload(self, built, generation=1):
        h = C.c_void_p()
        self._check(self.dll.kg_catalog_upload(self.ctx, C.addressof(built.desc), generation, C.byref(h)))
        return h

    def catalog_build_ec2(self, built, generation=1):
        """The List() miss path natively: raw EC2 facts -> instance types (offerings, requirements, capacity, overhead)."""
        h = C.c_void_p()
        self._check(self.dll.kg_catalog_build_ec2(self.ctx, C.addressof(built.desc), generation, C.byref(h)))
        return h

    def catalog_query(self, catalog, type_index, resources):
        """-> (name, {resource: capacity milli}, {resource: allocatable milli})"""
        arr = (C.c_char_p * len(resources))(*[r.encode() for r in resources])
        cap = (C.c_int64 * len(resources))()
        alloc = (C.c_int64 * len(resources))()
        name = C.create_string_buffer(128)
        st = self.dll.kg_catalog_query(catalog, type_index, arr, len(resources), cap, alloc, name, 128)
        if st != 0:
            raise KgError(st, "kg_catalog_query")
        return name.value.decode(), dict(zip(resources, cap)), dict(zip(resources, alloc))

    def catalog_requirements(self, catalog, type_index):
        """-> {label key: sorted value list} ([] = DoesNotExist) as the encoder holds them for the type."""
        f = self.dll.kg_catalog_query_requirements
        f.argtypes = [C.c_void_p, C.c_uint32, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]
        need = C.c_size_t(0)
        f(catalog, type_index, None, 0, C.byref(need))
        buf = C.create_string_buffer(need.value)
        st = f(catalog, type_index, buf, need.value, C.byref(need))
        if st != 0:
            raise KgError(st, "kg_catalog_query_requirements")
        out = {}
        for line in buf.value.decode().splitlines():
            k, op, vals = line.split("\t")
            out[k] = vals.split(",") if op == "In" else []
        return out

    def catalog_update_offerings(self, catalog, updates, generation):
        """updates: iterable of (type_index, zone, capacity_type, available or None, price or None)."""
        strtab, arr = abi.build_offering_updates(updates)
        self._check(self.dll.kg_catalog_update_offerings(self.ctx, catalog, C.addressof(strtab.desc), abi._ptr(arr), arr.shape[0], generation))

    def stats(self):
        st = abi.Stats()
        self._check(self.dll.kg_stats_get(self.ctx, C.byref(st)))
        return st

    def candidates_order(self, catalog, cluster, n_nodes, lifetime=None, blocked=None):
        """-> (node indices in candidate order, costs); catalog may be None."""
        import numpy as np

        class Inputs(C.Structure):
            _fields_ = [("lifetime_remaining", C.c_void_p), ("blocked", C.c_void_p), ("n_nodes", C.c_uint32)]
        lt = None if lifetime is None else np.ascontiguousarray(lifetime, "<f8")
        bl = None if blocked is None else np.ascontiguousarray(blocked, "u1")
        inp = Inputs(abi._ptr(lt), abi._ptr(bl), n_nodes)
        out, costs, n = np.zeros(max(n_nodes, 1), "<i4"), np.zeros(max(n_nodes, 1), "<f8"), C.c_uint32(0)
        f = self.dll.kg_candidates_order
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
        self._check(f(self.ctx, catalog, cluster, C.byref(inp), out.ctypes.data, costs.ctypes.data, n_nodes, C.byref(n)))
        return out[:n.value].tolist(), costs[:n.value].tolist()

    def consolidate_validate(self, catalogs, cluster, pending, nodepools, candidates, decision, type_ids):
        """-> valid (bool)"""
        import numpy as np
        arr = (C.c_void_p * len(catalogs))(*catalogs)
        ca, ta, v, ms = np.array(list(candidates), "<i4"), np.array(list(type_ids), "<u4"), C.c_uint8(0), C.c_double(0)
        f = self.dll.kg_consolidate_validate
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int32, C.c_void_p,
                      C.c_uint32, C.c_void_p, C.c_void_p]
        self._check(f(self.ctx, arr, len(catalogs), cluster, C.addressof(pending.desc), C.addressof(nodepools.desc), ca.ctypes.data, len(ca),
                      decision, abi._ptr(ta), len(ta), C.byref(v), C.byref(ms)))
        return bool(v.value)

    def catalog_size(self, catalog):
        n, g = C.c_uint32(0), C.c_uint64(0)
        self.dll.kg_catalog_size(catalog, C.byref(n), C.byref(g))
        return n.value, g.value

    def cluster_upload(self, built):
        h = C.c_void_p()
        self._check(self.dll.kg_cluster_upload(self.ctx, C.addressof(built.desc), C.byref(h)))
        return h

    def solve(self, catalogs, cluster, pods, nodepools, n_pods, cap_claims=None, noalloc=False, out=None):
        """kg_solve; noalloc=True -> kg_solve_noalloc (device tables in the library's static scratch array).  `out`: result
        buffers of an earlier call to fill again (they are the caller's: a controller allocates them once)."""
        arr = (C.c_void_p * len(catalogs))(*catalogs)
        rb = out if out is not None else abi.ResultBuffers(n_pods, cap_claims)
        fn = self.dll.kg_solve_noalloc if noalloc else self.dll.kg_solve
        self._check(fn(self.ctx, arr, len(catalogs), cluster, C.addressof(pods.desc), C.addressof(nodepools.desc),
                                      C.addressof(rb.res)))
        return rb

    def consolidate_records(self, catalogs, cluster, pending, nodepools, cands, multi, out=None):
        """One SimulateScheduling + computeConsolidation per unit of the shard -> (records, device_ms).
        `out`: the caller's record buffer (>= n_candidates rows of abi.CANDIDATE_RECORD), as a controller that polls would keep."""
        import numpy as np
        arr = (C.c_void_p * len(catalogs))(*catalogs)
        recs = out if out is not None and out.shape[0] >= max(cands.desc.n_candidates, 1) else np.zeros(max(cands.desc.n_candidates, 1), abi.CANDIDATE_RECORD)
        k, ms = C.c_uint32(0), C.c_double(0)
        fn = self.dll.kg_consolidate_multi if multi else self.dll.kg_consolidate_single
        self._check(fn(self.ctx, arr, len(catalogs), cluster, C.addressof(pending.desc), C.addressof(nodepools.desc), C.addressof(cands.desc),
                       recs.ctypes.data, recs.shape[0], C.byref(k), C.byref(ms)))
        return recs[:k.value], ms.value

    def consolidate_select(self, catalog, records, n_candidates, multi, cap_types=1024):
        """`catalog`: one handle, or the list of handles the records were computed with."""
        cats = list(catalog) if isinstance(catalog, (list, tuple)) else [catalog]
        arr = (C.c_void_p * len(cats))(*cats)
        db = abi.DecisionBuffer(cap_types)
        self._check(self.dll.kg_consolidate_select(self.ctx, arr, len(cats), 1 if multi else 0, records.ctypes.data if len(records) else None,
                                                   len(records), n_candidates, C.addressof(db.dec)))
        return db.decode()

    def launch_filter(self, catalog, requirements, requests, max_types=60):
        """resolveInstanceTypes -> filterInstanceTypes -> Truncate for one NodeClaim -> (type ids, minValues failed)."""
        import numpy as np
        pools, rr, qq = abi.build_launch(requirements, requests)
        out = np.zeros(4096, "<u4")
        n, mv = C.c_uint32(0), C.c_uint8(0)
        self._check(self.dll.kg_launch_filter(self.ctx, catalog, C.addressof(pools.desc), rr, qq, max_types, out.ctypes.data, out.shape[0],
                                              C.byref(n), C.byref(mv)))
        return out[:n.value].tolist(), bool(mv.value)

    def launch_overrides(self, catalog, requirements, type_ids, subnet_zones):
        """getCapacityType + getOverrides -> ("spot" | "on-demand", [(type id, zone)])."""
        import numpy as np
        zones = list(subnet_zones)
        pools, rr, _, zid = abi.build_launch(requirements, {}, zones or ["-"])
        ids, zs = np.array(list(type_ids), dtype="<u4"), np.array(zid[:len(zones)], dtype="<u4")
        out = np.zeros(max(len(ids), 1) * max(len(zones), 1) * 2, abi.OVERRIDE)
        n, ct = C.c_uint32(0), C.c_int32(-1)
        self._check(self.dll.kg_launch_overrides(self.ctx, catalog, C.addressof(pools.desc), rr, ids.ctypes.data if len(ids) else None, len(ids),
                                                 zs.ctypes.data if len(zs) else None, len(zs), C.byref(ct), out.ctypes.data, out.shape[0], C.byref(n)))
        return ("spot" if ct.value == 1 else "on-demand"), [(int(o["type_id"]), zones[int(o["subnet"])]) for o in out[:n.value]]

    def truncate(self, catalog, requirements, type_ids, max_types=60):
        import numpy as np
        pools, rr, _ = abi.build_launch(requirements, {})
        ids = np.array(list(type_ids), dtype="<u4")
        out = np.zeros(max(len(ids), 1), "<u4")
        n, mv = C.c_uint32(0), C.c_uint8(0)
        self._check(self.dll.kg_truncate(self.ctx, catalog, C.addressof(pools.desc), rr, ids.ctypes.data if len(ids) else None, len(ids), max_types,
                                         out.ctypes.data, C.byref(n), C.byref(mv)))
        return out[:n.value].tolist(), bool(mv.value)

    def close(self):
        if self.ctx:
            self.dll.kg_destroy(self.ctx)
            self.ctx = C.c_void_p()


class Group:
    """kg_group: one process, one context + worker thread per device (include/karpgpu.h, "several GPUs, one process")."""

    def __init__(self, path=None, devices=None, n_devices=0):
        path = path or DEFAULT_PATH
        if not os.path.exists(path):
            raise RuntimeError("libkarpgpu.so not built (%s): run __graft_entry__.build()" % path)
        self.dll = d = C.CDLL(path)
        d.kg_group_init.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p)]
        d.kg_group_destroy.argtypes = [C.c_void_p]
        d.kg_group_size.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]
        d.kg_group_context.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p)]
        d.kg_group_last_error.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
        d.kg_group_catalog_upload.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
        d.kg_group_catalog_update_offerings.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64]
        d.kg_group_catalog_free.argtypes = [C.c_void_p, C.c_void_p]
        d.kg_group_cluster_upload.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        d.kg_group_cluster_free.argtypes = [C.c_void_p, C.c_void_p]
        d.kg_group_consolidate.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                           C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32), C.c_void_p, C.POINTER(C.c_double)]
        d.kg_stats_get.argtypes = [C.c_void_p, C.c_void_p]
        d.kg_set_collect_stats.argtypes = [C.c_void_p, C.c_int32]
        self.g = C.c_void_p()
        if devices is not None:
            arr = (C.c_int32 * len(devices))(*devices)
            st = d.kg_group_init(arr, len(devices), C.byref(self.g))
        else:
            st = d.kg_group_init(None, n_devices, C.byref(self.g))
        if st != 0:
            raise KgError(st, "kg_group_init failed (no GPU?)")
        n = C.c_uint32(0)
        d.kg_group_size(self.g, C.byref(n))
        self.n = n.value

    def _check(self, st):
        if st != 0:
            buf = C.create_string_buffer(1024)
            self.dll.kg_group_last_error(self.g, buf, 1024)
            raise KgError(st, buf.value.decode("utf-8", "replace"))

    def context(self, i):
        h = C.c_void_p()
        self._check(self.dll.kg_group_context(self.g, i, C.byref(h)))
        return h

    def set_collect_stats(self, enabled):
        for i in range(self.n):
            self.dll.kg_set_collect_stats(self.context(i), int(enabled))

    def stats(self, i):
        s = abi.Stats()
        self.dll.kg_stats_get(self.context(i), C.addressof(s))
        return s

    def catalog_upload(self, built, generation=1):
        hs = (C.c_void_p * self.n)()
        self._check(self.dll.kg_group_catalog_upload(self.g, C.addressof(built.desc), generation, hs))
        return hs

    def catalog_build_ec2(self, built, generation=1):
        """kg_catalog_build_ec2 on every context of the group (each device encodes its own resident copy) -> handle array."""
        hs = (C.c_void_p * self.n)()
        self.dll.kg_catalog_build_ec2.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_void_p)]
        for i in range(self.n):
            h = C.c_void_p()
            st = self.dll.kg_catalog_build_ec2(self.context(i), C.addressof(built.desc), generation, C.byref(h))
            if st != 0:
                raise KgError(st, "kg_catalog_build_ec2 on context %d" % i)
            hs[i] = h
        return hs

    def catalog_update_offerings(self, handles, updates, generation):
        st, a = abi.build_offering_updates(updates)
        self._check(self.dll.kg_group_catalog_update_offerings(self.g, handles, C.addressof(st.desc), a.ctypes.data if len(a) else None, len(a), generation))

    def catalog_free(self, handles):
        self.dll.kg_group_catalog_free(self.g, handles)

    def cluster_upload(self, built):
        hs = (C.c_void_p * self.n)()
        self._check(self.dll.kg_group_cluster_upload(self.g, C.addressof(built.desc), hs))
        return hs

    def cluster_free(self, handles):
        self.dll.kg_group_cluster_free(self.g, handles)

    def consolidate(self, catalogs, clusters, pending, nodepools, cands, multi, cap_types=1024, want_records=True):
        """catalogs: list (one per catalog of the call) of per-device handle arrays -> (decision dict, records, device_ms)."""
        import numpy as np
        nc = len(catalogs)
        arr = (C.c_void_p * (self.n * nc))(*[catalogs[k][i] for i in range(self.n) for k in range(nc)])
        recs = np.zeros(max(cands.desc.n_candidates, 1) if want_records else 0, abi.CANDIDATE_RECORD)
        k, ms = C.c_uint32(0), C.c_double(0)
        db = abi.DecisionBuffer(cap_types)
        self._check(self.dll.kg_group_consolidate(self.g, arr, nc, clusters, C.addressof(pending.desc), C.addressof(nodepools.desc), C.addressof(cands.desc),
                                                  1 if multi else 0, recs.ctypes.data if want_records else None, recs.shape[0], C.byref(k),
                                                  C.addressof(db.dec), C.byref(ms)))
        return db.decode(), recs[:k.value], ms.value

    def close(self):
        if self.g:
            self.dll.kg_group_destroy(self.g)
            self.g = C.c_void_p()
