"""The drop-in boundary: libkarpgpu.so loads without a GPU, exports every entry point include/karpgpu.h declares, and its
element struct layouts are the ones the bindings (karpenter_amd/abi.py; a cgo shim in production) assume.  No compute call
is made here; kg_init must fail cleanly (KG_EDEVICE) when no GPU is present -- there is no CPU fallback in the product."""
import ctypes as C
import os
import re

import pytest

from karpenter_amd import abi, lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "karpgpu.h")

pytestmark = pytest.mark.skipif(not os.path.exists(lib.DEFAULT_PATH), reason="libkarpgpu.so missing: run __graft_entry__.build()")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"^\s*(?:int32_t|const char\*)\s+(kg_\w+)\s*\(", src, flags=re.M)))


def test_exports_every_declared_entry_point():
    dll = C.CDLL(lib.DEFAULT_PATH)
    names = declared_functions()
    assert len(names) >= 20, names
    assert set(lib.EXPORTS) <= set(names)
    for n in names:
        assert hasattr(dll, n), "libkarpgpu.so does not export %s" % n


def test_struct_layouts_match_bindings():
    dll = C.CDLL(lib.DEFAULT_PATH)
    out = (C.c_uint32 * 64)()
    n = dll.kg_abi_sizes(out, 64)
    assert n == len(abi.ABI_ELEMENTS)
    for (name, dt), size in zip(abi.ABI_ELEMENTS, list(out)[:n]):
        assert dt.itemsize == size, (name, dt.itemsize, size)


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(lib.KgError) as e:
        lib.Library()
    assert e.value.status == 3  # KG_EDEVICE


def test_roofline_row_sizes_are_the_frozen_ones():
    """roofline.json == the KG_ROW_BYTES_* constants the library prices row visits with."""
    import json
    rf = json.load(open(os.path.join(ROOT, "roofline.json")))
    src = open(HEADER).read()
    got = {k.lower(): int(v) for k, v in re.findall(r"#define KG_ROW_BYTES_(\w+) (\d+)", src)}
    assert got == rf["row_bytes"] == {"existing_node": 212, "node_claim": 400, "instance_type": 336}
    assert rf["peak"]["hbm_GBps"] == 8000.0 and rf["bound"] == "hbm"
