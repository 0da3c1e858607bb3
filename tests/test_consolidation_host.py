"""Consolidation through the TEST-ONLY emulation build (host compile + simulation batching + decision logic on CPU)."""
import os

import pytest

from karpenter_amd import lib
from tests.cons_common import check_consolidation
from tests.helpers import EMU_PATH

pytestmark = pytest.mark.skipif(not os.path.exists(EMU_PATH), reason="emulation build missing: run __graft_entry__.build()")


@pytest.fixture(scope="module")
def L():
    l = lib.Library(EMU_PATH)
    yield l
    l.close()


@pytest.mark.parametrize("name,n,s2s", [("dense", 120, False), ("dense", 120, True), ("sparse", 60, False), ("dense_pending", 80, True),
                                         ("uninitialized", 60, False), ("topology", 80, True), ("host_ports", 80, False), ("minvalues_tight", 80, True), ("minvalues_trunc_pending", 60, False), ("volumes", 90, True), ("pending_ties", 70, False)])
def test_consolidation(L, name, n, s2s):
    seen = check_consolidation(L, name, n, s2s)
    if name == "dense":
        assert seen == {0, 1, 2}  # no-op, delete and replace all occur


def test_candidates_order(L):
    from tests.cons_common import check_candidates_order
    check_candidates_order(L)


def test_validate_command(L):
    from tests.cons_common import check_validate
    check_validate(L)


def test_multi_node_more_than_100_candidates(L):
    from tests.cons_common import check_multi_over_100
    check_multi_over_100(L)


@pytest.mark.parametrize("name", ["dense", "topology"])
def test_rounds_reuse_the_prepared_snapshot(L, name):
    from tests.cons_common import check_plan_reuse
    check_plan_reuse(L, name)


@pytest.mark.parametrize("name", ["dense", "dense_pending"])
def test_device_group_in_one_process(name):
    """(emulation: every 'device' is the host; three contexts with a worker thread each)"""
    from tests.group_common import check_group
    assert check_group(EMU_PATH, [[0], [0, 0, 0]], name) == [1, 3]
