"""GPU parity: kg_consolidate_single / _multi / _select through the C ABI of libkarpgpu.so vs the CPU oracle."""
import pytest

from karpenter_amd import lib
from tests.cons_common import check_consolidation

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L():
    l = lib.Library()
    assert l.backend() == "hip-gfx950"
    yield l
    l.close()


@pytest.mark.parametrize("name,n,s2s", [("dense", 150, False), ("dense", 150, True), ("sparse", 100, False), ("dense_pending", 120, True),
                                         ("uninitialized", 90, False), ("dense", 400, True), ("topology", 150, True), ("topology", 400, False), ("host_ports", 150, False), ("minvalues_loose", 120, True), ("minvalues_tight", 150, False), ("minvalues_trunc_pending", 120, True), ("volumes", 150, True), ("pending_ties", 120, False)])
def test_consolidation(L, name, n, s2s):
    check_consolidation(L, name, n, s2s)


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


def test_device_group_in_one_process():
    """Every visible device (1 on the test box), then the same GPU three times: three contexts, three streams, three worker
    threads whose simulations overlap on the one device."""
    from karpenter_amd import lib as _lib
    from tests.group_common import check_group
    sizes = check_group(_lib.DEFAULT_PATH, [None, [0, 0, 0]])
    assert sizes[0] >= 1 and sizes[1] == 3
