"""CPU-only checks of the C-ABI library: it loads, exports every symbol the header declares, its host-side
topology functions match the reference's golden vectors, and the product path fails loudly without a GPU."""
import os
import re
from types import SimpleNamespace

import numpy as np
import pytest

from xanthos_amd import _hip
from xanthos_amd.routing import mrtm

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))


def _header_functions():
    src = open(os.path.join(ROOT, 'include', 'xanthos_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(xh_[a-z0-9_]+)\s*\(', src)))


def test_library_exports_every_declared_symbol():
    lib = _hip.lib()
    names = _header_functions()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), 'libxanthos_hip.so does not export ' + n
    assert set(names) == set(_hip.SIGNATURES), set(names) ^ set(_hip.SIGNATURES)
    assert lib.xh_abi_version() == _hip.ABI_VERSION >= 3


@pytest.mark.parametrize('tag', ['rand', 'tree'])
def test_host_topology_matches_reference_golden(golden, tag):
    g = golden('topo')
    st = SimpleNamespace(ngridrow=int(g['nrow']), ngridcol=int(g['ncol']))
    ds = mrtm.downstream(g[tag + '_coords'], g[tag + '_flowdir'], st)
    assert np.array_equal(ds, g[tag + '_dsid'])
    up = mrtm.upstream(g[tag + '_coords'], ds, st)
    assert np.array_equal(up, g[tag + '_upid'])
    um = mrtm.upstream_genmatrix(up)
    assert np.array_equal(um.indptr, g[tag + '_um_indptr'])
    assert np.array_equal(um.indices, g[tag + '_um_indices'])
    assert np.array_equal(um.sign, g[tag + '_um_data'])
    csr = um.tocsr()
    assert csr.shape == (len(ds), len(ds)) and csr.nnz == len(um.indices)


def test_topology_rejects_bad_grid():
    coords = np.array([[1, 0, 0, 5, 5]], dtype=float)
    with pytest.raises(_hip.HipError):
        mrtm.downstream(coords, np.array([1.0]), SimpleNamespace(ngridrow=2, ngridcol=2))


def test_no_cpu_fallback_without_device():
    """Without a GPU the compute entry points must raise, never silently compute on the host."""
    if _hip.device_count() > 0:
        pytest.skip('a GPU is present')
    with pytest.raises(_hip.HipUnavailable):
        _hip.Context(0)
    from xanthos_amd.runoff import abcd
    z = np.zeros((4, 36))
    with pytest.raises(_hip.HipUnavailable):
        abcd.abcd_execute(1, np.ones(4, dtype=int), z, z, None, np.ones((1, 5)) * 0.5, 36, 30)


def test_product_never_imports_oracle():
    bad = []
    for base, _, files in os.walk(os.path.join(ROOT, 'xanthos_amd')):
        for f in files:
            if f.endswith('.py') and re.search(r'^\s*(from|import)\s+oracle\b', open(os.path.join(base, f)).read(), re.M):
                bad.append(f)
    assert not bad, bad
