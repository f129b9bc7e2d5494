"""CPU-side checks of the drop-in boundary: libsamx.so loads and exports exactly the entry points
include/samx.h declares (no compute call is made here — there is no GPU in this tier)."""
import ctypes
import os
import re

import pytest

from sam_tf_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'samx.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(samx_[a-z0-9_]+)\s*\(', text)))


def test_header_declares_entry_points():
    syms = declared_symbols()
    assert 'samx_disc_grad' in syms and 'samx_per_sample' in syms and len(syms) >= 25


def test_library_exports_every_declared_symbol():
    assert os.path.exists(_lib.SO_PATH), "build libsamx.so first: python -m sam_tf_amd.build"
    so = ctypes.CDLL(_lib.SO_PATH)
    for s in declared_symbols():
        assert hasattr(so, s), "libsamx.so does not export %s" % s


def test_binding_table_matches_header():
    assert sorted(_lib.SIGNATURES) == declared_symbols()


def test_host_only_calls():
    L = _lib.lib()
    assert L.samx_version() >= 100
    d = _lib.MlpDesc(11, 64, 64, 3, 0)
    assert L.samx_mlp_param_count(ctypes.byref(d)) == 11 * 64 + 64 + 64 * 64 + 64 + 64 * 3 + 3
    d = _lib.MlpDesc(14, 64, 64, 1, 1)
    assert L.samx_mlp_param_count(ctypes.byref(d)) == 5185 + 256
    assert L.samx_adam_partials() > 0


def test_leaf_values_are_libm_pow():
    """The PER leaf transform runs on the host with libm pow: bit-equal to Python's float pow."""
    import numpy as np
    from sam_tf_amd import ops
    rs = np.random.RandomState(0)
    p = np.abs(rs.randn(4096)) + 1e-6
    idx = rs.randint(0, 100, size=4096)
    leaf, mp = ops.per_leaf_values_host(idx, p, 0.3, 10, 0.1, 1.0)
    exp = np.array([float(x + (0.1 if i < 10 else 0.0)) ** 0.3 for x, i in zip(p, idx)])
    assert np.array_equal(leaf, exp)
    assert mp == max(1.0, (p + (idx < 10) * 0.1).max())


def test_bad_arguments_are_rejected_without_a_gpu():
    L = _lib.lib()
    rc = L.samx_tree_init(None, None, None, 8, None)
    assert rc < 0 and b'null' in L.samx_last_error()
    with pytest.raises(_lib.SamxError):
        _lib.check(L.samx_tree_reduce(None, 8, 0, 0, 8, None, None))


def test_shipped_library_is_not_a_profiling_build():
    """-DSAMX_TRACE builds (tools/trace_fused.py) add timestamp stores to the fused step kernel and
    export samx_debug_trace; the library in the tree must not be one of them."""
    import ctypes
    from sam_tf_amd import _lib
    L = ctypes.CDLL(_lib.SO_PATH)
    assert not hasattr(L, 'samx_debug_trace')


def test_pow_routine_matches_this_machines_libm_on_the_host():
    """The host build of csrc/pow_bits.h against the machine's libm pow (samx_pow_selfcheck; no GPU involved):
    what the driver asks before it keeps the priority refresh on the device."""
    from sam_tf_amd import _lib
    assert int(_lib.lib().samx_pow_selfcheck(500000)) == 0


def test_committed_pow_tables_are_those_of_this_images_libm(tmp_path):
    """pow_tables.h is generated from the image's libm.so.6 (tools/extract_pow_tables.py); a different glibc in the
    image must show up here, not as a silent fall-back to the host refresh."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libm = '/lib/x86_64-linux-gnu/libm.so.6'
    if not os.path.exists(libm):
        import pytest
        pytest.skip('no libm.so.6 at the usual place')
    out = tmp_path / 'pow_tables.h'
    subprocess.check_call([sys.executable, os.path.join(root, 'tools', 'extract_pow_tables.py'), libm, str(out)])
    strip = lambda s: '\n'.join(l for l in s.splitlines() if not l.startswith('// GENERATED'))
    assert strip(out.read_text()) == strip(open(os.path.join(root, 'sam_tf_amd', 'csrc', 'pow_tables.h')).read())
