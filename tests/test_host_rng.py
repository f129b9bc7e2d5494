"""samx_host_mt19937_randint (HOST code of libsamx.so) against numpy's legacy global generator: the replay index
draws of a multi-row launch (memory.py:90 per batch) must consume np.random exactly as the reference's calls do."""
import numpy as np
import pytest


def _lib_ready():
    from sam_tf_amd import ops
    return ops.host_rng_ready()


def test_selfcheck_leaves_the_generator_where_it_was():
    import sam_tf_amd.ops as ops
    np.random.seed(123)
    np.random.randn(3)                      # (a cached gaussian is part of the state that must survive)
    before = np.random.get_state()
    ops._HOST_RNG = None
    assert ops.host_rng_ready()
    after = np.random.get_state()
    assert before[0] == after[0] and np.array_equal(before[1], after[1]) and before[2:] == after[2:]


@pytest.mark.parametrize('high', [1, 2, 7, 4097, 100000, 1000000, 1 << 20, (1 << 20) + 1, (1 << 32) - 1, 1 << 32])
def test_draw_for_draw_with_numpy_calls_in_between(high):
    from sam_tf_amd import ops
    assert _lib_ready()
    for seed, burn in ((0, 0), (5, 623), (9, 624), (77, 1000)):
        ref, got = [], []
        for mine in (False, True):
            np.random.seed(seed)
            np.random.randint(0, 1 << 30, size=burn)     # move the position inside the 624-word block
            out = ref if not mine else got
            for rows, row_len, width in ((4, 64, 64), (1, 1, 3), (20, 512, 1281), (3, 700, 700), (0, 5, 5)):
                if mine:
                    w = np.full((max(rows, 1), width), -7, dtype=np.int64)
                    ops.host_randint_rows(w, rows, row_len, high)
                    assert bool((w[:, row_len:] == -7).all()) and bool((w[rows:] == -7).all())
                    out.append(w[:rows, :row_len].copy())
                else:
                    out.append(np.random.randint(low=0, high=high, size=(rows, row_len)))
                out.append(np.random.randint(low=0, high=1 << 20, size=5))   # numpy's own calls in between
                out.append(np.random.uniform(size=2))
            out.append(np.random.get_state()[1].copy())
        for a, b in zip(ref, got):
            assert a.dtype == b.dtype and np.array_equal(a, b)


def test_bad_arguments_are_refused():
    from sam_tf_amd import _lib, ops
    assert _lib_ready()
    w = np.zeros((2, 8), dtype=np.int64)
    with pytest.raises(_lib.SamxError):
        ops.host_randint_rows(w, 2, 8, 0)
    with pytest.raises(_lib.SamxError):
        ops.host_randint_rows(w, 2, 8, (1 << 32) + 1)
