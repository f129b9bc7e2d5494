"""samx_d_input_stats_static — the entry point that allocates nothing (its per-group float64 partials live in
a file-scope device array of libsamx) — against samx_d_input_stats with a caller workspace: same bits.  Two
calls in a row on the shared scratch, the second with a smaller input; then the size limit."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _case(rs, ring, ob, B, start):
    store = torch.as_tensor(rs.randn(ring, ob).astype(np.float32)).cuda()
    pool = torch.as_tensor((rs.randn(B + 50, ob) + 0.5).astype(np.float32)).cuda()
    idx = torch.as_tensor(rs.randint(0, ring, size=B)).cuda()
    e_start = torch.tensor([37], dtype=torch.int64, device='cuda')
    return store, pool, idx, e_start, start


def _run(fn, case, B, sums0):
    store, pool, idx, e_start, start = case
    ob = store.shape[1]
    add = torch.zeros(2 * ob + 1, dtype=torch.float64, device='cuda')
    sums = torch.as_tensor(sums0.copy()).cuda()
    mean = torch.zeros(ob, device='cuda'); std = torch.zeros(ob, device='cuda')
    fn(store, start, idx, pool, e_start, B, add, commit=True, sums=sums, mean=mean, std=std)
    torch.cuda.synchronize()
    return [t.cpu().numpy() for t in (add, sums, mean, std)]


def test_static_scratch_entry_point_matches_workspace_one():
    from sam_tf_amd import ops
    from sam_tf_amd._lib import SamxError, lib
    from oracle import nets_oracle as NO
    rs = np.random.RandomState(5)
    # both sizes take the two-stage form (2*B*ob > 16384), i.e. the one that uses the scratch:
    # Humanoid-shaped first (128 groups x 376 columns), then a smaller one (32 groups x 100 columns)
    sizes = [(4096, 376, 1024, 123), (999, 100, 256, 7)]
    assert lib().samx_d_input_stats_static_max_ob() == 512
    for ring, ob, B, start in sizes:
        assert 2 * B * ob > 16384
        case = _case(rs, ring, ob, B, start)
        sums0 = NO.rms_init(ob)
        want = _run(ops.d_input_stats, case, B, sums0)
        got = _run(ops.d_input_stats_static, case, B, sums0)
        for w, g in zip(want, got):
            assert np.array_equal(w, g)   # float64 sums and float32 mean/std: bit for bit
        # and both against the CPU oracle on the same rows (1e-12: float64 sums in another order)
        store, pool, idx, e_start, _ = case
        rows = np.concatenate([store.cpu().numpy()[(start + idx.cpu().numpy()) % ring],
                               pool.cpu().numpy()[37:37 + B]])
        np.testing.assert_allclose(got[1], NO.rms_update(sums0, rows, ob), rtol=1e-12)
    # a small input (one-workgroup form, no scratch) goes through as well
    case = _case(rs, 300, 11, 64, 5)
    for w, g in zip(_run(ops.d_input_stats, case, 64, NO.rms_init(11)),
                    _run(ops.d_input_stats_static, case, 64, NO.rms_init(11))):
        assert np.array_equal(w, g)
    # beyond the static scratch: a clear error, nothing launched
    case = _case(rs, 64, 600, 32, 0)
    with pytest.raises(SamxError, match='exceeds the static scratch'):
        _run(ops.d_input_stats_static, case, 32, NO.rms_init(600))
