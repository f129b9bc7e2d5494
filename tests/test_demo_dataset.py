"""On-disk expert format and batching rule (imitation_algorithms/demo_dataset.py:6-177) — host logic,
no GPU.  The archive here is written by this test (ragged trajectories -> numpy object arrays)."""
import numpy as np
import pytest


def _write(path, ragged):
    rs = np.random.RandomState(0)
    lens = [5, 9, 7] if ragged else [6, 6, 6]
    tr = dict(obs0=[], acs=[], env_rews=[], dones1=[], obs1=[])
    for T in lens:
        tr['obs0'].append(rs.randn(T, 4)); tr['obs1'].append(rs.randn(T, 4))
        tr['acs'].append(rs.randn(T, 2)); tr['env_rews'].append(rs.rand(T))
        d = np.zeros(T); d[-1] = 1; tr['dones1'].append(d)
    blob = {}
    for k, v in tr.items():
        if ragged:
            a = np.empty(len(v), dtype=object)
            for i, x in enumerate(v):
                a[i] = x
            blob[k] = a
        else:
            blob[k] = np.stack(v)
    blob['ep_lens'] = np.array(lens)
    blob['ep_env_rets'] = np.array([float(np.sum(r)) for r in tr['env_rews']])
    np.savez(path, **blob)
    return tr, lens


@pytest.mark.parametrize('ragged', [False, True])
def test_demo_dataset_roundtrip(tmp_path, ragged):
    from sam_tf_amd.demo_dataset import DemoDataset
    path = str(tmp_path / 'demos.npz')
    tr, lens = _write(path, ragged)
    if ragged:
        with pytest.raises(ValueError):      # object arrays are not unpickled unless asked
            DemoDataset(path, size=2)
    np.random.seed(3)
    ds = DemoDataset(path, size=2, randomize=True, full=True, allow_pickle=ragged)
    n = lens[0] + lens[1]
    assert ds.obs0.shape == (n, 4) and ds.acs.shape == (n, 2)
    assert ds.env_rews.shape == (n, 1) and ds.dones1.shape == (n, 1)       # scalars become a column
    assert np.array_equal(ds.obs0, np.concatenate(tr['obs0'][:2]))
    assert ds.len_mean == np.mean(lens[:2]) and ds.ret_mean == np.mean(
        [np.sum(tr['env_rews'][0]), np.sum(tr['env_rews'][1])])
    # batching rule: shuffled once at construction, consecutive slices, reshuffle when fewer than
    # batch_size pairs remain (pointer + batch_size >= num_entries, the reference's >=)
    np.random.seed(3)
    idx = np.arange(n); np.random.shuffle(idx)
    o, a = ds.get_next_pair_batch(4)
    assert np.array_equal(o, np.concatenate(tr['obs0'][:2])[idx][:4])
    seen = 4
    while seen + 4 < n:
        ds.get_next_pair_batch(4); seen += 4
    before = ds.pair_dset.obs0.copy()
    ds.get_next_pair_batch(4)   # triggers init_pointer -> reshuffle
    assert ds.pair_dset.pointer == 4 and not np.array_equal(before, ds.pair_dset.obs0)
    o_all, _ = ds.get_next_pair_batch(-1)
    assert o_all.shape == (n, 4)
