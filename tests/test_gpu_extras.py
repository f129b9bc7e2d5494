"""Rows SURVEY.md 8(f) N3/N4 on the GPU: parameter-space noise, demo pool from an archive, pinned demo
rows in the replay ring, checkpoint round trip."""
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _agent(**kw):
    from sam_tf_amd.hps import make_hps, SyntheticEnv
    from sam_tf_amd.discriminator import Discriminator
    from sam_tf_amd.agent import SAMAgent
    from sam_tf_amd.comm import TorchComm
    comm = TorchComm('cuda')
    env = SyntheticEnv(6, 3, 1.0)
    hps = make_hps(batch_size=16, mem_size=256, **kw)
    d = Discriminator('d', env, hps, comm)
    mu = SAMAgent('mu', comm, env, hps, d)
    mu.initialize_target_nets()
    return mu, d, comm


def test_perturb_params_statistics_and_frozen_layernorm():
    """sam_agent.py:40-58: Gaussian noise of the requested std on kernels/biases, LayerNorm variables
    copied; the draw is a function of (seed, counter) only.  TF's tf.random_normal stream itself cannot
    be reproduced (parity unpinned at the RNG), so the check is distributional."""
    from sam_tf_amd import ops
    d = ops.mlp_desc(40, 256, 256, 8, True)
    n = ops.param_count(d)
    p = torch.randn(n, device='cuda')
    out = torch.empty_like(p)
    ops.perturb_params(d, p, out, 0.2, 123, 7)
    diff = (out - p).cpu().numpy()
    lo0 = 40 * 256 + 256; hi0 = lo0 + 512
    lo1 = hi0 + 256 * 256 + 256; hi1 = lo1 + 512
    frozen = np.zeros(n, bool); frozen[lo0:hi0] = True; frozen[lo1:hi1] = True
    assert np.all(diff[frozen] == 0.0)
    z = diff[~frozen] / 0.2
    assert abs(z.mean()) < 0.02 and abs(z.std() - 1.0) < 0.02
    assert abs(np.mean(z ** 3)) < 0.05 and abs(np.mean(z ** 4) - 3.0) < 0.15
    out2 = torch.empty_like(p)
    ops.perturb_params(d, p, out2, 0.2, 123, 7)
    assert torch.equal(out, out2)
    ops.perturb_params(d, p, out2, 0.2, 123, 8)
    assert not torch.equal(out, out2)


def test_adaptive_param_noise_loop():
    """predict / adapt_param_noise / reset_noise (sam_agent.py:592-619,727-749)."""
    from oracle import nets_oracle as NO
    np.random.seed(0); random.seed(0)
    mu, d, comm = _agent(noise_type='adaptive-param_0.2, normal_0.1', with_layernorm=True)
    assert mu.param_noise is not None and mu.ac_noise is not None and mu.pn_cur_std == 0.2
    rs = np.random.RandomState(1)
    for t in range(40):
        mu.store_transition(rs.randn(6).astype(np.float32), rs.uniform(-1, 1, 3).astype(np.float32), 0.5,
                            rs.randn(6).astype(np.float32), 0.0)
    ob = rs.randn(6).astype(np.float32)
    a_clean, q_clean = mu.predict(ob, apply_noise=False, compute_q=True)
    mu.ac_noise = None
    a_pert, q_pert = mu.predict(ob, apply_noise=True, compute_q=True)
    assert not np.allclose(a_clean, a_pert) and q_pert == q_clean   # Q stays that of the clean actor
    # the distance the adaption sees = rms difference of the two actors' actions on a sampled batch
    np.random.seed(5)
    st = np.random.get_state()
    mu.adapt_param_noise(comm)
    np.random.set_state(st)
    idx = np.random.randint(0, mu.replay_buffer.num_entries, size=16)
    obs0 = mu.replay_buffer.batchify(idx)['obs0']
    m, s = mu.obs_rms.mean.cpu().numpy(), mu.obs_rms.std.cpu().numpy()
    da = NO.MLP(6, 64, 64, 3, True)
    dc = NO.MLP(9, 64, 64, 1, True)
    a0, _ = NO.act_q(mu.actor.get_flat(), da, mu.critic.get_flat(), dc, obs0, m, s, 1.0)
    a1, _ = NO.act_q(mu.apnp_actor.get_flat(), da, mu.critic.get_flat(), dc, obs0, m, s, 1.0)
    exp = float(np.sqrt(np.mean((a0 - a1) ** 2)))
    assert abs(mu.pn_dist - exp) <= 1e-5 * max(exp, 1e-3)
    assert mu.pn_cur_std == pytest.approx(0.2 * 1.01 if exp < 0.2 else 0.2 / 1.01)
    before = mu.pnp_actor.get_flat().copy()
    mu.reset_noise()
    assert not np.array_equal(before, mu.pnp_actor.get_flat())


def test_checkpoint_roundtrip_and_demo_rows(tmp_path):
    from sam_tf_amd import checkpoint
    from sam_tf_amd.demo_dataset import DemoDataset
    np.random.seed(0); random.seed(0)
    # an archive in the reference's format (rectangular: loads without pickle)
    rs = np.random.RandomState(2)
    T, ntraj = 12, 3
    blob = dict(obs0=rs.randn(ntraj, T, 6), acs=rs.uniform(-1, 1, (ntraj, T, 3)), env_rews=rs.rand(ntraj, T),
                dones1=np.tile((np.arange(T) == T - 1).astype(float), (ntraj, 1)), obs1=rs.randn(ntraj, T, 6),
                ep_lens=np.full(ntraj, T), ep_env_rets=rs.rand(ntraj))
    path = str(tmp_path / 'demos.npz')
    np.savez(path, **blob)
    ds = DemoDataset(path, size=2, full=True)
    mu, d, comm = _agent(rmsify_rets=True, enable_popart=True)
    # demos pinned at the bottom of the ring (memory.py:186-198, sam.py:246)
    mu.replay_buffer.add_demo_transitions_to_mem(ds)
    assert mu.replay_buffer.num_demos == 2 * T == mu.replay_buffer.num_entries
    got = mu.replay_buffer.batchify(np.arange(2 * T))
    assert np.array_equal(got['obs0'], ds.obs0.astype(np.float32))
    pool = ds.pool('cuda')
    o, a = pool.get_next_pair_batch(8)
    ho, ha = ds.get_next_pair_batch(8)
    assert np.array_equal(o, ho.astype(np.float32)) and np.array_equal(a, ha.astype(np.float32))
    # a few updates, save, scramble, load: identical state
    for _ in range(3):
        mu.train(); mu.update_target_net()
    ck = str(tmp_path / 'ck')   # no '.npz' suffix: save and load must agree on the file name
    checkpoint.save_learner(ck, mu, d)
    L = mu.L
    keep = [t.clone() for t in (L.params, L.targets, L.m, L.v, L.steps, mu.obs_rms.sums, mu.obs_rms.mean,
                                d.obs_rms.sums, mu.ret_rms.sums, mu.ret_rms.std)]
    for t in (L.params, L.targets, L.m, L.v, mu.obs_rms.sums, d.obs_rms.sums, mu.ret_rms.sums):
        t.add_(1.0)
    L.steps.zero_()
    checkpoint.load_learner(ck, mu, d)
    for a_, b_ in zip(keep, (L.params, L.targets, L.m, L.v, L.steps, mu.obs_rms.sums, mu.obs_rms.mean,
                             d.obs_rms.sums, mu.ret_rms.sums, mu.ret_rms.std)):
        assert torch.equal(a_, b_)
    assert mu.actor_ops['optimizer'].t == 3


def test_layerwise_step_call_captures_on_the_origin_stream_and_serial_on_a_fork():
    """samx_step_grads[_part] on the layer-wise path forks two role streams from the stream it is handed.  Under
    stream capture that stream must be the capture's ORIGIN stream: a fork of a forked stream crashes
    hipStreamEndCapture in this HIP runtime — with torch streams alone, no libsamx involved
    (profiles/r03_nested_fork_capture.txt; include/samx.h).  The two supported forms, each in a
    child process (a regression here is a host crash, which must not take the test session with it)."""
    import os
    import subprocess
    import sys
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools', 'exp',
                        'nested_fork_capture.py')
    for variant, env_extra in (('samx-on-origin', {}), ('samx-on-fork-serial', {'SAMX_SERIAL_ROLES': '1'})):
        r = subprocess.run([sys.executable, tool, variant], env=dict(os.environ, **env_extra), capture_output=True,
                           text=True, timeout=240)
        assert r.returncode == 0 and r.stdout.strip().endswith('ok'), (variant, r.returncode, r.stdout[-300:],
                                                                       r.stderr[-300:])


@pytest.mark.parametrize('rows,row_len,stride', [(1, 26, 26), (3, 5, 7), (4, 6, 8), (64, 515, 1281), (0, 4, 4), (2, 0, 3)])
def test_copy_words_rows_from_pinned_staging(rows, row_len, stride):
    """samx_copy_words: the first row_len words of each row, pinned host -> device by a kernel; the rest of a row and
    the rows behind are not touched (odd strides take the word-by-word path, even ones the 16-byte one)."""
    from sam_tf_amd import ops
    n = max(rows, 1) * stride + 4
    src = torch.arange(1, n + 1, dtype=torch.int64).pin_memory()
    dst = torch.full((n,), -5, dtype=torch.int64, device='cuda')
    ops.copy_words(src, dst, rows, row_len, stride)
    torch.cuda.synchronize()
    want = torch.full((n,), -5, dtype=torch.int64)
    for r in range(rows):
        want[r * stride:r * stride + row_len] = src[r * stride:r * stride + row_len]
    assert torch.equal(dst.cpu(), want)
    dev_src = src.cuda()
    dst2 = torch.full((n,), -5, dtype=torch.int64, device='cuda')
    ops.copy_words(dev_src, dst2, rows, row_len, stride)   # device -> device too
    assert torch.equal(dst2.cpu(), want)
