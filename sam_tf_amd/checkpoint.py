"""Checkpoint save / restore of the learner (helpers/tf_util.py:112-134, sam.py:228-235,306-310).

The reference's tf.train.Saver stores every global variable: online and target weights, Adam slots and
beta powers, the running-statistics sums.  Replay contents and RNG state are NOT saved there either.
Here that is the flat learner vectors + the float64 sums, one .npz (no pickled objects)."""
import numpy as np
import torch

from . import ops

FORMAT = 1


def save_learner(path, mu, d):
    L = mu.L
    blob = dict(format=np.int64(FORMAT), params=L.params.cpu().numpy(), targets=L.targets.cpu().numpy(),
                adam_m=L.m.cpu().numpy(), adam_v=L.v.cpu().numpy(), steps=L.steps.cpu().numpy(),
                layout=np.array([L.Pa, L.Pc, L.Pd], dtype=np.int64))
    if mu.hps.rmsify_obs:
        blob['mu_obs_sums'] = mu.obs_rms.sums.cpu().numpy()
        blob['d_obs_sums'] = d.obs_rms.sums.cpu().numpy()
    if mu.hps.rmsify_rets:
        blob['ret_sums'] = mu.ret_rms.sums.cpu().numpy()
    if mu.d_params is not None and not mu.hps.tie_actor_reward_net:  # the agent's own frozen reward net
        blob['actor_reward_params'] = mu.d_params.cpu().numpy()
    # through an open handle: np.savez(path) would append '.npz' to a path without that suffix and
    # load_learner(path) with the same string would then miss the file
    with open(path, 'wb') as f:
        np.savez(f, **blob)


def load_learner(path, mu, d):
    z = np.load(path, allow_pickle=False)
    L = mu.L
    assert int(z['format']) == FORMAT, "unknown checkpoint format"
    assert tuple(z['layout']) == (L.Pa, L.Pc, L.Pd), "checkpoint was written for other network shapes"
    dev = L.device
    for name, t in (('params', L.params), ('targets', L.targets), ('adam_m', L.m), ('adam_v', L.v),
                    ('steps', L.steps)):
        t.copy_(torch.as_tensor(z[name]).to(dev))
    pairs = []
    if mu.hps.rmsify_obs:
        pairs += [(mu.obs_rms, 'mu_obs_sums'), (d.obs_rms, 'd_obs_sums')]
    if mu.hps.rmsify_rets:
        pairs += [(mu.ret_rms, 'ret_sums')]
    for r, key in pairs:
        r.sums.copy_(torch.as_tensor(z[key]).to(dev))
        ops.rms_commit(r.sums, None, r.mean, r.std)   # refresh the derived float32 mean / std
    if 'actor_reward_params' in z.files and mu.d_params is not None:
        mu.d_params.copy_(torch.as_tensor(z['actor_reward_params']).to(dev))
