"""Hyper-parameters and the tiny env surface the hot path reads.

Field names follow helpers/argparsers.py:6-50 (`disambiguate`: hid_widths / ent_reg_scale with the
actorcritic_/d_ prefix stripped); defaults are the canonical launcher
(launchers/imitation/mujoco_sam_imitate.sh:16-64) except that with_layernorm, rmsify_rets,
enable_popart and n_step_returns default to off here (the BASELINE.json configs); all four are built.
with_layernorm switches the step onto the layer-wise path; the other three run on the whole-network
path for 64x64 nets.
"""
from types import SimpleNamespace

import numpy as np

DEFAULTS = dict(
    hid_widths=(64, 64), hid_nonlin='leaky_relu', hid_w_init='he_normal', with_layernorm=False,
    ac_branch_in=1, wd_scale=1e-3, gamma=0.99, polyak=0.01, actor_lr=1e-4, critic_lr=1e-3,
    d_lr=3e-4, clip_norm=5.0, batch_size=32, mem_size=100000, prioritized_replay=False, alpha=0.3,
    beta=1.0, ranked=False, unreal=False, rmsify_obs=True, rmsify_rets=False, enable_popart=False,
    n_step_returns=False, n=96, td_loss_1_scale=1.0, td_loss_n_scale=1.0, q_actor_loss_scale=1.0,
    d_actor_loss_scale=0.0, reward_scale=1.0, noise_type='none', ent_reg_scale=0.0,
    label_smoothing=True, one_sided_label_smoothing=True, non_satur_grad=False,
    from_raw_pixels=False, seed=0,
    # build-side switch: the reference's agent graph instantiates its OWN copy of the reward net
    # (variables under `mu/sr/...`, sam_agent.py:88,192 with networks.py:229-230) which is never
    # trained; False reproduces that, True ties it to the discriminator's weights.
    tie_actor_reward_net=False,
)


def make_hps(**kw):
    unknown = set(kw) - set(DEFAULTS)
    if unknown:
        raise TypeError('unknown hyper-parameters: %s' % sorted(unknown))
    h = dict(DEFAULTS)
    h.update(kw)
    h['hid_widths'] = tuple(int(x) for x in h['hid_widths'])
    return SimpleNamespace(**h)


def check_supported(hps):
    if len(hps.hid_widths) != 2:
        raise NotImplementedError('two hidden layers only (all reference launchers use two)')
    if hps.hid_nonlin != 'leaky_relu':
        raise NotImplementedError('leaky_relu only (the reference launchers use nothing else)')
    if hps.ac_branch_in != 1:
        raise NotImplementedError('ac_branch_in=1 only (launcher default)')
    for k in ('ranked', 'unreal', 'from_raw_pixels'):
        if getattr(hps, k):
            raise NotImplementedError('%s is not built (SURVEY.md §2: out of scope)' % k)
    if hps.n_step_returns and hps.prioritized_replay:
        # PrioritizedRB.n_step_lookahead_sample returns ONE dict (memory.py:323-335) but SAMAgent.train
        # unpacks two (sam_agent.py:634-638): the reference raises on this combination
        raise NotImplementedError('n_step_returns with prioritized_replay fails in the reference itself '
                                  '(memory.py:323-335 vs sam_agent.py:634-638); not built')
    if hps.n_step_returns and not hps.n > 1:
        raise ValueError('n_step_returns needs n > 1 (sam_agent.py:85)')


class Box(object):
    def __init__(self, dim, high=np.inf):
        self.shape = (int(dim),)
        self.high = np.full(self.shape, high, dtype=np.float32)
        self.low = -self.high


class SyntheticEnv(object):
    """Shape-only stand-in for a gym env: what SAMAgent/Discriminator read from `env`
    (sam_agent.py:73-97, discriminator.py:28-46)."""

    def __init__(self, ob_dim, ac_dim, ac_high=1.0, env_id='Synthetic-v0'):
        self.observation_space = Box(ob_dim)
        self.action_space = Box(ac_dim, ac_high)
        self.spec = SimpleNamespace(id=env_id)
