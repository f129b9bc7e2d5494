"""Sweep of the reference-shaped calls (sam.py's use of mu / d) over network shapes and options: looks for
workspace / shape errors outside the configurations the parity tests pin.  Prints one line per case."""
import os
import random
import sys
import traceback

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sam_tf_amd.agent import SAMAgent  # noqa: E402
from sam_tf_amd.comm import TorchComm  # noqa: E402
from sam_tf_amd.discriminator import Discriminator  # noqa: E402
from sam_tf_amd.hps import make_hps, SyntheticEnv  # noqa: E402


TWIN = False


class TwinComm(object):
    def Get_rank(self):
        return 0

    def Get_size(self):
        return 2

    def Barrier(self):
        pass

    def allreduce_(self, t):
        return t.mul_(2)

    def bcast_(self, t, root=0):
        return t

    def Allreduce(self, sendbuf, recvbuf, op='sum'):
        recvbuf[...] = 2 * np.asarray(sendbuf)

    def Bcast(self, buf, root=0):
        pass


def run(ob, ac, hid, B, per, ln, ret, nstep, d_scale, noise):
    np.random.seed(0); random.seed(0)
    comm = TwinComm() if TWIN else TorchComm('cuda')
    env = SyntheticEnv(ob, ac, 1.0)
    hps = make_hps(batch_size=B, mem_size=4 * B + 64, prioritized_replay=per, noise_type=noise,
                   d_actor_loss_scale=d_scale, hid_widths=hid, with_layernorm=ln, rmsify_rets=ret,
                   enable_popart=ret, n_step_returns=nstep > 0, n=nstep or 96)
    d = Discriminator('d', env, hps, comm)
    mu = SAMAgent('mu', comm, env, hps, d)
    mu.initialize_target_nets(); mu.sync_from_root(); d.sync_from_root()
    rs = np.random.RandomState(1)
    o = rs.randn(ob).astype(np.float32)
    for t in range(2 * B + 8):
        a, q = mu.predict(o, apply_noise=True, compute_q=True)
        r = d.get_reward(o, a)
        o1 = rs.randn(ob).astype(np.float32)
        mu.store_transition(o, a, r, o1, float(t % 13 == 12))
        o = o1
    xp = mu.replay_buffer.sample_uniform(B) if per else mu.replay_buffer.sample(B)
    feeds = {d.p_obs: xp['obs0'], d.p_acs: xp['acs'], d.e_obs: rs.randn(B, ob), d.e_acs: rs.rand(B, ac)}
    d.obs_rms.update(np.concatenate((feeds[d.p_obs], feeds[d.e_obs])), comm)
    g = d.get_grads(feeds); l = d.get_losses(feeds); d.optimize(feeds)
    out = mu.train()
    mu.update_target_net()
    if mu.param_noise is not None:
        mu.adapt_param_noise(comm)
    mu.reset_noise()
    vals = [np.asarray(g), np.asarray(l), out['actor_grads'], out['critic_grads'],
            np.asarray(out['actor_losses']), np.asarray(out['critic_losses'])]
    assert all(np.isfinite(v).all() for v in vals)
    torch.cuda.synchronize()


def main(twin=False):
    global TWIN
    TWIN = twin
    cases = []
    for (ob, ac, hid, B) in [(4, 1, (64, 64), 32), (11, 3, (64, 64), 256), (17, 6, (64, 64), 256),
                             (376, 17, (256, 256), 64), (376, 17, (400, 300), 128), (5, 2, (33, 70), 37),
                             (50, 10, (64, 64), 32), (3, 1, (8, 8), 1), (111, 8, (256, 256), 16), (17, 6, (64, 64), 512),
                             (11, 3, (64, 64), 2048)]:
        for per in (False, True):
            for ln, ret, nstep in ((False, False, 0), (True, True, 0 if per else 5), (False, True, 0 if per else 3)):
                cases.append((ob, ac, hid, B, per, ln, ret, nstep, 0.0 if ln else 0.5,
                              'adaptive-param_0.2' if not per else 'ou_0.2'))
    bad = 0
    for c in cases:
        try:
            run(*c)
            print('ok  ', c, flush=True)
        except Exception as e:  # noqa: BLE001
            bad += 1
            print('FAIL', c, repr(e)[:300], flush=True)
            traceback.print_exc(limit=3)
    print('%d cases, %d failed' % (len(cases), bad), flush=True)
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main(len(sys.argv) > 1 and sys.argv[1] == 'twin'))
