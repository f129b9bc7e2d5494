"""CPU oracle (torch autograd, fp64 or fp32) for the network rows of the hot path
(SURVEY.md §8 A1-A12): actor / critic / reward-net graphs, their losses and gradients, the
gradient penalty, TF1 Adam, global-norm clipping, Polyak updates and running mean/std.

TEST INFRASTRUCTURE ONLY.  Imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg; never by sam_tf_amd/.

PARITY UNPINNED at the TensorFlow boundary: the arithmetic of these rows lives in
tensorflow==1.12.0 (requirements.txt:54), which is not in /root/reference, not installed and not
obtainable (no network); the reference holds no fixture or test for any of them (SURVEY.md §4,
§8c).  This file restates the published semantics of the TF ops at the reference's call sites; it is
cross-checked by finite differences (tests/test_oracle_nets.py) but not against TF outputs.

Restated call sites (reference file:line):
  obs normalise + clip .......... sam_agent.py:150-156, discriminator.py:67-72, math_util.py:8-12
  dense / leaky-relu / LayerNorm  networks.py:45-72,345-404 ; network_util.py:5-72
  actor head .................... networks.py:282-302 ; sam_agent.py:177
  TD target ..................... sam_agent.py:196-205
  critic loss / huber / wd ...... sam_agent.py:457-520 ; math_util.py:22-29 ; network_util.py:43-56
  actor loss .................... sam_agent.py:406-433
  D losses, GP, reward .......... discriminator.py:77-190 ; distributions.py:300-328
  flatgrad + clip ............... tf_util.py:137-151
  MpiAdam (mean, clip, Adam) .... mpi_adam.py:18-36 (+ tf.train.AdamOptimizer update rule)
  Polyak ........................ sam_agent.py:20-37
  running mean/std .............. mpi_running_mean_std.py:11-59

Flat parameter layout (= TF variable creation order = flatgrad order, SURVEY.md §7 cheat-sheet):
  W1[in,h1] b1[h1] (beta1[h1] gamma1[h1]) W2[h1,h2] b2[h2] (beta2 gamma2) W3[h2,out] b3[out]
"""
import math
from collections import namedtuple, OrderedDict

import numpy as np
import torch

LEAK = 0.2
LN_EPS = 1e-8
GP_COEFF = 10.0

MLP = namedtuple('MLP', 'in_dim h1 h2 out_dim layernorm')


def param_count(d):
    ln = 2 * (d.h1 + d.h2) if d.layernorm else 0
    return d.in_dim * d.h1 + d.h1 + d.h1 * d.h2 + d.h2 + d.h2 * d.out_dim + d.out_dim + ln


def param_slices(d):
    """OrderedDict name -> (offset, shape) in the flat vector."""
    out, off = OrderedDict(), 0

    def add(name, *shape):
        nonlocal off
        out[name] = (off, shape)
        off += int(np.prod(shape))
    add('W1', d.in_dim, d.h1); add('b1', d.h1)
    if d.layernorm:
        add('beta1', d.h1); add('gamma1', d.h1)
    add('W2', d.h1, d.h2); add('b2', d.h2)
    if d.layernorm:
        add('beta2', d.h2); add('gamma2', d.h2)
    add('W3', d.h2, d.out_dim); add('b3', d.out_dim)
    return out


def unflatten(flat, d):
    return {k: flat[o:o + int(np.prod(s))].reshape(s) for k, (o, s) in param_slices(d).items()}


def init_params(d, rs, kind):
    """Random-init flat params with the reference's initialiser family (networks.py:74-80,369;
    network_util.py:180-195).  Statistical match only (parity tests inject weights anyway)."""
    p = np.zeros(param_count(d), dtype=np.float32)
    sl = param_slices(d)

    def put(name, val):
        o, s = sl[name]
        p[o:o + val.size] = val.astype(np.float32).ravel()

    def he(fan_in, shape):
        std = math.sqrt(1.3 * 2.0 / fan_in)
        x = rs.randn(*shape)
        bad = np.abs(x) > 2
        while bad.any():
            x[bad] = rs.randn(int(bad.sum()))
            bad = np.abs(x) > 2
        return x * std
    for name, fan_in in (('W1', d.in_dim), ('W2', d.h1)):
        shape = sl[name][1]
        if kind == 'critic':  # quirk: critic hidden kernels use the output initialiser
            put(name, rs.uniform(-3e-3, 3e-3, shape))
        else:
            put(name, he(fan_in, shape))
    put('W3', rs.uniform(-3e-3, 3e-3, sl['W3'][1]))
    if d.layernorm:
        put('gamma1', np.ones(d.h1)); put('gamma2', np.ones(d.h2))
    return p


class _Kinks(object):
    """Leaky-ReLU kink bookkeeping for parity tests at size (test infrastructure, like the rest of oracle/).
    A pre-activation within fp32 rounding distance of 0 gets mask 1 or 0.2 depending on the last bits of an
    fp32 sum, which no fp64 restatement can predict: at batch 1024 x 512 units x 6 passes a few such units occur
    in every update.  While `active`, lrelu() numbers its calls, lists the units with |z| < delta as
    (call, row, unit), and takes the OTHER branch for the units named in `flip` — so a test can ask 'is the
    difference to the device result exactly the effect of these undecidable units?'."""
    active = False
    delta = 2e-6
    calls = 0
    near = []
    flip = frozenset()


KINKS = _Kinks


def kinks_begin(flip=(), delta=None):
    """Start bookkeeping for ONE gradient function call (call numbers restart at 0)."""
    KINKS.active, KINKS.calls, KINKS.near, KINKS.flip = True, 0, [], frozenset(flip)
    if delta is not None:
        KINKS.delta = delta


def kinks_end():
    KINKS.active = False
    return list(KINKS.near)


def lrelu(x):
    if not KINKS.active or x.dim() != 2:
        return torch.maximum(x, LEAK * x)
    call = KINKS.calls
    KINKS.calls += 1
    xd = x.detach()
    for r, u in (xd.abs() < KINKS.delta).nonzero().tolist():
        KINKS.near.append((call, r, u))
    pos = xd >= 0
    for c, r, u in KINKS.flip:
        if c == call:
            pos[r, u] = ~pos[r, u]
    return torch.where(pos, x, LEAK * x)


def layer_norm(x, beta, gamma):
    mu = x.mean(dim=-1, keepdim=True)
    var = ((x - mu) ** 2).mean(dim=-1, keepdim=True)  # biased, tf.nn.moments
    return gamma * (x - mu) / torch.sqrt(var + LN_EPS) + beta


def mlp(P, x, d):
    h = lrelu(x @ P['W1'] + P['b1'])
    if d.layernorm:
        h = layer_norm(h, P['beta1'], P['gamma1'])
    h = lrelu(h @ P['W2'] + P['b2'])
    if d.layernorm:
        h = layer_norm(h, P['beta2'], P['gamma2'])
    return h @ P['W3'] + P['b3']


def min_abs_preact(flat, d, x, dtype=torch.float64):
    """Per row, the smallest |pre-activation| over both hidden layers.  Leaky-ReLU has a kink at 0:
    a row whose pre-activation is within rounding distance of it has a precision-dependent gradient
    (mask 1 vs 0.2), so parity tests keep their inputs away from the kink by a stated margin."""
    P = unflatten(_t(flat, dtype), d)
    x = _t(x, dtype)
    z1 = x @ P['W1'] + P['b1']
    h = lrelu(z1)
    if d.layernorm:
        h = layer_norm(h, P['beta1'], P['gamma1'])
    z2 = h @ P['W2'] + P['b2']
    return torch.minimum(z1.abs().min(dim=1).values, z2.abs().min(dim=1).values).numpy()


def rms_mean_std(sums, dim):
    """sums = [sum(dim), sumsq(dim), count] float64 -> (mean f32, std f32)
    (mpi_running_mean_std.py:31-34: the subtraction happens in float32)."""
    s = np.asarray(sums, dtype=np.float64)
    cnt = s[2 * dim]
    mean = (s[:dim] / cnt).astype(np.float32)
    msq = (s[dim:2 * dim] / cnt).astype(np.float32)
    std = np.sqrt(np.maximum(msq - np.square(mean), np.float32(1e-2))).astype(np.float32)
    return mean, std


def rms_init(dim, eps=1e-2):
    s = np.zeros(2 * dim + 1, dtype=np.float64)
    s[dim:] = eps
    return s


def rms_update(sums, x, dim):
    """mpi_running_mean_std.py:47-59 for one worker (the all-reduce adds such vectors)."""
    x = np.asarray(x).astype(np.float64).reshape(-1, dim)
    add = np.concatenate([x.sum(axis=0), np.square(x).sum(axis=0), [float(len(x))]])
    return sums + add


def obz(obs, mean, std, rmsify=True):
    if not rmsify:
        return obs
    return torch.clamp((obs - mean) / std, -5.0, 5.0)


def _t(x, dtype):
    return torch.as_tensor(np.asarray(x), dtype=dtype)


def _params(flat, d, dtype, grad=False):
    f = _t(flat, dtype).clone().requires_grad_(grad)
    return f, unflatten(f, d)


def actor_act(Pa, x, d, high):
    a = high * torch.tanh(mlp(Pa, x, d))
    return torch.clamp(a, -high, high)


def critic_q(Pc, x, ac, d):
    return mlp(Pc, torch.cat([x, ac], dim=-1), d)  # ac_branch_in = 1


def _derms(x, ret):
    """math_util.py:15-19 with the scalar return statistics; ret = (mean, std) or None (rmsify_rets off)."""
    return x if ret is None else x * float(ret[1]) + float(ret[0])


def _rms(x, ret):
    """math_util.py:8-12 (sam_agent.py:161-172: tc1z / tcnz)."""
    return x if ret is None else (x - float(ret[0])) / float(ret[1])


def targ_q(pa_t, pc_t, da, dc, obs1, rews, dones1, td_len, mean, std, gamma, high,
           dtype=torch.float64, rmsify=True, ret=None):
    """sam_agent.py:196-205; ret = (ret_mean, ret_std) de-normalises Q' (lines 199-201)."""
    _, Pa = _params(pa_t, da, dtype)
    _, Pc = _params(pc_t, dc, dtype)
    x1 = obz(_t(obs1, dtype), _t(mean, dtype), _t(std, dtype), rmsify)
    qp = _derms(critic_q(Pc, x1, actor_act(Pa, x1, da, high), dc), ret)
    r, dn, tl = _t(rews, dtype), _t(dones1, dtype), _t(td_len, dtype)
    g = torch.pow(torch.tensor(gamma, dtype=dtype), tl)
    return (r + g * (1.0 - dn) * qp).detach().numpy()


def huber(x):
    return torch.where(x.abs() < 1.0, 0.5 * x * x, x.abs() - 0.5)


def clip_by_global_norm(g, clip_norm):
    n = float(np.sqrt(np.sum(np.square(g.astype(np.float64)))))
    return g * (clip_norm / max(n, clip_norm)), n


def critic_loss_grad(pc, dc, obs0, acs, iws, tc1s, mean, std, wd_scale, td_scale=1.0,
                     dtype=torch.float64, rmsify=True, need_grad=True, ret=None, tcns=None,
                     tdn_scale=1.0):
    """sam_agent.py:457-520.  iws None <=> uniform replay.  Returns losses
    [td_loss_1, wd, total] (+ [td_loss_n] appended when tcns is given), flat grad (unclipped),
    td_errors [B,1] (1-step).  ret = (ret_mean, ret_std): critic_pred is DE-normalised (lines 180-182)
    while the targets are normalised (lines 161-172) — the reference's mixed-scale TD error."""
    f, Pc = _params(pc, dc, dtype, grad=True)
    x0 = obz(_t(obs0, dtype), _t(mean, dtype), _t(std, dtype), rmsify)
    q = _derms(critic_q(Pc, x0, _t(acs, dtype), dc), ret)
    td = q - _rms(_t(tc1s, dtype), ret)
    h = huber(td)
    if iws is not None:
        h = h * _t(iws, dtype)
    td_loss = h.mean() * td_scale
    wd = wd_scale * 0.5 * ((Pc['W1'] ** 2).sum() + (Pc['W2'] ** 2).sum())
    loss = td_loss + wd
    out = [td_loss, wd]
    if tcns is not None:  # lines 483-498
        hn = huber(q - _rms(_t(tcns, dtype), ret))
        if iws is not None:
            hn = hn * _t(iws, dtype)
        tdn_loss = hn.mean() * tdn_scale
        loss = loss + tdn_loss
    g = torch.autograd.grad(loss, f)[0].numpy() if need_grad else None
    losses = [td_loss.item(), wd.item(), loss.item()] + ([tdn_loss.item()] if tcns is not None else [])
    return (np.array(losses), g, td.detach().numpy())


def td_errors(pc, dc, obs0, acs, tc1s, mean, std, dtype=torch.float64, rmsify=True, ret=None):
    _, Pc = _params(pc, dc, dtype)
    x0 = obz(_t(obs0, dtype), _t(mean, dtype), _t(std, dtype), rmsify)
    return (_derms(critic_q(Pc, x0, _t(acs, dtype), dc), ret) - _rms(_t(tc1s, dtype), ret)).detach().numpy()


def popart(w3, b3, old_mean, old_std, new_mean, new_std):
    """sam_agent.py:554-590: rescale of an output layer (w [h2,1], b [1]) when the return statistics
    move from (old_mean, old_std) to (new_mean, new_std)."""
    w3, b3 = np.asarray(w3, dtype=np.float64), np.asarray(b3, dtype=np.float64)
    return w3 * old_std / new_std, (b3 * old_std + old_mean - new_mean) / new_std


def actor_loss_grad(pa, da, pc, dc, pd, dd, obs0, mean, std, high, q_scale, d_scale,
                    dtype=torch.float64, rmsify=True, need_grad=True, ret=None):
    """sam_agent.py:406-433.  pd/dd: the reward net the agent graph evaluates (may be None when
    d_scale == 0).  Returns losses [q_loss, d_loss, total], flat grad wrt actor (unclipped)."""
    f, Pa = _params(pa, da, dtype, grad=True)
    _, Pc = _params(pc, dc, dtype)
    x0 = obz(_t(obs0, dtype), _t(mean, dtype), _t(std, dtype), rmsify)
    a = actor_act(Pa, x0, da, high)
    q_loss = -_derms(critic_q(Pc, x0, a, dc), ret).mean() * q_scale  # sam_agent.py:184-187
    if pd is not None:
        _, Pd = _params(pd, dd, dtype)
        d_loss = -_derms(critic_q(Pd, x0, a, dd), ret).mean() * d_scale  # sam_agent.py:191-194
    else:
        d_loss = torch.zeros((), dtype=dtype)
    loss = q_loss + d_loss
    g = torch.autograd.grad(loss, f)[0].numpy() if need_grad else None
    return np.array([q_loss.item(), d_loss.item(), loss.item()]), g


def act_q(pa, da, pc, dc, obs, mean, std, high, dtype=torch.float64, rmsify=True, ret=None):
    """sam_agent.py:209-212: (actor_pred, critic_pred_w_actor)."""
    _, Pa = _params(pa, da, dtype)
    _, Pc = _params(pc, dc, dtype)
    x0 = obz(_t(obs, dtype), _t(mean, dtype), _t(std, dtype), rmsify)
    a = actor_act(Pa, x0, da, high)
    return a.detach().numpy(), _derms(critic_q(Pc, x0, a, dc), ret).detach().numpy()


def sigmoid_ce(x, z):
    return torch.clamp(x, min=0) - x * z + torch.log1p(torch.exp(-x.abs()))


D_LOSS_NAMES = ['d_policy_loss', 'd_expert_loss', 'd_ent_mean', 'd_ent_loss', 'd_policy_acc',
                'd_expert_acc', 'd_grad_pen', 'd_total_loss']


def disc_loss_grad(pd, dd, p_obs, p_acs, e_obs, e_acs, eps_ob, eps_ac, real_labels, fake_labels,
                   mean, std, ent_reg_scale=0.0, dtype=torch.float64, rmsify=True, need_grad=True):
    """discriminator.py:77-161 with the TF-random tensors (GP eps, smoothed labels) injected.
    Returns the 8 losses in the reference's order and the flat grad (unclipped)."""
    f, Pd = _params(pd, dd, dtype, grad=True)
    m, s = _t(mean, dtype), _t(std, dtype)
    pz, ez = obz(_t(p_obs, dtype), m, s, rmsify), obz(_t(e_obs, dtype), m, s, rmsify)
    pa_, ea_ = _t(p_acs, dtype), _t(e_acs, dtype)
    ps = critic_q(Pd, pz, pa_, dd)
    es = critic_q(Pd, ez, ea_, dd)
    scores = torch.cat([ps, es], dim=0)
    ent_mean = sigmoid_ce(scores, torch.sigmoid(scores)).sum(dim=-1).mean()
    ent_loss = -ent_reg_scale * ent_mean
    p_acc, e_acc = torch.sigmoid(ps).mean(), torch.sigmoid(es).mean()
    p_loss = sigmoid_ce(ps, _t(fake_labels, dtype).reshape(ps.shape)).sum(dim=-1).mean()
    e_loss = sigmoid_ce(es, _t(real_labels, dtype).reshape(es.shape)).sum(dim=-1).mean()
    eo, ea = _t(eps_ob, dtype), _t(eps_ac, dtype)
    oi = (eo * pz + (1.0 - eo) * ez).detach().requires_grad_(True)
    ai = (ea * pa_ + (1.0 - ea) * ea_).detach().requires_grad_(True)
    si = critic_q(Pd, oi, ai, dd)
    go, ga = torch.autograd.grad(si.sum(), [oi, ai], create_graph=need_grad)
    gnorm = torch.sqrt((go ** 2).mean() + (ga ** 2).mean())
    gp = (gnorm - 1.0) ** 2
    loss = p_loss + e_loss + ent_loss + GP_COEFF * gp
    g = torch.autograd.grad(loss, f)[0].numpy() if need_grad else None
    losses = np.array([x.item() for x in
                       (p_loss, e_loss, ent_mean, ent_loss, p_acc, e_acc, gp, loss)])
    return losses, g


def reward(pd, dd, obs, acs, mean, std, non_satur_grad=False, dtype=torch.float64, rmsify=True):
    """discriminator.py:180-194."""
    _, Pd = _params(pd, dd, dtype)
    sc = critic_q(Pd, obz(_t(obs, dtype), _t(mean, dtype), _t(std, dtype), rmsify),
                  _t(acs, dtype), dd)
    if non_satur_grad:
        r = torch.nn.functional.logsigmoid(sc)
    else:
        r = -torch.log(1.0 - torch.sigmoid(sc) + 1e-8)
    return r.detach().numpy()


def smooth_labels(u_real, u_fake=None):
    """discriminator.py:91-109: tf.random_uniform(minval=1.2, maxval=0.7) = 1.2 - 0.5u ;
    fake = 0 (one-sided) or 0.3 - 0.3u."""
    real = 1.2 + (0.7 - 1.2) * np.asarray(u_real)
    fake = np.zeros_like(real) if u_fake is None else 0.3 + (0.0 - 0.3) * np.asarray(u_fake)
    return real, fake


def adam_tf1(theta, m, v, g, lr, t, b1=0.9, b2=0.999, eps=1e-8):
    """tf.train.AdamOptimizer dense update at step t (1-based).  TF casts lr/beta1/beta2/epsilon
    to the variable dtype (float32) before use, so 1-beta2 is 1-fl32(0.999), 1.3e-5 off 1e-3; the
    oracle keeps that rounding of the hyper-parameters and does the rest in float64."""
    theta, m, v, g = (np.asarray(a, dtype=np.float64) for a in (theta, m, v, g))
    lr, b1, b2, eps = (float(np.float32(x)) for x in (lr, b1, b2, eps))
    lr_t = lr * math.sqrt(1.0 - b2 ** t) / (1.0 - b1 ** t)
    m = b1 * m + (1.0 - b1) * g
    v = b2 * v + (1.0 - b2) * g * g
    theta = theta - lr_t * m / (np.sqrt(v) + eps)
    return theta, m, v


def mpi_adam_step(theta, m, v, grads_per_rank, lr, t, clip_norm):
    """mpi_adam.py:18-36: mean over workers, THEN global-norm clip, then Adam."""
    g = np.mean(np.stack([np.asarray(x, dtype=np.float64) for x in grads_per_rank]), axis=0)
    if clip_norm is not None:
        g, _ = clip_by_global_norm(g, clip_norm)
    return adam_tf1(theta, m, v, g, lr, t)


def polyak(targ, online, tau):
    targ, online = np.asarray(targ, dtype=np.float64), np.asarray(online, dtype=np.float64)
    return (1.0 - tau) * targ + tau * online
