"""HIP network passes vs the fp64 oracle on identical seeded inputs.
Bar (north_star): losses, gradients, rewards within 1e-5 relative — measured as max-norm relative
error max|a-b|/max|b| for vectors and plain relative error for scalars."""
import numpy as np
import pytest
import torch

from tests.nets_cases import CONFIGS, make_case, relerr

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _dev(x):
    return None if x is None else torch.as_tensor(np.ascontiguousarray(x)).cuda()


def _descs(c):
    from sam_tf_amd import ops
    mk = lambda d: ops.mlp_desc(d.in_dim, d.h1, d.h2, d.out_dim, d.layernorm)
    return mk(c['da']), mk(c['dc']), mk(c['dd'])


def _scal(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-3)))


SMALL = ['C1', 'C2', 'C3', 'ragged']
ALL = SMALL + ['C4', 'C5']


@pytest.mark.parametrize('rms', [True, False])
@pytest.mark.parametrize('ln', [False, True])
@pytest.mark.parametrize('name', ALL)
def test_targ_q_and_act(name, rms, ln):
    from sam_tf_amd import ops
    from oracle import nets_oracle as NO
    c = make_case(name, ln=ln); P, D = c['P'], c['D']
    da, dc, dd = _descs(c)
    mean, std = (c['mean'], c['std']) if rms else (None, None)
    ws = ops.nets_ws(da, dc, None, c['B'])
    for tl in (None, D['td_len']):
        exp = NO.targ_q(P['pa_t'], P['pc_t'], c['da'], c['dc'], D['obs1'], D['rews'][:, None],
                        D['dones1'][:, None], np.ones((c['B'], 1)) if tl is None else tl[:, None],
                        mean if rms else 0.0, std if rms else 1.0, 0.99, 1.0, rmsify=rms)
        got = ops.targ_q(da, _dev(P['pa_t']), dc, _dev(P['pc_t']), _dev(D['obs1']), _dev(D['rews']),
                         _dev(D['dones1']), _dev(tl), _dev(mean), _dev(std), 0.99, 1.0, ws)
        assert relerr(got.cpu().numpy(), exp.ravel()) < TOL
    ea, eq = NO.act_q(P['pa'], c['da'], P['pc'], c['dc'], D['obs0'], mean if rms else 0.0,
                      std if rms else 1.0, 0.4, rmsify=rms)
    ga, gq = ops.act_q(da, _dev(P['pa']), dc, _dev(P['pc']), _dev(D['obs0']), _dev(mean), _dev(std), 0.4, ws)
    assert relerr(ga.cpu().numpy(), ea) < TOL and relerr(gq.cpu().numpy(), eq.ravel()) < TOL


@pytest.mark.parametrize('per', [True, False])
@pytest.mark.parametrize('ln', [False, True])
@pytest.mark.parametrize('name', ALL)
def test_critic_grad(name, per, ln):
    from sam_tf_amd import ops
    from oracle import nets_oracle as NO
    c = make_case(name, ln=ln); P, D = c['P'], c['D']
    da, dc, dd = _descs(c)
    tc = (D['rews'] * 0.7 - 0.5).astype(np.float32)  # mixes |td| < 1 and > 1 (both huber branches)
    iws = D['iws'] if per else None
    el, eg, etd = NO.critic_loss_grad(P['pc'], c['dc'], D['obs0'], D['acs'],
                                      None if iws is None else iws[:, None], tc[:, None], c['mean'],
                                      c['std'], 1e-3)
    assert (np.abs(etd) < 1).any() and (np.abs(etd) > 1).any()
    ws = ops.nets_ws(da, dc, None, c['B'])
    gl, gg, gtd = ops.critic_grad(dc, _dev(P['pc']), _dev(D['obs0']), _dev(D['acs']), _dev(iws), _dev(tc),
                                  _dev(c['mean']), _dev(c['std']), 1.0, 1e-3, ws)
    assert _scal(gl.cpu().numpy(), el) < TOL
    assert relerr(gtd.cpu().numpy(), etd.ravel()) < TOL
    assert relerr(gg.cpu().numpy(), eg) < TOL
    td2 = ops.td_errors(dc, _dev(P['pc']), _dev(D['obs0']), _dev(D['acs']), _dev(tc), _dev(c['mean']),
                        _dev(c['std']), ws)
    assert relerr(td2.cpu().numpy(), etd.ravel()) < TOL


@pytest.mark.parametrize('d_scale', [0.0, 0.5])
@pytest.mark.parametrize('ln', [False, True])
@pytest.mark.parametrize('name', ALL)
def test_actor_grad(name, d_scale, ln):
    from sam_tf_amd import ops
    from oracle import nets_oracle as NO
    c = make_case(name, ln=ln); P, D = c['P'], c['D']
    da, dc, dd = _descs(c)
    el, eg = NO.actor_loss_grad(P['pa'], c['da'], P['pc'], c['dc'], P['pd'] if d_scale else None,
                                c['dd'], D['obs0'], c['mean'], c['std'], 0.4, 1.0, d_scale)
    ws = ops.nets_ws(da, dc, dd, c['B'])
    gl, gg = ops.actor_grad(da, _dev(P['pa']), dc, _dev(P['pc']), dd, _dev(P['pd']) if d_scale else None,
                            _dev(D['obs0']), _dev(c['mean']), _dev(c['std']), 0.4, 1.0, d_scale, ws)
    assert _scal(gl.cpu().numpy(), el) < TOL
    assert relerr(gg.cpu().numpy(), eg) < TOL


@pytest.mark.parametrize('ent', [0.0, 1e-3])
@pytest.mark.parametrize('ln', [False, True])
@pytest.mark.parametrize('name', ALL)
def test_disc_grad_and_reward(name, ent, ln):
    from sam_tf_amd import ops
    from oracle import nets_oracle as NO
    c = make_case(name, ln=ln); P, D = c['P'], c['D']
    da, dc, dd = _descs(c)
    el, eg = NO.disc_loss_grad(P['pd'], c['dd'], D['obs0'], D['acs'], D['e_obs'], D['e_acs'], D['eps_ob'],
                               D['eps_ac'], D['real'], D['fake'], c['mean'], c['std'], ent_reg_scale=ent)
    ws = ops.disc_ws(dd, c['B'])
    gl, gg = ops.disc_grad(dd, _dev(P['pd']), _dev(D['obs0']), _dev(D['acs']), _dev(D['e_obs']),
                           _dev(D['e_acs']), _dev(D['eps_ob']), _dev(D['eps_ac']), _dev(D['real']),
                           _dev(D['fake']), _dev(c['mean']), _dev(c['std']), ent, ws)
    assert _scal(gl.cpu().numpy(), el) < TOL, (gl.cpu().numpy(), el)
    assert relerr(gg.cpu().numpy(), eg) < TOL
    if ent == 0.0:
        # default labels (no smoothing): real = 1, fake = 0
        el2, eg2 = NO.disc_loss_grad(P['pd'], c['dd'], D['obs0'], D['acs'], D['e_obs'], D['e_acs'],
                                     D['eps_ob'], D['eps_ac'], np.ones(c['B']), np.zeros(c['B']),
                                     c['mean'], c['std'])
        gl2, gg2 = ops.disc_grad(dd, _dev(P['pd']), _dev(D['obs0']), _dev(D['acs']), _dev(D['e_obs']),
                                 _dev(D['e_acs']), _dev(D['eps_ob']), _dev(D['eps_ac']), None, None,
                                 _dev(c['mean']), _dev(c['std']), 0.0, ws)
        assert _scal(gl2.cpu().numpy(), el2) < TOL and relerr(gg2.cpu().numpy(), eg2) < TOL
        wsn = ops.nets_ws(da, dc, dd, c['B'])
        for ns in (False, True):
            er = NO.reward(P['pd'], c['dd'], D['obs0'], D['acs'], c['mean'], c['std'], non_satur_grad=ns)
            gr = ops.reward(dd, _dev(P['pd']), _dev(D['obs0']), _dev(D['acs']), _dev(c['mean']),
                            _dev(c['std']), ns, wsn)
            assert relerr(gr.cpu().numpy(), er.ravel()) < TOL


def test_workspace_and_shape_errors():
    from sam_tf_amd import ops, _lib
    c = make_case('C1'); P, D = c['P'], c['D']
    da, dc, dd = _descs(c)
    small = torch.empty(64, dtype=torch.uint8, device='cuda')
    with pytest.raises(_lib.SamxError):
        ops.critic_grad(dc, _dev(P['pc']), _dev(D['obs0']), _dev(D['acs']), None, _dev(D['rews']),
                        None, None, 1.0, 0.0, small)
    with pytest.raises(ValueError):
        ops.critic_grad(dc, _dev(P['pc'][:-1]), _dev(D['obs0']), _dev(D['acs']), None, _dev(D['rews']),
                        None, None, 1.0, 0.0, ops.nets_ws(da, dc, None, c['B']))


@pytest.mark.parametrize('ln', [False, True])
@pytest.mark.parametrize('name', ['ragged', 'C2', 'C4'])
def test_return_normalisation_and_nstep_loss(name, ln):
    """rmsify_rets (sam_agent.py:161-201) in every agent pass, and the n-step critic term
    (sam_agent.py:483-498) — same 1e-5 bar."""
    from sam_tf_amd import ops
    from oracle import nets_oracle as NO
    c = make_case(name, ln=ln); P, D = c['P'], c['D']
    da, dc, dd = _descs(c)
    B = c['B']
    ret = (0.7, 2.5)
    rt = (_dev(np.array([ret[0]], dtype=np.float32)), _dev(np.array([ret[1]], dtype=np.float32)))
    ws = ops.nets_ws(da, dc, dd, B)
    exp = NO.targ_q(P['pa_t'], P['pc_t'], c['da'], c['dc'], D['obs1'], D['rews'][:, None], D['dones1'][:, None],
                    D['td_len'][:, None], c['mean'], c['std'], 0.99, 1.0, ret=ret)
    got = ops.targ_q(da, _dev(P['pa_t']), dc, _dev(P['pc_t']), _dev(D['obs1']), _dev(D['rews']),
                     _dev(D['dones1']), _dev(D['td_len']), _dev(c['mean']), _dev(c['std']), 0.99, 1.0, ws, ret=rt)
    assert relerr(got.cpu().numpy(), exp.ravel()) < TOL
    ea, eq = NO.act_q(P['pa'], c['da'], P['pc'], c['dc'], D['obs0'], c['mean'], c['std'], 0.4, ret=ret)
    ga, gq = ops.act_q(da, _dev(P['pa']), dc, _dev(P['pc']), _dev(D['obs0']), _dev(c['mean']), _dev(c['std']),
                       0.4, ws, ret=rt)
    assert relerr(gq.cpu().numpy(), eq.ravel()) < TOL
    tc1 = (D['rews'] * 1.9 - 1.0).astype(np.float32)
    tcn = (D['rews'] * 2.3 + 0.4).astype(np.float32)
    el, eg, etd = NO.critic_loss_grad(P['pc'], c['dc'], D['obs0'], D['acs'], D['iws'][:, None], tc1[:, None],
                                      c['mean'], c['std'], 1e-3, td_scale=0.8, ret=ret, tcns=tcn[:, None],
                                      tdn_scale=0.6)
    gl, gg, gtd = ops.critic_grad(dc, _dev(P['pc']), _dev(D['obs0']), _dev(D['acs']), _dev(D['iws']), _dev(tc1),
                                  _dev(c['mean']), _dev(c['std']), 0.8, 1e-3, ws, ret=rt, tcns=_dev(tcn),
                                  tdn_scale=0.6)
    assert gl.numel() == 4 and _scal(gl.cpu().numpy(), el) < TOL
    assert relerr(gtd.cpu().numpy(), etd.ravel()) < TOL and relerr(gg.cpu().numpy(), eg) < TOL
    td2 = ops.td_errors(dc, _dev(P['pc']), _dev(D['obs0']), _dev(D['acs']), _dev(tc1), _dev(c['mean']),
                        _dev(c['std']), ws, ret=rt)
    assert relerr(td2.cpu().numpy(), etd.ravel()) < TOL
    al, ag = NO.actor_loss_grad(P['pa'], c['da'], P['pc'], c['dc'], P['pd'], c['dd'], D['obs0'], c['mean'],
                                c['std'], 0.4, 1.0, 0.5, ret=ret)
    gal, gag = ops.actor_grad(da, _dev(P['pa']), dc, _dev(P['pc']), dd, _dev(P['pd']), _dev(D['obs0']),
                              _dev(c['mean']), _dev(c['std']), 0.4, 1.0, 0.5, ws, ret=rt)
    assert _scal(gal.cpu().numpy(), al) < TOL and relerr(gag.cpu().numpy(), ag) < TOL


@pytest.mark.parametrize('two', [False, True])
def test_ret_update_and_popart(two):
    """ret_rms.update on one or two target batches, then PopArt on both critics
    (sam_agent.py:554-590,662-672): the critic's de-normalised output must be preserved."""
    from sam_tf_amd import ops
    from oracle import nets_oracle as NO
    c = make_case('ragged'); P = c['P']
    _, dc, _ = _descs(c)
    rs = np.random.RandomState(5)
    sums0 = NO.rms_update(NO.rms_init(1), rs.randn(50, 1) * 3 + 1.5, 1)
    tq1, tqn = (rs.randn(c['B']) * 2 + 4).astype(np.float32), (rs.randn(c['B']) * 5 - 1).astype(np.float32)
    om, os_ = NO.rms_mean_std(sums0, 1)
    sums1 = NO.rms_update(sums0, tq1[:, None], 1)
    if two:
        sums1 = NO.rms_update(sums1, tqn[:, None], 1)
    nm, ns = NO.rms_mean_std(sums1, 1)
    sl = NO.param_slices(c['dc'])
    (ow, _), (ob_, _) = sl['W3'], sl['b3']
    h2 = c['dc'].h2
    exp = {}
    for k in ('pc', 'pc_t'):
        w, b = NO.popart(P[k][ow:ow + h2], P[k][ob_:ob_ + 1], float(om[0]), float(os_[0]), float(nm[0]), float(ns[0]))
        e = P[k].astype(np.float64).copy(); e[ow:ow + h2] = w; e[ob_] = b[0]
        exp[k] = e
    sums = _dev(sums0.copy()); mean = _dev(om.copy()); std = _dev(os_.copy())
    a1, a2 = torch.zeros(3, dtype=torch.float64, device='cuda'), torch.zeros(3, dtype=torch.float64, device='cuda')
    ops.rms_batch_sums(_dev(tq1[:, None]), None, a1)
    ops.rms_batch_sums(_dev(tqn[:, None]), None, a2)
    pc, pct = _dev(P['pc'].copy()), _dev(P['pc_t'].copy())
    ops.ret_update_popart(sums, a1, a2 if two else None, mean, std, True, dc, pc, pct)
    assert relerr(sums.cpu().numpy(), sums1) < 1e-14
    assert abs(mean.item() - nm[0]) <= 1e-6 * abs(nm[0]) and abs(std.item() - ns[0]) <= 1e-6 * abs(ns[0])
    assert relerr(pc.cpu().numpy(), exp['pc']) < 1e-6 and relerr(pct.cpu().numpy(), exp['pc_t']) < 1e-6
    # untouched everywhere else
    keep = np.ones(P['pc'].size, bool); keep[ow:ow + h2] = False; keep[ob_] = False
    assert np.array_equal(pc.cpu().numpy()[keep], P['pc'][keep])
