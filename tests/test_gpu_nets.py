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
