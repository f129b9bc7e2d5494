"""samx_step_grads / samx_step_apply (whole learner, fused whole-network kernels for 64x64 nets) vs
the fp64 oracle, and the fused path vs the layer-wise path.  Bar: 1e-5 relative (max-norm)."""
import numpy as np
import pytest
import torch

from tests.nets_cases import assert_per_var, make_case, record_margin, relerr

pytestmark = pytest.mark.gpu
TOL = 1e-5

HP = dict(ac_high=0.4, gamma=0.99, polyak=0.01, clip_norm=5.0, actor_lr=1e-4, critic_lr=1e-3, d_lr=3e-4,
          beta1=0.9, beta2=0.999, adam_eps=1e-8, td_loss_scale=1.0, wd_scale=1e-3,
          q_actor_loss_scale=1.0, d_actor_loss_scale=0.0, ent_reg_scale=0.0)


def _dev(x):
    return None if x is None else torch.as_tensor(np.ascontiguousarray(x)).cuda()


def _setup(name, d_scale, ent, per, rms=True, ln=False):
    from sam_tf_amd import ops
    c = make_case(name, ln=ln); P, D = c['P'], c['D']
    mk = lambda d: ops.mlp_desc(d.in_dim, d.h1, d.h2, d.out_dim, d.layernorm)
    hp = dict(HP, d_actor_loss_scale=d_scale, ent_reg_scale=ent)
    L = ops.LearnerState(mk(c['da']), mk(c['dc']), mk(c['dd']), c['ob'], c['ac'], hp)
    L.params.copy_(_dev(np.concatenate([P['pa'], P['pc'], P['pd']])))
    L.targets.copy_(_dev(np.concatenate([P['pa_t'], P['pc_t']])))
    if rms:
        for k in ('mu_mean', 'd_mean'):
            L.set_ptr(k, _dev(c['mean']))
        for k in ('mu_std', 'd_std'):
            L.set_ptr(k, _dev(c['std']))
    if d_scale:
        L.set_ptr('actor_reward_params', _dev(P['pd'][::-1].copy()))  # a DIFFERENT (frozen) reward net
    B = c['B']
    batch = ops.make_batch(B, p_obs=_dev(D['obs0']), p_acs=_dev(D['acs']), e_obs=_dev(D['e_obs']),
                           e_acs=_dev(D['e_acs']), eps_ob=_dev(D['eps_ob']), eps_ac=_dev(D['eps_ac']),
                           real_labels=_dev(D['real']), fake_labels=_dev(D['fake']), obs0=_dev(D['obs0']),
                           acs=_dev(D['acs']), rews=_dev(D['rews']), dones1=_dev(D['dones1']),
                           obs1=_dev(D['obs1']), iws=_dev(D['iws']) if per else None)
    return c, L, batch, ops.make_step_out(B)


def _oracle(c, d_scale, ent, per, rms=True, dtype=torch.float64):
    from oracle import nets_oracle as NO
    P, D = c['P'], c['D']
    mean, std = (c['mean'], c['std']) if rms else (0.0, 1.0)
    B = c['B']
    dl, dg = NO.disc_loss_grad(P['pd'], c['dd'], D['obs0'], D['acs'], D['e_obs'], D['e_acs'], D['eps_ob'],
                               D['eps_ac'], D['real'], D['fake'], mean, std, ent_reg_scale=ent, rmsify=rms, dtype=dtype)
    tq = NO.targ_q(P['pa_t'], P['pc_t'], c['da'], c['dc'], D['obs1'], D['rews'][:, None],
                   D['dones1'][:, None], np.ones((B, 1)), mean, std, 0.99, 0.4, rmsify=rms, dtype=dtype)
    al, ag = NO.actor_loss_grad(P['pa'], c['da'], P['pc'], c['dc'], P['pd'][::-1].copy() if d_scale else None,
                                c['dd'], D['obs0'], mean, std, 0.4, 1.0, d_scale, rmsify=rms, dtype=dtype)
    cl, cg, td = NO.critic_loss_grad(P['pc'], c['dc'], D['obs0'], D['acs'],
                                     D['iws'][:, None] if per else None, tq, mean, std, 1e-3, rmsify=rms, dtype=dtype)
    return dict(d_losses=dl, actor_losses=al, critic_losses=cl, targ_q=tq.ravel(), td=td.ravel(),
                grads=np.concatenate([ag, cg, dg]))


@pytest.mark.parametrize('name,d_scale,ent,per,rms,ln', [
    ('C1', 0.0, 0.0, False, True, False), ('C2', 0.0, 0.0, False, True, False),
    ('C3', 0.0, 0.0, True, True, False), ('C3', 0.5, 1e-3, True, True, False),
    ('C2', 0.5, 0.0, False, False, False),
    ('ragged', 0.5, 0.0, True, True, False),   # 33x70 nets, B = 37: not fusable -> layer-wise fallback
    ('C4', 0.0, 0.0, False, True, False),
    ('C5', 0.0, 1e-3, True, True, False),      # 400x300 nets, batch 8192: the layer-wise path at size
    ('w_20x3', 0.0, 0.0, False, True, False), ('w_40x32', 0.0, 1e-3, True, True, False),
    ('w_7x17', 0.0, 0.0, True, False, False), ('w_20x3', 0.5, 0.0, True, True, False),
    # with_layernorm (launcher default, mujoco_sam_imitate.sh): two-stream GP backward; 64x64 nets on the
    # whole-network LayerNorm kernels (fused_ln.hip), the rest on the layer-wise path
    ('C2', 0.5, 1e-3, True, True, True), ('ragged', 0.5, 0.0, False, True, True),
    ('C4', 0.0, 0.0, False, True, True),
    ('C1', 0.0, 0.0, False, True, True), ('C3', 0.0, 1e-3, True, False, True), ('C2', 0.0, 0.0, False, True, True),
    ('s_1x1', 0.5, 0.0, False, True, True), ('s_7x5', 0.5, 1e-3, True, False, True),
    ('s_20x4', 0.5, 0.0, False, True, True), ('s_13x2', 0.0, 0.0, True, True, True),
    ('s_1x1', 0.5, 0.0, False, True, False), ('s_2x1', 0.0, 1e-3, True, True, False),
    ('s_7x5', 0.5, 1e-3, True, False, False), ('s_20x4', 0.5, 0.0, False, True, False),
    ('s_13x2', 0.0, 0.0, True, True, False),
    # with_layernorm on the role-fused 256-wide path (fused_wide.hip: LayerNorm mini-phases, two-stream GP)
    ('w_20x3', 0.0, 0.0, False, True, True), ('w_40x32', 0.0, 1e-3, True, True, True),
    ('w_7x17', 0.0, 0.0, True, False, True)])
def test_step_grads_vs_oracle(name, d_scale, ent, per, rms, ln):
    from sam_tf_amd import ops
    c, L, batch, out = _setup(name, d_scale, ent, per, rms, ln)
    assert L.fused(c['B']) == (name in ('C1', 'C2', 'C3') or name.startswith('s_'))
    # 256x256 nets without an actor loss through the reward net: the role-fused path, LayerNorm or not
    assert (L.path(c['B']) == 2) == ((name == 'C4' or name.startswith('w_')) and d_scale == 0.0)
    exp = _oracle(c, d_scale, ent, per, rms)
    ops.step_grads(L, batch, out)
    torch.cuda.synchronize()
    tag = 'step_grads/%s%s%s' % (name, '-ln' if ln else '', '-dscale' if d_scale else '')
    exp32 = None
    for k in ('d_losses', 'actor_losses', 'critic_losses'):
        got = out.t[k].cpu().numpy().astype(np.float64)[:len(exp[k])]
        err = np.max(np.abs(got - exp[k]) / np.maximum(np.abs(exp[k]), 1e-3))
        if err >= TOL:
            # a loss is a mean of signed terms (the actor's: -mean Q) and can sit far below its terms' scale: it may
            # then be off by 4 x what torch float32 ON THE CPU leaves on the same inputs (recorded)
            exp32 = exp32 or _oracle(c, d_scale, ent, per, rms, dtype=torch.float32)
            floor = np.max(np.abs(exp32[k] - exp[k]) / np.maximum(np.abs(exp[k]), 1e-3))
            record_margin(tag + '/loss_fp32cpu_floor', k, floor)
            record_margin(tag + '/loss_error', k, err)
            assert err < 4 * floor, (k, got, exp[k], exp32[k])
    assert relerr(out.t['targ_q'].cpu().numpy(), exp['targ_q']) < TOL
    assert relerr(out.t['td'].cpu().numpy(), exp['td']) < TOL
    g = L.grads.cpu().numpy()
    for k, (net, desc) in enumerate((('actor', c['da']), ('critic', c['dc']), ('reward', c['dd']))):
        # every variable of every network against its own max-norm
        assert_per_var(L.seg(torch.as_tensor(g), k).numpy(), L.seg(torch.as_tensor(exp['grads']), k).numpy(), desc,
                       TOL, '%s/%s' % (tag, net))
    assert L.steps.tolist() == [1, 1, 1]


@pytest.mark.parametrize('name', ['C2', 'C4', 'C3'])
def test_masks_without_kink_redraw(name):
    """The other parity cases redraw rows whose pre-activations come within 1e-4 of the leaky-ReLU kink
    (nets_cases.keep_off_the_kink).  Here nothing is redrawn.  The importance weights of the ABI select ONE
    row of the critic's batch (iws one-hot), so the step kernels return that row's own gradient; it is compared
    with the fp64 oracle's for every row within the margin of the kink and for a sample of the others.  A row
    whose gradient differs is a row whose lrelu mask differs from the oracle's: allowed only among the
    near-kink rows, and only for a handful (|z| below fp32 rounding, expected 0-1 per batch)."""
    from oracle import nets_oracle as NO
    from sam_tf_amd import ops
    c = make_case(name, off_kink=False)
    P, D, B = c['P'], c['D'], c['B']
    t64 = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float64)
    ob0 = NO.obz(t64(D['obs0']), t64(c['mean']), t64(c['std'])).numpy()
    margin = NO.min_abs_preact(P['pc'], c['dc'], np.concatenate([ob0, D['acs']], axis=1))
    near = np.where(margin < 1e-4)[0]
    rs = np.random.RandomState(5)
    far = rs.choice(np.where(margin >= 1e-4)[0], size=6, replace=False)
    assert near.size <= B // 8, 'case has implausibly many near-kink rows: %d' % near.size
    mk = lambda d: ops.mlp_desc(d.in_dim, d.h1, d.h2, d.out_dim, d.layernorm)
    L = ops.LearnerState(mk(c['da']), mk(c['dc']), mk(c['dd']), c['ob'], c['ac'], dict(HP, wd_scale=0.0))
    L.params.copy_(_dev(np.concatenate([P['pa'], P['pc'], P['pd']])))
    L.targets.copy_(_dev(np.concatenate([P['pa_t'], P['pc_t']])))
    for k in ('mu_mean', 'd_mean'):
        L.set_ptr(k, _dev(c['mean']))
    for k in ('mu_std', 'd_std'):
        L.set_ptr(k, _dev(c['std']))
    iws = torch.zeros(B, dtype=torch.float32, device='cuda')
    batch = ops.make_batch(B, p_obs=_dev(D['obs0']), p_acs=_dev(D['acs']), e_obs=_dev(D['e_obs']),
                           e_acs=_dev(D['e_acs']), eps_ob=_dev(D['eps_ob']), eps_ac=_dev(D['eps_ac']),
                           real_labels=_dev(D['real']), fake_labels=_dev(D['fake']), obs0=_dev(D['obs0']),
                           acs=_dev(D['acs']), rews=_dev(D['rews']), dones1=_dev(D['dones1']),
                           obs1=_dev(D['obs1']), iws=iws)
    out = ops.make_step_out(B)
    tq = NO.targ_q(P['pa_t'], P['pc_t'], c['da'], c['dc'], D['obs1'], D['rews'][:, None], D['dones1'][:, None],
                   np.ones((B, 1)), c['mean'], c['std'], 0.99, 0.4)
    differs = []
    for r in list(near) + list(far):
        iws.zero_()
        iws[r] = 1.0
        ops.step_grads(L, batch, out)
        torch.cuda.synchronize()
        got = L.seg(L.grads, 1).cpu().numpy()
        w = np.zeros((B, 1), dtype=np.float32); w[r] = 1.0
        _, exp, _ = NO.critic_loss_grad(P['pc'], c['dc'], D['obs0'], D['acs'], w, tq, c['mean'], c['std'], 0.0)
        if relerr(got, exp) >= 1e-5:
            differs.append(int(r))
    assert not set(differs) & set(int(r) for r in far), 'rows away from the kink differ: %r' % differs
    assert len(differs) <= 3, 'near-kink rows with a different mask: %r of %d candidates' % (differs, near.size)
    from tests.nets_cases import record_margin
    record_margin('masks_without_redraw/%s' % name, 'near_kink_rows', near.size)
    record_margin('masks_without_redraw/%s' % name, 'rows_with_other_mask', len(differs))


@pytest.mark.parametrize('world', [1, 4])
def test_step_apply_vs_oracle(world):
    """Optimiser pass for the three segments: clipped copies (local norm), all-reduce emulated by
    scaling, Adam + Polyak — against oracle mpi_adam_step / polyak; fused and unfused modes agree."""
    from sam_tf_amd import ops
    from oracle import nets_oracle as NO
    c, L, batch, out = _setup('C2', 0.0, 0.0, False)
    ops.step_grads(L, batch, out)
    g_local = L.grads.cpu().numpy().astype(np.float64)
    th0 = L.params.cpu().numpy().astype(np.float64)
    tg0 = L.targets.cpu().numpy().astype(np.float64)
    if world == 1:
        ops.step_apply(L, c['B'], 3, 1.0)
        g_mean = g_local
    else:
        ops.step_apply(L, c['B'], 1, 1.0)            # logging copy with the LOCAL norm
        other = np.random.RandomState(0).randn(g_local.size) * 0.3
        L.grads.copy_(torch.as_tensor((g_local * 1.0 + other).astype(np.float32)).cuda())  # "all-reduce"
        g_mean = L.grads.cpu().numpy().astype(np.float64) / world
        ops.step_apply(L, c['B'], 6, 1.0 / world)
    lrs = (1e-4, 1e-3, 3e-4)
    for k in range(3):
        sl = lambda a: L.seg(torch.as_tensor(a), k).numpy()
        clipped, _ = NO.clip_by_global_norm(sl(g_local), 5.0)
        assert relerr(sl(L.clipped.cpu().numpy()), clipped) < TOL
        th, m, v = NO.mpi_adam_step(sl(th0), np.zeros(sl(th0).size), np.zeros(sl(th0).size), [sl(g_mean)],
                                    lrs[k], 1, 5.0)
        assert relerr(sl(L.params.cpu().numpy()), th) < TOL
        assert relerr(sl(L.m.cpu().numpy()), m) < TOL and relerr(sl(L.v.cpu().numpy()), v) < 2e-5
        if k < 2:
            o = (0, L.Pa)[k]
            tg = NO.polyak(tg0[o:o + th.size], th, 0.01)
            assert relerr(L.targets.cpu().numpy()[o:o + th.size], tg) < TOL


@pytest.mark.parametrize('ln', [False, True])
def test_fused_matches_layerwise(ln):
    """The two implementations of the same step agree far inside the parity bar."""
    from sam_tf_amd import ops
    c, L, batch, out = _setup('C3', 0.5, 1e-3, True, ln=ln)
    assert L.fused(c['B'])
    ops.step_grads(L, batch, out)
    g_f = L.grads.clone()
    ws = ops.nets_ws(L.da, L.dc, L.dd, c['B'])
    P, D = c['P'], c['D']
    _, gd = ops.disc_grad(L.dd, _dev(P['pd']), _dev(D['obs0']), _dev(D['acs']), _dev(D['e_obs']),
                          _dev(D['e_acs']), _dev(D['eps_ob']), _dev(D['eps_ac']), _dev(D['real']),
                          _dev(D['fake']), _dev(c['mean']), _dev(c['std']), 1e-3, ops.disc_ws(L.dd, c['B']))
    assert relerr(L.seg(g_f, 2).cpu().numpy(), gd.cpu().numpy()) < 2e-6
    _, ga = ops.actor_grad(L.da, _dev(P['pa']), L.dc, _dev(P['pc']), L.dd, _dev(P['pd'][::-1].copy()),
                           _dev(D['obs0']), _dev(c['mean']), _dev(c['std']), 0.4, 1.0, 0.5, ws)
    assert relerr(L.seg(g_f, 0).cpu().numpy(), ga.cpu().numpy()) < 2e-6


def test_indirect_rows_and_kernel_randoms_match_dense():
    """samx_step_grads reading rows in place through ring indices / the expert slice and drawing its
    own D randoms == the same call on gathered rows with the materialised randoms (bit for bit on
    the fused paths, to rounding on the layer-wise path)."""
    from sam_tf_amd import ops
    for name in ('C2', 'ragged', 'w_20x3', 'C4'):   # fused 64-wide, layer-wise, role-fused 256-wide (two sizes)
        c, L, dense, out = _setup(name, 0.0, 0.0, True)
        B, ob, ac = c['B'], c['ob'], c['ac']
        rs = np.random.RandomState(5)
        ring, start = 4 * B + 3, 2 * B + 1
        st = {k: torch.as_tensor(rs.randn(ring, w).astype(np.float32)).cuda()
              for k, w in (('obs0', ob), ('acs', ac), ('rews', 1), ('dones1', 1), ('obs1', ob))}
        pool_o = torch.as_tensor(rs.randn(3 * B, ob).astype(np.float32)).cuda()
        pool_a = torch.as_tensor(rs.rand(3 * B, ac).astype(np.float32)).cuda()
        idx_p = torch.as_tensor(rs.randint(0, ring, size=B)).cuda()
        idx_a = torch.as_tensor(rs.randint(0, ring, size=B)).cuda()
        e_start = torch.tensor([B // 2], dtype=torch.int64, device='cuda')
        iws = _dev(c['D']['iws'])
        L.steps[2] = 5  # pretend five reward-net updates happened: the randoms depend on it
        ind = ops.make_batch(B, p_obs=st['obs0'], p_acs=st['acs'], e_obs=pool_o, e_acs=pool_a,
                             obs0=st['obs0'], acs=st['acs'], rews=st['rews'].view(-1),
                             dones1=st['dones1'].view(-1), obs1=st['obs1'], iws=iws, p_idx=idx_p, a_idx=idx_a,
                             e_start=e_start, ring_start=start, ring_len=ring, rng_seed=1234,
                             label_smoothing=2)
        ops.step_grads(L, ind, out)
        g_ind = L.grads.clone(); l_ind = {k: v.clone() for k, v in out.t.items()}
        rows = lambda t, ix: t[(start + ix) % ring].contiguous()
        eo, ea, real, fake = ops.disc_randoms(1234, 5, B, ob, ac, want_real=True, want_fake=True)
        L.steps[2] = 5
        e0 = B // 2
        den = ops.make_batch(B, p_obs=rows(st['obs0'], idx_p), p_acs=rows(st['acs'], idx_p),
                             e_obs=pool_o[e0:e0 + B].contiguous(), e_acs=pool_a[e0:e0 + B].contiguous(),
                             eps_ob=eo, eps_ac=ea, real_labels=real, fake_labels=fake,
                             obs0=rows(st['obs0'], idx_a), acs=rows(st['acs'], idx_a),
                             rews=rows(st['rews'], idx_a).view(-1), dones1=rows(st['dones1'], idx_a).view(-1),
                             obs1=rows(st['obs1'], idx_a), iws=iws)
        ops.step_grads(L, den, out)
        if name != 'ragged':
            assert torch.equal(g_ind, L.grads), name
            for k in l_ind:
                assert torch.equal(l_ind[k], out.t[k]), k
        else:
            assert relerr(g_ind.cpu().numpy(), L.grads.cpu().numpy()) < 1e-6


@pytest.mark.parametrize('name,ln', [('C4', False), ('w_20x3', False), ('w_20x3', True)])
def test_weight_images_kept_by_the_optimiser_pass(name, ln):
    """256-wide path: update steps whose optimiser pass keeps the forward products' weight images current
    (samx_step_apply mode bit 6, samx_step_grads_part bits 7/8) == the same steps with the images packed from the
    parameter vectors at the head of every call, bit for bit — parameters, targets, Adam moments, gradients, losses.
    Role subsets in between (a discriminator-only and an agents-only step) leave the other images valid."""
    from sam_tf_amd import ops
    runs = {}
    for keep in (False, True):
        c, L, batch, out = _setup(name, 0.0, 1e-3, True, ln=ln)
        assert L.path(c['B']) == 2
        seq = [(0, 0), (0, 0), (4, 8), (8, 16), (0, 0)]   # (part role bits, apply role bits): all, all, agents, D, all
        for k, (pr, ar) in enumerate(seq):
            img_part = (256 if k == 0 else 128) if keep else 0
            img_mode = 64 if keep and k < len(seq) - 1 else 0
            ops.step_grads_part(L, batch, out, 3 | pr | img_part)
            ops.step_apply(L, c['B'], 3 | ar | img_mode, 1.0)
        torch.cuda.synchronize()
        runs[keep] = dict(params=L.params.clone(), targets=L.targets.clone(), m=L.m.clone(), v=L.v.clone(),
                          grads=L.grads.clone(), **{k: v.clone() for k, v in out.t.items()})
    for k in runs[False]:
        assert torch.equal(runs[False][k], runs[True][k]), k


@pytest.mark.parametrize('name,ln,d_scale', [('C2', False, 0.0), ('C3', False, 0.5), ('C1', False, 0.0), ('C2', True, 0.0),
                                             ('C3', True, 0.5)])
def test_optimiser_pass_inside_the_reduction_launch(name, ln, d_scale):
    """Whole-network path: update steps whose optimiser pass runs inside the reduction launch (samx_step_grads_part
    bit 9: gradients stay in registers, the blocks exchange their sums of squares as tagged words) == the same steps
    with samx_step_grads + samx_step_apply(mode 3), bit for bit — parameters, targets, Adam moments, clipped copies,
    gradients, losses, step counters.  Role subsets in between (an agents-only and a discriminator-only step), and
    the two forms interleaved on one workspace (the tagged words of a two-launch step are stale, never current)."""
    from sam_tf_amd import ops
    runs = {}
    seq = [(0, 0), (0, 0), (4, 8), (8, 16), (0, 0), (0, 0)]   # (part role bits, apply role bits)
    for one in (0, 1, 2):   # two launches | one launch | alternating
        c, L, batch, out = _setup(name, d_scale, 1e-3, True, ln=ln)
        assert L.path(c['B']) == 1
        for k, (pr, ar) in enumerate(seq):
            if one == 1 or (one == 2 and k % 2 == 0):
                ops.step_grads_part(L, batch, out, 3 | pr | 512)
            else:
                ops.step_grads_part(L, batch, out, 3 | pr)
                ops.step_apply(L, c['B'], 3 | ar, 1.0)
        torch.cuda.synchronize()
        assert bool(torch.isfinite(L.params).all())
        runs[one] = dict(params=L.params.clone(), targets=L.targets.clone(), m=L.m.clone(), v=L.v.clone(),
                         grads=L.grads.clone(), clipped=L.clipped.clone(), steps=L.steps.clone(),
                         **{k: v.clone() for k, v in out.t.items()})
    assert runs[0]['steps'].tolist() == [5, 5, 5]
    for k in runs[0]:
        assert torch.equal(runs[0][k], runs[1][k]), k
        assert torch.equal(runs[0][k], runs[2][k]), k


def test_optimiser_pass_inside_the_reduction_launch_is_refused_elsewhere():
    """Bit 9 is for samx_step_path 1: the 256-wide and layer-wise paths refuse it, and so does an evaluation without the
    counter tick (bit 6)."""
    from sam_tf_amd import ops, _lib
    for name in ('w_20x3', 'ragged'):
        c, L, batch, out = _setup(name, 0.0, 0.0, True)
        assert L.path(c['B']) != 1
        with pytest.raises(_lib.SamxError):
            ops.step_grads_part(L, batch, out, 3 | 512)
    c, L, batch, out = _setup('C2', 0.0, 0.0, True)
    with pytest.raises(_lib.SamxError):
        ops.step_grads_part(L, batch, out, 3 | 64 | 512)
    with pytest.raises(_lib.SamxError):
        ops.step_grads_part(L, batch, out, 1 | 512)


@pytest.mark.parametrize('name', ['C2', 'C4'])
def test_masks_without_kink_redraw_actor_and_disc(name):
    """test_masks_without_kink_redraw for the other two roles.  Neither the actor's loss nor the discriminator's takes
    importance weights, so one row is selected by REPLICATING it over the batch (every row of the batch is row r: the
    mean gradient is r's own; the gradient penalty's scalar is then that row's too, in the oracle as on the device).
    Nothing is redrawn.  Rows whose pre-activations — the actor's on obz0 and the critic's on [obz0 | mu(obz0)]; the
    reward net's on the policy pair, the expert pair and their interpolate — come within 1e-4 of the leaky-ReLU kink
    may carry another mask than the fp64 oracle (a handful at most); rows away from it must agree to 1e-5."""
    from oracle import nets_oracle as NO
    from sam_tf_amd import ops
    c = make_case(name, off_kink=False)
    P, D, B = c['P'], c['D'], c['B']
    t64 = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float64)
    obz = lambda o: NO.obz(t64(o), t64(c['mean']), t64(c['std'])).numpy()
    ob0 = obz(D['obs0'])
    mu = NO.actor_act(NO._params(P['pa'], c['da'], torch.float64)[1], t64(ob0), c['da'], 0.4).detach().numpy()
    m_actor = np.minimum(NO.min_abs_preact(P['pa'], c['da'], ob0),
                         NO.min_abs_preact(P['pc'], c['dc'], np.concatenate([ob0, mu], axis=1)))
    xp = np.concatenate([ob0, D['acs']], axis=1)
    xe = np.concatenate([obz(D['e_obs']), D['e_acs']], axis=1)
    eps = np.concatenate([D['eps_ob'], D['eps_ac']], axis=1)
    m_disc = np.minimum(np.minimum(NO.min_abs_preact(P['pd'], c['dd'], xp), NO.min_abs_preact(P['pd'], c['dd'], xe)),
                        NO.min_abs_preact(P['pd'], c['dd'], eps * xp + (1.0 - eps) * xe))
    mk = lambda d: ops.mlp_desc(d.in_dim, d.h1, d.h2, d.out_dim, d.layernorm)
    L = ops.LearnerState(mk(c['da']), mk(c['dc']), mk(c['dd']), c['ob'], c['ac'], dict(HP, wd_scale=0.0))
    L.params.copy_(_dev(np.concatenate([P['pa'], P['pc'], P['pd']])))
    L.targets.copy_(_dev(np.concatenate([P['pa_t'], P['pc_t']])))
    for k in ('mu_mean', 'd_mean'):
        L.set_ptr(k, _dev(c['mean']))
    for k in ('mu_std', 'd_std'):
        L.set_ptr(k, _dev(c['std']))
    out = ops.make_step_out(B)
    rs = np.random.RandomState(7)
    rep = lambda a, r: np.ascontiguousarray(np.broadcast_to(a[r], a.shape))
    for role, margin in (('actor', m_actor), ('disc', m_disc)):
        near = np.where(margin < 1e-4)[0]
        assert near.size <= B // 8, '%s: implausibly many near-kink rows: %d' % (role, near.size)
        near = near[:12]   # (each row costs a device step and an fp64 autograd pass over the whole batch)
        far = rs.choice(np.where(margin >= 1e-4)[0], size=4, replace=False)
        differs = []
        for r in list(near) + list(far):
            Dr = dict(D)
            if role == 'actor':
                Dr['obs0'] = rep(D['obs0'], r)
            else:
                for k in ('obs0', 'acs', 'e_obs', 'e_acs', 'eps_ob', 'eps_ac', 'real', 'fake'):
                    Dr[k] = rep(D[k], r)
            batch = ops.make_batch(B, p_obs=_dev(Dr['obs0'] if role == 'disc' else D['obs0']),
                                   p_acs=_dev(Dr['acs']), e_obs=_dev(Dr['e_obs']), e_acs=_dev(Dr['e_acs']),
                                   eps_ob=_dev(Dr['eps_ob']), eps_ac=_dev(Dr['eps_ac']), real_labels=_dev(Dr['real']),
                                   fake_labels=_dev(Dr['fake']), obs0=_dev(Dr['obs0'] if role == 'actor' else D['obs0']),
                                   acs=_dev(D['acs']), rews=_dev(D['rews']), dones1=_dev(D['dones1']),
                                   obs1=_dev(D['obs1']))
            ops.step_grads(L, batch, out)
            torch.cuda.synchronize()
            if role == 'actor':
                got = L.seg(L.grads, 0).cpu().numpy()
                _, exp = NO.actor_loss_grad(P['pa'], c['da'], P['pc'], c['dc'], None, c['dd'], Dr['obs0'], c['mean'],
                                            c['std'], 0.4, 1.0, 0.0)
            else:
                got = L.seg(L.grads, 2).cpu().numpy()
                _, exp = NO.disc_loss_grad(P['pd'], c['dd'], Dr['obs0'], Dr['acs'], Dr['e_obs'], Dr['e_acs'], Dr['eps_ob'],
                                           Dr['eps_ac'], Dr['real'], Dr['fake'], c['mean'], c['std'], ent_reg_scale=0.0)
            if relerr(got, exp) >= 1e-5:
                differs.append(int(r))
        assert not set(differs) & set(int(r) for r in far), '%s: rows away from the kink differ: %r' % (role, differs)
        assert len(differs) <= 3, '%s: near-kink rows with a different mask: %r of %d' % (role, differs, near.size)
        record_margin('masks_without_redraw/%s' % name, '%s_near_kink_rows' % role, int(near.size))
        record_margin('masks_without_redraw/%s' % name, '%s_rows_with_other_mask' % role, len(differs))
