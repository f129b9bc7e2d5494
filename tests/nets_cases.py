"""Shared seeded inputs for the network parity tests (CPU oracle side and GPU side)."""
import numpy as np

from oracle import nets_oracle as NO

# BASELINE.json configs C1-C5: (ob, ac, h1, h2, B)
CONFIGS = {
    'C1': (4, 1, 64, 64, 32),
    'C2': (11, 3, 64, 64, 256),
    'C3': (17, 6, 64, 64, 256),
    'C4': (376, 17, 256, 256, 1024),
    'C5': (376, 17, 400, 300, 8192),
    'ragged': (5, 2, 33, 70, 37),   # nothing a multiple of a tile
    # whole-network path away from the benchmark shapes: 1-wide inputs, widths not a multiple of 4 (the
    # padded k4 rows of the LDS weight tiles), the widest input that fits, one tile, many tiles
    's_1x1': (1, 1, 64, 64, 16),
    's_2x1': (2, 1, 64, 64, 16),
    's_7x5': (7, 5, 64, 64, 80),
    's_20x4': (20, 4, 64, 64, 48),
    's_13x2': (13, 2, 64, 64, 1024),
    # 256x256 nets: the role-fused path (fused_wide.hip): narrow input, the widest action head it takes, an
    # input that is not a multiple of 4, one tile
    'w_20x3': (20, 3, 256, 256, 48),
    'w_40x32': (40, 32, 256, 256, 32),
    'w_7x17': (7, 17, 256, 256, 16),
}


def relerr(a, b):
    """max-norm relative error: max|a-b| / max|b| (the 1e-5 bar of north_star is on this)."""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    den = np.max(np.abs(b))
    return float(np.max(np.abs(a - b)) / (den if den > 0 else 1.0))


# ---- per-variable norm (north_star: "gradients ... within 1e-5 relative"): every variable of a network
# (W1, b1, [beta1, gamma1], W2, b2, [beta2, gamma2], W3, b3) is held to the bar against ITS OWN max-norm, so a
# small-magnitude variable cannot hide behind a large one.  The observed maxima are collected per test and
# written out at session end (tests/conftest.py -> gpurun_out/parity_margins.json; the kept copy lives in
# profiles/).
MARGINS = {}


def record_margin(tag, var, value):
    key = '%s/%s' % (tag, var)
    MARGINS[key] = max(MARGINS.get(key, 0.0), float(value))


def per_var_relerr(got, exp, desc):
    """name -> max|got - exp| / max|exp| over that variable of the flat vector of network `desc`."""
    got, exp = np.asarray(got, dtype=np.float64).ravel(), np.asarray(exp, dtype=np.float64).ravel()
    assert got.shape == exp.shape, (got.shape, exp.shape)
    out = {}
    for name, (off, shape) in NO.param_slices(desc).items():
        n = int(np.prod(shape))
        out[name] = relerr(got[off:off + n], exp[off:off + n])
    return out


FLOOR_FACTOR, FLOOR_CAP = 4.0, 1e-3


def assert_per_var(got, exp, desc, tol, tag, abs_tol=0.0, floors=None, tol_by_var=None):
    """Assert every variable within tol of the oracle (max|got - exp| <= tol * max|exp| + abs_tol); a variable
    that is identically zero in the oracle (e.g. nothing flows into it) must be zero to tol in absolute terms
    (relerr's denominator falls back to 1).  abs_tol: see adam_allowance.
    tol_by_var: {variable: tol} overrides for named variables.
    floors: {variable: per-variable error of torch float32 ON THE CPU against the same fp64 oracle} — the noise
    floor of fp32 arithmetic on this very input.  A variable may then be off by max(tol, FLOOR_FACTOR * floor)
    (capped at FLOOR_CAP): the device is held to a small multiple of what the reference's own precision leaves,
    never to less than float32 can deliver.  Both numbers are recorded."""
    got, exp = np.asarray(got, dtype=np.float64).ravel(), np.asarray(exp, dtype=np.float64).ravel()
    errs = per_var_relerr(got, exp, desc)
    bad = {}
    for name, (off, shape) in NO.param_slices(desc).items():
        n = int(np.prod(shape))
        record_margin(tag, name, errs[name])
        t = (tol_by_var or {}).get(name, tol)
        if floors is not None:
            record_margin(tag + '/fp32cpu_floor', name, floors[name])
            t = max(t, min(FLOOR_FACTOR * floors[name], FLOOR_CAP))
        den = np.max(np.abs(exp[off:off + n]))
        # (abs_tol: a scalar, or one allowance per element of the flat vector — OracleLearner.adam's)
        at = abs_tol[off:off + n] if isinstance(abs_tol, np.ndarray) else abs_tol
        if not np.max(np.abs(got[off:off + n] - exp[off:off + n]) - at) <= t * (den if den > 0 else 1.0):
            bad[name] = (errs[name], t)
    assert not bad, '%s: per-variable error beyond its bar (error, bar) (+ %s absolute): %r (all: %r; floors: %r)' % (
        tag, 'per-element' if isinstance(abs_tol, np.ndarray) else '%g' % abs_tol, bad, errs, floors)
    return errs


def adam_allowance(K, lr, frac=0.01):
    """Absolute allowance for WEIGHTS after K Adam steps.  Adam's step lr * m / (sqrt(v) + eps) normalises every
    element to O(lr) whatever its gradient's size, so an element whose gradient is far below its variable's max
    (or within reach of eps = 1e-8) turns the fp32-vs-fp64 rounding difference of that gradient — 1e-6 of the
    VARIABLE's max, i.e. possibly 100 % of the element — into a fraction of lr.  Zero-initialised biases and the
    U(+-3e-3) heads are all displacement after a few steps, so their own max-norm is K * lr itself.  Gradients,
    the well-conditioned quantity, are held to the plain per-variable bar at every step; weights get this
    allowance of `frac` (1 %) of the largest displacement K steps can produce."""
    return frac * K * lr


KINK_MARGIN = 1e-4


def _row_margin(c):
    """Smallest |pre-activation| per batch row over every pass the parity tests run (fp64)."""
    import torch
    P, D, da, dc, dd = c['P'], c['D'], c['da'], c['dc'], c['dd']
    t = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float64)
    ob0 = NO.obz(t(D['obs0']), t(c['mean']), t(c['std'])).numpy()
    obe = NO.obz(t(D['e_obs']), t(c['mean']), t(c['std'])).numpy()
    xp = np.concatenate([ob0, D['acs']], axis=1)
    xe = np.concatenate([obe, D['e_acs']], axis=1)
    eps = np.concatenate([D['eps_ob'], D['eps_ac']], axis=1).astype(np.float64)
    xi = eps * xp + (1 - eps) * xe
    Pa = NO.unflatten(t(P['pa']), da)
    a = NO.actor_act(Pa, t(ob0), da, 0.4).numpy()
    xa = np.concatenate([ob0, a], axis=1)
    m = NO.min_abs_preact(P['pa'], da, ob0)
    for flat, d, x in ((P['pc'], dc, xp), (P['pc'], dc, xa), (P['pd'], dd, xa), (P['pd'], dd, xp),
                       (P['pd'], dd, xe), (P['pd'], dd, xi)):
        m = np.minimum(m, NO.min_abs_preact(flat, d, x))
    return m


def keep_off_the_kink(c, rs, margin=KINK_MARGIN, max_iter=60):
    """Redraw the rows that put some leaky-ReLU pre-activation within `margin` of 0 (see
    nets_oracle.min_abs_preact): there the gradient depends on the last bit of an fp32 sum, and even
    torch's own fp32 result differs from the fp64 one by ~1e-4 of the gradient's max."""
    D, ob, ac = c['D'], c['ob'], c['ac']
    for _ in range(max_iter):
        bad = np.where(_row_margin(c) < margin)[0]
        if bad.size == 0:
            return c
        n = bad.size
        D['obs0'][bad] = (rs.randn(n, ob) * 1.5 + 0.3).astype(np.float32)
        D['acs'][bad] = rs.uniform(-1, 1, (n, ac)).astype(np.float32)
        D['e_obs'][bad] = (rs.randn(n, ob) + 0.5).astype(np.float32)
        D['e_acs'][bad] = rs.uniform(-1, 1, (n, ac)).astype(np.float32)
        D['eps_ob'][bad] = rs.rand(n, ob).astype(np.float32)
        D['eps_ac'][bad] = rs.rand(n, ac).astype(np.float32)
    raise RuntimeError('could not move every row off the kink')


def make_case(name, ln=False, seed=0, off_kink=True):
    ob, ac, h1, h2, B = CONFIGS[name]
    rs = np.random.RandomState(seed + 17 * len(name))
    da = NO.MLP(ob, h1, h2, ac, ln)
    dc = NO.MLP(ob + ac, h1, h2, 1, ln)
    dd = NO.MLP(ob + ac, h1, h2, 1, ln)

    def perturb(p, s):
        return (p + rs.randn(p.size).astype(np.float32) * s).astype(np.float32)
    P = dict(
        pa=perturb(NO.init_params(da, rs, 'actor'), 0.01),
        pa_t=perturb(NO.init_params(da, rs, 'actor'), 0.01),
        # trained-looking critic / reward nets (the U(+-3e-3) init would make every test trivial)
        pc=perturb(NO.init_params(dc, rs, 'reward'), 0.02),
        pc_t=perturb(NO.init_params(dc, rs, 'reward'), 0.02),
        pd=perturb(NO.init_params(dd, rs, 'reward'), 0.02),
    )
    for k in ('pc', 'pc_t', 'pd'):
        sl = NO.param_slices(dc)
        o, s = sl['W3']
        P[k][o:o + h2] = rs.randn(h2).astype(np.float32) * (1.0 / np.sqrt(h2))
    D = dict(
        obs0=(rs.randn(B, ob) * 1.5 + 0.3).astype(np.float32),
        obs1=(rs.randn(B, ob) * 1.5 + 0.3).astype(np.float32),
        acs=rs.uniform(-1, 1, (B, ac)).astype(np.float32),
        rews=rs.uniform(0, 5, (B,)).astype(np.float32),
        dones1=(rs.rand(B) < 0.1).astype(np.float32),
        iws=rs.uniform(0.05, 1.0, (B,)).astype(np.float32),
        e_obs=(rs.randn(B, ob) + 0.5).astype(np.float32),
        e_acs=rs.uniform(-1, 1, (B, ac)).astype(np.float32),
        eps_ob=rs.rand(B, ob).astype(np.float32),
        eps_ac=rs.rand(B, ac).astype(np.float32),
        td_len=rs.randint(1, 5, size=B).astype(np.float32),
    )
    real, fake = NO.smooth_labels(rs.rand(B), rs.rand(B))
    D['real'], D['fake'] = real.astype(np.float32), fake.astype(np.float32)
    sums = NO.rms_update(NO.rms_init(ob), rs.randn(4000, ob) * 1.3 + 0.2, ob)
    mean, std = NO.rms_mean_std(sums, ob)
    c = dict(da=da, dc=dc, dd=dd, P=P, D=D, mean=mean, std=std, B=B, ob=ob, ac=ac)
    return keep_off_the_kink(c, rs) if off_kink else c
