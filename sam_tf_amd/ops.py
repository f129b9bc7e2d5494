"""Tensor-level wrappers over the libsamx C ABI (include/samx.h).

Every function takes torch CUDA tensors (device memory owner = torch's allocator, stream = torch's
current stream), checks shapes/dtypes on the host — a wrong shape must never reach a kernel — and
enqueues the HIP work.  There is no CPU path here.
"""
import ctypes as C
import os

import numpy as np
import torch

from . import _lib
from ._lib import MlpDesc, check, ptr, stream_ptr


def _L():
    return _lib.lib()


def mlp_desc(in_dim, h1, h2, out_dim, layernorm=False):
    return MlpDesc(int(in_dim), int(h1), int(h2), int(out_dim), int(bool(layernorm)))


def param_count(d):
    return int(_L().samx_mlp_param_count(C.byref(d)))


def _chk(t, dtype, shape=None, name='tensor'):
    if not (isinstance(t, torch.Tensor) and t.is_cuda):
        raise TypeError('%s must be a CUDA tensor' % name)
    if t.dtype != dtype:
        raise TypeError('%s must be %s, got %s' % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise ValueError('%s must be contiguous' % name)
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise ValueError('%s has shape %s, expected %s' % (name, tuple(t.shape), tuple(shape)))
    return t


def _opt(t, dtype, shape, name):
    return None if t is None else _chk(t, dtype, shape, name)


F32, F64, I64, I32 = torch.float32, torch.float64, torch.int64, torch.int32


# ------------------------------------------------------------------ segment trees / PER

class DeviceTrees(object):
    """Sum + min tree pair in HBM (2*cap doubles each) plus the dedupe stamp array."""

    def __init__(self, cap, device='cuda'):
        assert cap > 0 and cap & (cap - 1) == 0
        self.cap = cap
        self.sum = torch.empty(2 * cap, dtype=F64, device=device)
        self.min = torch.empty(2 * cap, dtype=F64, device=device)
        self.stamp = torch.empty(cap, dtype=I32, device=device)
        self._out = torch.empty(1, dtype=F64, device=device)
        check(_L().samx_tree_init(ptr(self.sum), ptr(self.min), ptr(self.stamp), cap, stream_ptr()))

    def set(self, idxs, leaf):
        n = idxs.numel()
        _chk(idxs, I64, (n,), 'idxs'); _chk(leaf, F64, (n,), 'leaf')
        check(_L().samx_tree_set(ptr(self.sum), ptr(self.min), ptr(self.stamp), self.cap, ptr(idxs),
                                 ptr(leaf), n, stream_ptr()))

    def reduce(self, kind, start=0, end=None):
        tree = self.sum if kind == 'sum' else self.min
        check(_L().samx_tree_reduce(ptr(tree), self.cap, 0 if kind == 'sum' else 1, int(start),
                                    self.cap if end is None else int(end), ptr(self._out),
                                    stream_ptr()))
        return float(self._out.item())

    def find_prefixsum(self, prefix):
        n = prefix.numel()
        _chk(prefix, F64, (n,), 'prefix')
        out = torch.empty(n, dtype=I64, device=prefix.device)
        check(_L().samx_tree_find_prefixsum(ptr(self.sum), self.cap, ptr(prefix), n, ptr(out),
                                            stream_ptr()))
        return out

    def per_sample(self, num_entries, u, beta, idxs_out=None, iws_out=None, iws32_out=None):
        B = u.numel()
        _chk(u, F64, (B,), 'u')
        dev = u.device
        idxs_out = torch.empty(B, dtype=I64, device=dev) if idxs_out is None else idxs_out
        iws_out = torch.empty(B, dtype=F64, device=dev) if iws_out is None else iws_out
        _chk(idxs_out, I64, (B,), 'idxs_out'); _chk(iws_out, F64, (B,), 'iws_out')
        _opt(iws32_out, F32, (B,), 'iws32_out')
        check(_L().samx_per_sample(ptr(self.sum), ptr(self.min), self.cap, int(num_entries), ptr(u), B,
                                   float(beta), ptr(idxs_out), ptr(iws_out), ptr(iws32_out),
                                   stream_ptr()))
        return idxs_out, iws_out


def per_leaf_values_host(idxs, prios, alpha, num_demos, demos_eps, max_priority):
    """numpy in, numpy out; host libm pow (see include/samx.h)."""
    prios = np.ascontiguousarray(prios, dtype=np.float64)
    idxs = np.ascontiguousarray(idxs, dtype=np.int64)
    assert idxs.shape == prios.shape
    leaf = np.empty_like(prios)
    mp = C.c_double(float(max_priority))
    check(_L().samx_per_leaf_values_host(ptr(idxs), ptr(prios), prios.size, float(alpha),
                                         int(num_demos), float(demos_eps), ptr(leaf), C.byref(mp)))
    return leaf, mp.value


def per_leaves_from_td_host(td, idxs, num_entries, alpha, num_demos, demos_eps, max_priority, out):
    """td float32[n], idxs int64[n], out float64[n]: host numpy arrays (pinned staging is fine).  Returns
    the new max_priority; `out` receives the leaves (samx_per_leaves_from_td_host)."""
    mp = C.c_double(float(max_priority))
    check(_L().samx_per_leaves_from_td_host(td.ctypes.data, idxs.ctypes.data, td.size, int(num_entries),
                                            float(alpha), int(num_demos), float(demos_eps), out.ctypes.data,
                                            C.byref(mp)))
    return mp.value


_POW_OK = None


def pow_matches_host(n=200000):
    """True when the library's pow routine reproduces THIS machine's libm pow bit for bit on n host-side
    cases (samx_pow_selfcheck) — the condition for keeping the priority refresh on the device."""
    global _POW_OK
    if _POW_OK is None:
        _POW_OK = int(_L().samx_pow_selfcheck(int(n))) == 0
    return _POW_OK


_HOST_RNG = None   # (bit generator object, address of its state) once checked; False when not usable


def host_rng_ready():
    """True when the library's restatement of numpy's legacy randint (samx_host_mt19937_randint, include/samx.h)
    reproduces THIS numpy draw for draw on the global generator's own state — checked once per process, with the
    generator put back exactly where it was.  SAMX_HOST_RNG=0 keeps numpy's call."""
    global _HOST_RNG
    if _HOST_RNG is None:
        _HOST_RNG = False
        try:
            bg = np.random.mtrand._rand._bit_generator
            if os.environ.get('SAMX_HOST_RNG', '1') != '0' and type(bg).__name__ == 'MT19937':
                addr = int(bg.ctypes.state_address)
                saved = np.random.get_state()
                ok = True
                try:
                    for high in (1 << 20, 1000000, 100000, 7, 1, 1 << 32, 4097):
                        np.random.set_state(saved)
                        ref = np.random.randint(low=0, high=high, size=(3, 500))
                        tail = np.random.randint(low=0, high=1 << 30, size=4)
                        np.random.set_state(saved)
                        got = np.full((3, 520), -1, dtype=np.int64)
                        check(_L().samx_host_mt19937_randint(addr, high, 3, 500, got.ctypes.data, 520))
                        ok = ok and np.array_equal(ref, got[:, :500]) and bool((got[:, 500:] == -1).all()) and \
                            np.array_equal(tail, np.random.randint(low=0, high=1 << 30, size=4))
                finally:
                    np.random.set_state(saved)
                if ok:
                    _HOST_RNG = (bg, addr)
        except Exception:  # noqa: BLE001 - any surprise in numpy's internals: numpy's own call stays
            _HOST_RNG = False
    return bool(_HOST_RNG) and np.random.mtrand._rand._bit_generator is _HOST_RNG[0]


def host_randint_rows(words, rows, row_len, high):
    """np.random.randint(low=0, high=high, size=(rows, row_len)) of the global generator, written into the first
    row_len int64 words of the first `rows` rows of the 2-D int64 array `words` (its rows may be longer: pinned
    staging).  Only after host_rng_ready()."""
    assert words.dtype == np.int64 and words.ndim == 2 and words.strides[1] == 8 and words.strides[0] % 8 == 0
    assert rows <= words.shape[0] and row_len <= words.shape[1]
    check(_L().samx_host_mt19937_randint(_HOST_RNG[1], int(high), int(rows), int(row_len), words.ctypes.data,
                                         words.strides[0] // 8))


_POW_DEV_OK = {}


def pow_device_matches_host(alpha, beta, device='cuda', n=4096):
    """True when the DEVICE build of the pow routine gives this machine's libm bits for the exponents this replay
    buffer uses: priority ** alpha on float32-born bases (memory.py:341-342,376-377) and (p * N) ** -beta on
    double ratios (memory.py:301,308).  pow_matches_host checks the host build of the same source on fixed
    exponents; the device compiler lowers fma / division on its own, so the driver asks this before it keeps the
    priority refresh on the device (a mismatch falls back to the host refresh: slower, same results)."""
    dev = torch.device(device)
    key = (float(alpha), float(beta), dev.index if dev.index is not None else torch.cuda.current_device())
    if key not in _POW_DEV_OK:
        rs = np.random.RandomState(20241)
        f32 = np.exp(rs.uniform(np.log(1e-6), np.log(1e4), n)).astype(np.float32).astype(np.float64)
        f32[:4] = (1e-6, np.float64(np.float32(1e-6)), 1.0, np.float64(np.float32(1e4)))
        ratio = np.exp(rs.uniform(np.log(1e-9), np.log(1e3), n))
        ok = True
        for x, y in ((f32, float(alpha)), (f32 + 0.1, float(alpha)), (ratio, -float(beta))):
            got = pow_device(torch.as_tensor(x, dtype=F64).to(dev), y).cpu().numpy()
            exp = np.array([float(v) ** y for v in x], dtype=np.float64)   # CPython float.__pow__ -> libm pow
            ok = ok and bool(np.array_equal(got.view(np.int64), exp.view(np.int64)))
        _POW_DEV_OK[key] = ok
    return _POW_DEV_OK[key]


def pow_device(x, y):
    """x float64 device tensor -> x ** y with the host pow's bits (samx_pow_device)."""
    _chk(x, F64, (x.numel(),), 'x')
    out = torch.empty_like(x)
    check(_L().samx_pow_device(ptr(x), float(y), x.numel(), ptr(out), stream_ptr()))
    return out


def make_per_state(max_priority, device='cuda'):
    """The 16 state bytes of samx_per_leaves_from_td: [max_priority, error bits] as two float64 words."""
    st = torch.zeros(2, dtype=F64, device=device)
    st[0] = float(max_priority)
    return st


def per_leaves_from_td(td, idxs, cap, num_entries, alpha, num_demos, demos_eps, leaf_out, state):
    n = td.numel()
    _chk(td, F32, (n,), 'td'); _chk(idxs, I64, (n,), 'idxs'); _chk(leaf_out, F64, (n,), 'leaf_out')
    _chk(state, F64, (2,), 'state')
    check(_L().samx_per_leaves_from_td(ptr(td), ptr(idxs), n, int(cap), int(num_entries), float(alpha),
                                       int(num_demos), float(demos_eps), ptr(leaf_out), ptr(state), stream_ptr()))


def per_leaves_from_td_geom(td, idxs, cap, geom, alpha, num_demos, demos_eps, leaf_out, state):
    """per_leaves_from_td with num_entries read from the device words geom = int64[2] {start, num_entries}."""
    n = td.numel()
    _chk(td, F32, (n,), 'td'); _chk(idxs, I64, (n,), 'idxs'); _chk(leaf_out, F64, (n,), 'leaf_out')
    _chk(state, F64, (2,), 'state'); _chk(geom, I64, (2,), 'geom')
    check(_L().samx_per_leaves_from_td_geom(ptr(td), ptr(idxs), n, int(cap), ptr(geom), float(alpha), int(num_demos),
                                            float(demos_eps), ptr(leaf_out), ptr(state), stream_ptr()))


def per_sample_rows(trees, geom, ring_len, u, beta, idxs_out, iws_out, iws32_out, rows_out, idx_copy_out=None):
    """trees.per_sample on device geometry words; rows_out receives the storage rows (include/samx.h)."""
    B = u.numel()
    _chk(u, F64, (B,), 'u'); _chk(geom, I64, (2,), 'geom')
    _chk(idxs_out, I64, (B,), 'idxs_out'); _chk(iws_out, F64, (B,), 'iws_out'); _opt(iws32_out, F32, (B,), 'iws32_out')
    _chk(rows_out, I64, (B,), 'rows_out'); _opt(idx_copy_out, I64, (B,), 'idx_copy_out')
    check(_L().samx_per_sample_rows(ptr(trees.sum), ptr(trees.min), trees.cap, ptr(geom), int(ring_len), ptr(u), B,
                                    float(beta), ptr(idxs_out), ptr(iws_out), ptr(iws32_out), ptr(rows_out),
                                    ptr(idx_copy_out), stream_ptr()))


def per_row_prologue_rows(trees, upd_idx, upd_td, alpha, num_demos, demos_eps, state, geom, ring_len, u, beta,
                          idxs_out, iws_out, iws32_out, rows_out, idx_copy_out=None):
    """Pending priority update given as TD errors + the row's stratified sample, ring geometry in device words
    (samx_per_row_prologue_td_rows): refresh, tree update, sample; rows_out = storage rows of the sample."""
    n, B = upd_idx.numel(), u.numel()
    _chk(upd_idx, I64, (n,), 'upd_idx'); _chk(upd_td, F32, (n,), 'upd_td'); _chk(state, F64, (2,), 'state')
    _chk(geom, I64, (2,), 'geom'); _chk(u, F64, (B,), 'u')
    _chk(idxs_out, I64, (B,), 'idxs_out'); _chk(iws_out, F64, (B,), 'iws_out'); _opt(iws32_out, F32, (B,), 'iws32_out')
    _chk(rows_out, I64, (B,), 'rows_out'); _opt(idx_copy_out, I64, (B,), 'idx_copy_out')
    check(_L().samx_per_row_prologue_td_rows(
        ptr(trees.sum), ptr(trees.min), ptr(trees.stamp), trees.cap, ptr(upd_idx), ptr(upd_td), n, float(alpha),
        int(num_demos), float(demos_eps), ptr(state), ptr(geom), int(ring_len), ptr(u), B, float(beta), ptr(idxs_out),
        ptr(iws_out), ptr(iws32_out), ptr(rows_out), ptr(idx_copy_out), stream_ptr()))


# ------------------------------------------------------------------ replay rows

def replay_gather(store, start, idxs, want=('obs0', 'acs', 'rews', 'dones1', 'obs1'), out=None, host_out=False):
    """store: dict atom -> tensor [maxlen, dim] ; returns dict atom -> [B, dim].
    host_out: the tensors of `out` are PINNED HOST memory — the kernel stores the rows there directly."""
    B = idxs.numel()
    if host_out and not idxs.is_cuda:
        assert idxs.is_pinned() and idxs.dtype == I64 and idxs.is_contiguous()
    else:
        _chk(idxs, I64, (B,), 'idxs')
    maxlen, ob = store['obs0'].shape
    ac = store['acs'].shape[1]
    dims = dict(obs0=ob, acs=ac, rews=1, dones1=1, obs1=ob)
    out = {} if out is None else out
    for k in want:
        if host_out:
            t = out[k]
            assert t.is_pinned() and t.dtype == F32 and tuple(t.shape) == (B, dims[k]) and t.is_contiguous(), k
            continue
        if k not in out:
            out[k] = torch.empty(B, dims[k], dtype=F32, device=idxs.device)
        _chk(out[k], F32, (B, dims[k]), 'out[%s]' % k)
    for k in dims:
        _chk(store[k], F32, (maxlen, dims[k]), 'store[%s]' % k)
    check(_L().samx_replay_gather(ptr(store['obs0']), ptr(store['acs']), ptr(store['rews']),
                                  ptr(store['dones1']), ptr(store['obs1']), maxlen, int(start), ob, ac,
                                  ptr(idxs), B, ptr(out.get('obs0')), ptr(out.get('acs')),
                                  ptr(out.get('rews')), ptr(out.get('dones1')), ptr(out.get('obs1')),
                                  stream_ptr()))
    return out


def replay_write(store, packed, rows):
    n = rows.numel()
    maxlen, ob = store['obs0'].shape
    ac = store['acs'].shape[1]
    _chk(rows, I64, (n,), 'rows'); _chk(packed, F32, (n, 2 * ob + ac + 2), 'packed')
    check(_L().samx_replay_write(ptr(store['obs0']), ptr(store['acs']), ptr(store['rews']),
                                 ptr(store['dones1']), ptr(store['obs1']), ob, ac, ptr(packed),
                                 ptr(rows), n, stream_ptr()))


# ------------------------------------------------------------------ running mean / std

def rms_batch_sums(x1, x2, addvec, host_ok=False):
    """host_ok: x1 may be a PINNED host tensor (read once by the kernel over PCIe)."""
    dim = x1.shape[1]
    if host_ok and not x1.is_cuda:
        assert x1.is_pinned() and x1.dtype == F32 and x1.is_contiguous() and x2 is None
    else:
        _chk(x1, F32, (x1.shape[0], dim), 'x1')
    if x2 is not None:
        _chk(x2, F32, (x2.shape[0], dim), 'x2')
    _chk(addvec, F64, (2 * dim + 1,), 'addvec')
    check(_L().samx_rms_batch_sums(ptr(x1), x1.shape[0], ptr(x2), 0 if x2 is None else x2.shape[0],
                                   dim, ptr(addvec), stream_ptr()))


def rms_commit(sums, addvec, mean, std):
    dim = mean.numel()
    _chk(sums, F64, (2 * dim + 1,), 'sums'); _opt(addvec, F64, (2 * dim + 1,), 'addvec')
    _chk(mean, F32, (dim,), 'mean'); _chk(std, F32, (dim,), 'std')
    check(_L().samx_rms_commit(ptr(sums), ptr(addvec), dim, ptr(mean), ptr(std), stream_ptr()))


# ------------------------------------------------------------------ optimiser

def adam_partials(device='cuda'):
    return torch.empty(_L().samx_adam_partials(), dtype=F32, device=device)


def clip_adam(param, grad, m, v, lr, step, clip_norm, partials, grad_scale=1.0, targ=None, tau=0.0,
              beta1=0.9, beta2=0.999, eps=1e-8, step_dev=None):
    n = param.numel()
    for t, nm in ((param, 'param'), (grad, 'grad'), (m, 'm'), (v, 'v')):
        _chk(t, F32, (n,), nm)
    _opt(targ, F32, (n,), 'targ')
    _chk(partials, F32, (_L().samx_adam_partials(),), 'partials')
    check(_L().samx_clip_adam(ptr(param), ptr(grad), ptr(m), ptr(v), n, float(lr), float(beta1),
                              float(beta2), float(eps), int(step),
                              float(clip_norm) if clip_norm else 0.0, float(grad_scale), ptr(targ),
                              float(tau), ptr(partials), ptr(_opt(step_dev, I64, (1,), 'step_dev')),
                              stream_ptr()))


def polyak(targ, param, tau):
    n = param.numel()
    _chk(targ, F32, (n,), 'targ'); _chk(param, F32, (n,), 'param')
    check(_L().samx_polyak(ptr(targ), ptr(param), n, float(tau), stream_ptr()))


def clip_copy(grad, clip_norm, partials, out=None):
    n = grad.numel()
    _chk(grad, F32, (n,), 'grad')
    out = torch.empty_like(grad) if out is None else _chk(out, F32, (n,), 'out')
    check(_L().samx_clip_copy(ptr(grad), n, float(clip_norm) if clip_norm else 0.0, ptr(out),
                              ptr(partials), stream_ptr()))
    return out


def count_mismatch(a, b, out=None):
    n = a.numel()
    _chk(a, F32, (n,), 'a'); _chk(b, F32, (n,), 'b')
    out = torch.empty(1, dtype=I32, device=a.device) if out is None else out
    check(_L().samx_count_mismatch(ptr(a), ptr(b), n, ptr(out), stream_ptr()))
    return out


def perturb_params(d, params, out, stddev, seed, counter):
    _flat(params, d, 'params'); _flat(out, d, 'out')
    check(_L().samx_perturb_params(C.byref(d), ptr(params), ptr(out), float(stddev), int(seed), int(counter),
                                   stream_ptr()))
    return out


def rms_distance(a, b, out=None):
    n = a.numel()
    _chk(a, F32, tuple(a.shape), 'a'); _chk(b, F32, tuple(a.shape), 'b')
    out = torch.empty(1, dtype=F32, device=a.device) if out is None else _chk(out, F32, (1,), 'out')
    check(_L().samx_rms_distance(ptr(a), ptr(b), n, ptr(out), stream_ptr()))
    return out


# ------------------------------------------------------------------ network passes

def nets_ws(da, dc, dd, B, device='cuda'):
    n = _L().samx_nets_ws_bytes(C.byref(da), C.byref(dc), C.byref(dd) if dd is not None else None,
                                int(B))
    return torch.empty(int(n), dtype=torch.uint8, device=device)


def disc_ws(dd, B, device='cuda'):
    return torch.empty(int(_L().samx_disc_ws_bytes(C.byref(dd), int(B))), dtype=torch.uint8,
                       device=device)


def _ms(mean, std, ob):
    if mean is None:
        assert std is None
        return None, None
    return _chk(mean, F32, (ob,), 'ob_mean'), _chk(std, F32, (ob,), 'ob_std')


def _flat(p, d, name):
    return _chk(p, F32, (param_count(d),), name)


def _ret(ret):
    """ret: None or (mean[1], std[1]) CUDA float32 tensors -> POINTER(RetNorm) (kept alive by caller)."""
    if ret is None:
        return None
    m, s = ret
    _chk(m, F32, (1,), 'ret_mean'); _chk(s, F32, (1,), 'ret_std')
    r = _lib.RetNorm(m.data_ptr(), s.data_ptr())
    return C.byref(r)


def act_q(da, pa, dc, pc, obs, mean, std, high, ws, ac_out=None, q_out=None, want_q=True, ret=None):
    B, ob = obs.shape
    _chk(obs, F32, (B, da.in_dim), 'obs'); _flat(pa, da, 'actor params'); _flat(pc, dc, 'critic params')
    mean, std = _ms(mean, std, ob)
    ac_out = torch.empty(B, da.out_dim, dtype=F32, device=obs.device) if ac_out is None else ac_out
    _chk(ac_out, F32, (B, da.out_dim), 'ac_out')
    if want_q and q_out is None:
        q_out = torch.empty(B, dtype=F32, device=obs.device)
    _opt(q_out, F32, (B,), 'q_out')
    check(_L().samx_act_q(C.byref(da), ptr(pa), C.byref(dc), ptr(pc), ptr(obs), ptr(mean), ptr(std), B,
                          float(high), ptr(ac_out), ptr(q_out), _ret(ret), ptr(ws), ws.numel(),
                          stream_ptr()))
    return ac_out, q_out


def targ_q(da, pa_t, dc, pc_t, obs1, rews, dones1, td_len, mean, std, gamma, high, ws, out=None, ret=None):
    B, ob = obs1.shape
    _chk(obs1, F32, (B, da.in_dim), 'obs1'); _flat(pa_t, da, 'targ actor'); _flat(pc_t, dc, 'targ critic')
    rews = _chk(rews.reshape(B), F32, (B,), 'rews'); dones1 = _chk(dones1.reshape(B), F32, (B,), 'dones1')
    td_len = None if td_len is None else _chk(td_len.reshape(B), F32, (B,), 'td_len')
    mean, std = _ms(mean, std, ob)
    out = torch.empty(B, dtype=F32, device=obs1.device) if out is None else _chk(out, F32, (B,), 'out')
    check(_L().samx_targ_q(C.byref(da), ptr(pa_t), C.byref(dc), ptr(pc_t), ptr(obs1), ptr(rews),
                           ptr(dones1), ptr(td_len), ptr(mean), ptr(std), B, float(gamma), float(high),
                           ptr(out), _ret(ret), ptr(ws), ws.numel(), stream_ptr()))
    return out


def critic_grad(dc, pc, obs0, acs, iws, tc1s, mean, std, td_scale, wd_scale, ws, losses=None,
                grad=None, td=None, ret=None, tcns=None, tdn_scale=1.0, tdn=None):
    """losses: float32[4] = {td_loss_1, wd, total, td_loss_n}; the first three are returned unless
    tcns is given (then all four)."""
    B, ob = obs0.shape
    ac = dc.in_dim - ob
    _chk(obs0, F32, (B, ob), 'obs0'); _chk(acs, F32, (B, ac), 'acs'); _flat(pc, dc, 'critic params')
    iws = None if iws is None else _chk(iws.reshape(B), F32, (B,), 'iws')
    tc1s = _chk(tc1s.reshape(B), F32, (B,), 'tc1s')
    mean, std = _ms(mean, std, ob)
    dev = obs0.device
    l4 = torch.empty(4, dtype=F32, device=dev) if losses is None or losses.numel() != 4 else losses
    _chk(l4, F32, (4,), 'losses')
    grad = torch.empty(param_count(dc), dtype=F32, device=dev) if grad is None else _flat(grad, dc, 'grad')
    td = torch.empty(B, dtype=F32, device=dev) if td is None else _chk(td, F32, (B,), 'td')
    tcns = None if tcns is None else _chk(tcns.reshape(B), F32, (B,), 'tcns')
    _opt(tdn, F32, (B,), 'tdn')
    check(_L().samx_critic_grad(C.byref(dc), ptr(pc), ptr(obs0), ptr(acs), ptr(iws), ptr(tc1s), ptr(mean),
                                ptr(std), B, ob, float(td_scale), float(wd_scale), ptr(l4), ptr(grad),
                                ptr(td), _ret(ret), ptr(tcns), float(tdn_scale), ptr(tdn), ptr(ws),
                                ws.numel(), stream_ptr()))
    if losses is not None and losses.numel() == 3:
        _chk(losses, F32, (3,), 'losses').copy_(l4[:3])
        return losses, grad, td
    return (l4 if tcns is not None else l4[:3]), grad, td


def td_errors(dc, pc, obs0, acs, tc1s, mean, std, ws, out=None, ret=None):
    B, ob = obs0.shape
    ac = dc.in_dim - ob
    _chk(obs0, F32, (B, ob), 'obs0'); _chk(acs, F32, (B, ac), 'acs'); _flat(pc, dc, 'critic params')
    tc1s = _chk(tc1s.reshape(B), F32, (B,), 'tc1s')
    mean, std = _ms(mean, std, ob)
    out = torch.empty(B, dtype=F32, device=obs0.device) if out is None else _chk(out, F32, (B,), 'out')
    check(_L().samx_td_errors(C.byref(dc), ptr(pc), ptr(obs0), ptr(acs), ptr(tc1s), ptr(mean), ptr(std),
                              B, ob, ptr(out), _ret(ret), ptr(ws), ws.numel(), stream_ptr()))
    return out


_TDR_WS = {}


def td_errors_rows(dc, pc, obs0_store, acs_store, idx, ring_start, tc1s, mean, std, out, ret=None,
                   td_host=None, idx_host=None):
    """TD errors of the batch rows addressed in place in the replay storages (idx int64[B]).  td_host /
    idx_host: pinned host tensors that receive copies of the TD errors and of idx from the same launch."""
    if td_host is not None:
        # pinned host memory (the host refresh) or device memory (the device refresh): the kernel just stores
        assert idx_host is not None and (td_host.is_pinned() or td_host.is_cuda) and \
            (idx_host.is_pinned() or idx_host.is_cuda)
        assert td_host.dtype == F32 and idx_host.dtype == I64 and td_host.numel() == idx.numel() == idx_host.numel()
    B = idx.numel()
    ring_len, ob = obs0_store.shape
    _chk(obs0_store, F32, (ring_len, ob), 'obs0 store'); _chk(acs_store, F32, (ring_len, dc.in_dim - ob), 'acs store')
    _chk(idx, I64, (B,), 'idx'); _flat(pc, dc, 'critic params')
    tc1s = _chk(tc1s.reshape(B), F32, (B,), 'tc1s'); _chk(out, F32, (B,), 'out')
    mean, std = _ms(mean, std, ob)
    key = (dc.in_dim, dc.h1, dc.h2, dc.layernorm, B, str(idx.device))
    if key not in _TDR_WS:
        _TDR_WS[key] = torch.empty(int(_L().samx_td_errors_rows_ws_bytes(C.byref(dc), B)), dtype=torch.uint8,
                                   device=idx.device)
    ws = _TDR_WS[key]
    check(_L().samx_td_errors_rows(C.byref(dc), ptr(pc), ptr(obs0_store), ptr(acs_store), ptr(idx),
                                   int(ring_start), int(ring_len), ptr(tc1s), ptr(mean), ptr(std), B, ob,
                                   ptr(out), ptr(td_host), ptr(idx_host), _ret(ret), ptr(ws), ws.numel(),
                                   stream_ptr()))
    return out


def actor_grad(da, pa, dc, pc, dd, pd, obs0, mean, std, high, q_scale, d_scale, ws, losses=None,
               grad=None, ret=None):
    B, ob = obs0.shape
    _chk(obs0, F32, (B, da.in_dim), 'obs0'); _flat(pa, da, 'actor params'); _flat(pc, dc, 'critic params')
    if pd is not None:
        _flat(pd, dd, 'reward-net params')
    mean, std = _ms(mean, std, ob)
    dev = obs0.device
    losses = torch.empty(3, dtype=F32, device=dev) if losses is None else _chk(losses, F32, (3,), 'losses')
    grad = torch.empty(param_count(da), dtype=F32, device=dev) if grad is None else _flat(grad, da, 'grad')
    check(_L().samx_actor_grad(C.byref(da), ptr(pa), C.byref(dc), ptr(pc),
                               C.byref(dd) if dd is not None else None, ptr(pd), ptr(obs0), ptr(mean),
                               ptr(std), B, float(high), float(q_scale), float(d_scale), ptr(losses),
                               ptr(grad), _ret(ret), ptr(ws), ws.numel(), stream_ptr()))
    return losses, grad


def ret_update_popart(sums, addvec1, addvec2, ret_mean, ret_std, enable_popart, dc, critic_params,
                      targ_critic_params):
    """ret_rms.update (one or two target batches) + PopArt head rescale in one launch."""
    _chk(sums, F64, (3,), 'ret sums'); _chk(addvec1, F64, (3,), 'addvec1'); _opt(addvec2, F64, (3,), 'addvec2')
    _chk(ret_mean, F32, (1,), 'ret_mean'); _chk(ret_std, F32, (1,), 'ret_std')
    if enable_popart:
        _flat(critic_params, dc, 'critic params'); _flat(targ_critic_params, dc, 'target critic params')
    check(_L().samx_ret_update_popart(ptr(sums), ptr(addvec1), ptr(addvec2), ptr(ret_mean), ptr(ret_std),
                                      int(bool(enable_popart)), C.byref(dc), ptr(critic_params),
                                      ptr(targ_critic_params), stream_ptr()))


def nstep_lookahead(store, start, num_entries, idxs, n, gamma, want64=False):
    """RB.lookahead on device storage: returns end_idx int64[B], rews f32[B], dones f32[B], td_len
    f32[B] (+ rews f64[B] when want64)."""
    B = idxs.numel()
    _chk(idxs, I64, (B,), 'idxs')
    maxlen = store['rews'].shape[0]
    _chk(store['rews'], F32, (maxlen, 1), 'rews'); _chk(store['dones1'], F32, (maxlen, 1), 'dones1')
    dev = idxs.device
    end = torch.empty(B, dtype=I64, device=dev)
    rews, dones, tl = (torch.empty(B, dtype=F32, device=dev) for _ in range(3))
    r64 = torch.empty(B, dtype=F64, device=dev) if want64 else None
    check(_L().samx_nstep_lookahead(ptr(store['rews']), ptr(store['dones1']), maxlen, int(start),
                                    int(num_entries), ptr(idxs), B, int(n), float(gamma), ptr(end), ptr(rews),
                                    ptr(r64), ptr(dones), ptr(tl), stream_ptr()))
    return (end, rews, dones, tl, r64) if want64 else (end, rews, dones, tl)


def nstep_lookahead_many(store, start, num_entries, words, a_off, B, n, gamma, end, rews, dones, tl):
    """RB.lookahead for the K batches of a launch at once: batch k's indices are words[k, a_off:a_off+B];
    outputs are the caller's dense [K, B] tensors (end int64, rews / dones / tl float32)."""
    K, W = words.shape
    _chk(words, I64, (K, W), 'words')
    assert 0 <= a_off and a_off + B <= W
    maxlen = store['rews'].shape[0]
    _chk(store['rews'], F32, (maxlen, 1), 'rews'); _chk(store['dones1'], F32, (maxlen, 1), 'dones1')
    _chk(end, I64, (K, B), 'end')
    for t, nm in ((rews, 'rews'), (dones, 'dones'), (tl, 'tl')):
        _chk(t, F32, (K, B), nm)
    check(_L().samx_nstep_lookahead_many(ptr(store['rews']), ptr(store['dones1']), maxlen, int(start),
                                         int(num_entries), words.data_ptr() + 8 * a_off, W, K, B, int(n),
                                         float(gamma), ptr(end), ptr(rews), ptr(dones), ptr(tl), stream_ptr()))


def nstep_lookahead_many_rows(store, geom, words, a_off, B, n, gamma, end, rews, dones, tl):
    """nstep_lookahead_many on storage rows: words[k, a_off:a_off+B] hold (start + idx) % maxlen, geom is int64[2]
    {start, num_entries} in device memory, `end` receives storage rows (include/samx.h)."""
    K, W = words.shape
    _chk(words, I64, (K, W), 'words')
    _chk(geom, I64, (2,), 'geom')
    assert 0 <= a_off and a_off + B <= W
    maxlen = store['rews'].shape[0]
    _chk(store['rews'], F32, (maxlen, 1), 'rews'); _chk(store['dones1'], F32, (maxlen, 1), 'dones1')
    _chk(end, I64, (K, B), 'end')
    for t, nm in ((rews, 'rews'), (dones, 'dones'), (tl, 'tl')):
        _chk(t, F32, (K, B), nm)
    check(_L().samx_nstep_lookahead_many_rows(ptr(store['rews']), ptr(store['dones1']), maxlen, geom.data_ptr(),
                                              words.data_ptr() + 8 * a_off, W, K, B, int(n), float(gamma), ptr(end),
                                              ptr(rews), ptr(dones), ptr(tl), stream_ptr()))


def disc_grad(dd, pd, p_obs, p_acs, e_obs, e_acs, eps_ob, eps_ac, real, fake, mean, std, ent_scale, ws,
              losses=None, grad=None):
    B, ob = p_obs.shape
    ac = dd.in_dim - ob
    for t, s, nm in ((p_obs, (B, ob), 'p_obs'), (e_obs, (B, ob), 'e_obs'), (eps_ob, (B, ob), 'eps_ob'),
                     (p_acs, (B, ac), 'p_acs'), (e_acs, (B, ac), 'e_acs'), (eps_ac, (B, ac), 'eps_ac')):
        _chk(t, F32, s, nm)
    _flat(pd, dd, 'reward-net params')
    real = None if real is None else _chk(real.reshape(B), F32, (B,), 'real_labels')
    fake = None if fake is None else _chk(fake.reshape(B), F32, (B,), 'fake_labels')
    mean, std = _ms(mean, std, ob)
    dev = p_obs.device
    losses = torch.empty(8, dtype=F32, device=dev) if losses is None else _chk(losses, F32, (8,), 'losses')
    grad = torch.empty(param_count(dd), dtype=F32, device=dev) if grad is None else _flat(grad, dd, 'grad')
    check(_L().samx_disc_grad(C.byref(dd), ptr(pd), ptr(p_obs), ptr(p_acs), ptr(e_obs), ptr(e_acs),
                              ptr(eps_ob), ptr(eps_ac), ptr(real), ptr(fake), ptr(mean), ptr(std), B, ob,
                              float(ent_scale), ptr(losses), ptr(grad), ptr(ws), ws.numel(),
                              stream_ptr()))
    return losses, grad


def reward(dd, pd, obs, acs, mean, std, non_satur_grad, ws, out=None):
    B, ob = obs.shape
    ac = dd.in_dim - ob
    _chk(obs, F32, (B, ob), 'obs'); _chk(acs, F32, (B, ac), 'acs'); _flat(pd, dd, 'reward-net params')
    mean, std = _ms(mean, std, ob)
    out = torch.empty(B, dtype=F32, device=obs.device) if out is None else _chk(out, F32, (B,), 'out')
    check(_L().samx_reward(C.byref(dd), ptr(pd), ptr(obs), ptr(acs), ptr(mean), ptr(std), B, ob,
                           int(bool(non_satur_grad)), ptr(out), ptr(ws), ws.numel(), stream_ptr()))
    return out


# ------------------------------------------------------------------ whole-learner step

class LearnerState(object):
    """Owns the flat device vectors of the three optimisers ([actor | critic | reward-net]) and the
    `samx_learner` struct that points at them."""

    def __init__(self, da, dc, dd, ob_dim, ac_dim, hp, device='cuda'):
        self.da, self.dc, self.dd = da, dc, dd
        self.Pa, self.Pc, self.Pd = param_count(da), param_count(dc), param_count(dd)
        P = self.Pa + self.Pc + self.Pd
        z = lambda n: torch.zeros(n, dtype=F32, device=device)
        self.params, self.grads, self.clipped, self.m, self.v = z(P), z(P), z(P), z(P), z(P)
        self.targets = z(self.Pa + self.Pc)
        self.steps = torch.zeros(3, dtype=I64, device=device)
        self.device = torch.device(device)
        self.hp = dict(hp)
        self.c = _lib.Learner()
        self.c.actor, self.c.critic, self.c.reward = da, dc, dd
        self.c.ob_dim, self.c.ac_dim = int(ob_dim), int(ac_dim)
        for k, t in (('params', self.params), ('grads', self.grads), ('clipped', self.clipped),
                     ('adam_m', self.m), ('adam_v', self.v), ('targets', self.targets),
                     ('steps', self.steps)):
            setattr(self.c, k, t.data_ptr())
        for k, v in self.hp.items():
            setattr(self.c, k, int(v) if k in ('enable_popart', 'n_step') else float(v))
        self._keep = {}
        self._ws = {}

    def set_ptr(self, field, tensor):
        """Optional pointers: actor_reward_params, mu_mean, mu_std, d_mean, d_std, ret_mean, ret_std,
        ret_sums."""
        self._keep[field] = tensor
        setattr(self.c, field, None if tensor is None else tensor.data_ptr())

    def seg(self, t, k):
        o = (0, self.Pa, self.Pa + self.Pc)[k]
        n = (self.Pa, self.Pc, self.Pd)[k]
        return t[o:o + n]

    def fused(self, B):
        return bool(_L().samx_fused_supported(C.byref(self.c), int(B)))

    def path(self, B):
        """0 layer-wise, 1 whole-network fused (64x64), 2 role-fused 256x256 (samx_step_path)."""
        return int(_L().samx_step_path(C.byref(self.c), int(B)))

    def ws(self, B):
        if B not in self._ws:
            n = _L().samx_step_ws_bytes(C.byref(self.c), int(B))
            self._ws[B] = torch.zeros(int(n), dtype=torch.uint8, device=self.device)
        return self._ws[B]


_BATCH_F32 = ('p_obs', 'p_acs', 'e_obs', 'e_acs', 'eps_ob', 'eps_ac', 'real_labels', 'fake_labels', 'obs0',
              'acs', 'rews', 'dones1', 'obs1', 'iws', 'td_len', 'n_rews', 'n_dones', 'n_len')
_BATCH_I64 = ('p_idx', 'a_idx', 'e_start', 'n_end')
_BATCH_F64 = ('d_rms_sums', 'd_rms_addvec', 'ret_addvec')
_BATCH_INT = ('ring_start', 'ring_len', 'rng_seed', 'label_smoothing', 'num_entries')


def make_batch(B, host_ok=False, **fields):
    """Fill a `samx_batch`.  float32 / int64 CUDA tensors for the pointer fields, ints for the rest.
    Row-count checks apply to dense inputs only (indirect inputs address whole storages).
    host_ok: pointer fields may also be PINNED host tensors (read by the kernels over PCIe; for small inputs that
    are read once or twice)."""
    b = _lib.Batch()
    b.B = int(B)
    keep = []
    indirect_p = fields.get('p_idx') is not None
    indirect_a = fields.get('a_idx') is not None
    indirect_e = fields.get('e_start') is not None
    for k, t in fields.items():
        if t is None:
            continue
        if k in _BATCH_INT:
            setattr(b, k, int(t))
            continue
        want = F32 if k in _BATCH_F32 else (F64 if k in _BATCH_F64 else I64)
        assert k in _BATCH_F32 or k in _BATCH_I64 or k in _BATCH_F64, 'unknown batch field %s' % k
        assert (t.is_cuda or (host_ok and t.is_pinned())) and t.dtype == want and t.is_contiguous(), k
        dense = k not in _BATCH_F64 and not ((indirect_p and k in ('p_obs', 'p_acs')) or (indirect_e and k in ('e_obs', 'e_acs')) or
                     (indirect_a and k in ('obs0', 'acs', 'rews', 'dones1', 'obs1')) or k == 'e_start')
        if dense:
            assert t.shape[0] == B, '%s must have %d rows' % (k, B)
        keep.append(t)
        setattr(b, k, t.data_ptr())
    b._keep = keep
    return b


def disc_randoms(seed, step, B, ob, ac, want_real=True, want_fake=False, device='cuda'):
    """The tensors samx_step_grads draws itself for reward-net update number `step` (0-based)."""
    eo = torch.empty(B, ob, dtype=F32, device=device); ea = torch.empty(B, ac, dtype=F32, device=device)
    real = torch.empty(B, dtype=F32, device=device) if want_real else None
    fake = torch.empty(B, dtype=F32, device=device) if want_fake else None
    check(_L().samx_disc_randoms(int(seed), int(step), int(B), int(ob), int(ac), ptr(eo), ptr(ea), ptr(real),
                                 ptr(fake), stream_ptr()))
    return eo, ea, real, fake


_STATS_WS = {}


def _stats_ws(B, ob, device):
    key = (B, ob, str(device))
    if key not in _STATS_WS:
        n = int(_L().samx_d_input_stats_ws_bytes(B, ob))
        _STATS_WS[key] = torch.empty(n, dtype=torch.uint8, device=device)
    return _STATS_WS[key]


def d_input_stats(replay_obs0, ring_start, p_idx, pool_obs0, e_start, B, addvec, commit=False, sums=None,
                  mean=None, std=None):
    ring_len, ob = replay_obs0.shape
    _chk(replay_obs0, F32, (ring_len, ob), 'replay_obs0'); _chk(pool_obs0, F32, (pool_obs0.shape[0], ob), 'pool')
    _opt(p_idx, I64, (B,), 'p_idx'); _opt(e_start, I64, (1,), 'e_start')
    _chk(addvec, F64, (2 * ob + 1,), 'addvec')
    if commit:
        _chk(sums, F64, (2 * ob + 1,), 'sums'); _chk(mean, F32, (ob,), 'mean'); _chk(std, F32, (ob,), 'std')
    ws = _stats_ws(int(B), int(ob), replay_obs0.device)
    check(_L().samx_d_input_stats(ptr(replay_obs0), int(ring_start), int(ring_len), ptr(p_idx), ptr(pool_obs0),
                                  ptr(e_start), int(B), int(ob), ptr(addvec), int(bool(commit)), ptr(sums),
                                  ptr(mean), ptr(std), ptr(ws), ws.numel(), stream_ptr()))


def d_input_stats_static(replay_obs0, ring_start, p_idx, pool_obs0, e_start, B, addvec, commit=False, sums=None,
                         mean=None, std=None):
    """d_input_stats through the entry point that allocates nothing: the partials live in the library's
    static device scratch (ob <= samx_d_input_stats_static_max_ob()); no workspace tensor is made here."""
    ring_len, ob = replay_obs0.shape
    _chk(replay_obs0, F32, (ring_len, ob), 'replay_obs0'); _chk(pool_obs0, F32, (pool_obs0.shape[0], ob), 'pool')
    _opt(p_idx, I64, (B,), 'p_idx'); _opt(e_start, I64, (1,), 'e_start')
    _chk(addvec, F64, (2 * ob + 1,), 'addvec')
    if commit:
        _chk(sums, F64, (2 * ob + 1,), 'sums'); _chk(mean, F32, (ob,), 'mean'); _chk(std, F32, (ob,), 'std')
    check(_L().samx_d_input_stats_static(ptr(replay_obs0), int(ring_start), int(ring_len), ptr(p_idx),
                                         ptr(pool_obs0), ptr(e_start), int(B), int(ob), ptr(addvec),
                                         int(bool(commit)), ptr(sums), ptr(mean), ptr(std), stream_ptr()))


def d_input_sums_many(replay_obs0, ring_start, words, B, pool_obs0, addvecs):
    """Batch sums of K steps at once.  words: int64 [K, W] with idx_p at [:, :B] and e_start at
    [:, 2B]; addvecs: float64 [K, 2*ob+1]."""
    ring_len, ob = replay_obs0.shape
    K, W = words.shape
    _chk(replay_obs0, F32, (ring_len, ob), 'replay_obs0'); _chk(pool_obs0, F32, (pool_obs0.shape[0], ob), 'pool')
    _chk(words, I64, (K, W), 'words'); _chk(addvecs, F64, (K, 2 * ob + 1), 'addvecs')
    assert W >= 2 * B + 1
    ws = None
    if 2 * B * ob > 16384:   # two-stage sums: per-step partials
        key = (int(B), int(ob), int(K), str(replay_obs0.device))
        if key not in _STATS_WS:
            n = int(_L().samx_d_input_stats_ws_bytes(int(B), int(ob))) * int(K)
            _STATS_WS[key] = torch.empty(n, dtype=torch.uint8, device=replay_obs0.device)
        ws = _STATS_WS[key]
    check(_L().samx_d_input_sums_many_ws(ptr(replay_obs0), int(ring_start), int(ring_len),
                                         C.c_void_p(words.data_ptr()), ptr(pool_obs0),
                                         C.c_void_p(words.data_ptr() + 8 * 2 * B), W, int(B), int(ob),
                                         ptr(addvecs), 2 * ob + 1, K, ptr(ws), ws.numel() if ws is not None else 0,
                                         stream_ptr()))


def per_row_prologue(trees, upd_idx, upd_leaf, replay_obs0, ring_start, words_row, B, pool_obs0, addvec,
                     sample=None, upload=None, td=None):
    """Pending priority update (trees.set) + the row's D-input sums in one launch (samx_per_row_prologue).
    words_row: int64 [W] with idx_p at [:B] and e_start at [2B]; addvec: float64 [2*ob+1].
    sample = (num_entries, u f64[B], beta, idxs_out i64[B], iws_out f64[B], iws32_out f32[B] or None): the
    row's stratified sample on the updated trees in the same launch.  upload = (pinned int64 host tensor,
    device int64 tensor of the same size, skip_lo, skip_n): copied host -> device by the kernel before it reads
    anything, except words [skip_lo, skip_lo + skip_n) (where idxs_out may live); the other arguments may be
    views of the device tensor.
    td = (upd_td f32[n], alpha, num_demos, demos_eps, state f64[2]) instead of upd_leaf: the pending update as TD
    errors, refreshed on the device (samx_per_row_prologue_td; no upload)."""
    if td is not None:
        assert upload is None and upd_leaf is None
        upd_td, alpha, num_demos, demos_eps, state = td
        n = upd_idx.numel()
        _chk(upd_idx, I64, (n,), 'upd_idx'); _chk(upd_td, F32, (n,), 'upd_td'); _chk(state, F64, (2,), 'state')
        if replay_obs0 is None:   # no D-input sums in this launch
            ring_len = ob = 0
            pool_obs0 = addvec = None
        else:
            ring_len, ob = replay_obs0.shape
            _chk(addvec, F64, (2 * ob + 1,), 'addvec')
        n_ent, u, beta, idxs_out, iws_out, iws32_out = sample
        _chk(u, F64, (B,), 'u'); _chk(idxs_out, I64, (B,), 'idxs_out'); _chk(iws_out, F64, (B,), 'iws_out')
        _opt(iws32_out, F32, (B,), 'iws32_out')
        check(_L().samx_per_row_prologue_td(
            ptr(trees.sum), ptr(trees.min), ptr(trees.stamp), trees.cap, ptr(upd_idx), ptr(upd_td), n, float(alpha),
            int(num_demos), float(demos_eps), ptr(state), ptr(replay_obs0), int(ring_start), int(ring_len),
            C.c_void_p(words_row.data_ptr()) if ob else None, ptr(pool_obs0),
            C.c_void_p(words_row.data_ptr() + 8 * 2 * B) if ob else None, int(B),
            int(ob), ptr(addvec), int(n_ent), ptr(u), float(beta), ptr(idxs_out), ptr(iws_out), ptr(iws32_out),
            stream_ptr()))
        return
    src = dst = None
    n_copy = skip_lo = skip_n = 0
    if upload is not None:
        src, dst, skip_lo, skip_n = upload
        assert src.is_pinned() and dst.is_cuda and src.dtype == dst.dtype == I64 and src.numel() == dst.numel()
        assert src.is_contiguous() and dst.is_contiguous()
        n_copy = src.numel()
    ring_len, ob = replay_obs0.shape
    n = upd_idx.numel()
    _chk(upd_idx, I64, (n,), 'upd_idx'); _chk(upd_leaf, F64, (n,), 'upd_leaf')
    _chk(addvec, F64, (2 * ob + 1,), 'addvec')
    n_ent, u, beta, idxs_out, iws_out, iws32_out = sample if sample is not None else (0, None, 1.0, None, None, None)
    if sample is not None:
        _chk(u, F64, (B,), 'u'); _chk(idxs_out, I64, (B,), 'idxs_out'); _chk(iws_out, F64, (B,), 'iws_out')
        _opt(iws32_out, F32, (B,), 'iws32_out')
    check(_L().samx_per_row_prologue(ptr(trees.sum), ptr(trees.min), ptr(trees.stamp), trees.cap, ptr(upd_idx),
                                     ptr(upd_leaf), n, ptr(replay_obs0), int(ring_start), int(ring_len),
                                     C.c_void_p(words_row.data_ptr()), ptr(pool_obs0),
                                     C.c_void_p(words_row.data_ptr() + 8 * 2 * B), int(B), int(ob), ptr(addvec),
                                     int(n_ent), ptr(u), float(beta), ptr(idxs_out), ptr(iws_out), ptr(iws32_out),
                                     ptr(src), ptr(dst), int(n_copy), int(skip_lo), int(skip_n),
                                     stream_ptr()))


def to_host_pair(a, a_host, b, b_host):
    """a (float32, device) -> a_host, b (int64, device) -> b_host; the *_host tensors are pinned."""
    n = a.numel()
    assert a_host.is_pinned() and b_host.is_pinned() and b.numel() == n
    check(_L().samx_to_host_pair(ptr(a), C.c_void_p(a_host.data_ptr()), ptr(b), C.c_void_p(b_host.data_ptr()), n,
                                 stream_ptr()))


def make_step_out(B, device='cuda'):
    o = _lib.StepOut()
    o.t = dict(d_losses=torch.zeros(8, dtype=F32, device=device),
               actor_losses=torch.zeros(3, dtype=F32, device=device),
               critic_losses=torch.zeros(4, dtype=F32, device=device),
               targ_q=torch.zeros(B, dtype=F32, device=device), td=torch.zeros(B, dtype=F32, device=device),
               targ_q_n=torch.zeros(B, dtype=F32, device=device))
    for k, t in o.t.items():
        setattr(o, k, t.data_ptr())
    return o


def copy_words(src, dst, rows, row_len=None, row_stride=None):
    """The first row_len int64 words of `rows` rows (row_stride words apart; default: one contiguous run of `rows`
    words) of `src` (a pinned host tensor or a device tensor) into the same places of `dst` (device), by a kernel on
    the current stream (samx_copy_words)."""
    if row_len is None:
        rows, row_len, row_stride = 1, int(rows), int(rows)
    assert src.dtype == torch.int64 and dst.dtype == torch.int64 and dst.is_cuda
    need = (rows - 1) * row_stride + row_len if rows else 0
    assert src.is_contiguous() and dst.is_contiguous() and need <= src.numel() and need <= dst.numel()
    check(_L().samx_copy_words(src.data_ptr(), dst.data_ptr(), int(rows), int(row_len), int(row_stride), stream_ptr()))


def step_grads(L, batch, out):
    ws = L.ws(batch.B)
    check(_L().samx_step_grads(C.byref(L.c), C.byref(batch), C.byref(out), ptr(ws), ws.numel(),
                               stream_ptr()))


def step_grads_part(L, batch, out, part):
    """Measurement aid (include/samx.h): part 1 = network passes, 2 = reduction/losses, 3 = both."""
    ws = L.ws(batch.B)
    check(_L().samx_step_grads_part(C.byref(L.c), C.byref(batch), C.byref(out), ptr(ws), ws.numel(),
                                    int(part), stream_ptr()))


def step_apply(L, B, mode, grad_scale=1.0):
    ws = L.ws(B)
    check(_L().samx_step_apply(C.byref(L.c), int(B), int(mode), float(grad_scale), ptr(ws), ws.numel(),
                               stream_ptr()))


def wgrad_partials_f32(A, B, M, N, R, nsplit, a_ones_rows=-1, kernel=1, overread_ok=False):
    """partials[nsplit, M(+1), N] of the split weight-gradient product A[:R, :M]^T B[:R, :N] (row-major operands with
    their own row strides), by the register-direct kernel (1) or the staged one (0) — samx_wgrad_partials_f32."""
    Mo = M + (1 if a_ones_rows >= 0 else 0)
    P = torch.empty(nsplit, Mo, N, dtype=torch.float32, device=A.device)
    check(_L().samx_wgrad_partials_f32(ptr(A), A.stride(0), ptr(B), B.stride(0), ptr(P), M, N, R, a_ones_rows, nsplit,
                                       int(kernel), int(bool(overread_ok)), stream_ptr()))
    return P


def gemm_f32(A, a_strides, B, b_strides, M, N, R, epi=0, bias=None, aux=None, add=None, alpha=0.0,
             a_ones_rows=-1, nsplit=1, out=None, overread_ok=False):
    """C[M(+1), N] = epi(sum_r A(i,r) B(r,j)) with explicit (free, reduction) / (reduction, free) element
    strides — the building block of the layer-wise path (samx_gemm_f32)."""
    Mo = M + (1 if a_ones_rows >= 0 else 0)
    C_ = out if out is not None else torch.empty(Mo, N, dtype=torch.float32, device=A.device)
    partials = torch.empty(nsplit * Mo * N, dtype=torch.float32, device=A.device) if nsplit > 1 else None
    check(_L().samx_gemm_f32(ptr(A), a_strides[0], a_strides[1], ptr(B), b_strides[0], b_strides[1], ptr(C_),
                             C_.stride(0), M, N, R, a_ones_rows, epi, ptr(bias), ptr(aux),
                             aux.stride(0) if aux is not None else 0, ptr(add),
                             add.stride(0) if add is not None else 0, float(alpha), nsplit, ptr(partials),
                             int(bool(overread_ok)), stream_ptr()))
    return C_
