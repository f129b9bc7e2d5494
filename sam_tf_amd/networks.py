"""Flat-parameter network state for ActorNN / CriticNN / RewardNN (helpers/networks.py:222-422).

A network is a description (layer sizes, layernorm flag) plus ONE float32 vector in HBM in the
reference's variable order; `names`/`slices` give the per-variable views the reference exposes as
`trainable_vars`.  Initialisers follow networks.py:74-80,369 and network_util.py:180-195.
"""
import math
from collections import OrderedDict

import numpy as np
import torch

from . import ops


def var_slices(desc):
    out, off = OrderedDict(), 0

    def add(name, *shape):
        nonlocal off
        out[name] = (off, shape)
        off += int(np.prod(shape))
    add('fc1/kernel', desc.in_dim, desc.h1); add('fc1/bias', desc.h1)
    if desc.layernorm:
        add('layernorm1/beta', desc.h1); add('layernorm1/gamma', desc.h1)
    add('fc2/kernel', desc.h1, desc.h2); add('fc2/bias', desc.h2)
    if desc.layernorm:
        add('layernorm2/beta', desc.h2); add('layernorm2/gamma', desc.h2)
    add('final/kernel', desc.h2, desc.out_dim); add('final/bias', desc.out_dim)
    return out


def _truncated_normal(rs, shape, std):
    x = rs.randn(*shape)
    bad = np.abs(x) > 2
    while bad.any():
        x[bad] = rs.randn(int(bad.sum()))
        bad = np.abs(x) > 2
    return x * std


def init_flat(desc, rs, kind, hid_w_init='he_normal'):
    """kind: 'actor' | 'critic' | 'reward'.  The critic's HIDDEN kernels use the output
    initialiser U(-3e-3, 3e-3) (networks.py:369) — reproduced."""
    sl = var_slices(desc)
    p = np.zeros(ops.param_count(desc), dtype=np.float32)

    def put(name, val):
        o, s = sl[name]
        p[o:o + val.size] = np.asarray(val, dtype=np.float32).ravel()
    for name in ('fc1/kernel', 'fc2/kernel'):
        shape = sl[name][1]
        fan_in, fan_out = float(shape[0]), float(shape[1])
        if kind == 'critic':
            put(name, rs.uniform(-3e-3, 3e-3, shape))
        elif hid_w_init == 'he_normal':
            put(name, _truncated_normal(rs, shape, math.sqrt(1.3 * 2.0 / fan_in)))
        elif hid_w_init == 'he_uniform':
            b = math.sqrt(6.0 / fan_in); put(name, rs.uniform(-b, b, shape))
        elif hid_w_init == 'xavier_normal':
            put(name, _truncated_normal(rs, shape, math.sqrt(1.3 * 2.0 / (fan_in + fan_out))))
        elif hid_w_init == 'xavier_uniform':
            b = math.sqrt(6.0 / (fan_in + fan_out)); put(name, rs.uniform(-b, b, shape))
        else:
            raise RuntimeError("unknown weight init: '{}'".format(hid_w_init))
    put('final/kernel', rs.uniform(-3e-3, 3e-3, sl['final/kernel'][1]))
    if desc.layernorm:
        put('layernorm1/gamma', np.ones(desc.h1)); put('layernorm2/gamma', np.ones(desc.h2))
    return p


class Var(object):
    """A named view into a flat parameter vector (stands in for a tf.Variable)."""

    def __init__(self, name, tensor):
        self.name, self.tensor = name, tensor
        self.shape = tuple(tensor.shape)

    def eval(self):
        return self.tensor.cpu().numpy()


class FlatNet(object):

    def __init__(self, scope, name, desc, flat):
        self.scope, self.name, self.desc = scope, name, desc
        self.flat = flat  # 1-D float32 CUDA tensor (possibly a view into a larger buffer)
        assert flat.numel() == ops.param_count(desc)
        self.slices = var_slices(desc)

    @property
    def trainable_vars(self):
        return [Var('%s/%s/%s:0' % (self.scope, self.name, k),
                    self.flat[o:o + int(np.prod(s))].view(*s))
                for k, (o, s) in self.slices.items()]

    vars = trainable_vars

    def set_flat(self, values):
        self.flat.copy_(torch.as_tensor(np.asarray(values, dtype=np.float32)).to(self.flat.device))

    def get_flat(self):
        return self.flat.cpu().numpy()
