"""ctypes binding of libsamx.so (include/samx.h).  There is no fallback: if the HIP extension is
missing or a call fails, the product path raises."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(_HERE, 'libsamx.so')


class SamxError(RuntimeError):
    pass


class MlpDesc(C.Structure):
    """Mirror of `struct samx_mlp`."""
    _fields_ = [('in_dim', C.c_int32), ('h1', C.c_int32), ('h2', C.c_int32),
                ('out_dim', C.c_int32), ('layernorm', C.c_int32)]

    def __repr__(self):
        return 'MlpDesc(in=%d, h1=%d, h2=%d, out=%d, ln=%d)' % (
            self.in_dim, self.h1, self.h2, self.out_dim, self.layernorm)


_P, _I64, _I32, _F, _D, _SZ = C.c_void_p, C.c_int64, C.c_int32, C.c_float, C.c_double, C.c_size_t
_MP = C.POINTER(MlpDesc)


class Learner(C.Structure):
    """Mirror of `struct samx_learner`."""
    _fields_ = ([('actor', MlpDesc), ('critic', MlpDesc), ('reward', MlpDesc),
                 ('ob_dim', C.c_int32), ('ac_dim', C.c_int32)] +
                [(k, C.c_void_p) for k in ('params', 'grads', 'clipped', 'adam_m', 'adam_v', 'targets',
                                           'steps', 'actor_reward_params', 'mu_mean', 'mu_std', 'd_mean',
                                           'd_std')] +
                [(k, C.c_float) for k in ('ac_high', 'gamma', 'polyak', 'clip_norm', 'actor_lr',
                                          'critic_lr', 'd_lr', 'beta1', 'beta2', 'adam_eps',
                                          'td_loss_scale', 'wd_scale', 'q_actor_loss_scale',
                                          'd_actor_loss_scale', 'ent_reg_scale')] +
                [('ret_mean', C.c_void_p), ('ret_std', C.c_void_p), ('ret_sums', C.c_void_p),
                 ('enable_popart', C.c_int32), ('n_step', C.c_int32), ('td_loss_n_scale', C.c_float)])


class Batch(C.Structure):
    """Mirror of `struct samx_batch`."""
    _fields_ = ([('B', C.c_int64)] + [(k, C.c_void_p) for k in (
        'p_obs', 'p_acs', 'e_obs', 'e_acs', 'eps_ob', 'eps_ac', 'real_labels', 'fake_labels', 'obs0',
        'acs', 'rews', 'dones1', 'obs1', 'iws', 'td_len', 'p_idx', 'a_idx', 'e_start')] +
                [('ring_start', C.c_int64), ('ring_len', C.c_int64), ('rng_seed', C.c_uint64),
                 ('label_smoothing', C.c_int32), ('d_rms_sums', C.c_void_p), ('d_rms_addvec', C.c_void_p),
                 ('num_entries', C.c_int64), ('n_end', C.c_void_p), ('n_rews', C.c_void_p),
                 ('n_dones', C.c_void_p), ('n_len', C.c_void_p), ('ret_addvec', C.c_void_p)])


class StepOut(C.Structure):
    """Mirror of `struct samx_step_out`."""
    _fields_ = [(k, C.c_void_p) for k in ('d_losses', 'actor_losses', 'critic_losses', 'targ_q', 'td',
                                          'targ_q_n')]


class RetNorm(C.Structure):
    """Mirror of `struct samx_retnorm`."""
    _fields_ = [('mean', C.c_void_p), ('std', C.c_void_p)]


_LP, _BP, _OP, _RP = C.POINTER(Learner), C.POINTER(Batch), C.POINTER(StepOut), C.POINTER(RetNorm)

# name -> (restype, argtypes); must list every symbol include/samx.h declares
SIGNATURES = {
    'samx_last_error': (C.c_char_p, []),
    'samx_version': (C.c_int, []),
    'samx_stream_synchronize': (C.c_int, [_P]),
    'samx_mlp_param_count': (_I64, [_MP]),
    'samx_tree_init': (C.c_int, [_P, _P, _P, _I64, _P]),
    'samx_tree_set': (C.c_int, [_P, _P, _P, _I64, _P, _P, _I64, _P]),
    'samx_tree_reduce': (C.c_int, [_P, _I64, C.c_int, _I64, _I64, _P, _P]),
    'samx_tree_find_prefixsum': (C.c_int, [_P, _I64, _P, _I64, _P, _P]),
    'samx_per_sample': (C.c_int, [_P, _P, _I64, _I64, _P, _I64, _D, _P, _P, _P, _P]),
    'samx_per_leaf_values_host': (C.c_int, [_P, _P, _I64, _D, _I64, _D, _P, _P]),
    'samx_per_row_prologue': (C.c_int, [_P, _P, _P, _I64, _P, _P, _I64, _P, _I64, _I64, _P, _P, _P, _I64,
                                        C.c_int, _P, _I64, _P, _D, _P, _P, _P, _P, _P, _I64, _I64, _I64, _P]),
    'samx_per_row_prologue_td': (C.c_int, [_P, _P, _P, _I64, _P, _P, _I64, _D, _I64, _D, _P, _P, _I64, _I64, _P, _P,
                                           _P, _I64, C.c_int, _P, _I64, _P, _D, _P, _P, _P, _P]),
    'samx_per_leaves_from_td': (C.c_int, [_P, _P, _I64, _I64, _I64, _D, _I64, _D, _P, _P, _P]),
    'samx_per_leaves_from_td_geom': (C.c_int, [_P, _P, _I64, _I64, _P, _D, _I64, _D, _P, _P, _P]),
    'samx_per_sample_rows': (C.c_int, [_P, _P, _I64, _P, _I64, _P, _I64, _D, _P, _P, _P, _P, _P, _P]),
    'samx_per_row_prologue_td_rows': (C.c_int, [_P, _P, _P, _I64, _P, _P, _I64, _D, _I64, _D, _P, _P, _I64, _P, _I64,
                                                _D, _P, _P, _P, _P, _P, _P]),
    'samx_pow_device': (C.c_int, [_P, _D, _I64, _P, _P]),
    'samx_pow_device_f32_range': (C.c_int, [C.c_uint32, _I64, _D, _D, _P, _P]),
    'samx_pow_host_f32_range': (C.c_int, [C.c_uint32, _I64, _D, _D, _P]),
    'samx_pow_host': (C.c_int, [_P, _D, _I64, _P]),
    'samx_pow_selfcheck': (_I64, [_I64]),
    'samx_host_mt19937_randint': (C.c_int, [_P, C.c_uint64, _I64, _I64, _P, _I64]),
    'samx_to_host_pair': (C.c_int, [_P, _P, _P, _P, _I64, _P]),
    'samx_per_leaves_from_td_host': (C.c_int, [_P, _P, _I64, _I64, _D, _I64, _D, _P, _P]),
    'samx_copy_words': (C.c_int, [_P, _P, _I64, _I64, _I64, _P]),
    'samx_replay_gather': (C.c_int, [_P, _P, _P, _P, _P, _I64, _I64, C.c_int, C.c_int, _P, _I64,
                                     _P, _P, _P, _P, _P, _P]),
    'samx_replay_write': (C.c_int, [_P, _P, _P, _P, _P, C.c_int, C.c_int, _P, _P, _I64, _P]),
    'samx_rms_batch_sums': (C.c_int, [_P, _I64, _P, _I64, C.c_int, _P, _P]),
    'samx_rms_commit': (C.c_int, [_P, _P, C.c_int, _P, _P, _P]),
    'samx_adam_partials': (C.c_int, []),
    'samx_clip_adam': (C.c_int, [_P, _P, _P, _P, _I64, _F, _F, _F, _F, _I64, _F, _F, _P, _F, _P,
                                 _P, _P]),
    'samx_polyak': (C.c_int, [_P, _P, _I64, _F, _P]),
    'samx_clip_copy': (C.c_int, [_P, _I64, _F, _P, _P, _P]),
    'samx_count_mismatch': (C.c_int, [_P, _P, _I64, _P, _P]),
    'samx_perturb_params': (C.c_int, [_MP, _P, _P, _F, C.c_uint64, C.c_uint64, _P]),
    'samx_rms_distance': (C.c_int, [_P, _P, _I64, _P, _P]),
    'samx_gemm_f32': (C.c_int, [_P, _I64, _I64, _P, _I64, _I64, _P, _I64, C.c_int32, C.c_int32, C.c_int32,
                                C.c_int32, C.c_int32, _P, _P, _I64, _P, _I64, C.c_float, C.c_int32, _P, C.c_int32, _P]),
    'samx_wgrad_partials_f32': (C.c_int, [_P, _I64, _P, _I64, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                          C.c_int32, C.c_int32, C.c_int32, _P]),
    'samx_nets_ws_bytes': (_SZ, [_MP, _MP, _MP, _I64]),
    'samx_act_q': (C.c_int, [_MP, _P, _MP, _P, _P, _P, _P, _I64, _F, _P, _P, _RP, _P, _SZ, _P]),
    'samx_targ_q': (C.c_int, [_MP, _P, _MP, _P, _P, _P, _P, _P, _P, _P, _I64, _F, _F, _P, _RP, _P, _SZ,
                              _P]),
    'samx_critic_grad': (C.c_int, [_MP, _P, _P, _P, _P, _P, _P, _P, _I64, C.c_int, _F, _F, _P, _P,
                                   _P, _RP, _P, _F, _P, _P, _SZ, _P]),
    'samx_td_errors': (C.c_int, [_MP, _P, _P, _P, _P, _P, _P, _I64, C.c_int, _P, _RP, _P, _SZ, _P]),
    'samx_td_errors_rows_ws_bytes': (_SZ, [_MP, _I64]),
    'samx_td_errors_rows': (C.c_int, [_MP, _P, _P, _P, _P, _I64, _I64, _P, _P, _P, _I64, C.c_int, _P, _P, _P, _RP, _P,
                                      _SZ, _P]),
    'samx_actor_grad': (C.c_int, [_MP, _P, _MP, _P, _MP, _P, _P, _P, _P, _I64, _F, _F, _F, _P, _P,
                                  _RP, _P, _SZ, _P]),
    'samx_ret_update_popart': (C.c_int, [_P, _P, _P, _P, _P, C.c_int, _MP, _P, _P, _P]),
    'samx_nstep_lookahead': (C.c_int, [_P, _P, _I64, _I64, _I64, _P, _I64, C.c_int, _D, _P, _P, _P, _P,
                                       _P, _P]),
    'samx_nstep_lookahead_many': (C.c_int, [_P, _P, _I64, _I64, _I64, _P, _I64, _I64, _I64, C.c_int, _D, _P, _P,
                                            _P, _P, _P]),
    'samx_nstep_lookahead_many_rows': (C.c_int, [_P, _P, _I64, _P, _P, _I64, _I64, _I64, C.c_int, _D, _P, _P, _P, _P, _P]),
    'samx_disc_ws_bytes': (_SZ, [_MP, _I64]),
    'samx_disc_grad': (C.c_int, [_MP, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I64, C.c_int,
                                 _F, _P, _P, _P, _SZ, _P]),
    'samx_reward': (C.c_int, [_MP, _P, _P, _P, _P, _P, _I64, C.c_int, C.c_int, _P, _P, _SZ, _P]),
    'samx_fused_supported': (C.c_int, [_LP, _I64]),
    'samx_step_path': (C.c_int, [_LP, _I64]),
    'samx_disc_randoms': (C.c_int, [C.c_uint64, _I64, _I64, C.c_int, C.c_int, _P, _P, _P, _P, _P]),
    'samx_d_input_stats': (C.c_int, [_P, _I64, _I64, _P, _P, _P, _I64, C.c_int, _P, C.c_int, _P, _P, _P,
                                     _P, _SZ, _P]),
    'samx_d_input_stats_ws_bytes': (_SZ, [_I64, C.c_int]),
    'samx_d_input_stats_static_max_ob': (C.c_int, []),
    'samx_d_input_stats_static': (C.c_int, [_P, _I64, _I64, _P, _P, _P, _I64, C.c_int, _P, C.c_int, _P, _P, _P,
                                            _P]),
    'samx_d_input_sums_many': (C.c_int, [_P, _I64, _I64, _P, _P, _P, _I64, _I64, C.c_int, _P, _I64, _I64, _P]),
    'samx_d_input_sums_many_ws': (C.c_int, [_P, _I64, _I64, _P, _P, _P, _I64, _I64, C.c_int, _P, _I64, _I64, _P,
                                            _SZ, _P]),
    'samx_step_ws_bytes': (_SZ, [_LP, _I64]),
    'samx_step_grads': (C.c_int, [_LP, _BP, _OP, _P, _SZ, _P]),
    'samx_step_grads_part': (C.c_int, [_LP, _BP, _OP, _P, _SZ, C.c_int, _P]),
    'samx_step_apply': (C.c_int, [_LP, _I64, C.c_int, _F, _P, _SZ, _P]),
}

_lib = None


def lib():
    """The loaded library; raises SamxError when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise SamxError('libsamx.so not found at %s: build it with `python -m sam_tf_amd.build` '
                            '(there is no CPU fallback)' % SO_PATH)
        # torch ships its own HIP runtime (torch/lib/libamdhip64.so).  It must be resident BEFORE
        # libsamx.so is opened so that both bind to the same runtime instance: streams and device
        # pointers are only meaningful inside the runtime that created them.
        import torch  # noqa: F401
        L = C.CDLL(SO_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if the .so is stale
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        raise SamxError('libsamx error %d: %s' % (rc, lib().samx_last_error().decode()))


def ptr(t):
    """Device/host pointer of a torch tensor or numpy array (None -> NULL)."""
    if t is None:
        return None
    if hasattr(t, 'data_ptr'):
        assert t.is_contiguous(), 'libsamx needs contiguous tensors'
        return C.c_void_p(t.data_ptr())
    assert t.flags['C_CONTIGUOUS']
    return C.c_void_p(t.ctypes.data)


_RAW = None


def stream_ptr():
    """torch's current HIP stream on the current device as a void* (the raw-handle getter: torch.cuda.current_stream()
    builds a Stream object and costs ~8 us per call, which adds up over the 15 library calls of an object-API step)."""
    global _RAW
    if _RAW is None:
        import torch
        _RAW = (torch._C._cuda_getCurrentRawStream, torch._C._cuda_getDevice)
    return C.c_void_p(_RAW[0](_RAW[1]()))


def stream_sync():
    """Wait for torch's current stream (hipStreamSynchronize through the library: no Stream object)."""
    check(lib().samx_stream_synchronize(stream_ptr()))
