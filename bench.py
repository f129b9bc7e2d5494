#!/usr/bin/env python
"""SAM update steps/sec on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config C2] [--no-graph] [--no-cpu]

One "step" = 1 discriminator update + 1 actor/critic update (+ PER refresh) + Polyak target update on
one replayed minibatch (sam_tf_amd/driver.py; reference loop body sam.py:323-433 with
d_steps = g_steps = 1).  Inputs (replay storage, expert pool, weights, optimiser state) are resident
in HBM before the timed region.  N > 1: one process per GPU (torch.distributed / RCCL), each rank its
own replay shard and minibatch (weak scaling), gradients and running-mean sums all-reduced (RCCL on
the launch stream, sam_tf_amd/rccl.py; captured inside the K-step graph when that plan completes).

Prints ONE JSON line on rank 0 (contract in the task statement), with two extra objects:
  roofline     — the dominant device pass of the step, timed with HIP events on the launch stream
  cpu_baseline — the CPU restatement of the reference (oracle/, reference call structure, 1 thread)
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# (ob, ac, h1, h2, batch, prioritized, mem_size)
CONFIGS = {
    'C1': dict(ob=4, ac=1, h=(64, 64), B=32, per=False, mem=100000, name='InvertedPendulum-shaped'),
    'C2': dict(ob=11, ac=3, h=(64, 64), B=256, per=False, mem=1000000, name='Hopper-shaped'),
    'C3': dict(ob=17, ac=6, h=(64, 64), B=256, per=True, mem=1000000, name='HalfCheetah-shaped'),
    'C4': dict(ob=376, ac=17, h=(256, 256), B=1024, per=False, mem=1000000, name='Humanoid-shaped'),
    'C5': dict(ob=376, ac=17, h=(400, 300), B=8192, per=True, mem=4000000, name='Humanoid-stress'),
}
PEAK_FP32_TFLOPS = 157.3   # MI355X_MICROARCH.md: fp32 vector = fp32 MFMA (exact f32) peak
PEAK_HBM_GBS = 8000.0


def weight_elems(ob, ac, h):
    wa = ob * h[0] + h[0] * h[1] + h[1] * ac
    wc = (ob + ac) * h[0] + h[0] * h[1] + h[1]
    return wa, wc, wc


def flops_per_step(cfg):
    """F = B*(8 Wa + 12 Wc [+2 Wc if PER] + 22 Wd)."""
    wa, wc, wd = weight_elems(cfg['ob'], cfg['ac'], cfg['h'])
    return cfg['B'] * (8 * wa + (14 if cfg['per'] else 12) * wc + 22 * wd)


def role_flops(cfg):  # per network pass, SURVEY.md §8(d)
    wa, wc, wd = weight_elems(cfg['ob'], cfg['ac'], cfg['h'])
    B = cfg['B']
    return {'disc_grad': 22 * wd * B, 'targ_q': 2 * (wa + wc) * B, 'actor_grad': (6 * wa + 4 * wc) * B,
            'critic_grad': 6 * wc * B}


def build_learner(cfg, comm, device, seed, use_graph):
    from sam_tf_amd.hps import make_hps, SyntheticEnv
    from sam_tf_amd.discriminator import Discriminator
    from sam_tf_amd.agent import SAMAgent
    from sam_tf_amd.demo_pool import DevicePairPool
    from sam_tf_amd.driver import UpdateStepDriver
    rank = comm.Get_rank()
    env = SyntheticEnv(cfg['ob'], cfg['ac'], 1.0)
    hps = make_hps(hid_widths=cfg['h'], batch_size=cfg['B'], mem_size=cfg['mem'],
                   prioritized_replay=cfg['per'], seed=seed,
                   with_layernorm=bool(cfg.get('ln', False)),
                   rmsify_rets=bool(cfg.get('ret', False)), enable_popart=bool(cfg.get('ret', False)),
                   n_step_returns=bool(cfg.get('nstep', 0)), n=int(cfg.get('nstep', 0)) or 96,
                   **cfg.get('hps', {}))
    d = Discriminator('d', env, hps, comm, device=device)
    mu = SAMAgent('mu', comm, env, hps, d, device=device)
    # synthetic transitions: obs ~ N(0,1), acs ~ U(-1,1), rews ~ U(0,5),
    # dones ~ Bernoulli(1/1000); each rank its own shard (run_sam.py:30-34 seeds by rank)
    g = torch.Generator(device=device); g.manual_seed(1234 + rank)
    n = min(cfg['mem'], 1 << 20)
    r = lambda *s: torch.randn(*s, generator=g, device=device)
    u = lambda *s: torch.rand(*s, generator=g, device=device)
    obs0, obs1 = r(n, cfg['ob']), r(n, cfg['ob'])
    acs_, rews_, dones_ = u(n, cfg['ac']) * 2 - 1, u(n) * 5, (u(n) < cfg.get('p_done', 1e-3)).float()
    n_demo = int(cfg.get('demos', 0))
    if n_demo:
        # add_demos_to_mem (sam.py:244-246, memory.py:186-198): the first n_demo transitions are pinned in front of
        # the ring as demonstrations; the rest go in one by one as rollouts would store them (the bulk fill assumes a
        # ring without demonstrations), plus cfg['demos_extra'] more so that the ring's start moves inside its span
        import types
        c = lambda t: t.cpu().numpy()
        host = [c(t) for t in (obs0, acs_, rews_, obs1, dones_)]
        mu.replay_buffer.add_demo_transitions_to_mem(types.SimpleNamespace(
            obs0=host[0][:n_demo], acs=host[1][:n_demo], env_rews=host[2][:n_demo], obs1=host[3][:n_demo],
            dones1=host[4][:n_demo]))
        for i in range(n_demo, n + int(cfg.get('demos_extra', 0))):
            mu.replay_buffer.append(*(h_[i % n] for h_ in host))
    else:
        mu.replay_buffer.extend_device(obs0, acs_, rews_, obs1, dones_)
    if hps.rmsify_obs:  # what store_transition would have accumulated (sam_agent.py:627-629)
        mu.obs_rms.update_device(obs0, None, comm)
    del obs0, obs1
    rs = np.random.RandomState(4321 + rank)
    np.random.seed(seed + 1000003 * rank)
    import random
    random.seed(seed + 1000003 * rank)
    pool = DevicePairPool(rs.randn(32000, cfg['ob']) + 0.5, rs.uniform(-1, 1, (32000, cfg['ac'])),
                          device=device)
    mu.initialize_target_nets()
    mu.sync_from_root()
    d.sync_from_root()
    if cfg['per']:  # warm the priorities (200 rounds of |N(0,1)| + 1e-6)
        rb = mu.replay_buffer
        for _ in range(200):
            idx = rs.randint(0, rb.num_entries, size=cfg['B'])
            rb.update_priorities(idx, np.abs(rs.randn(cfg['B'])) + 1e-6)
    drv = UpdateStepDriver(mu, d, pool, use_graph=use_graph, d_steps=int(cfg.get('d_steps', 1)),
                           g_steps=int(cfg.get('g_steps', 1)), window=cfg.get('window'),
                           pn_adapt_frequency=cfg.get('pn_adapt_frequency'),
                           steps_per_launch=int(os.environ.get('SAMX_STEPS_PER_LAUNCH', 0)) or
                           # measured on MI355X (tools/exp/rows_per_launch_sweep*.sh): 64 rows of the 2-launch plain step
                           # (128 nodes; 32 -> 64 rows +3.8 %, 128 rows +1 % more), 32 rows of the 3..5-launch rows of the
                           # option variants (+1.5..2.5 % over 16), 16 rows with prioritized replay (5-6 launches per row;
                           # 24 / 32 measure the same)
                           (16 if cfg['per'] else
                            (64 if not (cfg.get('ret') or cfg.get('nstep') or cfg.get('ln')) else 32)))
    return mu, d, pool, drv


def object_api_loop(mu, d, expert_dataset, comm, n_training_steps, d_steps=1, g_steps=1):
    """The body of the reference's training loop, call for call (sam.py:369-433, window = None, no parameter
    noise): per d_step  sample(_uniform) -> get_next_pair_batch -> d.obs_rms.update -> d.get_grads ->
    d.get_losses -> d.optimize ; per g_step  mu.train() -> mu.update_target_net().  Everything the reference
    receives as numpy / python objects is received as such here (and kept, as its deques do)."""
    batch_size = mu.hps.batch_size
    keep = []
    for _ in range(n_training_steps):
        for _ in range(d_steps):
            sample_ = mu.replay_buffer.sample
            if hasattr(mu.replay_buffer, 'sample_uniform'):
                sample_ = mu.replay_buffer.sample_uniform
            xp_batch_ = sample_(batch_size=batch_size)
            ob_pol_, ac_pol_ = xp_batch_['obs0'], xp_batch_['acs']
            ob_exp_, ac_exp_ = expert_dataset.get_next_pair_batch(batch_size=batch_size)
            assert len(ob_exp_) == len(ob_pol_), "length mismatch"
            if hasattr(d, "obs_rms"):
                d.obs_rms.update(np.concatenate((ob_pol_, ob_exp_), axis=0), comm)
            feeds_ = {d.p_obs: ob_pol_, d.p_acs: ac_pol_, d.e_obs: ob_exp_, d.e_acs: ac_exp_}
            d_grads = d.get_grads(feeds_)
            d_losses = d.get_losses(feeds_)
            d.optimize(feeds_)
            keep = [d_grads, d_losses]
        for _ in range(g_steps):
            losses_and_grads = mu.train()
            keep.append(losses_and_grads)
            mu.update_target_net()
    return keep


class HostPairs(object):
    """expert_dataset of the loop above: the reference's host-side PairDataset behind get_next_pair_batch
    (demo_dataset.py:6-41,94-101)."""

    def __init__(self, obs0, acs):
        from sam_tf_amd.demo_dataset import PairDataset
        self.pairs = PairDataset(np.asarray(obs0, dtype=np.float32), np.asarray(acs, dtype=np.float32), True)

    def get_next_pair_batch(self, batch_size):
        return self.pairs.get_next_batch(batch_size)


def object_api_rate(cfg, comm, device, seed, n, warm, d_steps=1, g_steps=1):
    """Update steps/s through the OBJECT API (the drop-in surface, one host synchronisation per returned
    object): the loop body of sam.py:369-433 with d_steps = g_steps = 1, synthetic inputs as everywhere else."""
    import gc
    mu, d, pool, drv = build_learner(cfg, comm, device, seed, False)
    del drv
    rs = np.random.RandomState(4321)
    data = HostPairs(rs.randn(32000, cfg['ob']) + 0.5, rs.uniform(-1, 1, (32000, cfg['ac'])))
    object_api_loop(mu, d, data, comm, warm, d_steps, g_steps)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    object_api_loop(mu, d, data, comm, n, d_steps, g_steps)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    ok = all(bool(torch.isfinite(t).all()) for t in (mu.params, d.params))
    path = {0: 'layer-wise', 1: 'fused-64', 2: 'role-fused-256'}[mu.L.path(cfg['B'])]
    del mu, d, pool
    gc.collect()
    torch.cuda.empty_cache()
    return {'steps_per_s': n / el, 'ms_per_step': el / n * 1e3, 'steps': n, 'warmup': warm, 'step_path': path,
            'finite_params': ok, 'calls_per_step': 'sample_uniform|sample, get_next_pair_batch, d.obs_rms.update, '
            'd.get_grads, d.get_losses, d.optimize, mu.train, mu.update_target_net (sam.py:369-433)'}


def time_passes(drv, cfg, iters=50):
    """Average device time of the step's passes, HIP events recorded on the launch stream (torch's
    current stream is the stream every libsamx call is handed)."""
    from sam_tf_amd import ops
    st = torch.cuda.current_stream()
    L, B = drv.L, drv.B
    keep = [L.params, L.targets, L.m, L.v, L.steps]
    if drv.d.hps.rmsify_obs:
        keep += [drv.d.obs_rms.sums, drv.d.obs_rms.mean, drv.d.obs_rms.std]
    saved = [t.clone() for t in keep]
    fns = [('network_passes', lambda: ops.step_grads_part(L, drv.batch, drv.out, 1)),
           ('reduce_losses', lambda: ops.step_grads_part(L, drv.batch, drv.out, 2)),
           ('step_apply', lambda: ops.step_apply(L, B, 3, 1.0)),
           ('d_input_sums(K steps)', lambda: (drv._seg_sums(drv.K), drv._seg_inputs()))]
    if drv.fused and drv.world == 1 and drv.apply_in_reduction:
        # what the driver's rows launch instead of the two passes above (samx_step_grads_part bit 9)
        fns.insert(3, ('reduce_losses_and_step_apply_in_one_launch',
                       lambda: ops.step_grads_part(L, drv.batch, drv.out, 2 | 512)))
    out = {}
    for name, fn in fns:
        fn(); fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(iters):
            fn()
        e1.record(st)
        e1.synchronize()
        out[name] = e0.elapsed_time(e1) / iters * 1e-3  # seconds
    for t, sv in zip(keep, saved):
        t.copy_(sv)
    return out


def pmc_traffic(config):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes
    (profiles/pmc_<config>.json, written by profiles/collect_pmc.py on the GPU box); None when that
    config has not been measured."""
    path = os.path.join(ROOT, 'profiles', 'pmc_%s.json' % config)
    if not os.path.exists(path):
        return None
    with open(path) as f:
        return json.load(f).get('traffic_bytes_per_launch')


def cpu_baseline(cfg, budget_s=12.0, n_thread=True):
    from oracle.cpu_step import CpuReferenceStep
    torch.set_num_threads(1)
    ref = CpuReferenceStep(cfg['ob'], cfg['ac'], cfg['h'], cfg['B'], per=cfg['per'],
                           mem=min(cfg['mem'], 1 << 17))
    ref.step()  # warm-up
    n, t0 = 0, time.perf_counter()
    while True:
        ref.step(); n += 1
        el = time.perf_counter() - t0
        if el > budget_s or n >= 2000:
            break
    out = {'value': n / el, 'unit': 'steps/s', 'cores': 1, 'kind': 'port',
           'sample': '%d update steps of the same workload (CPU restatement of the reference in its '
                     'own call structure: 3 D passes + 8 agent passes per step, fp32 torch, 1 thread, '
                     'Python replay loops; replay of %d rows on the CPU side against %d resident on the GPU '
                     '- fewer tree levels for prioritized replay, which favours the CPU), %.1f s'
                     % (n, min(cfg['mem'], 1 << 17), min(cfg['mem'], 1 << 20), el)}
    if not n_thread:
        return out
    # the N-thread variant beside it (the reference itself pins one thread per worker,
    # run_sam.py:18); a short sample, reported next to the 1-core figure, never instead of it
    nt = min(16, os.cpu_count() or 1)
    if nt > 1:
        torch.set_num_threads(nt)
        ref.step()
        n2, t0 = 0, time.perf_counter()
        while True:
            ref.step(); n2 += 1
            el2 = time.perf_counter() - t0
            if el2 > budget_s / 3 or n2 >= 2000:
                break
        torch.set_num_threads(1)
        out['n_thread_variant'] = {'value': n2 / el2, 'cores': nt, 'sample': '%d steps, %.1f s' % (n2, el2)}
    return out


OTHER_RUNS = (('C3', 2000, 200), ('C4', 600, 100), ('C5', 40, 8))   # (config, steps, warm-up)


def other_configs(comm, device, args):
    """Short runs of C3 / C4 / C5 (graph-captured, inputs resident, same timing rule as the headline:
    synchronize - K steps - synchronize), each with its roofline pass and a bounded CPU-restatement sample."""
    import gc
    out = {}
    for name, steps, warm in OTHER_RUNS:
        cfg = dict(CONFIGS[name], ln=False, ret=False, nstep=0)
        try:
            if os.environ.get('SAMX_BENCH_FAIL_OTHER') == name:   # test hook for the "degraded" exit below
                raise RuntimeError('forced by SAMX_BENCH_FAIL_OTHER')
            mu, d, pool, drv = build_learner(cfg, comm, device, args.seed, True)
            drv.prepare_all()
            drv.run(warm)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            drv.run(steps)
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
            passes = time_passes(drv, cfg, iters=20 if name != 'C5' else 5)
            dom = passes['network_passes']
            ach = flops_per_step(cfg) / dom / 1e12
            o = {'steps_per_s': steps / el, 'ms_per_step': el / steps * 1e3, 'steps': steps, 'warmup': warm,
                 'fused_whole_network_path': bool(drv.fused),
                 'step_path': 'fused-64' if drv.fused else ('role-fused-256' if drv.wide else 'layer-wise'),
                 'finite_params': all(bool(torch.isfinite(t).all()) for t in (mu.params, d.params)),
                 'roofline': {'bound': 'latency' if drv.fused else 'mfma', 'frac': ach / PEAK_FP32_TFLOPS,
                              'achieved_tflops': ach, 'kernel_us': dom * 1e6,
                              'flops_per_launch': flops_per_step(cfg),
                              # the same flops over the whole timed step (reduction, optimiser pass, replay /
                              # tree work, launch gaps included): the figure nothing can flatter
                              'frac_whole_step': flops_per_step(cfg) / (el / steps) / 1e12 / PEAK_FP32_TFLOPS,
                              'traffic': pmc_traffic(name)}}
            if not (drv.fused or drv.wide):
                # layer-wise path: the network passes run on three streams, so their HIP-event time is no
                # kernel's duration — `frac` is the whole-step figure there, the event-timed one is kept beside it;
                # the per-kernel figure (sum of GEMM flops / sum of GEMM kernel time) is in profiles/ (rocprofv3)
                o['roofline']['frac_network_passes_hip_events'] = o['roofline']['frac']
                o['roofline']['frac'] = o['roofline']['frac_whole_step']
                o['roofline']['achieved_tflops'] = o['roofline']['frac'] * PEAK_FP32_TFLOPS
            del mu, d, pool, drv
            gc.collect()
            torch.cuda.empty_cache()
            if not args.no_cpu:
                cb = cpu_baseline(cfg, budget_s=5.0, n_thread=False)
                o['cpu_baseline'] = {k: cb[k] for k in ('value', 'unit', 'cores', 'kind', 'sample')}
                o['vs_cpu_1core'] = o['steps_per_s'] / cb['value']
            out[name] = o
        except Exception as e:  # noqa: BLE001 - the headline line must still come out
            out[name] = {'error': '%s: %s' % (type(e).__name__, str(e)[:200])}
            gc.collect()
            torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=2000)
    ap.add_argument('--warmup', type=int, default=200)
    ap.add_argument('--config', default='C2', choices=sorted(CONFIGS))
    ap.add_argument('--no-graph', action='store_true')
    ap.add_argument('--no-cpu', action='store_true')
    ap.add_argument('--layernorm', action='store_true', help='with_layernorm nets (layer-wise path)')
    ap.add_argument('--launcher', action='store_true',
                    help='the canonical launcher options (mujoco_sam_imitate.sh): layernorm, rmsify_rets, '
                         'popart, n-step returns n=96')
    ap.add_argument('--argparse-defaults', action='store_true',
                    help="the reference's argparse defaults (argparsers.py:279-309): rmsify_rets, popart, n-step "
                         'returns n=10, no layernorm')
    ap.add_argument('--seed', type=int, default=0)
    ap.add_argument('--no-others', action='store_true',
                    help='skip the short C3/C4/C5 runs the default (C2, one GPU) invocation appends as '
                         'config.other_configs')
    ap.add_argument('--object-api', action='store_true',
                    help='only measure the update step through the object API (sam.py:369-433 call for call) for '
                         '--config, print its JSON and exit')
    ap.add_argument('--no-training-step', action='store_true',
                    help='skip the extra d_steps=1 : g_steps=3 training-step figure (config.training_steps_per_s)')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU: the update step has no CPU path')
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    rehearse = world == 1 and os.environ.get('SAMX_BENCH_REHEARSE') == '1'
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group('nccl', device_id=device)
    assert world == args.gpus or world == 1, 'launch with torchrun --nproc-per-node N for --gpus N'

    from sam_tf_amd.comm import TorchComm
    if rehearse:
        # one-GPU rehearsal of the N > 1 flow below (tools/rehearse_nccl.py): a real 1-rank RCCL group
        # carries every collective, the communicator reports 2 identical ranks.  Not a scaling number.
        import torch.distributed as dist
        from tools.rehearse_nccl import TwinOverRccl
        dist.init_process_group('nccl', init_method='tcp://127.0.0.1:29617', rank=0, world_size=1,
                                device_id=device)
        comm = TwinOverRccl(device, False)
        world = 2
    else:
        comm = TorchComm(device)
    nstep = 96 if args.launcher else (10 if args.argparse_defaults else 0)
    cfg = dict(CONFIGS[args.config], ln=args.layernorm or args.launcher, ret=args.launcher or args.argparse_defaults,
               nstep=nstep if not CONFIGS[args.config]['per'] else 0)
    if args.object_api:
        print(json.dumps({'config': args.config, 'object_api': object_api_rate(cfg, comm, device, args.seed,
                                                                                args.steps, args.warmup)}), flush=True)
        return
    mu, d, pool, drv = build_learner(cfg, comm, device, args.seed, not args.no_graph)

    def timed():
        """W warm-up steps, then EXACTLY K steps between barrier + synchronize; max over ranks."""
        drv.prepare_all()
        # the sizes the two runs below end on get plans of their own, so that e.g. 20 timed steps are ONE launch of
        # 20 rows and not 16 + 4 (capturing is not part of a step; run() only uses sizes that have a plan)
        if os.environ.get('SAMX_BENCH_TAIL_PLANS', '1') != '0':
            for k in sorted({args.steps % drv.K, args.warmup % drv.K}):
                if k > 1 and k % drv.S == 0:
                    drv.prepare(k)
        drv.run(args.warmup)
        torch.cuda.synchronize(); comm.Barrier(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        drv.run(args.steps)   # K consecutive steps per graph launch where nothing needs the host in between
        torch.cuda.synchronize(); comm.Barrier()
        el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=device)
        if world > 1:
            import torch.distributed as dist
            dist.all_reduce(el, op=dist.ReduceOp.MAX)
        return float(el.item())

    def result_for(el, passes, collectives):
        finite = all(bool(torch.isfinite(t).all()) for t in (mu.params, d.params))
        # Dominant kernel: the network passes of the step.  Fused path (64x64 nets): ONE launch of
        # samx::fused::step_kernel carries every matmul of the step, so its algorithmic work is the whole
        # F of SURVEY.md 8(d) and `achieved` = F / its launch duration.  Layer-wise path: the same F over
        # the duration of the three concurrent GEMM chains.
        dom = passes['network_passes']
        achieved = flops_per_step(cfg) / dom / 1e12
        return {
            'metric': 'SAM update steps/sec (actor+critic+disc, batch %d)' % cfg['B'],
            'value': world * args.steps / el, 'unit': 'steps/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': el / args.steps * 1e3, 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': '%s %s: obs %d, act %d, %dx%d MLPs, batch %d per GPU, %s replay (%d rows '
                                   'resident), D gradient penalty, rmsify_obs%s' %
                                   (args.config, cfg['name'], cfg['ob'], cfg['ac'], cfg['h'][0], cfg['h'][1],
                                    cfg['B'], 'prioritized' if cfg['per'] else 'uniform',
                                    mu.replay_buffer.num_entries, (', with_layernorm' if cfg.get('ln') else '') +
                                    (', rmsify_rets+popart' if cfg.get('ret') else '') +
                                    (', n-step returns n=%d' % cfg['nstep'] if cfg.get('nstep') else '')),
                       'parallelism': ('dp%d' % world) + (' (one-GPU rehearsal with a twin rank)' if rehearse else ''), 'hip_graph': not args.no_graph,
                       'steps_per_graph_launch': drv.K, 'collectives': collectives, 'finite_params': finite},
            'roofline': {'bound': 'latency' if drv.fused else 'mfma',
                         'kernel': ('samx::fused::step_kernel (whole-network fused, all three roles)' if drv.fused
                                    else ('samx::wide::roles_kernel (the layer chains of every role, one launch) + '
                                          'samx::wgrad_direct_kernel (all weight-gradient products, one launch)')
                                    if drv.wide
                                    else 'layer-wise fp32 MFMA GEMM chains (samx::gemm_kernel), three roles '
                                         'on three streams'),
                         'achieved': achieved, 'peak': PEAK_FP32_TFLOPS, 'unit': 'TFLOP/s',
                         'frac': achieved / PEAK_FP32_TFLOPS,
                         'traffic': pmc_traffic(args.config + ('-ln' if cfg.get('ln') else '')),
                         'kernel_us': dom * 1e6, 'flops_per_launch': flops_per_step(cfg),
                         'note': ('latency-bound: %.1f MFLOP and < 1 MB per launch; neither the MFMA nor the HBM '
                                  'roofline binds (DESIGN.md 3.1)' % (flops_per_step(cfg) / 1e6)) if drv.fused
                                 else 'fp32 MFMA (exact f32) peak 157.3 TFLOP/s at 2.4 GHz; with every CU on MFMAs the chip holds 2.1 GHz: '
                                      '138 TFLOP/s measured with nothing but MFMAs (profiles/r03_mfma_rate.txt)',
                         'pass_us': {k: v * 1e6 for k, v in passes.items()}},
        }

    # N > 1: three plans, from the most ordinary to the fastest, each measured over the same K steps; the
    # line reports the fastest one that completed and carries the others' ms/step in `config`.
    #   1. torch.distributed all_reduce between graph segments (the process group's own stream);
    #   2. RCCL called directly on the launch stream (sam_tf_amd/rccl.py), issued by the host between segments;
    #   3. those RCCL all-reduces captured INSIDE the K-step hipGraph.
    # No multi-GPU box was available while this was written: plans 2 and 3 run under a watchdog, and if
    # they do not come back the rank reports the measurement it already has instead of hanging the job.
    # Every N > 1 stage runs under a watchdog.  A stage that hangs or raises is never silent: the line gets a
    # top-level "degraded" (= "failed_stage") field naming it, and the process exits non-zero (3) when not even
    # the plain torch.distributed plan completed — then `value` is null.  When only one of the optional faster
    # plans failed, the line carries the measurement of the plans that did complete, "degraded" says which
    # stage was lost, and the exit code is 0 unless SAMX_BENCH_STRICT=1.
    import threading
    stage = ['']
    result = None
    strict = os.environ.get('SAMX_BENCH_STRICT') == '1'

    def degraded_exit(why):
        """Print what there is (rank 0), flagged, and end this rank."""
        line = result if result is not None else {
            'metric': 'SAM update steps/sec (actor+critic+disc, batch %d)' % cfg['B'], 'value': None,
            'unit': 'steps/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': None,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': args.config}}
        line['degraded'] = line['failed_stage'] = '%s: %s' % (stage[0], why)
        if rank == 0:
            print(json.dumps(line), flush=True)
        os._exit(3 if (result is None or strict) else 0)

    def watchdog(seconds, what):
        stage[0] = what
        t = threading.Timer(seconds, lambda: degraded_exit('did not complete within %.0f s (watchdog)' % seconds))
        t.daemon = True
        t.start()
        return t

    wd_s = float(os.environ.get('SAMX_BENCH_WATCHDOG_S', '120'))
    drv.set_collectives(False)
    dog = watchdog(wd_s + 0.05 * (args.steps + args.warmup), 'plan with torch.distributed all_reduce') if world > 1 else None
    try:
        el = timed()
    except Exception as e:  # noqa: BLE001
        if world == 1:
            raise
        degraded_exit('%s: %s' % (type(e).__name__, str(e)[:200]))
    if dog:
        dog.cancel()
    passes = time_passes(drv, cfg)
    result = result_for(el, passes, 'none' if world == 1 else 'torch.distributed all_reduce between graph segments')

    if world > 1 and os.environ.get('SAMX_RCCL_DIRECT', '1') != '0':
        dog = watchdog(wd_s, 'direct RCCL set-up')
        try:
            tried = {'torch_distributed': el}
            if comm.enable_direct():   # all ranks or none (agreed in stages, sam_tf_amd/comm.py)
                stage[0] = 'plan with host-issued direct RCCL'
                labels = {'host_issued_rccl': 'rccl on the launch stream, issued by the host between graph segments',
                          'in_graph_rccl': 'rccl all-reduces captured inside the %d-step hipGraph' % drv.K}
                tried['host_issued_rccl'] = timed()
                if drv.set_collectives(True):
                    stage[0] = 'plan with in-graph RCCL'
                    tried['in_graph_rccl'] = timed()
                best = min(tried, key=tried.get)
                if best != 'torch_distributed':
                    if best != 'in_graph_rccl':
                        drv.set_collectives(False)
                    result = result_for(tried[best], passes, labels[best])
                for k, v in tried.items():
                    if k != best:
                        result['config']['ms_per_step_' + k] = v / args.steps * 1e3
            else:
                result['config']['direct_rccl'] = 'unavailable: ' + str(getattr(comm, 'direct_error', ''))[:160]
            dog.cancel()
        except Exception as e:  # noqa: BLE001 - the measurement we have is printed, flagged; peers' watchdogs end them
            dog.cancel()
            degraded_exit('%s: %s' % (type(e).__name__, str(e)[:200]))
    if world > 1:
        # which collective implementation the numbers belong to
        info = {k: os.environ.get(k) for k in ('NCCL_ALGO', 'NCCL_PROTO', 'NCCL_MIN_NCHANNELS', 'NCCL_MAX_NCHANNELS',
                                               'RCCL_MSCCL_ENABLE', 'HSA_ENABLE_IPC_MODE_LEGACY')}
        try:
            info['rccl_version'] = '.'.join(str(x) for x in torch.cuda.nccl.version())
        except Exception:  # noqa: BLE001
            info['rccl_version'] = None
        result['config']['rccl'] = info
    drv3 = None
    if world == 1 and not args.no_training_step and not rehearse:
        # the reference's loop runs d_steps = 1 discriminator update and g_steps = 3 agent
        # updates per training step (mujoco_sam_imitate.sh:26-27); `value` above is the 1:1 step of the
        # metric, this is the same learner on that schedule (one D+agent row, two agent-only rows)
        from sam_tf_amd.driver import UpdateStepDriver
        drv3 = UpdateStepDriver(mu, d, pool, use_graph=not args.no_graph, d_steps=1, g_steps=3)
        drv3.prepare_all()
        n_ts = max(10, args.steps // 4)
        drv3.run(3 * max(5, args.warmup // 4))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        drv3.run(3 * n_ts)
        torch.cuda.synchronize()
        result['config']['training_steps_per_s_d1_g3'] = n_ts / (time.perf_counter() - t0)
    if world > 1:
        # all-reduce latency vs bucket bytes, back to back on the launch stream (what a
        # step pays in series is the first entry at or above this config's gradient bucket)
        lat = {}
        dog = watchdog(60.0, 'all-reduce latency sweep')
        for nbytes in (1 << 8, 1 << 12, 1 << 14, 1 << 16, 1 << 18, 1 << 20, 1 << 22):
            buf = torch.zeros(nbytes // 4, dtype=torch.float32, device=device)
            for _ in range(10):
                comm.allreduce_(buf)
            torch.cuda.synchronize(); comm.Barrier(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(50):
                comm.allreduce_(buf)
            torch.cuda.synchronize()
            lat[str(nbytes)] = (time.perf_counter() - t0) / 50 * 1e6
        dog.cancel()
        result['config']['allreduce_us_by_bytes'] = lat
        result['config']['grad_bucket_bytes'] = int(mu.L.grads.numel() * 4)
    plain = not (args.layernorm or args.launcher or args.argparse_defaults or args.no_graph)
    if world == 1 and not rehearse and args.config == 'C2' and plain and not args.no_others:
        # the other single-GPU configurations of BASELINE.json, short runs in the same process so that the
        # driver-timed line carries them (C4 is the configuration the north_star's >= 50x target is stated on)
        del drv, drv3, mu, d, pool
        import gc
        gc.collect()
        torch.cuda.empty_cache()
        result['config']['other_configs'] = other_configs(comm, device, args)
        # the same update step through the reference's own object calls (SAMAgent.train, Discriminator.get_grads /
        # get_losses / optimize, replay sample, obs_rms.update) instead of the update-step driver
        oa = {}
        for name, n, warm in (('C2', 1000, 100), ('C4', 300, 30)):
            try:
                oa[name] = object_api_rate(dict(CONFIGS[name], ln=False, ret=False, nstep=0), comm, device,
                                           args.seed, n, warm)
            except Exception as e:  # noqa: BLE001
                oa[name] = {'error': '%s: %s' % (type(e).__name__, str(e)[:200])}
        result['config']['object_api_steps_per_s'] = oa
        # a side run that raised is never silent: the headline line still comes out (its own measurement is
        # complete), carries a top-level "degraded" naming what was lost, and the process exits 3
        lost = ['other_configs.%s: %s' % (k, v['error']) for k, v in result['config']['other_configs'].items()
                if 'error' in v]
        lost += ['object_api_steps_per_s.%s: %s' % (k, v['error']) for k, v in oa.items()
                 if isinstance(v, dict) and 'error' in v]
        if lost:
            result['degraded'] = result['failed_stage'] = '; '.join(lost)
    if rank == 0:
        if world == 1 and not args.no_cpu:
            result['cpu_baseline'] = cpu_baseline(cfg)
        print(json.dumps(result), flush=True)
    if world == 1 and result.get('degraded'):
        sys.stdout.flush()
        sys.exit(3)
    if world > 1:
        import torch.distributed as dist
        # the line is out and complete: a stuck teardown must not hold the job; it is reported on stderr
        # (and by exit code 4 under SAMX_BENCH_STRICT=1)
        def stuck():
            sys.stderr.write('bench.py: rank %d: communicator teardown did not complete in 30 s\n' % rank)
            sys.stderr.flush()
            os._exit(4 if strict else 0)
        bye = threading.Timer(30.0, stuck)
        bye.daemon = True
        bye.start()
        if comm.direct is not None:
            comm.direct.destroy()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
