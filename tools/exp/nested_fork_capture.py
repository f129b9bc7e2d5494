"""Is a fork of a forked stream capturable on this ROCm?  (round-2 advice: hipStreamEndCapture crashed when
samx_step_grads_part — which forks two role streams from the stream it is given — was issued on a stream that was
itself forked from the capture's origin stream.)  Each variant runs in a child process; a crash is an exit code.

  torch-nested      : origin -> A -> (B, C) with torch streams / events only, no libsamx
  samx-on-origin    : the layer-wise step call (forks its role streams) on the origin stream      [shipped order]
  samx-on-fork      : the same call on a stream forked from the origin                            [round 2's crash]
  samx-on-fork-serial: as samx-on-fork with SAMX_SERIAL_ROLES=1 (the call forks nothing)
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def child(variant):
    import torch
    sys.path.insert(0, ROOT)
    dev = torch.device('cuda', 0)
    if variant == 'torch-nested':
        x = torch.zeros(1 << 16, device=dev)
        y, z = torch.zeros_like(x), torch.zeros_like(x)   # (no allocation inside the capture)
        a, b, c = (torch.cuda.Stream(dev) for _ in range(3))
        origin = torch.cuda.Stream(dev)
        with torch.cuda.stream(origin):
            for rep in range(2):   # warm-up, then capture
                g = torch.cuda.CUDAGraph() if rep else None
                ctx = torch.cuda.graph(g, stream=origin, capture_error_mode='thread_local') if rep else None
                if ctx:
                    ctx.__enter__()
                cur = torch.cuda.current_stream()
                for _ in range(3):
                    a.wait_stream(cur)
                    with torch.cuda.stream(a):
                        x.add_(1.0)
                        b.wait_stream(a); c.wait_stream(a)
                        with torch.cuda.stream(b):
                            torch.mul(x, 2.0, out=y)
                        with torch.cuda.stream(c):
                            torch.mul(x, 3.0, out=z)
                        a.wait_stream(b); a.wait_stream(c)
                        x.add_(y).add_(z)
                    cur.wait_stream(a)
                if ctx:
                    print('ending the capture', flush=True)
                    ctx.__exit__(None, None, None)
                    print('captured', flush=True)
            g.replay()
        torch.cuda.synchronize()
        print('ok', float(x[0]), flush=True)
        return
    import bench
    from sam_tf_amd import ops
    from sam_tf_amd.comm import TorchComm
    cfg = dict(ob=5, ac=2, h=(33, 70), B=64, per=False, mem=2048, name='ragged')
    mu, d, pool, drv = bench.build_learner(cfg, TorchComm('cuda'), dev, 3, False)
    assert not drv.fused and not drv.wide
    drv.prepare()
    drv.step()
    torch.cuda.synchronize()
    side = torch.cuda.Stream(dev)

    def row():
        cur = torch.cuda.current_stream()
        if variant == 'samx-on-origin':
            ops.step_grads_part(drv.L, drv.batches[0], drv.out, 3)
        else:
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                ops.step_grads_part(drv.L, drv.batches[0], drv.out, 3)
            cur.wait_stream(side)
    s = torch.cuda.Stream(dev)
    with torch.cuda.stream(s):
        row()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode='thread_local'):
        for _ in range(3):
            row()
    print('captured', flush=True)
    g.replay()
    torch.cuda.synchronize()
    print('ok', flush=True)


if __name__ == '__main__':
    if len(sys.argv) == 2:
        child(sys.argv[1])
        sys.exit(0)
    for v in sys.argv[2:] or ('torch-nested', 'samx-on-origin', 'samx-on-fork-serial', 'samx-on-fork'):
        env = dict(os.environ)
        if v.endswith('-serial'):
            env['SAMX_SERIAL_ROLES'] = '1'
        r = subprocess.run([sys.executable, os.path.abspath(__file__), v], env=env, capture_output=True, text=True,
                           timeout=200)
        tail = (r.stdout.strip().splitlines() or [''])[-1]
        err = [l for l in r.stderr.strip().splitlines() if 'amdgpu.ids' not in l][-3:]
        print('%-20s exit %4d  %s  %s' % (v, r.returncode, tail, ' | '.join(err)[:300]), flush=True)
