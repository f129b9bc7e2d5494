"""GEMM throughput at the layer shapes of C5 (and a large square):
   python tools/gemm_shapes.py                    (the shipped choice between the ring of three and the ring of two)
   SAMX_GEMM_FOUR=0 python tools/gemm_shapes.py   (ring of three everywhere; 2: ring of two everywhere)
Prints TFLOP/s (HIP events around 20 calls) and the max relative error against torch's fp32 matmul."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sam_tf_amd import ops  # noqa: E402
from sam_tf_amd import _lib  # noqa: E402,F401

SHAPES = [(8192, 400, 376, 'NN', 1), (8192, 300, 400, 'NN', 1), (8192, 376, 400, 'NT', 1), (8192, 400, 300, 'NT', 1),
          (376, 400, 8192, 'TN', 12), (376, 400, 8192, 'TN', 18), (376, 400, 8192, 'TN', 6), (400, 300, 8192, 'TN', 16), (400, 300, 8192, 'TN', 21), (400, 300, 8192, 'TN', 7), (8192, 8192, 2048, 'NN', 1),
          (8192, 256, 393, 'NN', 1), (393, 256, 8192, 'TN', 8), (393, 256, 8192, 'TN', 27), (393, 256, 8192, 'TN', 16)]


def main():
    dev = 'cuda'
    g = torch.Generator(device=dev); g.manual_seed(0)
    print('SAMX_GEMM_FOUR = %s' % os.environ.get('SAMX_GEMM_FOUR', '1 (default)'))
    for M, N, R, layout, nsplit in SHAPES:
        if layout == 'NN':
            A = torch.randn(M, R, generator=g, device=dev); B = torch.randn(R, N, generator=g, device=dev)
            call = lambda: ops.gemm_f32(A, (R, 1), B, (N, 1), M, N, R)
            ref = A @ B
        elif layout == 'NT':
            A = torch.randn(M, R, generator=g, device=dev); B = torch.randn(N, R, generator=g, device=dev)
            call = lambda: ops.gemm_f32(A, (R, 1), B, (1, R), M, N, R)
            ref = A @ B.t()
        else:
            A = torch.randn(R, M, generator=g, device=dev); B = torch.randn(R, N, generator=g, device=dev)
            call = lambda: ops.gemm_f32(A, (1, M), B, (N, 1), M, N, R, nsplit=nsplit, a_ones_rows=R)
            ref = torch.cat([A.t() @ B, B.sum(dim=0, keepdim=True)], dim=0)
        out = call()
        err = (out - ref).abs().max().item() / ref.abs().max().item()
        for _ in range(5):
            call()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            call()
        e1.record(); e1.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        print('%-2s %5d x %5d x %5d  nsplit %2d  %8.1f us  %6.1f TFLOP/s  relerr %.1e' %
              (layout, M, N, R, nsplit, us, 2.0 * M * N * R / us / 1e6, err))


if __name__ == '__main__':
    main()
