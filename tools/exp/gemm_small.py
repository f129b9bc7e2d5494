"""The layer products of the 256-wide LayerNorm configuration (1024 rows): time per call against the split count."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sam_tf_amd import ops  # noqa: E402

dev = 'cuda'
g = torch.Generator(device=dev); g.manual_seed(0)
for M, N, R in [(1024, 256, 376), (1024, 256, 256), (3072, 256, 393), (3072, 256, 256)]:
    A = torch.randn(M, R, generator=g, device=dev); B = torch.randn(R, N, generator=g, device=dev)
    ref = A @ B
    for ns in (1, 2, 3, 4):
        call = lambda: ops.gemm_f32(A, (R, 1), B, (N, 1), M, N, R, nsplit=ns)
        out = call()
        err = (out - ref).abs().max().item() / ref.abs().max().item()
        for _ in range(5):
            call()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            call()
        e1.record(); e1.synchronize()
        print('NN %5d x %4d x %4d nsplit %d  %6.1f us  relerr %.1e' % (M, N, R, ns, e0.elapsed_time(e1) / 50 * 1e3, err))
