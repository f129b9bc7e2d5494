"""samx_gemm_f32 (the fp32 MFMA GEMM of the layer-wise path) against a float64 product of the same
float32 operands.  Tolerance 1e-5 relative to the largest output magnitude (fp32 fma chain of length R
vs fp64).  Shapes cover the 16-byte and the dword staging of either operand (extent % 4), partial tiles,
reductions that are not a multiple of the 32-deep stage, the 16-wave small-launch variant and the
4-wave one (>= 512 workgroups), its ring-of-two form (unsplit launches of more than 768 workgroups: four per CU; reductions of
one, two and four stages), every epilogue, the split reduction and the virtual ones row."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = [(64, 64, 32), (37, 70, 33), (1024, 256, 376), (1024, 256, 393), (1000, 17, 256), (128, 1, 300),
          (2048, 400, 376), (2050, 398, 402), (4096, 300, 400), (8192, 400, 100), (8192, 400, 37),
          (8192, 400, 20), (12288, 400, 50), (3, 5, 7), (256, 256, 4)]


def _padded(x):
    """Device copy of x with 16 readable bytes behind its last element (overread_ok contract)."""
    buf = torch.full((x.size + 4,), float('nan'), dtype=torch.float32, device='cuda')
    buf[:x.size] = torch.as_tensor(np.ascontiguousarray(x).ravel()).cuda()
    return buf[:x.size].view(*x.shape)


def _operands(layout, M, N, R, rs):
    a = rs.randn(M, R).astype(np.float32)
    b = rs.randn(R, N).astype(np.float32)
    A = _padded(a if layout != 'TN' else a.T)
    B = _padded(b if layout != 'NT' else b.T)
    a_str = (R, 1) if layout != 'TN' else (1, M)
    b_str = (N, 1) if layout != 'NT' else (1, R)
    return a, b, A, B, a_str, b_str


@pytest.mark.parametrize('overread', [False, True])
@pytest.mark.parametrize('layout', ['NN', 'NT', 'TN'])
@pytest.mark.parametrize('M,N,R', SHAPES)
def test_gemm_layouts_and_shapes(layout, M, N, R, overread):
    """overread=True: 16-byte staging also for extents that are not a multiple of 4 (what nets.hip uses;
    the bytes behind each operand are NaN here, so any use of an over-read value would show)."""
    from sam_tf_amd import ops
    rs = np.random.RandomState(M + 3 * N + 7 * R)
    a, b, A, B, a_str, b_str = _operands(layout, M, N, R, rs)
    got = ops.gemm_f32(A, a_str, B, b_str, M, N, R, overread_ok=overread).cpu().numpy()
    want = a.astype(np.float64) @ b.astype(np.float64)
    assert np.max(np.abs(got - want)) <= 1e-5 * np.max(np.abs(want))


@pytest.mark.parametrize('epi', [1, 2, 3, 4, 5, 6])
@pytest.mark.parametrize('M,N,R', [(1024, 256, 376), (77, 45, 50)])
def test_gemm_epilogues(epi, M, N, R):
    from sam_tf_amd import ops
    rs = np.random.RandomState(epi)
    a, b, A, B, a_str, b_str = _operands('NN', M, N, R, rs)
    a *= 0.1; A *= 0.1
    bias = rs.randn(N).astype(np.float32)
    aux = rs.uniform(-0.9, 0.9, (M, N)).astype(np.float32)
    add = rs.randn(M, N).astype(np.float32)
    alpha = 1.5
    t = lambda x: torch.as_tensor(x).cuda()
    got = ops.gemm_f32(A, a_str, B, b_str, M, N, R, epi=epi, bias=t(bias), aux=t(aux),
                       add=t(add) if epi == 5 else None, alpha=alpha).cpu().numpy()
    acc = a.astype(np.float64) @ b.astype(np.float64)
    if epi == 1:
        want = acc + bias
    elif epi == 2:
        want = np.maximum(acc + bias, 0.2 * (acc + bias))
    elif epi == 3:
        want = np.clip(alpha * np.tanh(acc + bias), -alpha, alpha)
    elif epi == 4:
        want = acc * np.where(aux >= 0, 1.0, 0.2)
    elif epi == 5:
        want = (acc + add) * (alpha * (1.0 - (aux.astype(np.float64) / alpha) ** 2))
    else:
        want = acc + alpha * aux
    assert np.max(np.abs(got - want)) <= 1e-5 * max(1.0, np.max(np.abs(want)))


@pytest.mark.parametrize('M,N,R,nsplit,ones', [(256, 256, 1024, 8, 1024), (393, 400, 8192, 14, 8192),
                                                 (376, 256, 1000, 4, 600), (17, 300, 2048, 16, -1),
                                                 (64, 64, 4096, 1, 4096)])
@pytest.mark.parametrize('overread', [False, True])
def test_gemm_weight_gradient_form(M, N, R, nsplit, ones, overread):
    """TN product with the reduction split over workgroups and the virtual ones row (bias gradient =
    column sums of the first `ones` rows of B)."""
    from sam_tf_amd import ops
    rs = np.random.RandomState(M + N)
    a, b, A, B, a_str, b_str = _operands('TN', M, N, R, rs)
    got = ops.gemm_f32(A, a_str, B, b_str, M, N, R, a_ones_rows=ones, nsplit=nsplit,
                       overread_ok=overread).cpu().numpy()
    want = a.astype(np.float64) @ b.astype(np.float64)
    if ones >= 0:
        want = np.concatenate([want, b[:ones].astype(np.float64).sum(0, keepdims=True)])
    assert got.shape == want.shape
    assert np.max(np.abs(got - want)) <= 2e-5 * np.max(np.abs(want))


def test_gemm_rejects_bad_arguments():
    from sam_tf_amd import ops, _lib
    A = torch.zeros(8, 8, device='cuda'); B = torch.zeros(8, 8, device='cuda')
    with pytest.raises(_lib.SamxError):
        ops.gemm_f32(A, (8, 2), B, (8, 1), 8, 8, 4)          # no unit stride on A
    with pytest.raises(_lib.SamxError):
        ops.gemm_f32(A, (8, 1), B, (8, 1), 8, 8, 8, epi=9)   # unknown epilogue
    with pytest.raises(_lib.SamxError):
        ops.gemm_f32(A, (8, 1), B, (8, 1), 8, 8, 8, epi=1)   # bias epilogue without bias
