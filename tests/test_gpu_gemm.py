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


def _split_chunk(R, nsplit):
    return ((R + nsplit - 1) // nsplit + 31) // 32 * 32


@pytest.mark.parametrize('kernel', [1, 0])
@pytest.mark.parametrize('M,N,R,nsplit,ones', [
    (256, 256, 1024, 2, 1024),    # C4's W2 products: main loop only, bias row over every row
    (393, 256, 2048, 4, 1024),    # reward net W1 with LayerNorm: 2B rows, bias row over the primal half, 9-wide edge tile
    (376, 256, 1024, 2, -1),      # no bias row
    (256, 17, 1024, 2, 1024),     # action head: odd N (4-byte stores), one narrow column tile
    (64, 64, 64, 2, 64),          # 8 rows per wave: the k-step-at-a-time loop only
    (100, 70, 160, 2, 100),       # 24 rows per wave: one group + the remainder; bias limit inside a group; odd extents
    (33, 2, 96, 3, 50),           # a second tile of one column of X, the narrowest B, a bias limit inside chunk 1
    (20, 20, 32, 2, 32),          # chunk 1 has no rows: zeros
    (130, 130, 2, 2, 2),          # one k-step in all
    (4, 6, 1000, 5, 777),
])
def test_weight_gradient_partials(M, N, R, nsplit, ones, kernel):
    """samx_wgrad_partials_f32: every chunk's partial sums and bias row of the register-direct kernel (and of the
    staged one it replaced) against float64, on padded row strides; the two kernels agree to fp32 rounding."""
    from sam_tf_amd import ops
    if kernel == 0 and min(M, N) < 4:
        pytest.skip('the staged kernel stages 16 bytes at a time')
    rs = np.random.RandomState(M * 7 + N + R)
    lda, ldb = M + 3, N + 5   # rows keep 4-byte alignment only; the tail of each row is readable (overread_ok)
    a = rs.standard_normal((R + 1, lda)).astype(np.float32)
    b = rs.standard_normal((R + 1, ldb)).astype(np.float32)
    A = torch.from_numpy(a).cuda(); B = torch.from_numpy(b).cuda()
    got = ops.wgrad_partials_f32(A, B, M, N, R, nsplit, a_ones_rows=ones, kernel=kernel, overread_ok=True).cpu().numpy()
    chunk = _split_chunk(R, nsplit)
    Mo = M + (1 if ones >= 0 else 0)
    assert got.shape == (nsplit, Mo, N)
    for z in range(nsplit):
        r0, r1 = min(R, z * chunk), min(R, (z + 1) * chunk)
        want = a[r0:r1, :M].astype(np.float64).T @ b[r0:r1, :N].astype(np.float64)
        if ones >= 0:
            lim = max(r0, min(r1, ones))
            want = np.concatenate([want, b[r0:lim, :N].astype(np.float64).sum(0, keepdims=True)])
        scale = max(1.0, np.max(np.abs(want)))
        assert np.max(np.abs(got[z] - want)) <= 2e-5 * scale, (z, np.max(np.abs(got[z] - want)), scale)


def test_weight_gradient_partials_rejects_what_does_not_fit():
    from sam_tf_amd import ops, _lib
    A = torch.zeros(65, 8, device='cuda'); B = torch.zeros(65, 8, device='cuda')
    with pytest.raises(_lib.SamxError):
        ops.wgrad_partials_f32(A, B, 8, 8, 63, 2, kernel=1, overread_ok=True)    # odd R: row pairs
    with pytest.raises(_lib.SamxError):
        ops.wgrad_partials_f32(A, B, 7, 8, 64, 2, kernel=1, overread_ok=False)   # odd M without readable bytes behind
    with pytest.raises(_lib.SamxError):
        ops.wgrad_partials_f32(A, B, 8, 8, 64, 1, kernel=1)                      # a split product has >= 2 chunks
    ops.wgrad_partials_f32(A, B, 8, 8, 64, 2, kernel=1)                          # even extents need no overread
