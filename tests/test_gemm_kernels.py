"""Kernel-level GEMM tests: the reference's sweep of its tensor-core GEMM against h_mmult
(Cuda/main.cu:14-15 -> Cuda/qr.cu:1944-1959 test_iterator_template_tensorcore_mmult_tiled, comparator
Cuda/mmult.cuh:387-435, tolerance 5e-4 at mmult.cuh:413) and its two known answers (Cuda/mmult.cu:312-356: a = b = column
index, 16 x 16 -> C[i][j] == 120 j;  mmult.cu:652-696: 32 x 32 -> 496 j), driven through mpqr_gemm_test_f32 so that every MFMA
GEMM kernel of the library sees odd shapes directly (edge tiles, masked rows / columns, K padded to the k step), not only
through whole factorisations.  The host comparator plays the part of h_mmult (Cuda/mmult.cuh:70-89) on the rounded operands."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

KERNELS = {1: "gemm_f16 (128 tile)", 2: "gemm2 A_F32T (256 tile, fp32 source transposed while staged)", 6: "gemm6 (256 tile ping-pong)",
           16: "gemm6 on v_mfma_f32_16x16x32_f16 (operands swapped: 16-byte epilogue accesses)"}


def _h16(x):
    return x.astype(np.float16).astype(np.float32)


def _sweep():
    """m = 20 .. 1280 (x2), n = m/4 .. < m (x2), k = n/4 .. < n (x2): the loop nest of qr.cu:1944-1959."""
    m = 20
    while m < 2000:
        n = m // 4
        while n < m:
            k = n // 4
            while k < n:
                if k >= 1:
                    yield m, n, k
                k *= 2
            n *= 2
        m *= 2


@pytest.mark.parametrize("kernel", sorted(KERNELS))
def test_known_answers(kernel):
    """mmult.cu:312-356 and :652-696: a[i][j] = b[i][j] = j  =>  C[i][j] = j * sum(k), exactly representable."""
    import mixedprecisionblockqr_amd as mp
    for nn, s in ((16, 120.0), (32, 496.0)):
        A = np.tile(np.arange(nn, dtype=np.float32), (nn, 1))
        C = np.zeros((nn, nn), np.float32)
        mp.gemm_test(A, A.copy(), C, kernel, 0)
        assert np.array_equal(C, np.tile(np.arange(nn, dtype=np.float32) * s, (nn, 1))), KERNELS[kernel]


@pytest.mark.parametrize("kernel", sorted(KERNELS))
def test_reference_sweep_against_h_mmult(kernel):
    """qr.cu:1944-1959 with the <half, half, float> instantiation: U[0,1) operands rounded to fp16, fp32 accumulation,
    |C_dev - C_host| <= 5e-4 (mmult.cuh:413) element-wise."""
    import mixedprecisionblockqr_amd as mp
    rng = np.random.default_rng(1234)
    worst = 0.0
    for m, n, k in _sweep():
        A = rng.random((m, k), dtype=np.float32); B = rng.random((k, n), dtype=np.float32)
        C = np.zeros((m, n), np.float32)
        mp.gemm_test(A, B, C, kernel, 0)
        ref = _h16(A).astype(np.float64) @ _h16(B).astype(np.float64)
        err = float(np.abs(C - ref).max())
        worst = max(worst, err)
        assert err <= 5e-4, (KERNELS[kernel], m, n, k, err)
    print(f"{KERNELS[kernel]}: max |C - h_mmult| over the sweep {worst:.2e}")


@pytest.mark.parametrize("kernel", [1, 6, 16])
@pytest.mark.parametrize("shape", [(320, 200, 130), (288, 511, 64), (1024, 130, 200), (160, 80, 37), (544, 260, 1030), (768, 512, 192)])
def test_in_place_update_on_odd_shapes(kernel, shape):
    """C -= A B with the read-modify-write epilogue (the trailing update's form, Cuda/mmult.cu:236-288 + the copy-back
    mmult.cuh:104-151) on shapes that end inside a tile: N and K arbitrary, M a multiple of 32 only (the epilogue's documented
    precondition: the library pads the rows of the matrix it updates)."""
    import mixedprecisionblockqr_amd as mp
    m, n, k = shape
    rng = np.random.default_rng(7)
    A = rng.standard_normal((m, k)).astype(np.float32); B = rng.standard_normal((k, n)).astype(np.float32)
    C0 = rng.standard_normal((m, n)).astype(np.float32)
    C = C0.copy()
    mp.gemm_test(A, B, C, kernel, 2)
    ref = C0.astype(np.float64) - _h16(A).astype(np.float64) @ _h16(B).astype(np.float64)
    tol = 2e-6 * k * 4 + 1e-5                          # fp32 accumulation of k products of O(1) magnitude
    assert np.abs(C - ref).max() <= tol * max(1.0, np.abs(ref).max()), (KERNELS[kernel], shape)


@pytest.mark.parametrize("shape", [(256, 256, 128), (320, 200, 130), (992, 260, 500)])
@pytest.mark.parametrize("mode", [0, 2])
def test_fp8_kernel_against_e4m3_emulation(shape, mode):
    """The e4m3 kernel of BASELINE config 5 (v_mfma_scale_f32_32x32x64_f8f6f4): operands rounded to fp16 and then to OCP e4m3
    by the library's quantisation kernel; the host applies the same two roundings (oracle/pyoracle.round_e4m3)."""
    import mixedprecisionblockqr_amd as mp
    from oracle import pyoracle as po
    m, n, k = shape
    rng = np.random.default_rng(11)
    A = rng.random((m, k), dtype=np.float32); B = rng.random((k, n), dtype=np.float32)
    C0 = rng.standard_normal((m, n)).astype(np.float32)
    C = C0.copy() if mode == 2 else np.zeros((m, n), np.float32)
    mp.gemm_test(A, B, C, 8, mode)
    prod = po.round_e4m3(_h16(A)).astype(np.float64) @ po.round_e4m3(_h16(B)).astype(np.float64)
    ref = C0 - prod if mode == 2 else prod
    assert np.abs(C - ref).max() <= 1e-5 * k + 1e-4, shape


def test_rmw_epilogue_precondition_is_checked():
    """M % 32 != 0 with the read-modify-write epilogue is refused, not silently mis-stored."""
    import mixedprecisionblockqr_amd as mp
    A = np.ones((300, 64), np.float32); B = np.ones((64, 64), np.float32); C = np.zeros((300, 64), np.float32)
    with pytest.raises(mp.MpqrError):
        mp.gemm_test(A, B, C, 1, 2)


@pytest.mark.parametrize("kernel", sorted(KERNELS))
def test_device_cast_round_trip(kernel):
    """The reference's cast check (Cuda/mmult.cuh:482-555 test_dev_cpy_and_cast_array: fp32 -> fp16 -> fp32 through its copy-and-cast
    kernel, |x - roundtrip(x)| <= 4e-4 for U[0,1) data, mmult.cuh:541).  Here the casts are fused into the GEMM kernels' operand staging
    (no cast kernel exists), so the DEVICE-side conversion is driven through each kernel's staging path with B = I: C = A I has exactly one
    non-zero product per entry, i.e. C[i][j] = fp16(A[i][j]) as the staging rounded it -- bit for bit the host's round-to-nearest-even
    conversion, and within the reference's 4e-4 of the fp32 input.  Odd shapes: masked rows / columns and a K that is padded to the k step."""
    import mixedprecisionblockqr_amd as mp
    rng = np.random.default_rng(7)
    for (m, k) in ((97, 90), (256, 256), (129, 80), (600, 400)):
        A = rng.random((m, k), dtype=np.float32)
        C = np.zeros((m, k), np.float32)
        mp.gemm_test(A, np.eye(k, dtype=np.float32), C, kernel, 0)
        assert np.array_equal(C, _h16(A)), (KERNELS[kernel], m, k, float(np.abs(C - _h16(A)).max()))
        assert float(np.abs(C - A).max()) <= 4e-4, (KERNELS[kernel], m, k)
