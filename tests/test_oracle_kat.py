"""The reference's own known-answer unit tests, run through the CPU oracle (pins the oracle a second way)."""
import numpy as np
import pytest

import kat_cases
import oracle_lib as orc

EMPTY_C = (np.zeros(1, np.int64), np.zeros(0, np.int32), np.zeros(0, np.float32))


@pytest.mark.parametrize("name", sorted(kat_cases.cases()))
def test_lsqr_known_answers(name):
    c = kat_cases.cases()[name]
    S = kat_cases.dense_to_csr(c["A"])
    x, it, r = orc.lsqr(S, EMPTY_C, c["A"].shape[1], c["b"], c["niter"], c["rmin"])
    kat_cases.check(c, x)


def test_wavelet_calculate_data():
    """tests_wavelet_compression.f90:70-130: A.x == Haar(A).Haar(x) on 3x4x5 (orthonormality)."""
    nx, ny, nz, nrows = 3, 4, 5, 5
    n = nx * ny * nz
    i = np.arange(1, n + 1, dtype=np.float64)
    A = np.stack([(2 * i - j) / (i + j) for j in range(1, nrows + 1)])
    x = np.full(n, 2.0 * (nrows + 1) + 1.0)     # the reference uses the stale loop index j = nrows + 1 (:103)
    b = A @ x
    Aw = np.stack([orc.wavelet(r, nx, ny, nz, 1) for r in A])
    b2 = Aw @ orc.wavelet(x, nx, ny, nz, 1)
    assert np.all(np.abs(b - b2) <= 0.5 * 1e-6 * (np.abs(b) + np.abs(b2)))


def test_wavelet_diagonal_matrix_nnz():
    """tests_wavelet_compression.f90:140-181: Haar of the 10^3 identity has exactly 46656 non-zeros."""
    n = 1000
    nnz = 0
    for j in range(n):
        a = np.zeros(n)
        a[j] = 1.0
        nnz += int(np.count_nonzero(orc.wavelet(a, 10, 10, 10, 1)))
    assert nnz == 46656


@pytest.mark.parametrize("wtype", [1, 2])
def test_wavelet_norm_preserving(wtype):
    """tests_wavelet_compression.f90:187-238: x = 1..N on 10x11x12, |norm(x) - norm(W x)| within tol = 1e-6 (relative)."""
    x = np.arange(1, 10 * 11 * 12 + 1, dtype=np.float64)
    w = orc.wavelet(x, 10, 11, 12, wtype)
    n0, n1 = np.linalg.norm(x), np.linalg.norm(w)
    assert abs(n0 - n1) <= 0.5 * 1e-6 * (n0 + n1)


@pytest.mark.parametrize("wtype", [1, 2])
def test_wavelet_invertible(wtype):
    """tests_wavelet_compression.f90:244-300: inverse(forward(e_j)) == e_j on 10x11x12 (abs 1e-15 off-diagonal);
    here on a random vector, abs 1e-13."""
    rng = np.random.default_rng(3)
    a = rng.standard_normal(10 * 11 * 12)
    back = orc.wavelet(orc.wavelet(a, 10, 11, 12, wtype), 10, 11, 12, wtype, inverse=True)
    assert np.max(np.abs(back - a)) <= 1e-13


def test_sparse_matrix_column_extraction():
    """tests_sparse_matrix.f90:39-: mult_vector with unit vectors returns the matrix columns."""
    rng = np.random.default_rng(5)
    A = rng.standard_normal((7, 9))
    A[2] = 0.0
    S = kat_cases.dense_to_csr(A)
    for j in range(9):
        e = np.zeros(9)
        e[j] = 1.0
        assert np.array_equal(orc.spmv(*S, e), A.astype(np.float32)[:, j].astype(np.float64))
        assert np.array_equal(orc.spmtv(*S, np.eye(7)[min(j, 6)], 9), A.astype(np.float32)[min(j, 6)].astype(np.float64))


def test_normalize_columns_like_the_reference_unit_test():
    """tests_sparse_matrix.f90:39-104 (test_normalize_columns): 30 x 10 matrix, entry = running counter in the first five columns,
    zero columns after; column_norm = norm2 of the dense column, normalised non-zero columns have unit length, zero columns stay."""
    nrows, ncols = 30, 10
    A = np.zeros((nrows, ncols))
    counter = 0
    for j in range(nrows):
        for i in range(ncols):
            counter += 1
            if i < ncols // 2:
                A[j, i] = float(counter)
    S = kat_cases.dense_to_csr(A)
    norm, vals = orc.normalize_columns(*S, ncols)
    tol = kat_cases.TOL
    for i in range(ncols):
        want = np.linalg.norm(A[:, i])
        assert abs(norm[i] - want) <= 0.5 * tol * (abs(norm[i]) + abs(want))
        e = np.zeros(ncols)
        e[i] = 1.0
        col = orc.spmv(S[0], S[1], vals, e)
        got = np.linalg.norm(col)
        want1 = 1.0 if want != 0.0 else 0.0
        assert abs(got - want1) <= 0.5 * tol * (abs(got) + want1)

