"""Known-answer systems of the reference's own unit tests (src/tests/tests_lsqr.f90), restated as data so the same
cases can be run through the CPU oracle (test_oracle_kat.py) and through the HIP path (test_gpu_parity.py).
Each case: dense A (nrows x ncols), b, expected x, tolerance, niter, rmin."""
import numpy as np

TOL = 1e-6          # src/global_typedefs.F90:55 (matrix stored in fp32)


def cases():
    out = {}
    # tests_lsqr.f90:71-125  rank-1 1440 x 1440, rows j*(1..1) ; b_j = j*N ; x = all ones
    n = 1440
    A = np.repeat(np.arange(1, n + 1, dtype=np.float64)[:, None], n, axis=1)
    out["determined"] = dict(A=A, b=np.arange(1, n + 1, dtype=np.float64) * n, x=np.ones(n), tol=TOL, niter=100,
                             rmin=1e-13, rel=True)
    # tests_lsqr.f90:144-212  quadratic regression 1000 x 3 -> (1, -3, 0)
    xi = np.arange(1, 1001, dtype=np.float64) / 1000.0
    A = np.stack([xi ** 0, xi ** 1, xi ** 2], 1)
    out["overdetermined_1"] = dict(A=A, b=1.0 - 3.0 * xi, x=np.array([1.0, -3.0, 0.0]), tol=TOL, niter=100, rmin=1e-14,
                                   rel=False)
    # tests_lsqr.f90:227-340  Wunsch 5 x 3 -> (157.611, -38.0747, 96.0291) +- 1e-2
    a = np.array([[1.2550, 1.6731, -1.3927], [0.4891, 0.0943, -0.7829], [-0.1755, 1.8612, 1.0972],
                  [0.4189, 0.2469, -0.5990], [-0.2900, 0.7677, 0.8188]])
    out["overdetermined_2"] = dict(A=a, b=np.array([0.3511, -1.6710, 6.838, -0.8843, 3.7018]),
                                   x=np.array([157.611, -38.0747, 96.0291]), tol=1e-2, niter=100, rmin=1e-13, rel=False)
    # tests_lsqr.f90:366-440  minimum-norm (0, 1, 1)
    out["underdetermined_1"] = dict(A=np.array([[1.0, 1.0, 0.0], [2.0, 1.0, -1.0]]), b=np.array([1.0, 0.0]),
                                    x=np.array([0.0, 1.0, 1.0]), tol=TOL, niter=100, rmin=1e-13, rel=False)
    # tests_lsqr.f90:461-515  Menke 1/4 row -> d1 everywhere
    out["underdetermined_2"] = dict(A=np.full((1, 4), 0.25), b=np.array([1.0]), x=np.ones(4), tol=TOL, niter=100,
                                    rmin=1e-14, rel=True)
    # tests_lsqr.f90:532-620  (0, 1/2, 1/2, 0)
    out["underdetermined_3"] = dict(A=np.array([[1.0, 1.0, 1.0, 1.0], [1.0, -1.0, -1.0, 1.0]]), b=np.array([1.0, -1.0]),
                                    x=np.array([0.0, 0.5, 0.5, 0.0]), tol=TOL, niter=100, rmin=1e-14, rel=False)
    return out


def dense_to_csr(A):
    """Row-by-row matrix%add (sparse_matrix.f90:213-229 drops exact zeros), fp32 values, 1-based columns."""
    A32 = A.astype(np.float32)
    rp, cols, vals = [0], [], []
    for r in range(A.shape[0]):
        nz = np.nonzero(A[r] != 0.0)[0]
        cols.append((nz + 1).astype(np.int32))
        vals.append(A32[r, nz])
        rp.append(rp[-1] + nz.size)
    return np.array(rp, np.int64), np.concatenate(cols), np.concatenate(vals)


def check(case, x):
    """ftnunit's assert_comparable_real (src/libs/ftnunit.f90:353-365): |v1-v2| <= 0.5*margin*(|v1|+|v2|);
    absolute where the reference's test uses assert_true(abs(...) < tol)."""
    ref, tol = case["x"], case["tol"]
    for xi, ri in zip(x, ref):
        if case["rel"] or ri != 0.0 and tol == TOL:
            assert abs(xi - ri) <= 0.5 * tol * (abs(xi) + abs(ri)), (xi, ri)
        else:
            assert abs(xi - ri) < tol, (xi, ri)
