"""Host-side constraint builders (numpy, no GPU): the vectorised product versions in tomofast-x_amd/inversion.py against the
test oracle's cell-by-cell restatements, which tests/test_oracle_golden.py pins to the reference's own runs."""
import importlib
import os

import numpy as np
import pytest

import oracle_inversion as oinv

tfx = importlib.import_module("tomofast-x_amd")


def bits_equal(a, b):
    return a.dtype == b.dtype and a.shape == b.shape and a.tobytes() == b.tobytes()


def case(golden_dir, name):
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    dims = (int(g["nx"]), int(g["ny"]), int(g["nz"]))
    grid = [g[k] for k in ("X1", "X2", "Y1", "Y2", "Z1", "Z2")]
    sh = (dims[2], dims[1], dims[0])
    spacing = (np.abs(grid[1] - grid[0]).reshape(sh)[0, 0, :], np.abs(grid[3] - grid[2]).reshape(sh)[0, :, 0],
               np.abs(grid[5] - grid[4]).reshape(sh)[:, 0, 0])
    return g, dims, grid, spacing


def test_gradient_damping_rows_match_the_restatement(golden_dir):
    g, dims, grid, spacing = case(golden_dir, "e2e_dgrad")
    N = int(np.prod(dims))
    m = np.random.default_rng(3).standard_normal(N)
    cw = g["np1_column_weight"]
    A, ra = tfx.inversion.gradient_damping_rows(m, dims, spacing, cw, 0.7, 1.3e-3)
    B, rb = oinv.gradient_damping_rows(m, dims, grid, cw, 0.7, 1.3e-3)
    assert np.array_equal(A[0], B[0]) and np.array_equal(A[1], B[1]) and bits_equal(A[2], B[2]) and bits_equal(ra, rb)
    assert A[0].size == 3 * N + 1 and int(A[0][-1]) == 2 * (3 * N - dims[1] * dims[2] - dims[0] * dims[2] - dims[0] * dims[1])


@pytest.mark.parametrize("der_type", [1, 2])
def test_cross_gradient_rows_match_the_restatement(golden_dir, der_type):
    g, dims, grid, spacing = case(golden_dir, "e2e_xgrad")
    N = int(np.prod(dims))
    rng = np.random.default_rng(4)
    m1, m2 = rng.standard_normal(N), rng.standard_normal(N)
    cw1, cw2 = g["np1_grav_column_weight"], g["np1_magn_column_weight"]
    A, ra, ca = tfx.inversion.cross_gradient_rows(m1, m2, dims, spacing, cw1, cw2, 0.37, der_type)
    B, rb, cb = oinv.cross_gradient_rows(m1, m2, dims, grid, cw1, cw2, 0.37, der_type)
    assert np.array_equal(A[0], B[0]) and np.array_equal(A[1], B[1]) and bits_equal(A[2], B[2]) and bits_equal(ra, rb)
    assert np.allclose(ca, cb, rtol=1e-13)
    # parallel gradients: tau = 0 everywhere, the right-hand side vanishes
    _, r0, c0 = tfx.inversion.cross_gradient_rows(m1, 3.0 * m1, dims, spacing, cw1, cw2, 0.37, der_type)
    assert np.abs(r0).max() <= 1e-12 * np.abs(ra).max() and c0.max() <= 1e-24 * ca.max()
    # keepModelConstant drops exactly the entries of that model's column block
    K, rk, _ = tfx.inversion.cross_gradient_rows(m1, m2, dims, spacing, cw1, cw2, 0.37, der_type, keep_constant=(False, True))
    assert bits_equal(rk, ra) and K[1].max() <= N and np.array_equal(K[1], A[1][A[1] <= N]) and bits_equal(K[2], A[2][A[1] <= N])


@pytest.mark.parametrize("weights", [(1e-3, 2e-3), (1e-3, 0.0), (0.0, 2e-3)])
@pytest.mark.parametrize("opt_type", [1, 2])
@pytest.mark.parametrize("local", [False, True])
def test_clustering_rows_match_the_restatement(golden_dir, weights, opt_type, local):
    g, dims, grid, spacing = case(golden_dir, "e2e_clust_normal")
    N = int(np.prod(dims))
    rng = np.random.default_rng(5)
    m1, m2 = rng.uniform(-100, 400, N), rng.uniform(-0.01, 0.05, N)
    m1[:3] = 1e5                                          # far from every cluster: the exp(-100) floor (clustering.F90:576-582)
    cw1, cw2 = g["np1_grav_column_weight"], g["np1_magn_column_weight"]
    cellw = tfx.inversion.clustering_cell_weights(g["mixtures"], N, g["cell_weights"] if local else None)
    assert bits_equal(cellw, oinv.clustering_setup(g["mixtures"], N, g["cell_weights"] if local else None))
    A, ra, ca = tfx.inversion.clustering_rows(m1, m2, cw1, cw2, weights, g["mixtures"], cellw, opt_type)
    B, rb, cb = oinv.clustering_rows(m1, m2, cw1, cw2, weights, g["mixtures"], cellw, opt_type)
    assert np.array_equal(A[0], B[0]) and np.array_equal(A[1], B[1])
    assert np.allclose(A[2], B[2], rtol=1e-6, atol=0) and np.allclose(ra, rb, rtol=1e-12, atol=1e-300) and np.allclose(ca, cb, rtol=1e-12)
    assert A[0].size == 2 * N + 1
    for i in range(2):                                    # a problem without a clustering weight contributes empty rows and zero rhs
        if weights[i] == 0.0:
            assert int(A[0][(i + 1) * N] - A[0][i * N]) == 0 and not ra[i * N:(i + 1) * N].any()


def test_admm_projection_matches_the_restatement():
    rng = np.random.default_rng(6)
    n = 500
    bounds = np.array([-20.0, 20.0, 90.0, 130.0, 220.0, 260.0])
    st = tfx.inversion.AdmmState(n)
    z, u = np.zeros(n), np.zeros(n)
    for _ in range(4):
        x = rng.uniform(-60, 300, n)
        x0 = st.iterate_admm_arrays(x, bounds)
        r0 = oinv.admm_iterate(z, u, x, bounds)
        assert bits_equal(x0, r0) and bits_equal(st.z, z) and bits_equal(st.u, u)


def test_column_restriction_of_constraint_blocks(golden_dir):
    """restrict_columns / restrict_columns_blocks: the column-partitioned form of a constraint block (rows replicated, each rank
    its own cells of every model) - the pieces of all ranks put back together give the block."""
    g, dims, grid, spacing = case(golden_dir, "e2e_xgrad")
    N = int(np.prod(dims))
    rng = np.random.default_rng(8)
    m1, m2 = rng.standard_normal(N), rng.standard_normal(N)
    G, rhs, _ = tfx.inversion.cross_gradient_rows(m1, m2, dims, spacing, g["np1_grav_column_weight"], g["np1_magn_column_weight"], 0.5, 2)
    nrows = G[0].size - 1
    dense = np.zeros((nrows, 2 * N))
    for r in range(nrows):
        dense[r, G[1][G[0][r]:G[0][r + 1]] - 1] = G[2][G[0][r]:G[0][r + 1]]
    cuts = [0, 37, 150, N]
    back = np.zeros_like(dense)
    for c0, c1 in zip(cuts[:-1], cuts[1:]):
        L = tfx.inversion.restrict_columns_blocks(G, c0, c1, N, 2)
        nl = c1 - c0
        assert L[0].size == nrows + 1 and (L[1].size == 0 or (L[1].min() >= 1 and L[1].max() <= 2 * nl))
        for r in range(nrows):
            cols = L[1][L[0][r]:L[0][r + 1]] - 1
            assert np.all(np.diff(cols) > 0)                       # ascending inside a row, as the upload wants
            blk, cell = cols // nl, cols % nl
            back[r, blk * N + c0 + cell] = L[2][L[0][r]:L[0][r + 1]]
    assert np.array_equal(back, dense)
    # one model: restrict_columns
    Gd, _ = tfx.inversion.gradient_damping_rows(m1, dims, spacing, g["np1_grav_column_weight"], 1.0, 1e-3)
    one = tfx.inversion.restrict_columns(Gd, 37, 150)
    keep = (Gd[1] > 37) & (Gd[1] <= 150)
    assert np.array_equal(one[1], Gd[1][keep] - 37) and np.array_equal(one[2], Gd[2][keep]) and int(one[0][-1]) == int(keep.sum())


def test_canonical_csr_of_rows_built_with_add(tmp_path):
    """tfx_reference_api::api_canonical_csr (what the drop-in sparse_matrix / lsqr_solver modules pass to tfx_matrix_upload_csr): rows
    as the reference's add() builds them - any column order, repeated columns, empty rows (sparse_matrix.f90:213-229) - come out with
    strictly ascending columns, repeated columns merged by adding their values in the order they were added, and `where` maps every
    input entry to its output entry.  Compiled here with amdflang against the host objects (no GPU call is made)."""
    import subprocess
    host = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tomofast-x_amd", "host")
    fc = "/opt/rocm/bin/amdflang"
    if not os.path.isfile(fc) or not os.path.isfile(os.path.join(host, "tfx_reference_api.o")):
        pytest.skip("no Fortran compiler / host objects (build() makes them where amdflang exists)")
    exe = str(tmp_path / "canonical_csr_check")
    src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fortran", "canonical_csr_check.f90")
    cmd = [fc, "-O1", "-I" + host, src] + [os.path.join(host, o) for o in ("tfx_binding.o", "tfx_host_mpi.o", "tfx_reference_api.o")] + \
          ["-L" + os.path.dirname(host), "-ltfx", "-L" + os.path.join(host, "mpilib"), "-lmpifort", "-lmpi",
           "-Wl,-rpath," + os.path.dirname(host), "-Wl,-rpath," + os.path.join(host, "mpilib"), "-o", exe]
    p = subprocess.run(cmd, capture_output=True, text=True, cwd=str(tmp_path))
    assert p.returncode == 0, p.stderr[-3000:]
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = out.stdout.split("\n")
    rp = [int(v) for v in next(l for l in lines if l.startswith("rp")).split()[1:]]
    ent = [(int(l.split()[1]), int(l.split()[2]), float(l.split()[3])) for l in lines if l.startswith("entry")]
    where = [int(v) for v in next(l for l in lines if l.startswith("where")).split()[1:]]
    assert rp == [0, 3, 3, 4, 5, 8]
    assert ent == [(1, 1, 4.0), (1, 2, 2.0), (1, 7, 1.5), (3, 3, 0.25), (4, 9, 8.0), (5, 1, 5.0), (5, 4, 6.0), (5, 5, 4.0)]
    assert where == [3, 2, 3, 1, 4, 4, 4, 5, 8, 7, 8, 7, 6]
