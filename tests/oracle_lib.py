"""ctypes binding of oracle/libtfx_oracle.so (the CPU checker).  Test infrastructure only:
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg - never by the product."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ODIR = os.path.join(ROOT, "oracle")
_lib = None

c_dp = C.POINTER(C.c_double)
c_fp = C.POINTER(C.c_float)
c_ip = C.POINTER(C.c_int32)
c_lp = C.POINTER(C.c_int64)


def build():
    subprocess.run(["make", "-s", "-C", ODIR], check=True)


def lib():
    global _lib
    if _lib is None:
        so = os.path.join(ODIR, "libtfx_oracle.so")
        if not os.path.isfile(so) or os.path.getmtime(so) < os.path.getmtime(os.path.join(ODIR, "tfx_oracle.c")):
            build()
        _lib = C.CDLL(so)
        _lib.orc_compress_row.restype = C.c_int64
        _lib.orc_build_row_grav.restype = C.c_int64
    return _lib


def dp(a):
    return a.ctypes.data_as(c_dp)


def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def graviprism_z(grid, xd, yd, zd):
    X1, X2, Y1, Y2, Z1, Z2 = [f64(g) for g in grid]
    n = X1.size
    line = np.empty(n)
    ierr = lib().orc_graviprism_z(C.c_int64(n), dp(X1), dp(X2), dp(Y1), dp(Y2), dp(Z1), dp(Z2),
                                  C.c_double(xd), C.c_double(yd), C.c_double(zd), dp(line))
    return ierr, line


def graviprism_full(grid, xd, yd, zd):
    """-> (ierr, lines[3, n]); components X, Y, Z (LineX, LineY, LineZ of gravity_field.f90:41-126)."""
    X1, X2, Y1, Y2, Z1, Z2 = [f64(g) for g in grid]
    n = X1.size
    line = np.empty(3 * n)
    ierr = lib().orc_graviprism_full(C.c_int64(n), dp(X1), dp(X2), dp(Y1), dp(Y2), dp(Z1), dp(Z2),
                                     C.c_double(xd), C.c_double(yd), C.c_double(zd), dp(line))
    return ierr, line.reshape(3, n)


def dircos(incl, decl, azim):
    magv = np.empty(3)
    lib().orc_dircos(C.c_double(incl), C.c_double(decl), C.c_double(azim), dp(magv))
    return magv


def magprism_tmi(grid, xd, yd, zd, magv, intensity):
    X1, X2, Y1, Y2, Z1, Z2 = [f64(g) for g in grid]
    n = X1.size
    line = np.empty(n)
    magv = f64(magv)
    ierr = lib().orc_magprism_tmi(C.c_int64(n), dp(X1), dp(X2), dp(Y1), dp(Y2), dp(Z1), dp(Z2), C.c_double(xd), C.c_double(yd),
                                  C.c_double(zd), dp(magv), C.c_double(intensity), dp(line))
    return ierr, line


def magprism(grid, xd, yd, zd, magv, intensity, ncm, ncd):
    """-> (ierr, lines[ncd, ncm, n])  (Fortran sensit_line(n, ncm, ncd))."""
    X1, X2, Y1, Y2, Z1, Z2 = [f64(g) for g in grid]
    n = X1.size
    line = np.empty(n * ncm * ncd)
    magv = f64(magv)
    ierr = lib().orc_magprism(C.c_int64(n), ncm, ncd, dp(X1), dp(X2), dp(Y1), dp(Y2), dp(Z1), dp(Z2), C.c_double(xd),
                              C.c_double(yd), C.c_double(zd), dp(magv), C.c_double(intensity), dp(line))
    return ierr, line.reshape(ncd, ncm, n)


def gradiprism(grid, xd, yd, zd, only_zz):
    """-> (ierr, lines[1 or 6, n]); component order XX, YY, ZZ, XY, YZ, ZX."""
    X1, X2, Y1, Y2, Z1, Z2 = [f64(g) for g in grid]
    n = X1.size
    nc = 1 if only_zz else 6
    line = np.empty(n * nc)
    ierr = lib().orc_gradiprism(C.c_int64(n), 1 if only_zz else 0, dp(X1), dp(X2), dp(Y1), dp(Y2), dp(Z1), dp(Z2),
                                C.c_double(xd), C.c_double(yd), C.c_double(zd), dp(line))
    return ierr, line.reshape(nc, n)


def compress_line(line, cw, dims, ctype, K):
    """One build-loop line: weight -> wavelet -> threshold -> compaction.  -> cols (1-based), vals, error_r."""
    line = f64(line).copy()
    cw = f64(cw)
    N = line.size
    cols = np.empty(N, np.int32)
    vals = np.empty(N, np.float32)
    err = C.c_double()
    lib().orc_compress_line.restype = C.c_int64
    nel = lib().orc_compress_line(C.c_int64(N), dims[0], dims[1], dims[2], dp(cw), ctype, C.c_int64(K), dp(line),
                                  cols.ctypes.data_as(c_ip), vals.ctypes.data_as(c_fp), C.byref(err))
    return cols[:nel].copy(), vals[:nel].copy(), err.value


def rowgen(kind, grid, o, field=None, ncm=1, ncd=1):
    """Lines of one observation, [ncd, ncm, N].  kind: 'gz' | 'g3' | 'gzz' | 'ftg' | 'mag'."""
    if kind == "gz":
        ierr, line = graviprism_z(grid, o[0], o[1], o[2])
        lines = line.reshape(1, 1, -1)
    elif kind == "g3":
        ierr, l = graviprism_full(grid, o[0], o[1], o[2])
        lines = l.reshape(3, 1, -1)
    elif kind in ("gzz", "ftg"):
        ierr, l = gradiprism(grid, o[0], o[1], o[2], kind == "gzz")
        lines = l.reshape(l.shape[0], 1, -1)
    else:
        ierr, lines = magprism(grid, o[0], o[1], o[2], dircos(*field[:3]), field[3], ncm, ncd)
    assert ierr == 0, ierr
    return lines


def build_matrix_comp(kind, grid, dims, cw, obs, ctype, rate, field=None, ncm=1, ncd=1):
    """Multi-component kernel -> CSR with ndata*ncd rows and ncm*N columns (1-based; model component k occupies
    columns k*N + cell, sensitivity_gravmag.F90:829-846), nnz histogram over cells, mean compression error."""
    N = int(np.prod(dims))
    K = int(rate * N) if ctype > 0 else N
    rp, cs, vs, errs = [0], [], [], []
    hist = np.zeros(N, np.int64)
    for o in np.asarray(obs):
        lines = rowgen(kind, grid, o, field, ncm, ncd)
        for d in range(lines.shape[0]):
            n_row = 0
            for k in range(lines.shape[1]):
                c, v, e = compress_line(lines[d, k], cw, dims, ctype, K)
                hist += np.bincount(c - 1, minlength=N)
                cs.append(c + k * N)
                vs.append(v)
                errs.append(e)
                n_row += c.size
            rp.append(rp[-1] + n_row)
    return (np.array(rp, np.int64), np.concatenate(cs).astype(np.int32), np.concatenate(vs), hist.astype(np.int32),
            float(np.sum(errs) / len(errs)))


def column_weight_type1(grid, power=2.0, Z0=0.0, multiplier=4.0e3):
    X1, X2, Y1, Y2, Z1, Z2 = [f64(g) for g in grid]
    n = X1.size
    cw = np.empty(n)
    ierr = lib().orc_column_weight_type1(C.c_int64(n), dp(X1), dp(X2), dp(Y1), dp(Y2), dp(Z1), dp(Z2),
                                         C.c_double(power), C.c_double(Z0), C.c_double(multiplier), dp(cw))
    assert ierr == 0, ierr
    return cw


def column_weight_type2(grid, obs, power=2.0, beta=1.0, multiplier=4.0e3):
    X1, X2, Y1, Y2, Z1, Z2 = [f64(g) for g in grid]
    obs = np.asarray(obs, np.float64)
    xd, yd, zd = f64(obs[:, 0]), f64(obs[:, 1]), f64(obs[:, 2])
    n = X1.size
    cw = np.empty(n)
    ierr = lib().orc_column_weight_type2(C.c_int64(n), dp(X1), dp(X2), dp(Y1), dp(Y2), dp(Z1), dp(Z2), C.c_int64(xd.size), dp(xd),
                                         dp(yd), dp(zd), C.c_double(power), C.c_double(beta), C.c_double(multiplier), dp(cw))
    assert ierr == 0, ierr
    return cw


def column_weight_type3(grid, obs, power=2.0, multiplier=4.0e3):
    X1, X2, Y1, Y2, Z1, Z2 = [f64(g) for g in grid]
    obs = np.asarray(obs, np.float64)
    xd, yd, zd = f64(obs[:, 0]), f64(obs[:, 1]), f64(obs[:, 2])
    n = X1.size
    cw = np.empty(n)
    ierr = lib().orc_column_weight_type3(C.c_int64(n), dp(X1), dp(X2), dp(Y1), dp(Y2), dp(Z1), dp(Z2), C.c_int64(xd.size), dp(xd),
                                         dp(yd), dp(zd), C.c_double(power), C.c_double(multiplier), dp(cw))
    assert ierr == 0, ierr
    return cw


def wavelet(a, n1, n2, n3, wtype, inverse=False):
    s = f64(a).copy()
    assert s.size == n1 * n2 * n3
    fn = lib().orc_inverse_wavelet if inverse else lib().orc_forward_wavelet
    ierr = fn(dp(s), n1, n2, n3, wtype)
    assert ierr == 0
    return s


def compress_row(row, K):
    row = f64(row)
    N = row.size
    cols = np.empty(N, np.int32)
    vals = np.empty(N, np.float32)
    thr = C.c_double()
    cd = C.c_double()
    nel = lib().orc_compress_row(dp(row), C.c_int64(N), C.c_int64(K), cols.ctypes.data_as(c_ip),
                                 vals.ctypes.data_as(c_fp), C.byref(thr), C.byref(cd))
    return cols[:nel].copy(), vals[:nel].copy(), thr.value, cd.value


def build_row_grav(grid, dims, cw, obs, ctype, K):
    X1, X2, Y1, Y2, Z1, Z2 = [f64(g) for g in grid]
    cw = f64(cw)
    N = X1.size
    nx, ny, nz = dims
    work = np.empty(N)
    cols = np.empty(N, np.int32)
    vals = np.empty(N, np.float32)
    err = C.c_double()
    ierr = C.c_int()
    nel = lib().orc_build_row_grav(C.c_int64(N), nx, ny, nz, dp(X1), dp(X2), dp(Y1), dp(Y2), dp(Z1), dp(Z2), dp(cw),
                                   C.c_double(obs[0]), C.c_double(obs[1]), C.c_double(obs[2]), ctype, C.c_int64(K),
                                   dp(work), cols.ctypes.data_as(c_ip), vals.ctypes.data_as(c_fp), C.byref(err),
                                   C.byref(ierr))
    assert ierr.value == 0, ierr.value
    return cols[:nel].copy(), vals[:nel].copy(), err.value


def build_matrix_grav(grid, dims, cw, obs, ctype, rate):
    """All rows -> CSR (rowptr int64 0-based, cols int32 1-based, vals fp32), nnz histogram, mean error."""
    N = int(np.prod(dims))
    K = int(rate * N) if ctype > 0 else N          # sensitivity_gravmag.F90:64-77
    rp = [0]
    cs, vs, errs = [], [], []
    for o in np.asarray(obs):
        c, v, e = build_row_grav(grid, dims, cw, o, ctype, K)
        cs.append(c)
        vs.append(v)
        errs.append(e)
        rp.append(rp[-1] + c.size)
    cols = np.concatenate(cs)
    hist = np.bincount(cols - 1, minlength=N).astype(np.int32)
    return np.array(rp, np.int64), cols, np.concatenate(vs), hist, float(np.sum(errs) / len(errs))


def build_matrix_mag(grid, dims, cw, obs, field, ctype, rate):
    """Magnetic (TMI, scalar) rows -> CSR like build_matrix_grav."""
    N = int(np.prod(dims))
    K = int(rate * N) if ctype > 0 else N
    magv = dircos(*field[:3])
    rp, cs, vs = [0], [], []
    for o in np.asarray(obs):
        ierr, row = magprism_tmi(grid, o[0], o[1], o[2], magv, field[3])
        assert ierr == 0
        row = row * f64(cw)
        if ctype > 0:
            c, v, _, _ = compress_row(wavelet(row, dims[0], dims[1], dims[2], ctype), K)
        else:
            c, v = np.arange(1, N + 1, dtype=np.int32), row.astype(np.float32)
        cs.append(c)
        vs.append(v)
        rp.append(rp[-1] + c.size)
    return np.array(rp, np.int64), np.concatenate(cs), np.concatenate(vs)


def partition(nnz, P):
    nnz = np.ascontiguousarray(nnz, np.int32)
    nel = np.zeros(P, np.int32)
    nz = np.zeros(P, np.int64)
    lib().orc_partition(nnz.ctypes.data_as(c_ip), C.c_int64(nnz.size), P, nel.ctypes.data_as(c_ip),
                        nz.ctypes.data_as(c_lp))
    return nel, nz


def _csr(rowptr, cols, vals):
    return (np.ascontiguousarray(rowptr, np.int64), np.ascontiguousarray(cols, np.int32),
            np.ascontiguousarray(vals, np.float32))


def spmv(rowptr, cols, vals, x, b=None):
    rowptr, cols, vals = _csr(rowptr, cols, vals)
    nrows = rowptr.size - 1
    x = f64(x)
    b = np.zeros(nrows) if b is None else f64(b).copy()
    lib().orc_spmv_add(C.c_int64(nrows), rowptr.ctypes.data_as(c_lp), cols.ctypes.data_as(c_ip),
                       vals.ctypes.data_as(c_fp), dp(x), dp(b))
    return b


def spmtv(rowptr, cols, vals, x, ncols, b=None):
    rowptr, cols, vals = _csr(rowptr, cols, vals)
    nrows = rowptr.size - 1
    x = f64(x)
    b = np.zeros(ncols) if b is None else f64(b).copy()
    lib().orc_spmtv_add(C.c_int64(nrows), rowptr.ctypes.data_as(c_lp), cols.ctypes.data_as(c_ip),
                        vals.ctypes.data_as(c_fp), dp(x), dp(b))
    return b


def normalize_columns(rowptr, cols, vals, ncols):
    """sparse_matrix.f90:414-443.  Returns (column_norm, normalised fp32 values)."""
    rowptr, cols, vals = _csr(rowptr, cols, vals)
    vals = vals.copy()
    norm = np.zeros(ncols)
    lib().orc_normalize_columns(C.c_int64(rowptr.size - 1), C.c_int64(ncols), rowptr.ctypes.data_as(c_lp), cols.ctypes.data_as(c_ip),
                                vals.ctypes.data_as(c_fp), dp(norm))
    return norm, vals


def lsqr(S, Cm, ncols, b, niter, rmin=1e-13, gamma=0.0, target_misfit=0.0, spatial=None):
    """S, Cm: (rowptr, cols, vals) CSR.  Returns x, iters, r.
    spatial = (wavelet_type, n1, n2, n3): WAVELET_DOMAIN = false (unknowns spatial, S applied through the transform)."""
    s_rp, s_c, s_v = _csr(*S)
    c_rp, c_c, c_v = _csr(*Cm)
    nl_s, nl_c = s_rp.size - 1, c_rp.size - 1
    u = f64(b).copy()
    assert u.size == nl_s + nl_c
    x = np.zeros(ncols)
    r = C.c_double()
    lib().orc_lsqr_solve_sensit_wd.restype = C.c_int
    wt, n1, n2, n3 = spatial if spatial else (0, 0, 0, 0)
    it = lib().orc_lsqr_solve_sensit_wd(C.c_int64(nl_s), C.c_int64(nl_c), C.c_int64(ncols), int(niter),
                                        C.c_double(rmin), C.c_double(gamma), C.c_double(target_misfit),
                                        s_rp.ctypes.data_as(c_lp), s_c.ctypes.data_as(c_ip), s_v.ctypes.data_as(c_fp),
                                        c_rp.ctypes.data_as(c_lp), c_c.ctypes.data_as(c_ip), c_v.ctypes.data_as(c_fp),
                                        dp(u), dp(x), C.byref(r), int(wt), int(n1), int(n2), int(n3))
    return x, it, r.value


def diag_csr(d):
    """CSR of diag(d) (fp32 values, zero entries dropped like sparse_matrix.f90:219)."""
    d = np.asarray(d, np.float32)
    nzm = d != 0
    rc = nzm.astype(np.int64)
    return np.concatenate([[0], np.cumsum(rc)]).astype(np.int64), (np.nonzero(nzm)[0] + 1).astype(np.int32), d[nzm]


def calc_data(model, cw, dims, ctype, S, problem_weight, data_weight):
    s_rp, s_c, s_v = _csr(*S)
    nd = s_rp.size - 1
    N = int(np.prod(dims))
    model, cw, dw = f64(model), f64(cw), f64(data_weight)
    work = np.empty(N)
    out = np.empty(nd)
    ierr = lib().orc_calc_data(C.c_int64(N), dims[0], dims[1], dims[2], C.c_int64(nd), dp(model), dp(cw), ctype,
                               s_rp.ctypes.data_as(c_lp), s_c.ctypes.data_as(c_ip), s_v.ctypes.data_as(c_fp),
                               C.c_double(problem_weight), dp(dw), dp(work), dp(out))
    assert ierr == 0
    return out


def rc_to_rowptr(rc):
    return np.concatenate([[0], np.cumsum(np.asarray(rc, np.int64))]).astype(np.int64)
