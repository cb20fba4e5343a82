"""Major inversion loop restated on top of the oracle's C functions (test infrastructure).

Follows src/problem_joint_gravmag.F90:473-547 and src/inversion/joint_inverse_problem.F90:393-573 for a
single gravity problem on one rank with WAVELET_DOMAIN = true (or compression off):
  residuals (problem_joint_gravmag.F90:666-675) -> RHS + damping / ADMM diagonal blocks
  (damping.F90:97-234, admm_method.F90:70-134) -> lsqr_solve_sensit -> inverse wavelet + rescale
  (joint_inverse_problem.F90:559-571) -> model update -> calculate_data (model.F90:220-307).
"""
import math

import numpy as np

import oracle_lib as orc



def libm_pow(a, p):
    """a ** p element by element through the C library's pow (what the reference's `**` compiles to), not numpy's vectorised one (which
    differs from it in the last bit on some arguments)."""
    import math
    return np.array([math.pow(float(v), float(p)) for v in np.asarray(a, np.float64).ravel()], np.float64).reshape(np.shape(a))

def admm_iterate(z, u, x, bounds):
    """admm_method.F90:70-134 with global bounds [lo1 hi1 lo2 hi2 ...]; updates z, u in place; returns x0."""
    lo, hi = np.asarray(bounds[0::2]), np.asarray(bounds[1::2])
    arg = x + u
    inside = ((lo[None, :] <= arg[:, None]) & (arg[:, None] <= hi[None, :])).any(1)
    cand = np.stack([np.abs(np.stack([lo, hi], 1).ravel()[None, :] - arg[:, None])], 0)[0]
    ends = np.stack([lo, hi], 1).ravel()
    closest = ends[np.argmin(cand, axis=1)]          # first minimum wins, like the strict '<' of :113-121
    z[:] = np.where(inside, arg, closest)
    u[:] = u + x - z
    return z - u


def wavelet_comp(v, dims, ctype, ncm, inverse=False):
    """Per-component 3-D transform of a component-major vector [k*N + cell] (wavelet_utils.F90:37-72 loops k)."""
    N = int(np.prod(dims))
    return np.concatenate([orc.wavelet(v[k * N:(k + 1) * N], dims[0], dims[1], dims[2], ctype, inverse=inverse)
                           for k in range(ncm)])


def calc_data_comp(model, cw, dims, ctype, S, pw, dw, ncm):
    """model_calculate_data (model.F90:220-307) for ncm model components; data index = idata*ncd + d."""
    scaled = np.where(cw != 0.0, model / cw, 0.0)
    if ctype > 0:
        scaled = wavelet_comp(scaled, dims, ctype, ncm)
    return orc.spmv(S[0], S[1], S[2], scaled) / pw / dw


def run_inversion(S, cw, dims, ctype, d_obs, nmajor, nminor, alpha=0.0, rmin=1e-13, pw=1.0, m0=None, m_prior=None,
                  admm=None, lsqr=None, calc_data=None, ncm=1, data_weight=None):
    """S = (rowptr, cols, vals) CSR.  admm = dict(bounds=..., rho=...) or None.
    lsqr / calc_data: replaceable callables (default: oracle) so that tests can run the SAME loop on the HIP path.
    ncm > 1: model vectors are component-major [k*N + cell] (the reference's model%val(:, k) flattened), the data vector is
    [idata*ncd + d]; the damping block repeats per component (joint_inverse_problem.F90:456-463).
    Returns final model, calculated data and per-iteration records."""
    N1 = int(np.prod(dims))
    if ncm > 1:
        cw = np.tile(np.asarray(cw, np.float64), ncm)
        assert admm is None
        dims_w = dims
        class _W:                                   # per-component transforms behind the single-component call sites
            @staticmethod
            def wavelet(v, n1, n2, n3, t, inverse=False):
                return wavelet_comp(v, dims_w, t, ncm, inverse)
        worc = _W
    else:
        worc = orc
    N = N1 * ncm
    nd = d_obs.size
    dw = np.ones(nd) if data_weight is None else np.asarray(data_weight, np.float64)   # S must carry float32(pw * dw) per row
    m = np.zeros(N) if m0 is None else np.array(m0, np.float64)
    mp = np.zeros(N) if m_prior is None else np.asarray(m_prior, np.float64)
    lsqr = lsqr or (lambda blocks, b, niter: orc.lsqr(S, _blocks_csr(blocks, N), N, b, niter, rmin)[:3])
    if calc_data is None:
        if ncm > 1:
            calc_data = lambda model: calc_data_comp(model, cw, dims, ctype, S, pw, dw, ncm)
        else:
            calc_data = lambda model: orc.calc_data(model, cw, dims, ctype, S, pw, dw)
    d_calc = calc_data(m)
    z, u = np.zeros(N), np.zeros(N)
    hist = []
    for it in range(nmajor):
        res = dw * (d_obs - d_calc)                                       # problem_joint_gravmag.F90:666-675
        rhs = [pw * res]                                                  # joint_inverse_problem.F90:379-387
        blocks = []
        if alpha != 0.0:                                                  # damping.F90:97-234
            md = (m - mp) / cw
            if ctype > 0:
                md = worc.wavelet(md, dims[0], dims[1], dims[2], ctype)
            blocks.append(np.full(N, np.float32(alpha * pw), np.float32))
            rhs.append(-alpha * pw * md)
        if admm is not None:                                              # joint_inverse_problem.F90:497-527
            x0 = admm_iterate(z, u, m, admm["bounds"])
            md = (m - x0) / cw
            if ctype > 0:
                md = worc.wavelet(md, dims[0], dims[1], dims[2], ctype)
            blocks.append(np.full(N, np.float32(admm["rho"] * pw), np.float32))
            rhs.append(-admm["rho"] * pw * md)
        x, iters, r = lsqr(blocks, np.concatenate(rhs), nminor)
        dm = worc.wavelet(x, dims[0], dims[1], dims[2], ctype, inverse=True) if ctype > 0 else x.copy()
        dm = dm * cw                                                      # joint_inverse_problem.F90:570
        m = m + dm
        d_calc = calc_data(m)
        hist.append(dict(iters=iters, r=r, cost=float(np.linalg.norm(d_calc - d_obs) / np.linalg.norm(d_obs))))
    return m, d_calc, hist


def _blocks_csr(blocks, N):
    """Stack of diagonal blocks -> one CSR (what damping%add builds row by row)."""
    rp, cs, vs = [np.zeros(1, np.int64)], [], []
    off = 0
    for d in blocks:
        r, c, v = orc.diag_csr(d)
        rp.append(r[1:] + off)
        off += int(r[-1])
        cs.append(c)
        vs.append(v)
    if not blocks:
        return np.zeros(1, np.int64), np.zeros(0, np.int32), np.zeros(0, np.float32)
    return np.concatenate(rp), np.concatenate(cs), np.concatenate(vs)


def run_joint_inversion(problems, dims, ctype, nmajor, nminor, rmin=1e-13, lsqr=None, calc_data=None):
    """Joint inversion of several problems on one grid without structural coupling (cross-gradient / clustering weights 0):
    one LSQR system  [pw_1 S_1, 0; 0, pw_2 S_2; damping_1; damping_2] [x_1; x_2] = [pw_1 res_1; pw_2 res_2; ...]
    (joint_inverse_problem.F90:393-573: block layout :712-739, right-hand side :379-387, damping per problem :448-463).
    problems: list of dict(S=(rowptr, cols, vals) already scaled by float32(pw) like read_sensitivity_kernel :834-843,
    cw, d_obs, pw, alpha).  lsqr / calc_data replaceable (HIP path).  Returns models, data, history."""
    N = int(np.prod(dims))
    P = len(problems)
    m = [np.zeros(N) for _ in problems]
    if lsqr is None:
        # block-diagonal CSR: problem i occupies columns [i*N, (i+1)*N)
        rp = [np.zeros(1, np.int64)]
        cs, vs = [], []
        off = 0
        for i, pr in enumerate(problems):
            r, c, v = pr["S"]
            rp.append(np.asarray(r[1:], np.int64) + off)
            off += int(r[-1])
            cs.append(np.asarray(c, np.int64) + i * N)
            vs.append(v)
        Sj = (np.concatenate(rp), np.concatenate(cs).astype(np.int32), np.concatenate(vs))
        lsqr = lambda blocks, b, niter: orc.lsqr(Sj, _blocks_csr(blocks, P * N), P * N, b, niter, rmin)[:3]
    if calc_data is None:
        calc_data = lambda i, model: orc.calc_data(model, problems[i]["cw"], dims, ctype, problems[i]["S"], problems[i]["pw"],
                                                   np.ones(problems[i]["d_obs"].size))
    d = [calc_data(i, m[i]) for i in range(P)]
    hist = []
    for it in range(nmajor):
        rhs = [pr["pw"] * (pr["d_obs"] - d[i]) for i, pr in enumerate(problems)]
        blocks = []
        for i, pr in enumerate(problems):
            if pr["alpha"] != 0.0:
                md = m[i] / pr["cw"]
                if ctype > 0:
                    md = orc.wavelet(md, dims[0], dims[1], dims[2], ctype)
                blk = np.zeros(P * N, np.float32)
                blk[i * N:(i + 1) * N] = np.float32(pr["alpha"] * pr["pw"])
                r = np.zeros(P * N)
                r[i * N:(i + 1) * N] = -pr["alpha"] * pr["pw"] * md
                blocks.append(blk)
                rhs.append(r)
        x, iters, r = lsqr(blocks, np.concatenate(rhs), nminor)
        for i, pr in enumerate(problems):
            xi = x[i * N:(i + 1) * N]
            dm = orc.wavelet(xi, dims[0], dims[1], dims[2], ctype, inverse=True) if ctype > 0 else xi.copy()
            m[i] = m[i] + dm * pr["cw"]
            d[i] = calc_data(i, m[i])
        hist.append(dict(iters=iters, r=r))
    return m, d, hist


def gradient_damping_rows(m, dims, grid, cw, pw, beta, reference_order=False):
    """damping_gradient%add for the three directions (src/inversion/damping_gradient.F90:94-205; forward differences
    gradient.F90:77-81; spacings grid.F90:371-391): 3 N rows, two entries each (none in the last layer of a direction),
    values cast to fp32 like sparse_matrix.f90:226.  Returns (rowptr, cols 1-based ascending, vals fp32), rhs."""
    nx, ny, nz = dims
    N = nx * ny * nz
    X1, X2, Y1, Y2, Z1, Z2 = grid
    idx = np.arange(N).reshape(nz, ny, nx)
    hx = np.abs(X2 - X1).reshape(nz, ny, nx)[0, 0, :]
    hy = np.abs(Y2 - Y1).reshape(nz, ny, nx)[0, :, 0]
    hz = np.abs(Z2 - Z1).reshape(nz, ny, nx)[:, 0, 0]
    f = np.asarray(m, np.float64).reshape(nz, ny, nx)
    rp, cols, vals, rhs = [0], [], [], []
    for direction in (1, 2, 3):
        for k in range(nz):
            for j in range(ny):
                for i in range(nx):
                    last = (i == nx - 1, j == ny - 1, k == nz - 1)[direction - 1]
                    if last:
                        rp.append(rp[-1])
                        rhs.append(0.0)
                        continue
                    delta = (hx[i], hy[j], hz[k])[direction - 1]
                    nb = (idx[k, j, i + 1] if direction == 1 else idx[k, j + 1, i] if direction == 2 else idx[k + 1, j, i])
                    me = idx[k, j, i]
                    gval = (f.ravel()[nb] - f.ravel()[me]) / delta
                    v1 = (1.0 / delta) * pw * beta * cw[nb]
                    v2 = -(1.0 / delta) * pw * beta * cw[me]
                    if reference_order:    # the order damping_gradient%add puts them into the row (:185-192): f(i + 1) first, then f(i) - the order
                        cols += [nb + 1, me + 1]                    # the reference's products sum them in (sparse_matrix.f90:320, :399)
                        vals += [np.float32(v1), np.float32(v2)]
                    else:                  # ascending columns, as the device layout takes them (me < nb always)
                        cols += [me + 1, nb + 1]
                        vals += [np.float32(v2), np.float32(v1)]
                    rp.append(rp[-1] + 2)
                    rhs.append(-pw * beta * gval)
    return (np.array(rp, np.int64), np.array(cols, np.int32), np.array(vals, np.float32)), np.array(rhs)


def admm_iterate_local(z, u, x, bounds):
    """admm_method.F90:70-134 with per-cell intervals bounds[cell] = [lo1 hi1 lo2 hi2 ...]; plain loops (the checker)."""
    for p in range(x.size):
        a = x[p] + u[p]
        b = bounds[p]
        inside = False
        for j in range(0, b.size, 2):
            if b[j] <= a <= b[j + 1]:
                inside = True
                z[p] = a
                break
        if not inside:
            best, mind = a, 1e30
            for j in range(b.size):
                if abs(b[j] - a) < mind:
                    mind, best = abs(b[j] - a), b[j]
            z[p] = best
    u[:] = u + x - z
    return z - u


def run_inversion_gradient_damping(S, cw, dims, grid, ctype, d_obs, nmajor, nminor, alpha, beta, rmin=1e-13, pw=1.0, lsqr=None,
                                   calc_data=None, norm_power=2.0, admm=None, damping_weight=None):
    """Major loop with model damping + gradient damping: WAVELET_DOMAIN = false (joint_inverse_problem.F90:189-198), i.e. the
    unknowns are the spatial (depth-weighted) model update and nothing is transformed back after the solve (:559-571)."""
    N = int(np.prod(dims))
    m = np.zeros(N)
    lsqr = lsqr or (lambda Cm, b, niter: orc.lsqr(S, Cm, N, b, niter, rmin, spatial=(ctype, dims[0], dims[1], dims[2]) if ctype > 0 else None)[:3])
    calc_data = calc_data or (lambda model: orc.calc_data(model, cw, dims, ctype, S, pw, np.ones(d_obs.size)))
    d = calc_data(m)
    hist = []
    for it in range(nmajor):
        rhs = [pw * (d_obs - d)]
        blocks = []
        if alpha != 0.0:                                   # damping.F90:97-234 without the transform
            md = np.where(cw != 0.0, m / np.where(cw != 0.0, cw, 1.0), 0.0)     # damping.F90:129-135
            mult = np.ones(N)
            if norm_power != 2.0:                          # Lp norm multiplier, damping.F90:171-175, :250-262
                nzm = md != 0.0
                mult[nzm] = libm_pow(np.abs(md[nzm]), norm_power / 2.0 - 1.0)
            val, r = alpha * pw * mult, -alpha * pw * md * mult   # the reference's order: alpha * pw, Lp multiplier, local weight
            if damping_weight is not None:                 # local weight = local alpha (damping.F90:177-180, :264-267)
                val, r = val * damping_weight, r * damping_weight
            blocks.append(orc.diag_csr(val.astype(np.float32)))
            rhs.append(r)
        if beta != 0.0:
            G, grhs = gradient_damping_rows(m, dims, grid, cw, pw, beta, reference_order=True)
            blocks.append(G)
            rhs.append(grhs)
        if admm is not None:                               # local bounds + local weight (joint_inverse_problem.F90:497-527)
            if it == 0:
                az, au = np.zeros(N), np.zeros(N)
            x0 = admm_iterate_local(az, au, m, admm["bounds"])
            lw = admm["weight"]
            blocks.append(orc.diag_csr((admm["rho"] * pw * lw).astype(np.float32)))
            rhs.append(-admm["rho"] * pw * ((m - x0) / cw) * lw)
        rp = [np.zeros(1, np.int64)]
        off = 0
        for b in blocks:
            rp.append(b[0][1:] + off)
            off += int(b[0][-1])
        Cm = (np.concatenate(rp), np.concatenate([b[1] for b in blocks]), np.concatenate([b[2] for b in blocks]))
        x, iters, r = lsqr(Cm, np.concatenate(rhs), nminor)
        m = m + x * cw                                      # joint_inverse_problem.F90:570
        d = calc_data(m)
        hist.append(dict(iters=iters, r=r))
    return m, d, hist


# ---------------------------------------------------------------------------------------------------------------
# cross-gradient constraint (structural coupling of two models), loop-by-loop restatement
def _grad(f, dims, h, i, j, k, kind):
    """get_grad (src/inversion/gradient.F90:68-86) with grad_get_par's zero outside the grid (:196-225); i, j, k 1-based."""
    nx, ny, nz = dims

    def par(a, b, c):
        if a == 0 or b == 0 or c == 0 or a == nx + 1 or b == ny + 1 or c == nz + 1:
            return 0.0
        return f[((c - 1) * ny + (b - 1)) * nx + (a - 1)]
    dx, dy, dz = h[0][i - 1], h[1][j - 1], h[2][k - 1]
    if kind == "BWD":
        return ((par(i, j, k) - par(i - 1, j, k)) / dx, (par(i, j, k) - par(i, j - 1, k)) / dy, (par(i, j, k) - par(i, j, k - 1)) / dz)
    if kind == "FWD":
        return ((par(i + 1, j, k) - par(i, j, k)) / dx, (par(i, j + 1, k) - par(i, j, k)) / dy, (par(i, j, k + 1) - par(i, j, k)) / dz)
    return ((par(i + 1, j, k) - par(i - 1, j, k)) / 2.0 / dx, (par(i, j + 1, k) - par(i, j - 1, k)) / 2.0 / dy,
            (par(i, j, k + 1) - par(i, j, k - 1)) / 2.0 / dz)


def cross_gradient_rows(m1, m2, dims, grid, cw1, cw2, weight, der_type=1):
    """cross_gradient_calculate (src/inversion/cross_gradient.F90:220-391) with calculate_tau (:457-577) and
    calculate_tau_backward (:675-743): 3 rows per cell (x, y, z component of grad m1 x grad m2) over the columns of both models
    (model 2 at + N), values dm * column_weight * weight cast to fp32, right-hand side -tau * weight.
    Returns (rowptr, cols 1-based ascending, vals), rhs, cost[3]."""
    nx, ny, nz = dims
    N = nx * ny * nz
    X1, X2, Y1, Y2, Z1, Z2 = grid
    sh = (nz, ny, nx)
    h = (np.abs(X2 - X1).reshape(sh)[0, 0, :], np.abs(Y2 - Y1).reshape(sh)[0, :, 0], np.abs(Z2 - Z1).reshape(sh)[:, 0, 0])

    def ind(a, b, c):
        return ((c - 1) * ny + (b - 1)) * nx + a            # 1-based cell index
    rp, cols, vals, rhs = [0], [], [], []
    cost = np.zeros(3)
    for k in range(1, nz + 1):
        for j in range(1, ny + 1):
            for i in range(1, nx + 1):
                left = i == 1 or j == 1 or k == 1
                right = i == nx or j == ny or k == nz
                rows = None
                if left and right:
                    tau = (0.0, 0.0, 0.0)
                elif right or (der_type == 2 and left):
                    backward = right
                    kind = "BWD" if backward else "FWD"
                    g1, g2 = _grad(m1, dims, h, i, j, k, kind), _grad(m2, dims, h, i, j, k, kind)
                    sx, sy, sz = h[0][i - 1], h[1][j - 1], h[2][k - 1]
                    if backward:                                  # :700-735
                        rows = [
                            [(ind(i, j - 1, k), -g2[2] / sy, g1[2] / sy), (ind(i, j, k - 1), g2[1] / sz, -g1[1] / sz),
                             (ind(i, j, k), g2[2] / sy - g2[1] / sz, g1[1] / sz - g1[2] / sy)],
                            [(ind(i - 1, j, k), g2[2] / sx, -g1[2] / sx), (ind(i, j, k - 1), -g2[0] / sz, g1[0] / sz),
                             (ind(i, j, k), g2[0] / sz - g2[2] / sx, g1[2] / sx - g1[0] / sz)],
                            [(ind(i - 1, j, k), -g2[1] / sx, g1[1] / sx), (ind(i, j - 1, k), g2[0] / sy, -g1[0] / sy),
                             (ind(i, j, k), g2[1] / sx - g2[0] / sy, g1[0] / sy - g1[1] / sx)]]
                    else:                                         # forward scheme on the left boundary of a central run
                        rows = _tau_rows(g1, g2, (sx, sy, sz), i, j, k, ind, 1)
                    tau = (g1[1] * g2[2] - g1[2] * g2[1], g1[2] * g2[0] - g1[0] * g2[2], g1[0] * g2[1] - g1[1] * g2[0])
                else:
                    kind = "FWD" if der_type == 1 else "CNT"
                    g1, g2 = _grad(m1, dims, h, i, j, k, kind), _grad(m2, dims, h, i, j, k, kind)
                    step = (h[0][i - 1], h[1][j - 1], h[2][k - 1])
                    if der_type != 1:
                        step = tuple(2.0 * v for v in step)
                    rows = _tau_rows(g1, g2, step, i, j, k, ind, der_type)
                    tau = (g1[1] * g2[2] - g1[2] * g2[1], g1[2] * g2[0] - g1[0] * g2[2], g1[0] * g2[1] - g1[1] * g2[0])
                cost += np.array(tau) ** 2
                for comp in range(3):
                    entries = {}
                    if rows is not None:
                        for (cell, d1, d2) in rows[comp]:
                            v1 = np.float32(d1 * cw1[cell - 1] * weight)
                            v2 = np.float32(d2 * cw2[cell - 1] * weight)
                            if v1 != 0:
                                entries[cell] = v1
                            if v2 != 0:
                                entries[cell + N] = v2
                    for c in sorted(entries):
                        cols.append(c)
                        vals.append(entries[c])
                    rp.append(rp[-1] + len(entries))
                    rhs.append(-tau[comp] * weight)
    return (np.array(rp, np.int64), np.array(cols, np.int32), np.array(vals, np.float32)), np.array(rhs), cost


def _tau_rows(g1, g2, step, i, j, k, ind, der_type):
    """calculate_tau's derivative table (cross_gradient.F90:485-559): per component a list of (cell, d/dm1, d/dm2)."""
    sx, sy, sz = step
    x = [(ind(i, j + 1, k), g2[2] / sy, -g1[2] / sy), (ind(i, j, k + 1), -g2[1] / sz, g1[1] / sz)]
    y = [(ind(i + 1, j, k), -g2[2] / sx, g1[2] / sx), (ind(i, j, k + 1), g2[0] / sz, -g1[0] / sz)]
    z = [(ind(i + 1, j, k), g2[1] / sx, -g1[1] / sx), (ind(i, j + 1, k), -g2[0] / sy, g1[0] / sy)]
    if der_type == 1:
        x.append((ind(i, j, k), -(g2[2] / sy - g2[1] / sz), -(g1[1] / sz - g1[2] / sy)))
        y.append((ind(i, j, k), -(g2[0] / sz - g2[2] / sx), -(g1[2] / sx - g1[0] / sz)))
        z.append((ind(i, j, k), -(g2[1] / sx - g2[0] / sy), -(g1[0] / sy - g1[1] / sx)))
    else:
        x += [(ind(i, j - 1, k), -x[0][1], -x[0][2]), (ind(i, j, k - 1), -x[1][1], -x[1][2])]
        y += [(ind(i - 1, j, k), -y[0][1], -y[0][2]), (ind(i, j, k - 1), -y[1][1], -y[1][2])]
        z += [(ind(i - 1, j, k), -z[0][1], -z[0][2]), (ind(i, j - 1, k), -z[1][1], -z[1][2])]
    return [x, y, z]


def _gaussian(mu, sigma, wloc, val):
    """clustering_calculate_Gaussian (src/inversion/clustering.F90:539-584): 2-D Gaussian (:519-533) when both problems carry a
    clustering weight, else the 1-D one of the active problem (:505-511); clamped to exp(-100) for tiny arguments."""
    x, y = val
    mu1, mu2 = mu
    s11, s22, s12 = sigma
    if wloc[0] != 0.0 and wloc[1] != 0.0:
        s12_4 = s12 * s12 * s12 * s12          # sigma12**4 as the reference's compiler evaluates it: ((x x) x) x (measured, round 6)
        arg = (-((-mu2 + y) * (mu2 * s11**2 - mu1 * s12**2 + s12**2 * x - s11**2 * y)) / (s12_4 - s11**2 * s22**2)
               - ((-mu1 + x) * (mu2 * s12**2 - mu1 * s22**2 + s22**2 * x - s12**2 * y)) / (-s12_4 + s11**2 * s22**2)) / 2.0
        norm = 2.0 * math.pi * math.sqrt(-s12_4 + s11**2 * s22**2)
    elif wloc[1] == 0.0:
        arg = -(x - mu1)**2 / s11**2 / 2.0
        norm = math.sqrt(2.0 * math.pi * s11**2)
    else:
        arg = -(y - mu2)**2 / s22**2 / 2.0
        norm = math.sqrt(2.0 * math.pi * s22**2)
    if arg < -100.0:
        return math.exp(-100.0)
    return math.exp(arg) / norm


def _mixture(mix, wloc, val, cluster_weight):
    """clustering_calculate_Gaussian_mixture (:591-642): value and the two partial derivatives."""
    gauss, deriv = 0.0, [0.0, 0.0]
    x, y = val
    for i in range(mix.shape[0]):
        mu1, s11, mu2, s22, s12 = mix[i, 1], mix[i, 2], mix[i, 3], mix[i, 4], mix[i, 5]
        gl = cluster_weight[i] * _gaussian((mu1, mu2), (s11, s22, s12), wloc, val)
        gauss = gauss + gl
        s12_4 = s12 * s12 * s12 * s12          # sigma12**4 as the reference's compiler evaluates it: ((x x) x) x (measured, round 6)
        c1 = (s22**2 * (-mu1 + x) + s12**2 * (mu2 - y)) / (s12_4 - s11**2 * s22**2)
        c2 = (s12**2 * (mu1 - x) + s11**2 * (-mu2 + y)) / (s12_4 - s11**2 * s22**2)
        deriv[0] = deriv[0] + c1 * gl
        deriv[1] = deriv[1] + c2 * gl
    return gauss, deriv


def clustering_setup(mixtures, N, cell_weights=None):
    """clustering_read_mixtures (:159-283): per-cell cluster weights (global ones normalised to sum 1, local ones as read)."""
    mixtures = np.asarray(mixtures, np.float64)
    if cell_weights is None:
        w = mixtures[:, 0] / np.sum(mixtures[:, 0])
        return np.tile(w[None, :], (N, 1))
    return np.asarray(cell_weights, np.float64)


def clustering_rows(m1, m2, cw1, cw2, weight_glob, mixtures, cell_weight, opt_type=2):
    """clustering_add for problem 1 then 2 (joint_inverse_problem.F90:613-631, clustering.F90:393-499): 2 N rows, row p of block i
    holds weight_i * column_weight_i[p] * dP/dm_i (or -dP/dm_i / P for the logarithmic objective) at column p (+ N for i = 2);
    right-hand side -weight_i * (P - P_max) resp. -weight_i * (log P_max - log P).  Returns CSR, rhs, cost[2]."""
    N = m1.size
    mix = np.asarray(mixtures, np.float64)
    wloc = [0.0 if w == 0.0 else 1.0 for w in weight_glob]
    cw = (cw1, cw2)
    rp, cols, vals, rhs = [0], [], [], []
    cost = np.zeros(2)
    pmax = np.zeros(N)
    for p in range(N):                                      # calculate_Gaussian_mixture_max (:647-674): the largest value at a cluster centre
        for i in range(mix.shape[0]):
            gval, _ = _mixture(mix, wloc, (mix[i, 1], mix[i, 3]), cell_weight[p])
            pmax[p] = max(pmax[p], gval)
    for i in range(2):
        for p in range(N):
            gauss, deriv = _mixture(mix, wloc, (m1[p], m2[p]), cell_weight[p])
            if opt_type == 2:
                deriv = [-d / gauss for d in deriv] if gauss != 0.0 else [0.0, 0.0]
            v = np.float32(weight_glob[i] * cw[i][p] * deriv[i] * wloc[i])
            if v != 0:
                cols.append(p + 1 + i * N)
                vals.append(v)
            rp.append(len(cols))
            if opt_type == 1:
                f = gauss - pmax[p]
            else:
                f = -math.log(gauss) + math.log(pmax[p]) if gauss > 0.0 else 0.0
            b = -weight_glob[i] * f * wloc[i]
            rhs.append(b)
            cost[i] += b * b
    return (np.array(rp, np.int64), np.array(cols, np.int32), np.array(vals, np.float32)), np.array(rhs), cost


def run_joint_inversion_xgrad(problems, dims, grid, ctype, nmajor, nminor, xgrad_weight, der_type=1, rmin=1e-13, lsqr=None,
                              calc_data=None, rows_fn=None, coupling=None):
    """Joint inversion with the cross-gradient constraint: WAVELET_DOMAIN = false, unknowns [x1; x2] spatial
    (joint_inverse_problem.F90:189-198, :393-573)."""
    N = int(np.prod(dims))
    P = 2
    rows_fn = rows_fn or cross_gradient_rows
    m = [np.zeros(N), np.zeros(N)]
    rp = [np.zeros(1, np.int64)]
    cs, vs = [], []
    off = 0
    for i, pr in enumerate(problems):
        r, c, v = pr["S"]
        rp.append(np.asarray(r[1:], np.int64) + off)
        off += int(r[-1])
        cs.append(np.asarray(c, np.int64) + i * N)
        vs.append(v)
    Sj = (np.concatenate(rp), np.concatenate(cs).astype(np.int32), np.concatenate(vs))
    lsqr = lsqr or (lambda Cm, b, niter: orc.lsqr(Sj, Cm, P * N, b, niter, rmin, spatial=(ctype, dims[0], dims[1], dims[2]) if ctype > 0 else None)[:3])
    calc_data = calc_data or (lambda i, model: orc.calc_data(model, problems[i]["cw"], dims, ctype, problems[i]["S"], problems[i]["pw"],
                                                             np.ones(problems[i]["d_obs"].size)))
    d = [calc_data(i, m[i]) for i in range(P)]
    hist = []
    for it in range(nmajor):
        rhs = [pr["pw"] * (pr["d_obs"] - d[i]) for i, pr in enumerate(problems)]
        blocks = []
        for i, pr in enumerate(problems):                  # damping, spatial (damping.F90:135-150)
            if pr["alpha"] != 0.0:
                blk = np.zeros(P * N, np.float32)
                blk[i * N:(i + 1) * N] = np.float32(pr["alpha"] * pr["pw"])
                blocks.append(orc.diag_csr(blk))
                r = np.zeros(P * N)
                r[i * N:(i + 1) * N] = -pr["alpha"] * pr["pw"] * (m[i] / pr["cw"])
                rhs.append(r)
        if coupling is not None:                           # another coupling constraint in place of the cross-gradient (clustering)
            G, grhs, cost = coupling(m[0], m[1])
        else:
            G, grhs, cost = rows_fn(m[0], m[1], dims, grid, problems[0]["cw"], problems[1]["cw"], xgrad_weight, der_type)
        blocks.append(G)
        rhs.append(grhs)
        rpc = [np.zeros(1, np.int64)]
        off = 0
        for b in blocks:
            rpc.append(b[0][1:] + off)
            off += int(b[0][-1])
        Cm = (np.concatenate(rpc), np.concatenate([b[1] for b in blocks]), np.concatenate([b[2] for b in blocks]))
        x, iters, r = lsqr(Cm, np.concatenate(rhs), nminor)
        for i, pr in enumerate(problems):
            m[i] = m[i] + x[i * N:(i + 1) * N] * pr["cw"]
            d[i] = calc_data(i, m[i])
        hist.append(dict(iters=iters, r=r, xgrad_cost=cost))
    return m, d, hist
