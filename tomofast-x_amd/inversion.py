"""Major inversion loop on top of the HIP path - the slice of solve_problem_joint_gravmag
(src/problem_joint_gravmag.F90:65-613) and joint_inversion_solve (src/inversion/joint_inverse_problem.F90:393-573)
that drives the hot path, for one gravity problem with WAVELET_DOMAIN = true (or compression off):

  residuals (problem_joint_gravmag.F90:666-675) -> right-hand side + damping / ADMM diagonal blocks
  (src/inversion/damping.F90:97-234, src/inversion/admm_method.F90:70-134) -> lsqr_solve_sensit ->
  inverse wavelet + column un-weighting (joint_inverse_problem.F90:559-571) -> model update ->
  model_calculate_data (src/inversion/model.F90:220-307).

All O(N) and O(nnz) work (wavelets, products, LSQR) runs on the device through Context; this file is control flow.
Multi-rank (col_range given): every rank keeps the full model (N doubles), the LSQR unknowns are its column slice of the
wavelet-domain vector exactly like the reference's nelements_at_cpu partition, and the slices of the model update are
gathered once per major iteration before the inverse transform, which every rank runs redundantly (the
reference gathers to rank 0, transforms there and scatters: wavelet_utils.F90:37-72)."""
import numpy as np


class AdmmState:
    """admm_method.F90:25-66: z, u persist across major iterations."""

    def __init__(self, n):
        self.z = np.zeros(n)
        self.u = np.zeros(n)

    def iterate_admm_arrays(self, x, bounds):
        """admm_method.F90:70-134.  bounds: [lo1 hi1 lo2 hi2 ...] for all cells (boundType 1) or an (N, 2 nlithos) array of
        per-cell intervals (boundType 2, model_IO.F90:311-372)."""
        ends = np.asarray(bounds, np.float64)
        if ends.ndim == 1:
            ends = ends[None, :]
        lo, hi = ends[:, 0::2], ends[:, 1::2]
        arg = x + self.u
        inside = ((lo <= arg[:, None]) & (arg[:, None] <= hi)).any(1)
        # closest boundary; np.argmin returns the first minimum = the strict '<' scan order xmin(1), xmax(1), xmin(2)...
        ends_b = np.broadcast_to(ends, (arg.size, ends.shape[1]))
        closest = np.take_along_axis(ends_b, np.argmin(np.abs(ends_b - arg[:, None]), axis=1)[:, None], 1)[:, 0]
        self.z = np.where(inside, arg, closest)
        self.u = self.u + x - self.z
        return self.z - self.u


def restrict_columns(G, c0, c1):
    """Rows of a constraint block over the columns (c0, c1] (1-based, as uploaded) of one rank, renumbered from 1: the
    column-partitioned form of matrix_cons - every rank holds all rows and its own columns; LSQR sums the partial products."""
    rowptr, cols, vals = G
    keep = (cols > c0) & (cols <= c1)
    csum = np.concatenate([[0], np.cumsum(keep)])
    return csum[rowptr].astype(np.int64), (cols[keep] - c0).astype(np.int32), vals[keep]


def gradient_damping_rows(m, dims, spacing, cw, pw, beta):
    """damping_gradient%add for the three directions (src/inversion/damping_gradient.F90:94-205): forward differences
    (gradient.F90:77-81) of the model along x, y, z; 3 N rows of two entries (none in the last layer of a direction) with
    values pw * beta * cw(column) * (+-1 / delta) cast to fp32, right-hand side -pw * beta * gradient.
    Returns (rowptr, cols 1-based ascending, vals), rhs - what Context.cons_upload_csr takes."""
    nx, ny, nz = dims
    N = nx * ny * nz
    idx = np.arange(N, dtype=np.int64).reshape(nz, ny, nx)
    f = np.asarray(m, np.float64)
    cnts, cs, vs, rh = [], [], [], []
    for axis, h in ((2, spacing[0]), (1, spacing[1]), (0, spacing[2])):       # direction 1 = x (fastest index), 2 = y, 3 = z
        n_ax = idx.shape[axis]
        me = np.take(idx, np.arange(n_ax - 1), axis=axis)
        nb = np.take(idx, np.arange(1, n_ax), axis=axis)
        shape = [1, 1, 1]
        shape[axis] = n_ax - 1
        delta = np.broadcast_to(np.asarray(h, np.float64)[:n_ax - 1].reshape(shape), me.shape)
        me, nb, delta = me.ravel(), nb.ravel(), delta.ravel()
        gval = (f[nb] - f[me]) / delta
        cnt = np.zeros(N, np.int64)
        cnt[me] = 2
        order = np.argsort(me, kind="stable")                                   # rows are cells in i-fastest order
        me, nb, delta, gval = me[order], nb[order], delta[order], gval[order]
        c = np.empty(2 * me.size, np.int32)
        v = np.empty(2 * me.size, np.float32)
        c[0::2] = me + 1
        c[1::2] = nb + 1
        v[0::2] = (-(1.0 / delta) * pw * beta * cw[me]).astype(np.float32)
        v[1::2] = ((1.0 / delta) * pw * beta * cw[nb]).astype(np.float32)
        r = np.zeros(N)
        r[me] = -pw * beta * gval
        cnts.append(cnt)
        cs.append(c)
        vs.append(v)
        rh.append(r)
    rowptr = np.concatenate([[0], np.cumsum(np.concatenate(cnts))]).astype(np.int64)
    return (rowptr, np.concatenate(cs), np.concatenate(vs)), np.concatenate(rh)


def cross_gradient_rows(m1, m2, dims, spacing, cw1, cw2, weight, der_type=1, keep_constant=(False, False)):
    """The cross-gradient constraint tau = grad m1 x grad m2 = 0 of a joint inversion (src/inversion/cross_gradient.F90:220-391;
    derivative tables :457-577 forward / central, :675-743 backward; boundary rules :255-285; gradients gradient.F90:68-86 with
    zeros outside the grid): 3 rows per cell over the columns of both models (model 2 at + N), values
    d tau / d m * column_weight * weight cast to fp32, right-hand side -tau * weight.
    keep_constant[i]: no derivative entries for model i (:294-295).
    Returns (rowptr, cols 1-based ascending, vals), rhs, cost[3] - what Context.cons_upload_csr takes."""
    nx, ny, nz = dims
    N = nx * ny * nz
    sh = (nz, ny, nx)
    hx = np.broadcast_to(np.asarray(spacing[0], np.float64)[None, None, :], sh)
    hy = np.broadcast_to(np.asarray(spacing[1], np.float64)[None, :, None], sh)
    hz = np.broadcast_to(np.asarray(spacing[2], np.float64)[:, None, None], sh)
    idx = np.arange(N, dtype=np.int64).reshape(sh)
    kk, jj, ii = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
    left = (ii == 0) | (jj == 0) | (kk == 0)
    right = (ii == nx - 1) | (jj == ny - 1) | (kk == nz - 1)

    def grads(f):
        P = np.pad(np.asarray(f, np.float64).reshape(sh), 1)
        c = P[1:-1, 1:-1, 1:-1]
        xp, xm = P[1:-1, 1:-1, 2:], P[1:-1, 1:-1, :-2]
        yp, ym = P[1:-1, 2:, 1:-1], P[1:-1, :-2, 1:-1]
        zp, zm = P[2:, 1:-1, 1:-1], P[:-2, 1:-1, 1:-1]
        return dict(F=((xp - c) / hx, (yp - c) / hy, (zp - c) / hz), B=((c - xm) / hx, (c - ym) / hy, (c - zm) / hz),
                    C=((xp - xm) / 2.0 / hx, (yp - ym) / 2.0 / hy, (zp - zm) / 2.0 / hz))
    G1, G2 = grads(m1), grads(m2)
    modes = []                                              # (mask, scheme, gradient kind, step multiplier)
    if der_type == 1:
        modes.append((~right, "fwd", "F", 1.0))
        modes.append((right & ~left, "bwd", "B", 1.0))
    else:
        modes.append((~right & ~left, "cnt", "C", 2.0))
        modes.append((left & ~right, "fwd", "F", 1.0))
        modes.append((right & ~left, "bwd", "B", 1.0))
    rows_l, cols_l, vals_l = [], [], []
    tau = np.zeros((3,) + sh)
    nbr = {"x+": (0, 0, 1), "x-": (0, 0, -1), "y+": (0, 1, 0), "y-": (0, -1, 0), "z+": (1, 0, 0), "z-": (-1, 0, 0), "0": (0, 0, 0)}
    for mask, scheme, kind, mult in modes:
        if not mask.any():
            continue
        g1, g2 = G1[kind], G2[kind]
        sx, sy, sz = mult * hx, mult * hy, mult * hz
        t = (g1[1] * g2[2] - g1[2] * g2[1], g1[2] * g2[0] - g1[0] * g2[2], g1[0] * g2[1] - g1[1] * g2[0])
        for comp in range(3):
            tau[comp][mask] = t[comp][mask]
        if scheme == "bwd":                                 # :700-735
            table = [
                [("y-", -g2[2] / sy, g1[2] / sy), ("z-", g2[1] / sz, -g1[1] / sz), ("0", g2[2] / sy - g2[1] / sz, g1[1] / sz - g1[2] / sy)],
                [("x-", g2[2] / sx, -g1[2] / sx), ("z-", -g2[0] / sz, g1[0] / sz), ("0", g2[0] / sz - g2[2] / sx, g1[2] / sx - g1[0] / sz)],
                [("x-", -g2[1] / sx, g1[1] / sx), ("y-", g2[0] / sy, -g1[0] / sy), ("0", g2[1] / sx - g2[0] / sy, g1[0] / sy - g1[1] / sx)]]
        else:                                               # :485-559
            tx = [("y+", g2[2] / sy, -g1[2] / sy), ("z+", -g2[1] / sz, g1[1] / sz)]
            ty = [("x+", -g2[2] / sx, g1[2] / sx), ("z+", g2[0] / sz, -g1[0] / sz)]
            tz = [("x+", g2[1] / sx, -g1[1] / sx), ("y+", -g2[0] / sy, g1[0] / sy)]
            if scheme == "fwd":
                tx.append(("0", -(g2[2] / sy - g2[1] / sz), -(g1[1] / sz - g1[2] / sy)))
                ty.append(("0", -(g2[0] / sz - g2[2] / sx), -(g1[2] / sx - g1[0] / sz)))
                tz.append(("0", -(g2[1] / sx - g2[0] / sy), -(g1[0] / sy - g1[1] / sx)))
            else:
                tx += [("y-", -tx[0][1], -tx[0][2]), ("z-", -tx[1][1], -tx[1][2])]
                ty += [("x-", -ty[0][1], -ty[0][2]), ("z-", -ty[1][1], -ty[1][2])]
                tz += [("x-", -tz[0][1], -tz[0][2]), ("y-", -tz[1][1], -tz[1][2])]
            table = [tx, ty, tz]
        p = idx[mask]
        for comp in range(3):
            for where, d1, d2 in table[comp]:
                dk, dj, di = nbr[where]
                cell = p + (dk * ny + dj) * nx + di
                for off, dm, cw, fixed in ((0, d1, cw1, keep_constant[0]), (N, d2, cw2, keep_constant[1])):
                    if fixed:
                        continue
                    v = (dm[mask] * cw[cell] * weight).astype(np.float32)
                    keep = v != 0
                    rows_l.append((3 * p + comp)[keep])
                    cols_l.append((cell + off + 1)[keep])
                    vals_l.append(v[keep])
    rows = np.concatenate(rows_l) if rows_l else np.zeros(0, np.int64)
    cols = np.concatenate(cols_l) if cols_l else np.zeros(0, np.int64)
    vals = np.concatenate(vals_l) if vals_l else np.zeros(0, np.float32)
    order = np.lexsort((cols, rows))
    rows, cols, vals = rows[order], cols[order], vals[order]
    rowptr = np.concatenate([[0], np.cumsum(np.bincount(rows, minlength=3 * N))]).astype(np.int64)
    tflat = np.stack([tau[0].ravel(), tau[1].ravel(), tau[2].ravel()], 1)       # (cell, component)
    rhs = (-tflat * weight).ravel()
    cost = (tflat ** 2).sum(0)
    return (rowptr, cols.astype(np.int32), vals), rhs, cost


def clustering_cell_weights(mixtures, N, cell_weights=None):
    """Per-cell cluster weights (src/inversion/clustering.F90:159-283): the global ones of the mixture file normalised to sum 1
    (constraintsType 1), or the per-cell table as read (constraintsType 2)."""
    mixtures = np.asarray(mixtures, np.float64)
    if cell_weights is None:
        return np.tile((mixtures[:, 0] / np.sum(mixtures[:, 0]))[None, :], (N, 1))
    return np.asarray(cell_weights, np.float64)


def _gaussian_mixture(mix, wloc, x, y, cell_weight):
    """Gaussian mixture P(x, y) and its two partial derivatives, vectorised over the cells (clustering.F90:591-642 with the
    Gaussians of :505-584: 2-D when both problems carry a clustering weight, else 1-D in the weighted one; exp(-100) floor)."""
    gauss = np.zeros_like(x)
    d1, d2 = np.zeros_like(x), np.zeros_like(x)
    for i in range(mix.shape[0]):
        mu1, s11, mu2, s22, s12 = mix[i, 1], mix[i, 2], mix[i, 3], mix[i, 4], mix[i, 5]
        s12_4 = (s12 * s12) * (s12 * s12)
        if wloc[0] != 0.0 and wloc[1] != 0.0:
            arg = (-((-mu2 + y) * (mu2 * s11**2 - mu1 * s12**2 + s12**2 * x - s11**2 * y)) / (s12_4 - s11**2 * s22**2)
                   - ((-mu1 + x) * (mu2 * s12**2 - mu1 * s22**2 + s22**2 * x - s12**2 * y)) / (-s12_4 + s11**2 * s22**2)) / 2.0
            norm = 2.0 * np.pi * np.sqrt(-s12_4 + s11**2 * s22**2)
        elif wloc[1] == 0.0:
            arg = -(x - mu1)**2 / s11**2 / 2.0
            norm = np.sqrt(2.0 * np.pi * s11**2)
        else:
            arg = -(y - mu2)**2 / s22**2 / 2.0
            norm = np.sqrt(2.0 * np.pi * s22**2)
        if norm == 0.0:
            raise ValueError("zero norm of a clustering Gaussian (clustering.F90:571-574)")
        val = np.where(arg < -100.0, np.exp(-100.0), np.exp(np.maximum(arg, -100.0)) / norm)
        gl = cell_weight[:, i] * val
        gauss = gauss + gl
        d1 = d1 + (s22**2 * (-mu1 + x) + s12**2 * (mu2 - y)) / (s12_4 - s11**2 * s22**2) * gl
        d2 = d2 + (s12**2 * (mu1 - x) + s11**2 * (-mu2 + y)) / (s12_4 - s11**2 * s22**2) * gl
    return gauss, d1, d2


def clustering_rows(m1, m2, cw1, cw2, weight_glob, mixtures, cell_weight, opt_type=2):
    """The clustering (petrophysical) constraint of a joint inversion (clustering.F90:393-499, called for problem 1 then 2,
    joint_inverse_problem.F90:613-631): 2 N rows, row p of block i holds weight_i * column_weight_i[p] * dP/dm_i - or
    -dP/dm_i / P for the logarithmic objective (opt_type 2) - at column p (+ N for i = 2), cast to fp32; right-hand side
    -weight_i * (P - P_max) resp. -weight_i * (log P_max - log P), P_max = the largest mixture value at a cluster centre
    (:647-674).  mixtures[c] = (cluster weight, mu1, sigma1, mu2, sigma2, sigma12).
    Returns (rowptr, cols 1-based, vals), rhs, cost[2] - what Context.cons_upload_csr takes."""
    N = m1.size
    mix = np.asarray(mixtures, np.float64)
    wloc = [0.0 if w == 0.0 else 1.0 for w in weight_glob]
    pmax = np.zeros(N)
    for i in range(mix.shape[0]):
        gc, _, _ = _gaussian_mixture(mix, wloc, np.full(N, mix[i, 1]), np.full(N, mix[i, 3]), cell_weight)
        pmax = np.maximum(pmax, gc)
    gauss, d1, d2 = _gaussian_mixture(mix, wloc, np.asarray(m1, np.float64), np.asarray(m2, np.float64), cell_weight)
    if opt_type == 2:
        nzm = gauss != 0.0
        safe = np.where(nzm, gauss, 1.0)
        d1, d2 = np.where(nzm, -d1 / safe, 0.0), np.where(nzm, -d2 / safe, 0.0)
        pos = gauss > 0.0
        func = np.where(pos, -np.log(np.where(pos, gauss, 1.0)) + np.log(pmax), 0.0)
    elif opt_type == 1:
        func = gauss - pmax
    else:
        raise ValueError("wrong optimization type of the clustering constraint: %r" % (opt_type,))
    vals = np.concatenate([(weight_glob[0] * cw1 * d1 * wloc[0]).astype(np.float32), (weight_glob[1] * cw2 * d2 * wloc[1]).astype(np.float32)])
    keep = vals != 0
    rowptr = np.concatenate([[0], np.cumsum(keep)]).astype(np.int64)
    cols = (np.nonzero(keep)[0] + 1).astype(np.int32)
    rhs = np.concatenate([-weight_glob[0] * func * wloc[0], -weight_glob[1] * func * wloc[1]])
    cost = np.array([np.sum(rhs[:N] ** 2), np.sum(rhs[N:] ** 2)])
    return (rowptr, cols, vals[keep]), rhs, cost


def solve_problem_gravity(ctx, column_weight, compression_type, data_obs, nmajor, nminor, alpha=0.0, rmin=1e-13,
                          problem_weight=1.0, data_weight=None, model_start=None, model_prior=None, admm=None,
                          gamma=0.0, target_misfit=0.0, log=None, nmodel_components=1, col_range=None, beta=0.0,
                          norm_power=2.0, damping_weight=None):
    """ctx: Context holding the sensitivity matrix S (already scaled by problem_weight * data_weight) over ALL columns.
    admm: dict(bounds=[...], rho=...) or None.  Returns (model, data_calc, history).
    One problem of either kind (the name is historical).  nmodel_components = 3 (magnetisation vector): model vectors are
    component-major [k*N + cell] (model%val(:, k) flattened), the wavelet transform runs per component
    (wavelet_utils.F90:37-72) and the damping block repeats per component (joint_inverse_problem.F90:456-463);
    data vectors are [idata*ndata_components + d]."""
    nx, ny, nz = ctx.dims
    ncm = int(nmodel_components)
    N1 = nx * ny * nz
    N = N1 * ncm
    # gradient damping (beta != 0) acts in space: WAVELET_DOMAIN = false (joint_inverse_problem.F90:189-198) - the unknowns
    # are the spatial depth-weighted update, S is applied through the per-iteration device transform, the damping block is not
    # transformed (damping.F90:135-150) and nothing is transformed back after the solve (joint_inverse_problem.F90:559-567)
    # ... and so does an Lp norm of the model damping (norm_power != 2: multiplier |m - m_prior|^(p/2 - 1), damping.F90:171-175)
    # ... and so do local ADMM bounds (admm["bounds"] per cell, optional admm["weight"] per cell = bound_weight)
    admm_local = admm is not None and (np.ndim(admm["bounds"]) == 2 or admm.get("weight") is not None)
    # ... and local model-damping weights (damping_weight per cell: model%damping_weight, model_IO.F90:425-476)
    spatial = beta != 0.0 or norm_power != 2.0 or admm_local or damping_weight is not None

    def unweight(v):                                 # v / column_weight with the reference's zero guard (damping.F90:129-135)
        return np.where(cw != 0.0, v / np.where(cw != 0.0, cw, 1.0), 0.0)
    if spatial and ncm != 1:
        raise NotImplementedError("gradient / Lp damping: one model component in this host")
    if col_range is None:
        loc = lambda v: v
        gather = lambda v: v
    else:
        from .distributed import allreduce_numpy
        c0, c1 = col_range

        def loc(v):                                  # this rank's cells of every model component
            return np.concatenate([v[k * N1 + c0:k * N1 + c1] for k in range(ncm)])

        def gather(v_loc):                           # all slices -> the full vector (disjoint supports: a sum is a gather)
            full = np.zeros(N)
            nl = c1 - c0
            for k in range(ncm):
                full[k * N1 + c0:k * N1 + c1] = v_loc[k * nl:(k + 1) * nl]
            return allreduce_numpy(full)
    cw = np.tile(np.asarray(column_weight, np.float64), ncm)
    if admm is not None and ncm != 1:
        raise NotImplementedError("ADMM bounds act on Mz only for vector models (joint_inverse_problem.F90:497-506)")
    pw = float(problem_weight)
    dw = np.ones(data_obs.size) if data_weight is None else np.asarray(data_weight, np.float64)
    m = np.zeros(N) if model_start is None else np.array(model_start, np.float64)
    mp = np.zeros(N) if model_prior is None else np.asarray(model_prior, np.float64)

    def to_wavelet(v):
        return ctx.forward_wavelet(v, nx, ny, nz, compression_type) if compression_type > 0 else v

    def to_unknowns(v):                              # the domain of the LSQR unknowns
        return v if spatial else to_wavelet(v)

    def calculate_data(model):                       # model.F90:242-305
        scaled = np.where(cw != 0.0, model / np.where(cw != 0.0, cw, 1.0), 0.0)
        return ctx.calc_data(loc(to_wavelet(scaled)), pw, dw)

    d_calc = calculate_data(m)
    st = AdmmState(N) if admm is not None else None
    hist = []
    for it in range(1, nmajor + 1):
        res = dw * (data_obs - d_calc)               # problem_joint_gravmag.F90:666-675
        b_data = pw * res                            # joint_inverse_problem.F90:379-387
        diag, rhs = [], []
        if alpha != 0.0:                             # damping.F90:97-234 (L2, no local weights)
            md = loc(to_unknowns(unweight(m - mp)))
            lp = np.ones(md.size)
            if norm_power != 2.0:                    # damping.F90:250-262
                nzm = md != 0.0
                lp[nzm] = np.abs(md[nzm]) ** (norm_power / 2.0 - 1.0)
            val, r = alpha * pw * lp, -alpha * pw * md * lp          # :160-166, :218-223: alpha * pw, then the Lp multiplier ...
            if damping_weight is not None:                           # ... then the local weight (:168-171, :225-228)
                lw = loc(np.asarray(damping_weight, np.float64))
                val, r = val * lw, r * lw
            diag.append(val.astype(np.float32))                      # one cast, like matrix%add(value) (sparse_matrix.f90:226)
            rhs.append(r)
        if admm is not None:                         # joint_inverse_problem.F90:497-527
            x0 = st.iterate_admm_arrays(m, admm["bounds"])
            md = loc(to_unknowns(unweight(m - x0)))
            lw = np.ones(md.size) if admm.get("weight") is None else loc(np.asarray(admm["weight"], np.float64))
            diag.append((admm["rho"] * pw * lw).astype(np.float32))     # local weight = local rho (damping.F90:177-180, :264-267)
            rhs.append(-admm["rho"] * pw * md * lw)
        if spatial:
            if beta != 0.0:
                G, grhs = gradient_damping_rows(m, (nx, ny, nz), ctx.spacing, cw, pw, beta)
                if col_range is not None:            # rows replicated, columns of this rank (joint_inverse_problem.F90:332: ncolumns local)
                    G = restrict_columns(G, col_range[0], col_range[1])
                ctx.cons_upload_csr(G[0], G[1], G[2], grhs)
            ctx.lsqr_set_wavelet_domain(False, compression_type)
            if col_range is not None:                # every product with S gathers the slices, transforms, keeps its own
                ctx.lsqr_set_partition(col_range[0], 1)
        try:
            x, iters, r = ctx.lsqr_solve_sensit(b_data, nminor, rmin, gamma, target_misfit, diag, rhs)
        finally:
            if spatial:
                ctx.lsqr_set_wavelet_domain(True)
                ctx.cons_clear()
        x = gather(x)
        dm = ctx.inverse_wavelet(x, nx, ny, nz, compression_type) if (compression_type > 0 and not spatial) else x
        m = m + dm * cw                              # joint_inverse_problem.F90:570, model update :500
        d_calc = calculate_data(m)
        cost = float(np.linalg.norm(d_calc - data_obs) / np.linalg.norm(data_obs))   # data_gravmag.f90:123-129
        hist.append(dict(it=it, iters=iters, r=r, cost=cost))
        if log:
            log("it %d: lsqr iters %d r %.6e data cost %.6e" % (it, iters, r, cost))
    return m, d_calc, hist


def restrict_columns_blocks(G, c0, c1, N, nblocks):
    """restrict_columns for a constraint block over `nblocks` models of N cells each (joint system): the rank keeps the cells
    (c0, c1] of every model; its unknown vector is [model 1 slice; model 2 slice; ...]."""
    rowptr, cols, vals = G
    cell = (cols - 1) % N + 1
    blk = (cols - 1) // N
    keep = (cell > c0) & (cell <= c1)
    csum = np.concatenate([[0], np.cumsum(keep)])
    return csum[rowptr].astype(np.int64), (blk[keep] * (c1 - c0) + cell[keep] - c0).astype(np.int32), vals[keep]


def solve_problem_joint(ctx, problems, compression_type, nmajor, nminor, rmin=1e-13, gamma=0.0, target_misfit=0.0, log=None,
                        cross_gradient=None, clustering=None, col_range=None):
    """Joint inversion of two problems on one grid (gravity + magnetic) without structural coupling: both sensitivity
    kernels in ONE LSQR system, S = blockdiag(slot 0, slot 1) (src/inversion/joint_inverse_problem.F90:393-573; block layout
    :712-739, right-hand side :379-387, one damping block per problem :448-463).  Coupling constraints built on the host
    (cross-gradient, clustering) enter through ctx.cons_upload_csr over both column blocks.

    problems: two dicts(column_weight, data_obs, problem_weight, alpha[, model_start, model_prior]); slot i of ctx holds
    problem i's kernel, built with its problem_weight.  cross_gradient = dict(weight, der_type 1 | 2): the structural
    coupling constraint; clustering = dict(weight (2), mixtures, opt_type 1 | 2[, cell_weights]): the petrophysical one.
    Either switches the solver to spatial unknowns (WAVELET_DOMAIN = false, :189-198); the rows go cross-gradient first (:529-541).
    col_range = (c0, c1): this rank holds the cells (c0, c1] of both kernels (and the all-reduce hook is set); every rank keeps
    the full models, the LSQR unknowns are [model 1 slice; model 2 slice].
    Returns (models, data_calc, history)."""
    nx, ny, nz = ctx.dims
    N = nx * ny * nz
    P = len(problems)
    if P != 2:
        raise ValueError("two problems (gravity, magnetic)")
    cw = [np.asarray(p["column_weight"], np.float64) for p in problems]
    pw = [float(p["problem_weight"]) for p in problems]
    m = [np.zeros(N) if p.get("model_start") is None else np.array(p["model_start"], np.float64) for p in problems]
    mp = [np.zeros(N) if p.get("model_prior") is None else np.asarray(p["model_prior"], np.float64) for p in problems]

    def to_wavelet(v):
        return ctx.forward_wavelet(v, nx, ny, nz, compression_type) if compression_type > 0 else v

    spatial = cross_gradient is not None or clustering is not None
    clust_w = clustering_cell_weights(clustering["mixtures"], N, clustering.get("cell_weights")) if clustering is not None else None
    c0, c1 = (0, N) if col_range is None else col_range
    nloc = c1 - c0

    def gather(v_loc):                                # slices of all ranks -> full vector (disjoint supports: a sum is a gather)
        if col_range is None:
            return v_loc
        from .distributed import allreduce_numpy
        full = np.zeros(N)
        full[c0:c1] = v_loc
        return allreduce_numpy(full)

    def calculate_data(i):                            # model.F90:242-305 on problem i's rows / columns
        ctx.select_problem(i)
        try:
            return ctx.calc_data(to_wavelet(np.where(cw[i] != 0.0, m[i] / cw[i], 0.0))[c0:c1], pw[i], None)
        finally:
            ctx.select_problem(0)

    d = [calculate_data(i) for i in range(P)]
    hist = []
    for it in range(1, nmajor + 1):
        b_data = np.concatenate([pw[i] * (np.asarray(problems[i]["data_obs"], np.float64) - d[i]) for i in range(P)])
        diag, rhs = [], []
        for i, p in enumerate(problems):
            if p.get("alpha", 0.0) != 0.0:            # damping block of problem i: its column block only (:448-463)
                md = (m[i] - mp[i]) / cw[i]
                if not spatial:
                    md = to_wavelet(md)
                blk = np.zeros(P * nloc, np.float32)
                blk[i * nloc:(i + 1) * nloc] = np.float32(p["alpha"] * pw[i])
                r = np.zeros(P * nloc)
                r[i * nloc:(i + 1) * nloc] = -p["alpha"] * pw[i] * md[c0:c1]
                diag.append(blk)
                rhs.append(r)
        xcost = ccost = None
        if spatial:
            blocks = []
            if cross_gradient is not None:
                G, grhs, xcost = cross_gradient_rows(m[0], m[1], (nx, ny, nz), ctx.spacing, cw[0], cw[1], float(cross_gradient["weight"]),
                                                     int(cross_gradient.get("der_type", 1)), cross_gradient.get("keep_constant", (False, False)))
                blocks.append((G, grhs))
            if clustering is not None:
                G, grhs, ccost = clustering_rows(m[0], m[1], cw[0], cw[1], clustering["weight"], clustering["mixtures"], clust_w,
                                                 int(clustering.get("opt_type", 2)))
                blocks.append((G, grhs))
            if col_range is not None:                 # rows replicated, every rank fills its own columns
                blocks = [(restrict_columns_blocks(b[0], c0, c1, N, P), b[1]) for b in blocks]
            rp = np.concatenate([[0]] + [b[0][0][1:] + off for b, off in zip(blocks, np.cumsum([0] + [int(b[0][0][-1]) for b in blocks])[:-1])])
            ctx.cons_upload_csr(rp.astype(np.int64), np.concatenate([b[0][1] for b in blocks]), np.concatenate([b[0][2] for b in blocks]),
                                np.concatenate([b[1] for b in blocks]))
            ctx.lsqr_set_wavelet_domain(False, compression_type)
            if col_range is not None:
                ctx.lsqr_set_partition(c0, P)
        try:
            x, iters, r = ctx.lsqr_solve_sensit(b_data, nminor, rmin, gamma, target_misfit, diag, rhs)
        finally:
            if spatial:
                ctx.lsqr_set_wavelet_domain(True)
                ctx.cons_clear()
        for i in range(P):
            xi = gather(x[i * nloc:(i + 1) * nloc])
            dm = ctx.inverse_wavelet(xi, nx, ny, nz, compression_type) if (compression_type > 0 and not spatial) else xi
            m[i] = m[i] + dm * cw[i]                  # joint_inverse_problem.F90:559-571
            d[i] = calculate_data(i)
        costs = [float(np.linalg.norm(d[i] - problems[i]["data_obs"]) / np.linalg.norm(problems[i]["data_obs"])) for i in range(P)]
        hist.append(dict(it=it, iters=iters, r=r, costs=costs, xgrad_cost=xcost, clustering_cost=ccost))
        if log:
            log("it %d: lsqr iters %d r %.6e data costs %s" % (it, iters, r, costs))
    return m, d, hist
