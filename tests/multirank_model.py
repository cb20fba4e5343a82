"""Numpy restatement of the column-partitioned LSQR scheme that lsqr.hip implements for nranks > 1 (test infrastructure).
Local products use the CPU oracle; the two reductions per iteration go through a caller-supplied all-reduce (gloo in
the CPU tests).  Constraint rows stay rank-local; only their squared norm joins reduction 1."""
import numpy as np

import oracle_lib as orc


def lsqr_column_partitioned(S_loc, ncols_loc, nrows, b_data, diag_blocks, rhs_blocks, niter, rmin, rank, allreduce):
    u = np.array(b_data, np.float64)
    uc = [np.array(r, np.float64) for r in rhs_blocks]
    dg = [np.asarray(d, np.float32).astype(np.float64) for d in diag_blocks]
    x = np.zeros(ncols_loc)

    def norm_u():
        ucsq = allreduce(np.array([sum(float(np.dot(c, c)) for c in uc)]))[0]
        return np.sqrt(float(np.dot(u, u)) + ucsq)

    beta = norm_u()
    if beta == 0.0:
        return x, 0, 0.0
    u /= beta
    uc = [c / beta for c in uc]
    b1 = beta
    v = orc.spmtv(*S_loc, u, ncols_loc)
    for d, c in zip(dg, uc):
        v += d * c
    alpha = np.sqrt(allreduce(np.array([float(np.dot(v, v))]))[0])
    v /= alpha
    rhobar, phibar = alpha, beta
    w = v.copy()
    it, r = 0, 1.0
    while it < niter and r > rmin:
        part = orc.spmv(*S_loc, v)
        if rank == 0:
            part = part - alpha * u
        uc = [-alpha * c + d * v for d, c in zip(dg, uc)]
        buf = allreduce(np.concatenate([part, [sum(float(np.dot(c, c)) for c in uc)]]))
        u = buf[:nrows]
        beta = np.sqrt(float(np.dot(u, u)) + buf[nrows])
        if beta != 0.0:
            u = u / beta
            uc = [c / beta for c in uc]
        v = -beta * v + orc.spmtv(*S_loc, u, ncols_loc)
        for d, c in zip(dg, uc):
            v += d * c
        alpha = np.sqrt(allreduce(np.array([float(np.dot(v, v))]))[0])
        if alpha != 0.0:
            v = v / alpha
        rho = np.sqrt(rhobar * rhobar + beta * beta)
        if rho == 0.0:
            break
        c_, s_ = rhobar / rho, beta / rho
        theta = s_ * alpha
        rhobar = -c_ * alpha
        phi = c_ * phibar
        phibar = s_ * phibar
        x = x + (phi / rho) * w
        w = v - (theta / rho) * w
        r = phibar / b1
        it += 1
        if abs(rhobar) < 1e-30:
            break
    return x, it, r


def column_slice(S, c0, c1):
    """Columns [c0, c1) (0-based) of a CSR with 1-based columns, re-based to local 1-based indices."""
    rp, cols, vals = S
    keep = (cols > c0) & (cols <= c1)
    cnt = np.add.reduceat(keep.astype(np.int64), rp[:-1]) if cols.size else np.zeros(rp.size - 1, np.int64)
    cnt = np.where(np.diff(rp) == 0, 0, cnt)
    return (np.concatenate([[0], np.cumsum(cnt)]).astype(np.int64), (cols[keep] - c0).astype(np.int32), vals[keep])
