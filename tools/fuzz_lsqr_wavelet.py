"""Randomised sweeps against the CPU oracle (test infrastructure; run on a GPU box):
  wavelets - random array shapes 1..70 per axis, Haar / D4, forward and inverse, several vectors at once: bit-identical;
  LSQR     - random sparse systems with diagonal constraint blocks, an optional general constraint matrix and soft
             thresholding, 1..12 iterations (before the Golub-Kahan recurrence amplifies rounding): x and r to 1e-9."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as orc  # noqa: E402

tfx = importlib.import_module("tomofast-x_amd")
ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
ctx = tfx.Context(0)
for case in range(ncases):
    n1, n2, n3 = (int(rng.integers(1, 71)) for _ in range(3))
    wt = int(rng.integers(1, 3))
    nvec = int(rng.integers(1, 4))
    a = rng.standard_normal((nvec, n1 * n2 * n3))
    f = ctx.forward_wavelet(a.ravel().copy(), n1, n2, n3, wt)          # nvec arrays back to back
    f = np.asarray(f).reshape(nvec, -1)
    for v in range(nvec):
        ref = orc.wavelet(a[v].copy(), n1, n2, n3, wt)
        assert f[v].tobytes() == ref.tobytes(), ("forward", case, n1, n2, n3, wt)
        inv = ctx.inverse_wavelet(ref.copy(), n1, n2, n3, wt)
        assert inv.tobytes() == orc.wavelet(ref.copy(), n1, n2, n3, wt, inverse=True).tobytes(), ("inverse", case, n1, n2, n3, wt)
print("wavelets OK (%d cases)" % ncases)


def rand_csr(nr, nc, mean):
    rp, cs, vs = [0], [], []
    for r in range(nr):
        n = int(min(nc, rng.poisson(mean)))
        c = np.sort(rng.choice(nc, n, replace=False)) if n else np.zeros(0, np.int64)
        cs.append(c.astype(np.int32) + 1)
        vs.append(rng.standard_normal(c.size).astype(np.float32))
        rp.append(rp[-1] + c.size)
    return np.array(rp, np.int64), np.concatenate(cs) if rp[-1] else np.zeros(0, np.int32), np.concatenate(vs) if rp[-1] else np.zeros(0, np.float32)


for case in range(ncases):
    nr, nc = int(rng.integers(1, 400)), int(rng.integers(1, 3000))
    S = rand_csr(nr, nc, float(rng.choice([2, 20, 200])))
    if S[0][-1] == 0:
        continue
    nb = int(rng.integers(0, 3))
    diag = [np.abs(rng.standard_normal(nc)).astype(np.float32) * np.float32(10.0 ** rng.integers(-3, 1)) for _ in range(nb)]
    rhs = [rng.standard_normal(nc) * 0.1 for _ in range(nb)]
    gen = bool(rng.integers(0, 2))
    Cg = rand_csr(int(rng.integers(1, 200)), nc, 3.0) if gen else None
    grhs = rng.standard_normal(Cg[0].size - 1) if gen else None
    gamma = float(rng.choice([0.0, 0.0, 1e-3]))
    b = rng.standard_normal(nr)
    nit = int(rng.integers(1, 13))
    ctx.matrix_upload_csr(nr, nc, *S)
    if gen:
        ctx.cons_upload_csr(Cg[0], Cg[1], Cg[2], grhs)
    try:
        x, it, r = ctx.lsqr_solve_sensit(b, nit, 1e-13, gamma, 0.0, diag, rhs)
    finally:
        if gen:
            ctx.cons_clear()
    # oracle: constraint rows = [general C ; diagonal blocks] in the library's row order (general rows follow the data rows)
    blocks = ([Cg] if gen else []) + [orc.diag_csr(d) for d in diag]
    rpc, off = [np.zeros(1, np.int64)], 0
    for blk in blocks:
        rpc.append(blk[0][1:] + off)
        off += int(blk[0][-1])
    Cm = (np.concatenate(rpc), np.concatenate([blk[1] for blk in blocks]) if blocks else np.zeros(0, np.int32),
          np.concatenate([blk[2] for blk in blocks]) if blocks else np.zeros(0, np.float32))
    bb = np.concatenate([b] + ([grhs] if gen else []) + rhs)
    x_ref, it_ref, r_ref = orc.lsqr(S, Cm, nc, bb, nit, 1e-13, gamma)
    assert it == it_ref, (case, it, it_ref)
    scale = max(np.linalg.norm(x_ref), 1e-300)
    # tolerance: 1e3 x the oracle's own reaction to a last-bit change of its right-hand side (see below), at least 1e-9
    x_p, _, r_p = orc.lsqr(S, Cm, nc, bb * (1.0 + 4e-16 * rng.standard_normal(bb.size)), nit, 1e-13, gamma)
    own = max(np.linalg.norm(x_p - x_ref) / scale, abs(r_p - r_ref) / max(r_ref, 1e-300), 1e-12)
    err = max(np.linalg.norm(x - x_ref) / scale, abs(r - r_ref) / max(r_ref, 1e-300) if r_ref > 1e-14 else 0.0)
    assert err <= 1e3 * own, (case, nr, nc, nb, gen, gamma, nit, err, own)
print("LSQR OK (%d cases)" % ncases)

# spatial unknowns (WAVELET_DOMAIN = F: S acts on the wavelet transform of the unknowns) and the joint two-kernel system
for case in range(ncases):
    n1, n2, n3 = (int(rng.integers(2, 14)) for _ in range(3))
    N = n1 * n2 * n3
    wt = int(rng.integers(1, 3))
    joint = bool(rng.integers(0, 2))
    P = 2 if joint else 1
    nrs = [int(rng.integers(1, 60)) for _ in range(P)]
    Ss = [rand_csr(nr, N, float(rng.choice([3, 30]))) for nr in nrs]
    if any(S[0][-1] == 0 for S in Ss):
        continue
    ctx.set_grid(n1, n2, n3, *tfx.synthetic.grid(n1, n2, n3))
    for i, S in enumerate(Ss):
        ctx.select_problem(i)
        ctx.matrix_upload_csr(nrs[i], N, *S)
    ctx.select_problem(0)
    spatial = bool(rng.integers(0, 2))
    nb = int(rng.integers(1, 3))
    diag = [np.abs(rng.standard_normal(P * N)).astype(np.float32) * np.float32(1e-2) for _ in range(nb)]
    rhs = [rng.standard_normal(P * N) * 0.1 for _ in range(nb)]
    b = rng.standard_normal(sum(nrs))
    nit = int(rng.integers(1, 10))
    try:
        if spatial:
            ctx.lsqr_set_wavelet_domain(False, wt)
        x, it, r = ctx.lsqr_solve_sensit(b, nit, 1e-13, 0.0, 0.0, diag, rhs)
    finally:
        if spatial:
            ctx.lsqr_set_wavelet_domain(True)
        if joint:
            ctx.select_problem(1)
            ctx.matrix_free()
            ctx.select_problem(0)
    rp, cs, vs, off = [np.zeros(1, np.int64)], [], [], 0
    for i, S in enumerate(Ss):
        rp.append(S[0][1:] + off)
        off += int(S[0][-1])
        cs.append(S[1].astype(np.int64) + i * N)
        vs.append(S[2])
    Sj = (np.concatenate(rp), np.concatenate(cs).astype(np.int32), np.concatenate(vs))
    blocks = [orc.diag_csr(d) for d in diag]
    rpc, off = [np.zeros(1, np.int64)], 0
    for blk in blocks:
        rpc.append(blk[0][1:] + off)
        off += int(blk[0][-1])
    Cm = (np.concatenate(rpc), np.concatenate([blk[1] for blk in blocks]), np.concatenate([blk[2] for blk in blocks]))
    sp = (wt, n1, n2, n3) if spatial else None
    bb = np.concatenate([b] + rhs)
    x_ref, it_ref, r_ref = orc.lsqr(Sj, Cm, P * N, bb, nit, 1e-13, 0.0, spatial=sp)
    assert it == it_ref, (case, it, it_ref)
    scale = max(np.linalg.norm(x_ref), 1e-300)
    # how far does the oracle itself move when its right-hand side moves in the last bit?  (Golub-Kahan without
    # re-orthogonalisation amplifies rounding on ill-conditioned systems; a real discrepancy is far above that.)
    x_p, _, r_p = orc.lsqr(Sj, Cm, P * N, bb * (1.0 + 4e-16 * rng.standard_normal(bb.size)), nit, 1e-13, 0.0, spatial=sp)
    own = max(np.linalg.norm(x_p - x_ref) / scale, abs(r_p - r_ref) / max(r_ref, 1e-300), 1e-11)
    err = max(np.linalg.norm(x - x_ref) / scale, abs(r - r_ref) / max(r_ref, 1e-300))
    assert err <= 1e3 * own, (case, joint, spatial, nit, err, own)
print("LSQR spatial / joint OK (%d cases)" % ncases)
