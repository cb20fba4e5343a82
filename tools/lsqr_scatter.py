import sys, os, importlib, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import oracle_lib as orc
tfx = importlib.import_module("tomofast-x_amd")
g = np.load("/root/repo/tests/golden/lsqr.npz")
ctx = tfx.Context(0)
for case in ("damp", "noC"):
    nl_s, ncols = int(g[case + "_nl_s"]), int(g[case + "_ncols"])
    S = (orc.rc_to_rowptr(g[case + "_S_rc"]), g[case + "_S_cols"], g[case + "_S_vals"])
    ctx.matrix_upload_csr(nl_s, ncols, *S)
    b = g[case + "_b"]
    diag, rhs = ([g[case + "_C_vals"]], [b[nl_s:]]) if case == "damp" else ([], [])
    worst = {}
    for rep in range(300):
        for (niter, rmin, gamma), xref, rref, itref in zip(g[case + "_runs"], g[case + "_x"], g[case + "_r"], g[case + "_iters"]):
            x, it, r = ctx.lsqr_solve_sensit(b[:nl_s], int(niter), rmin, gamma, 0.0, diag, rhs)
            k = (int(niter), float(rmin), float(gamma), int(itref))
            ex = np.linalg.norm(x - xref) / np.linalg.norm(xref); er = abs(r - rref) / abs(rref)
            w = worst.get(k, (0, 0, 0))
            worst[k] = (max(w[0], ex), max(w[1], er), max(w[2], abs(it - itref)))
    for k, v in worst.items():
        print(case, k, "x %.2e r %.2e dit %d" % v)
