#!/usr/bin/env python3
"""(oracle/_ref/hamersley_xgrad_SENSIT: `KEEP_SENSIT=1 python tests/golden/make_golden.py hamersley_conv` in the development container.)
CPU only. The first LSQR solve of the reference's joint Hamersley example (parfiles/hamersley/Parfile_hamersley_xgrad_joint.txt: 226 data,
2 x 57 057 unknowns, model damping 5.92e-8 / 2.8e2, problem weights 1 / 2.5e-6; the cross-gradient rows are zero at the zero start model) in
three arithmetics on the REFERENCE'S OWN kernel files (oracle/_ref/hamersley_xgrad_SENSIT, written by oracle/_ref/tomofastx): fp64 with the
reference's sequential sums (the C oracle), fp64 with numpy's pairwise / blocked sums, 80-bit long double.  r after 100 / 400 iterations:
    the compiled reference              1.497770582e-02   4.801500480e-03
    sequential fp64 (C oracle)          1.497770582e-02   4.801500480e-03     <- ALL 16 digits of the reference's (1.497770582171863e-02,
                                                                                4.801500480366518e-03) since round 6: norm2(u) evaluated like the
                                                                                reference's Fortran runtime (oracle/tfx_oracle.c norm2_flang);
                                                                                with a plain sum for norm2 (rounds 1-5): 1.506765081e-02 / 4.801600271e-03
    numpy fp64                          1.251982061e-02   4.761320369e-03
    the HIP path (MI355X)               1.250521353e-02   4.761308399e-03
    80-bit long double                  1.231753998e-02   4.743435043e-03
    (converged, 1600 iterations: 4.737197450e-03 on the reference and on the HIP path, 4.4e-11 apart)
The solve is stopped where its residual still falls fast; the less the sums round, the further the recurrence has got.
  python tools/hamersley_precision.py"""
import sys, os
sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo')
import numpy as np
import oracle_lib as orc
g = np.load('/root/repo/tests/golden/hamersley.npz')
N = 13 * 133 * 33
sd = '/root/repo/oracle/_ref/hamersley_xgrad_SENSIT'
def read_kernel(fn):
    b = open(fn, 'rb').read()
    hdr = np.frombuffer(b[:20], '>i4'); nd = int(hdr[0]); pos = 20
    A = np.zeros((nd, N), np.float32)
    for r in range(nd):
        idata, nel, kc, dc = np.frombuffer(b[pos:pos + 16], '>i4'); pos += 16
        cols = np.frombuffer(b[pos:pos + 4 * nel], '>i4').astype(np.int64) - 1; pos += 4 * nel
        vals = np.frombuffer(b[pos:pos + 4 * nel], '>f4').astype(np.float32); pos += 4 * nel
        A[idata - 1, cols] = vals
    return A
A1, A2 = read_kernel(sd + '/sensit_grav_1_0'), read_kernel(sd + '/sensit_magn_1_0')
pw1, pw2 = 1.0, 2.5e-6
a1, a2 = 5.92e-8, 2.8e2
d1, d2 = g['data_grav'][:, -1], g['data_magn'][:, -1]
print('data shapes', g['data_grav'].shape, d1[:3], d2[:3])
S = np.zeros((226, 2 * N), np.float32)
S[:113, :N] = A1 * np.float32(pw1); S[113:, N:] = A2 * np.float32(pw2)
b = np.concatenate([pw1 * d1, pw2 * d2])
diag = np.concatenate([np.full(N, np.float32(a1 * pw1)), np.full(N, np.float32(a2 * pw2))]).astype(np.float32)
# CSR (dense rows, zeros dropped)
nz = S != 0
rp = np.concatenate([[0], np.cumsum(nz.sum(1))]).astype(np.int64)
cols = (np.nonzero(nz)[1] + 1).astype(np.int32); vals = S[nz]
for K in (100, 400):
    x, it, r = orc.lsqr((rp, cols, vals), orc.diag_csr(diag), 2 * N, np.concatenate([b, np.zeros(2 * N)]), K, 1e-300)
    print('seq fp64 oracle: K', K, 'r', r, flush=True)
LD = np.longdouble
SL = S.astype(LD); dL = diag.astype(LD)
def lsqr_ld(K, dtype):
    Sx = S.astype(dtype); dg = diag.astype(dtype)
    u = np.concatenate([b, np.zeros(2 * N)]).astype(dtype)
    nrm = lambda v: np.sqrt(np.sum(v * v))
    A = lambda x: np.concatenate([Sx @ x, dg * x])
    AT = lambda u: Sx.T @ u[:226] + dg * u[226:]
    beta = nrm(u); u /= beta; b1 = beta
    v = AT(u); al = nrm(v); v /= al
    w = v.copy(); x = np.zeros(2 * N, dtype); phibar = beta; rhobar = al
    out = {}
    for k in range(1, K + 1):
        u = A(v) - al * u; beta = nrm(u); u /= beta
        v = AT(u) - beta * v; al = nrm(v); v /= al
        rho = np.sqrt(rhobar * rhobar + beta * beta)
        c, s_ = rhobar / rho, beta / rho
        theta = s_ * al; rhobar = -c * al; phi = c * phibar; phibar = s_ * phibar
        x = x + (phi / rho) * w
        w = v - (theta / rho) * w
        if k in (100, 400): out[k] = float(phibar / b1); print(dtype.__name__, 'K', k, 'r', out[k], flush=True)
    return out
lsqr_ld(400, np.float64)       # numpy fp64 (pairwise / BLAS sums): a third arithmetic
lsqr_ld(400, np.longdouble)
