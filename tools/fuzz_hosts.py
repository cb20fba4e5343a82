"""Cross-check of the two hosts on random Parfiles: the Fortran host (`tomofastx_amd -p Parfile`: parser, ASCII readers / writers,
weights, damping, ADMM, units) against the Python host (`inversion.solve_problem_gravity`) over the same `libtfx.so`.  Random
grids, data, compression, depth weighting type 1 / 2, prior / starting values, L2 damping, global ADMM bounds, unit multipliers;
two major iterations of a few LSQR iterations (before rounding is amplified).  Test infrastructure; GPU box."""
import importlib
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
tfx = importlib.import_module("tomofast-x_amd")
EXE = os.path.join(ROOT, "tomofast-x_amd", "host", "tomofastx_amd")
ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 51)
ctx = tfx.Context(0)


def read_col(path, ntok):
    """last column of a list-directed file: a count, then records of ntok numbers (the Fortran runtime may wrap a record)"""
    tok = open(path).read().split()
    return np.array(tok[1:], np.float64).reshape(-1, ntok)[:, -1]


for case in range(ncases):
    nx, ny, nz = (int(rng.integers(3, 11)) for _ in range(3))
    h = float(rng.uniform(40.0, 160.0))
    grid = tfx.synthetic.grid(nx, ny, nz, h=h)
    N = nx * ny * nz
    nd = int(rng.integers(3, 13))
    xs, ys = rng.uniform(0, nx * h, nd) + 0.123, rng.uniform(0, ny * h, nd) + 0.321
    zs = -rng.uniform(0.5, 40.0, nd)
    k, j, i = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
    mtrue = np.where((k.ravel() >= nz // 3) & (j.ravel() >= ny // 3) & (i.ravel() < 2 * nx // 3), 250.0, 0.0)
    ctype = int(rng.integers(0, 3))
    rate = float(rng.choice([0.2, 0.5, 1.0])) if ctype else 1.0
    if ctype and int(rate * N) == 0:
        continue
    dwt = int(rng.integers(1, 3))
    power = float(rng.choice([2.0, 3.0]))
    alpha = float(rng.choice([1e-7, 1e-5, 1e-3]))
    prior, start = float(rng.choice([0.0, 20.0])), float(rng.choice([0.0, 5.0]))
    admm = bool(rng.integers(0, 2))
    rho = 1e-5
    nmajor, nminor = 2, int(rng.integers(2, 6))
    mag = bool(rng.integers(0, 2))
    normp = float(rng.choice([2.0, 2.0, 1.5]))
    beta = float(rng.choice([0.0, 0.0, 1e-6]))
    field = (float(rng.uniform(40, 80)), float(rng.uniform(-20, 20)), 0.0, 50000.0)
    tag, sfx = ("magn", "mag") if mag else ("grav", "grav")
    if mag:
        mtrue = mtrue * 1e-4
    with tempfile.TemporaryDirectory() as wd:
        with open(os.path.join(wd, "grid.txt"), "w") as f:
            f.write("%d\n" % N)
            for p in range(N):
                f.write("%.17g %.17g %.17g %.17g %.17g %.17g %d %d %d\n" % (grid[0][p], grid[1][p], grid[2][p], grid[3][p], grid[4][p],
                                                                          grid[5][p], i.ravel()[p] + 1, j.ravel()[p] + 1, k.ravel()[p] + 1))
        with open(os.path.join(wd, "data_grid.txt"), "w") as f:
            f.write("%d\n" % nd)
            for o in zip(xs, ys, zs):
                f.write("%.17g %.17g %.17g 0.0\n" % o)
        with open(os.path.join(wd, "model_true.txt"), "w") as f:
            f.write("%d\n" % N)
            f.write("\n".join("%.17g" % v for v in mtrue) + "\n")
        par = """global.outputFolderPath     = out/
modelGrid.size                      = %d %d %d
modelGrid.TAG.file                 = grid.txt
forward.data.TAG.nData             = %d
forward.data.TAG.dataGridFile      = data_grid.txt
forward.data.TAG.useSyntheticModelForDataValues = 1
forward.data.TAG.syntheticModelFile = model_true.txt
forward.depthWeighting.type         = %d
forward.depthWeighting.TAG.power   = %.17g
forward.depthWeighting.TAG.beta    = 1.0d0
forward.depthWeighting.TAG.Z0      = 0.d0
sensit.readFromFiles                = 0
forward.matrixCompression.type      = %d
forward.matrixCompression.rate      = %.17g
inversion.priorModel.type           = 1
inversion.priorModel.TAG.value     = %.17g
inversion.startingModel.type        = 1
inversion.startingModel.TAG.value  = %.17g
inversion.nMajorIterations          = %d
inversion.nMinorIterations          = %d
inversion.minResidual               = 1.d-13
inversion.modelDamping.TAG.weight  = %.17g
inversion.modelDamping.normPower    = %.17g
inversion.dampingGradient.TAG.weight = %.17g
inversion.joint.grav.problemWeight  = %s
inversion.joint.magn.problemWeight  = %s
inversion.joint.TAG.columnWeightMultiplier = 4.d+3
forward.magneticField.inclination   = %.17g
forward.magneticField.declination   = %.17g
forward.magneticField.intensity_nT  = %.17g
forward.magneticField.XaxisDeclination = 0.d0
""" % (nx, ny, nz, nd, dwt, power, ctype, rate, prior, start, nmajor, nminor, alpha, normp, beta, "0.d0" if mag else "1.d0",
       "1.d0" if mag else "0.d0", field[0], field[1], field[3])
        par = par.replace("TAG", tag)
        if admm:
            par += "inversion.admm.enableADMM = 1\ninversion.admm.nLithologies = 2\ninversion.admm.%s.bounds = -10. 10. 200. 300.\ninversion.admm.%s.weight = %.17g\n" % (tag, tag, rho)
        open(os.path.join(wd, "Parfile.txt"), "w").write(par)
        out = subprocess.run([EXE, "-p", "Parfile.txt"], cwd=wd, capture_output=True, text=True, timeout=600, env=dict(os.environ, TFX_WRITE_SENSIT="0"))
        assert out.returncode == 0 and "THE END." in out.stdout, (case, out.stdout[-1500:], out.stderr[-1500:])
        m_f = read_col(os.path.join(wd, "out", "model", sfx + "_final_model_full.txt"), 1)
        d_f = read_col(os.path.join(wd, "out", "data", sfx + "_final.txt"), 4)
    # the same run through the Python host
    ctx.set_grid(nx, ny, nz, *grid)
    cw = ctx.calculate_depth_weight(power, 0.0, 4.0e3) if dwt == 1 else ctx.calculate_distance_weight(xs, ys, zs, power, 1.0, 4.0e3)
    ctx.calculate_sensit(xs, ys, zs, cw, ctype, rate, mag_field=field if mag else None)
    scaled = np.where(cw != 0.0, mtrue / np.where(cw != 0.0, cw, 1.0), 0.0)
    d_obs = ctx.calc_data(ctx.forward_wavelet(scaled, nx, ny, nz, ctype) if ctype else scaled, 1.0, None)
    m_p, d_p, hist = tfx.inversion.solve_problem_gravity(ctx, cw, ctype, d_obs, nmajor, nminor, alpha=alpha, model_start=np.full(N, start),
                                                        model_prior=np.full(N, prior), beta=beta, norm_power=normp,
                                                        admm=dict(bounds=[-10.0, 10.0, 200.0, 300.0], rho=rho) if admm else None)
    em = np.linalg.norm(m_f - m_p) / max(np.linalg.norm(m_p), 1e-300)
    ed = np.linalg.norm(d_f - d_p) / max(np.linalg.norm(d_p), 1e-300)
    tol = 1e-8
    if max(em, ed) > tol:
        # a weakly damped system amplifies the run-to-run rounding of the products (LDS atomics) within a few iterations: how far
        # apart are two runs of the SAME host?
        own = 0.0
        for _ in range(3):
            m_q, d_q, _ = tfx.inversion.solve_problem_gravity(ctx, cw, ctype, d_obs, nmajor, nminor, alpha=alpha, model_start=np.full(N, start),
                                                             model_prior=np.full(N, prior), beta=beta, norm_power=normp,
                                                             admm=dict(bounds=[-10.0, 10.0, 200.0, 300.0], rho=rho) if admm else None)
            own = max(own, np.linalg.norm(m_q - m_p) / max(np.linalg.norm(m_p), 1e-300))
        tol = max(tol, 30.0 * own)
    if em <= tol < ed:
        # The data of a model dominated by a uniform start / prior value cancel (a uniform magnetic slab has no anomaly away from
        # its edges): |S| |m| / |S m| reaches 1e3, so models that agree to 1e-10 give data that agree to 1e-7 only.  The
        # well-conditioned statement is that the Fortran host's data are the forward response of ITS OWN final model.
        sc_f = np.where(cw != 0.0, m_f / np.where(cw != 0.0, cw, 1.0), 0.0)
        d_chk = ctx.calc_data(ctx.forward_wavelet(sc_f, nx, ny, nz, ctype) if ctype else sc_f, 1.0, None)
        ed = np.linalg.norm(d_f - d_chk) / max(np.linalg.norm(d_chk), 1e-300)
    assert em <= tol and ed <= tol, (case, em, ed, tol, (nx, ny, nz), nd, ctype, rate, dwt, alpha, prior, start, admm, nminor, tag, normp, beta)
    print("case %2d %s %2dx%2dx%2d nd %2d ctype %d rate %.1f dw %d alpha %.0e prior %g start %g admm %d Lp %.1f beta %.0e nminor %d: model %.1e data %.1e" % (
        case, tag, nx, ny, nz, nd, ctype, rate, dwt, alpha, prior, start, admm, normp, beta, nminor, em, ed))
print("OK (%d cases)" % ncases)
