"""Deterministic synthetic gravity problems: uniform 100 m cells, a lattice of observations 1 m above
the surface (never on a cell face), a 300 kg/m3 block.  Used by bench.py, smoke() and the parity tests."""
import numpy as np


def grid(nx, ny, nz, h=100.0):
    k, j, i = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
    i, j, k = i.ravel().astype(np.float64), j.ravel().astype(np.float64), k.ravel().astype(np.float64)   # i fastest
    return i * h, (i + 1) * h, j * h, (j + 1) * h, k * h, (k + 1) * h


def observations(nx, ny, ox, oy, h=100.0):
    a, b = np.meshgrid(np.arange(ox), np.arange(oy), indexing="xy")
    xs = (a.ravel() + 0.5) * nx * h / ox + 0.37
    ys = (b.ravel() + 0.5) * ny * h / oy + 0.41
    return xs, ys, np.full(xs.size, -1.0)


def true_model(nx, ny, nz, rho=300.0):
    k, j, i = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
    i, j, k = i.ravel(), j.ravel(), k.ravel()
    inside = (k >= nz // 4) & (k < nz // 2) & (j >= ny // 3) & (j < 2 * ny // 3) & (i >= nx // 3) & (i < 2 * nx // 3)
    return np.where(inside, rho, 0.0)


PARFILE_TEMPLATE = """global.outputFolderPath     = output/synth/
global.description          = synthetic gravity inversion (SURVEY 8d generator)
modelGrid.size                      = {nx} {ny} {nz}
modelGrid.grav.file                 = grid.txt
forward.data.grav.nData             = {nd}
forward.data.grav.dataGridFile      = data_grid.txt
forward.data.grav.useSyntheticModelForDataValues = 1
forward.data.grav.syntheticModelFile = model_true.txt
forward.depthWeighting.type         = 1
forward.depthWeighting.grav.power   = 2.0d0
sensit.readFromFiles                = {sensit_read}
sensit.folderPath                   = output/synth/SENSIT/
forward.matrixCompression.type      = {ctype}
forward.matrixCompression.rate      = {rate}
inversion.priorModel.type           = 1
inversion.priorModel.grav.value     = 0.d0
inversion.startingModel.type        = 1
inversion.startingModel.grav.value  = 0.d0
inversion.nMajorIterations          = {nmajor}
inversion.nMinorIterations          = {nminor}
inversion.writeModelEveryNiter      = 0
inversion.minResidual               = 1.d-13
inversion.modelDamping.grav.weight  = 1.d-7
inversion.modelDamping.normPower    = 2.0d0
inversion.joint.grav.problemWeight  = 1.d0
inversion.joint.magn.problemWeight  = 0.d0
inversion.joint.grav.columnWeightMultiplier = 4.d+3
"""


def write_parfile_inputs(wd, nx, ny, nz, ox, oy, ctype, rate, nmajor=1, nminor=100, sensit_read=0):
    """The synthetic problem as the files a `tomofastx -p Parfile` run reads (reference ASCII formats: model grid
    src/inversion/model_IO.F90:135-241, model values :87-130, data grid src/forward/gravmag/data_gravmag.f90:204-239), so that the
    compiled reference and this repo's Fortran host can be run on the same inputs.  Returns the Parfile path."""
    import os
    X1, X2, Y1, Y2, Z1, Z2 = grid(nx, ny, nz)
    k, j, i = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
    n = nx * ny * nz
    # (vectorised: a 1e7-cell grid file is 10 columns x 1e7 lines; %.17g of these coordinates is exact)
    try:
        import pandas as pd
        df = pd.DataFrame({"a": X1, "b": X2, "c": Y1, "d": Y2, "e": Z1, "f": Z2, "g": i.ravel() + 1, "h": j.ravel() + 1, "i": k.ravel() + 1})
        with open(os.path.join(wd, "grid.txt"), "w") as f:
            f.write("%d\n" % n)
            df.to_csv(f, sep=" ", header=False, index=False, float_format="%.17g")
        with open(os.path.join(wd, "model_true.txt"), "w") as f:
            f.write("%d\n" % n)
            pd.DataFrame({"v": true_model(nx, ny, nz)}).to_csv(f, header=False, index=False, float_format="%.17g")
    except ImportError:
        cols = np.column_stack([X1, X2, Y1, Y2, Z1, Z2, i.ravel() + 1, j.ravel() + 1, k.ravel() + 1])
        with open(os.path.join(wd, "grid.txt"), "w") as f:
            f.write("%d\n" % n)
            np.savetxt(f, cols, fmt=["%.17g"] * 6 + ["%d"] * 3)
        with open(os.path.join(wd, "model_true.txt"), "w") as f:
            f.write("%d\n" % n)
            np.savetxt(f, true_model(nx, ny, nz), fmt="%.17g")
    xs, ys, zs = observations(nx, ny, ox, oy)
    with open(os.path.join(wd, "data_grid.txt"), "w") as f:
        f.write("%d\n" % xs.size)
        for a, b, c in zip(xs, ys, zs):
            f.write("%.17g %.17g %.17g 0.0\n" % (a, b, c))
    return write_parfile_text(wd, nx, ny, nz, xs.size, ctype, rate, nmajor=nmajor, nminor=nminor, sensit_read=sensit_read)


def write_parfile_text(wd, nx, ny, nz, nd, ctype, rate, nmajor=1, nminor=100, sensit_read=0):
    """Only the Parfile (the grid / model / data-grid files of write_parfile_inputs stay as they are): a 4e6-cell grid file is 400 MB of
    text and need not be rewritten to change an iteration count."""
    import os
    path = os.path.join(wd, "Parfile.txt")
    open(path, "w").write(PARFILE_TEMPLATE.format(nx=nx, ny=ny, nz=nz, nd=nd, ctype=ctype, rate=rate, nmajor=nmajor,
                                                 nminor=nminor, sensit_read=sensit_read))
    return path
