"""BASELINE config 1 driven the reference's way: `tomofastx_amd -p Parfile` (Fortran host, amdflang + iso_c_binding over
libtfx.so) on the mansf_slice example (2 x 128 x 32 cells, 256 data, Haar 0.15, ADMM with 3 lithologies, 60 x 100 LSQR
iterations), reading the reference's ASCII input formats and writing its output files.  Compared with the files the
reference itself wrote for the same Parfile (tests/golden/mansf.npz)."""
import os
import re
import subprocess

import numpy as np
import pytest
import ref_binaries
from parity_report import report

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tomofast-x_amd", "host", "tomofastx_amd")

# The keys / values of parfiles/Parfile_mansf_slice.txt (section banners and comments of the original omitted)
PARFILE = """
global.outputFolderPath     = output/mansf_slice/
global.description          = Gravity inversion with ADMM constraints (Mansfield area)
modelGrid.size                      = 2 128 32
modelGrid.grav.file                 = data/gravmag/mansf_slice/true_model_grav_3litho-grid.txt
forward.data.grav.nData             = 256
forward.data.grav.dataGridFile      = data/gravmag/mansf_slice/data_grid.txt
forward.data.grav.useSyntheticModelForDataValues = 1
forward.data.grav.syntheticModelFile = data/gravmag/mansf_slice/true_model_grav_3litho-values.txt
forward.depthWeighting.type         = 1
forward.depthWeighting.grav.power   = 2.0d0
sensit.readFromFiles                = 0
sensit.folderPath                   = output/mansf_slice/SENSIT/
forward.matrixCompression.type      = 1
forward.matrixCompression.rate      = 0.15
inversion.priorModel.type           = 1
inversion.priorModel.grav.value     = 0.d0
inversion.startingModel.type        = 1
inversion.startingModel.grav.value  = 0.d0
inversion.nMajorIterations          = 60
inversion.nMinorIterations          = 100
inversion.writeModelEveryNiter      = 0
inversion.minResidual               = 1.d-13
inversion.modelDamping.grav.weight  = 0.d0
inversion.modelDamping.normPower    = 2.0d0
inversion.joint.grav.problemWeight  = 1.d0
inversion.joint.magn.problemWeight  = 0.d0
inversion.admm.enableADMM           = 1
inversion.admm.nLithologies         = 3
inversion.admm.grav.bounds          = -20. 20. 90. 130. 220. 260.
inversion.admm.grav.weight          = 1.d-5
"""



class _Members(dict):
    files = property(lambda self: list(self))


def _load_npz(path):
    """The members of an .npz as a dict of arrays: indexing an NpzFile decompresses the member on every access, and the input writers
    below index the grid arrays cell by cell."""
    with np.load(path) as z:
        return _Members({k: z[k] for k in z.files})



def _sub_run(cmd, **kw):
    """subprocess.run; a run that does not come back is a failure that shows what the host had printed (its phase banners), not a bare
    TimeoutExpired after a quarter of an hour."""
    try:
        return subprocess.run(cmd, **kw)
    except subprocess.TimeoutExpired as e:
        def tail(b):
            return (b.decode("utf-8", "replace") if isinstance(b, bytes) else (b or ""))[-4000:]
        pytest.fail("%s did not finish within %s s\nstdout tail:\n%s\nstderr tail:\n%s" % (" ".join(str(c) for c in cmd[-6:]), kw.get("timeout"), tail(e.stdout), tail(e.stderr)))


def write_inputs(wd, g):
    dd = os.path.join(wd, "data", "gravmag", "mansf_slice")
    os.makedirs(dd)
    n = g["X1"].size
    nx, ny, nz = int(g["nx"]), int(g["ny"]), int(g["nz"])
    k, j, i = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
    with open(os.path.join(dd, "true_model_grav_3litho-grid.txt"), "w") as f:
        f.write("%d\n" % n)
        for p in range(n):
            f.write("%.17g %.17g %.17g %.17g %.17g %.17g %d %d %d\n" % (g["X1"][p], g["X2"][p], g["Y1"][p], g["Y2"][p], g["Z1"][p],
                                                                      g["Z2"][p], i.ravel()[p] + 1, j.ravel()[p] + 1, k.ravel()[p] + 1))
    with open(os.path.join(dd, "true_model_grav_3litho-values.txt"), "w") as f:
        f.write("%d\n" % n)
        f.write("\n".join("%.17g" % v for v in g["model_true"]) + "\n")
    with open(os.path.join(dd, "data_grid.txt"), "w") as f:
        f.write("%d\n" % g["obs"].shape[0])
        for o in g["obs"]:
            f.write("%.17g %.17g %.17g 0.0\n" % tuple(o))
    open(os.path.join(wd, "Parfile.txt"), "w").write("# tomofastx_amd parity run\n" + PARFILE)


def test_config1_from_parfile_matches_reference_outputs(tmp_path, golden_dir):
    if not os.path.isfile(EXE):
        ref_binaries.missing("Fortran host not built (no amdflang)")
    g = _load_npz(os.path.join(golden_dir, "mansf.npz"))
    wd = str(tmp_path)
    write_inputs(wd, g)
    out = _sub_run([EXE, "-p", "Parfile.txt"], cwd=wd, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "THE END." in out.stdout, out.stdout[-3000:] + out.stderr[-2000:]
    assert "nnz_total =" in out.stdout
    nnz = int(out.stdout.split("nnz_total =")[1].split()[0])
    assert abs(nnz - 314368) <= 16
    model = np.loadtxt(os.path.join(wd, "output", "mansf_slice", "model", "grav_final_model_full.txt"), skiprows=1)
    ref = g["model_final"]
    assert model.size == ref.size
    rel = np.linalg.norm(model - ref) / np.linalg.norm(ref)
    assert rel <= 1e-6, rel
    dfin = np.loadtxt(os.path.join(wd, "output", "mansf_slice", "data", "grav_final.txt"), skiprows=1)
    assert np.allclose(dfin[:, :3], g["obs"], rtol=1e-12)
    assert np.allclose(dfin[:, 3], g["data_final"], rtol=1e-6, atol=1e-9 * np.abs(g["data_final"]).max())
    dobs = np.loadtxt(os.path.join(wd, "output", "mansf_slice", "data", "grav_observed.txt"), skiprows=1)
    assert np.allclose(dobs[:, 3], g["data_observed"], rtol=1e-9, atol=1e-12 * np.abs(g["data_observed"]).max())
    costs = [l.split() for l in open(os.path.join(wd, "output", "mansf_slice", "costs.txt")) if not l.lstrip().startswith("#")]
    assert len(costs) == 61 and int(costs[-1][0]) == 60
    # final data cost: SURVEY 8d asks for <= 1e-5 relative; the reference's own 1 / 2 / 4-rank runs scatter by 1.5e-6
    dcost = abs(float(costs[-1][1]) - 9.339172972115141e-11) / 9.339172972115141e-11
    print("config 1 (Fortran host): final model rel-L2 %.3e, final data cost %s (relative distance %.3e)" % (rel, costs[-1][1], dcost))
    report("config1_mansf_end_to_end[shipping Fortran host]", model_rel_l2=float(rel), data_cost=float(costs[-1][1]), data_cost_rel_distance=float(dcost))
    assert dcost <= 5e-6, dcost          # measured 1.9-2.0e-6 on either host
    assert abs(float(costs[-1][2]) - 0.22595168071843558) <= 1e-5 * 0.22595168071843558            # final model cost


def read_tokens(path, ncol):
    """List-directed output wraps long records: all numbers after the count line, reshaped."""
    t = open(path).read().split()
    n = int(t[0])
    return np.array([float(v) for v in t[1:1 + n * ncol]]).reshape(n, ncol)


@pytest.mark.parametrize("name", ["e2e_ftg", "e2e_mag13", "e2e_mag31", "e2e_mag33"])
def test_multicomponent_parfiles_match_reference_outputs(tmp_path, golden_dir, name):
    """The Parfile the reference ran for the multi-component fixtures (gradiometry full tensor, three-component magnetic data,
    magnetisation vector; the Parfile text is ours, tests/golden/make_golden.py) through the Fortran host: same input
    files, same output files."""
    if not os.path.isfile(EXE):
        ref_binaries.missing("Fortran host not built (no amdflang)")
    g = _load_npz(os.path.join(golden_dir, name + ".npz"))
    wd = str(tmp_path)
    n = g["X1"].size
    nx, ny, nz = int(g["nx"]), int(g["ny"]), int(g["nz"])
    ncm, ncd = int(g["ncm"]), int(g["ncd"])
    k, j, i = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
    with open(os.path.join(wd, "grid.txt"), "w") as f:
        f.write("%d\n" % n)
        for p in range(n):
            f.write("%.17g %.17g %.17g %.17g %.17g %.17g %d %d %d\n" % (g["X1"][p], g["X2"][p], g["Y1"][p], g["Y2"][p], g["Z1"][p],
                                                                      g["Z2"][p], i.ravel()[p] + 1, j.ravel()[p] + 1, k.ravel()[p] + 1))
    with open(os.path.join(wd, "model_true.txt"), "w") as f:
        f.write("%d\n" % n)
        for v in g["model_true"]:
            f.write(" ".join("%.17g" % x for x in v) + "\n")
    with open(os.path.join(wd, "data_grid.txt"), "w") as f:
        f.write("%d\n" % g["obs"].shape[0])
        for o in g["obs"]:
            f.write("%.17g %.17g %.17g" % tuple(o) + " 0.0" * ncd + "\n")
    open(os.path.join(wd, "Parfile.txt"), "w").write(str(g["parfile"]))
    out = _sub_run([EXE, "-p", "Parfile.txt"], cwd=wd, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "THE END." in out.stdout, out.stdout[-3000:] + out.stderr[-2000:]
    sfx = "grav" if int(g["prob"]) == 1 else "mag"
    nnz = int(out.stdout.split("nnz_total =")[1].split()[0])
    assert abs(nnz - int(g["np1_nnz_total"])) <= 2 * g["obs"].shape[0] * ncm * ncd
    err = float(out.stdout.split("COMPRESSION ERROR, r =")[1].split()[0])
    assert abs(err - float(g["np1_comp_error"])) <= 1e-6 * float(g["np1_comp_error"])
    model = read_tokens(os.path.join(wd, "out", "model", sfx + "_final_model_full.txt"), ncm)
    ref, ref2 = g["np1_model_final"], g["np2_model_final"]
    self_diff = np.linalg.norm(ref2 - ref) / np.linalg.norm(ref)
    tol = max(1e-6, 100.0 * self_diff)
    assert np.linalg.norm(model - ref) <= tol * np.linalg.norm(ref), (np.linalg.norm(model - ref) / np.linalg.norm(ref), self_diff)
    dobs = read_tokens(os.path.join(wd, "out", "data", sfx + "_observed.txt"), 3 + ncd)
    assert np.allclose(dobs[:, :3], g["obs"], rtol=1e-12)
    assert np.allclose(dobs[:, 3:], g["np1_data_observed"], rtol=1e-9, atol=1e-11 * np.abs(g["np1_data_observed"]).max())
    dfin = read_tokens(os.path.join(wd, "out", "data", sfx + "_final.txt"), 3 + ncd)
    assert np.linalg.norm(dfin[:, 3:] - g["np1_data_final"]) <= 10.0 * tol * np.linalg.norm(g["np1_data_final"])


def write_case_inputs(wd, g):
    n = g["X1"].size
    nx, ny, nz = int(g["nx"]), int(g["ny"]), int(g["nz"])
    ncd = int(g["ncd"]) if "ncd" in g.files else 1
    k, j, i = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
    with open(os.path.join(wd, "grid.txt"), "w") as f:
        f.write("%d\n" % n)
        for p in range(n):
            f.write("%.17g %.17g %.17g %.17g %.17g %.17g %d %d %d\n" % (g["X1"][p], g["X2"][p], g["Y1"][p], g["Y2"][p], g["Z1"][p],
                                                                      g["Z2"][p], i.ravel()[p] + 1, j.ravel()[p] + 1, k.ravel()[p] + 1))
    mt = g["model_true"] if g["model_true"].ndim == 2 else g["model_true"][:, None]
    with open(os.path.join(wd, "model_true.txt"), "w") as f:
        f.write("%d\n" % n)
        for v in mt:
            f.write(" ".join("%.17g" % x for x in v) + "\n")
    with open(os.path.join(wd, "data_grid.txt"), "w") as f:
        f.write("%d\n" % g["obs"].shape[0])
        for o in g["obs"]:
            f.write("%.17g %.17g %.17g" % tuple(o) + " 0.0" * ncd + "\n")


@pytest.mark.parametrize("name", ["e2e_ftg", "e2e_mag31"])
def test_sensit_files_written_like_the_reference_and_reloaded(tmp_path, golden_dir, name):
    """calculate_and_write_sensit / read_sensitivity_kernel through the Fortran host: (1) the SENSIT set it writes parses
    with the reference's layout and matches the reference's lines; (2) a SENSIT set holding exactly the reference's kernel
    (written from the fixture) is re-loaded with sensit.readFromFiles = 1 and gives the reference's model."""
    import importlib
    if not os.path.isfile(EXE):
        ref_binaries.missing("Fortran host not built (no amdflang)")
    sio = importlib.import_module("tomofast-x_amd").sensit_io
    g = _load_npz(os.path.join(golden_dir, name + ".npz"))
    wd = str(tmp_path)
    write_case_inputs(wd, g)
    par = str(g["parfile"])
    open(os.path.join(wd, "Parfile.txt"), "w").write(par)
    out = _sub_run([EXE, "-p", "Parfile.txt"], cwd=wd, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "THE END." in out.stdout, out.stdout[-3000:] + out.stderr[-2000:]
    prob = int(g["prob"])
    ncm, ncd = int(g["ncm"]), int(g["ncd"])
    N = g["X1"].size
    got = sio.read_sensit(os.path.join(wd, "out", "SENSIT"), prob)
    assert got["meta"]["nmodel_components"] == ncm and got["meta"]["ndata_components"] == ncd
    assert got["column_weight"].tobytes() != b"" and np.max(np.abs(got["column_weight"] - g["np1_column_weight"]) / g["np1_column_weight"]) <= 1e-14
    sub_rp = g["np1_row_ptr"]
    kk = np.repeat(np.tile(np.arange(ncm), (sub_rp.size - 1) // ncm), np.diff(sub_rp))
    ref = (sub_rp[::ncm], (g["np1_cols"] + kk * N).astype(np.int32), g["np1_vals"])
    same = tot = 0
    for r in range(ref[0].size - 1):
        cb = got["cols"][got["rowptr"][r]:got["rowptr"][r + 1]]
        cr = ref[1][ref[0][r]:ref[0][r + 1]]
        same += np.intersect1d(cb, cr).size
        tot += max(cb.size, cr.size)
    assert same >= 0.995 * tot
    assert int(got["nnz_hist"].sum()) == got["meta"]["nnz_total"] == int(got["rowptr"][-1])
    # (2) reload the reference's own kernel
    wd2 = os.path.join(wd, "reload")
    os.makedirs(wd2)
    write_case_inputs(wd2, g)
    sio.write_sensit(os.path.join(wd2, "SENSIT_REF"), prob, ref, N, (int(g["nx"]), int(g["ny"]), int(g["nz"])), g["np1_column_weight"],
                     int(g["ctype"]), float(g["np1_comp_error"]), depth_weighting_type=int(g["dwtype"]), ndata_components=ncd,
                     nmodel_components=ncm)
    par2 = par.replace("sensit.readFromFiles                = 0", "sensit.readFromFiles                = 1")
    par2 = par2.replace("sensit.folderPath                   = out/SENSIT/", "sensit.folderPath                   = SENSIT_REF/")
    assert par2 != par
    open(os.path.join(wd2, "Parfile.txt"), "w").write(par2)
    out = _sub_run([EXE, "-p", "Parfile.txt"], cwd=wd2, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "Finished reading the sensitivity kernel." in out.stdout, out.stdout[-3000:] + out.stderr[-2000:]
    sfx = "grav" if prob == 1 else "mag"
    model = read_tokens(os.path.join(wd2, "out", "model", sfx + "_final_model_full.txt"), ncm)
    refm, ref2 = g["np1_model_final"], g["np2_model_final"]
    self_diff = np.linalg.norm(ref2 - refm) / np.linalg.norm(refm)
    tol = max(1e-6, 10.0 * self_diff)
    assert np.linalg.norm(model - refm) <= tol * np.linalg.norm(refm), (np.linalg.norm(model - refm) / np.linalg.norm(refm), self_diff)


def write_joint_inputs(wd, g):
    n = g["X1"].size
    nx, ny, nz = int(g["nx"]), int(g["ny"]), int(g["nz"])
    k, j, i = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
    with open(os.path.join(wd, "grid.txt"), "w") as f:
        f.write("%d\n" % n)
        for p in range(n):
            f.write("%.17g %.17g %.17g %.17g %.17g %.17g %d %d %d\n" % (g["X1"][p], g["X2"][p], g["Y1"][p], g["Y2"][p], g["Z1"][p],
                                                                      g["Z2"][p], i.ravel()[p] + 1, j.ravel()[p] + 1, k.ravel()[p] + 1))
    for tag in ("grav", "magn"):
        with open(os.path.join(wd, "data_grid_%s.txt" % tag), "w") as f:
            f.write("%d\n" % g["obs_" + tag].shape[0])
            for o in g["obs_" + tag]:
                f.write("%.17g %.17g %.17g 0.0\n" % tuple(o))
        with open(os.path.join(wd, "model_true_%s.txt" % tag), "w") as f:
            f.write("%d\n" % n)
            f.write("\n".join("%.17g" % v for v in g["model_true_" + tag]) + "\n")
    open(os.path.join(wd, "Parfile.txt"), "w").write(str(g["parfile"]))


def test_joint_parfile_matches_reference_outputs(tmp_path, golden_dir):
    """Joint gravity + magnetic inversion from the Parfile (both problem weights non-zero): two kernels, one LSQR system."""
    if not os.path.isfile(EXE):
        ref_binaries.missing("Fortran host not built (no amdflang)")
    g = _load_npz(os.path.join(golden_dir, "e2e_joint.npz"))
    wd = str(tmp_path)
    write_joint_inputs(wd, g)
    out = _sub_run([EXE, "-p", "Parfile.txt"], cwd=wd, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "THE END." in out.stdout and "JOINT inversion" in out.stdout, out.stdout[-3000:] + out.stderr[-2000:]
    for tag, sfx in (("grav", "grav"), ("magn", "mag")):
        model = read_tokens(os.path.join(wd, "out", "model", sfx + "_final_model_full.txt"), 1)[:, 0]
        ref = g["np1_%s_model_final" % tag]
        self_diff = np.linalg.norm(g["np2_%s_model_final" % tag] - ref) / np.linalg.norm(ref)
        tol = max(1e-6, 100.0 * self_diff)
        assert np.linalg.norm(model - ref) <= tol * np.linalg.norm(ref), (tag, np.linalg.norm(model - ref) / np.linalg.norm(ref))
        dfin = read_tokens(os.path.join(wd, "out", "data", sfx + "_final.txt"), 4)[:, 3]
        assert np.linalg.norm(dfin - g["np1_%s_data_final" % tag]) <= 10 * tol * np.linalg.norm(g["np1_%s_data_final" % tag])
    costs = [l.split() for l in open(os.path.join(wd, "out", "costs.txt")) if l.strip() and not l.lstrip().startswith("#")]
    assert len(costs) == int(g["nmajor"]) + 1 and len(costs[-1]) == 9            # iteration + 4 columns per problem


@pytest.mark.parametrize("name", ["e2e_xgrad", "e2e_xgrad_cnt"])
def test_cross_gradient_parfile_matches_reference(tmp_path, golden_dir, name):
    """inversion.crossGradient.weight /= 0 on a joint run: the host builds the 3 N coupling rows over both models' columns
    (forward or central differences), uploads them as the general constraint matrix, WAVELET_DOMAIN = F."""
    if not os.path.isfile(EXE):
        ref_binaries.missing("Fortran host not built (no amdflang)")
    g = _load_npz(os.path.join(golden_dir, name + ".npz"))
    wd = str(tmp_path)
    write_joint_inputs(wd, g)
    out = _sub_run([EXE, "-p", "Parfile.txt"], cwd=wd, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "THE END." in out.stdout and "WAVELET_DOMAIN = F" in out.stdout, out.stdout[-3000:] + out.stderr[-2000:]
    for tag, sfx in (("grav", "grav"), ("magn", "mag")):
        model = read_tokens(os.path.join(wd, "out", "model", sfx + "_final_model_full.txt"), 1)[:, 0]
        ref = g["np1_%s_model_final" % tag]
        assert np.linalg.norm(model - ref) <= 1e-6 * np.linalg.norm(ref), (tag, np.linalg.norm(model - ref) / np.linalg.norm(ref))
    rs = [float(t.split()[0]) for t in out.stdout.split("End of subroutine lsqr_solve_sensit, r =")[1:]]
    assert np.allclose(rs, g["np1_lsqr_r"], rtol=1e-5)
    xc = np.array([[float(v) for v in t.split()[:3]] for t in out.stdout.split("cross-grad cost =")[1:]])
    assert np.allclose(xc[2:], g["np1_xgrad_cost"][2:], rtol=1e-4)


@pytest.mark.parametrize("name", ["e2e_clust", "e2e_clust_normal", "e2e_clust_grav"])
def test_clustering_parfile_matches_reference(tmp_path, golden_dir, name):
    """inversion.clustering.*.weight /= 0 on a joint run: the host reads the mixture (and per-cell weight) files, builds the 2 N
    Gaussian-mixture rows each major iteration and uploads them as the general constraint matrix, WAVELET_DOMAIN = F."""
    if not os.path.isfile(EXE):
        ref_binaries.missing("Fortran host not built (no amdflang)")
    g = _load_npz(os.path.join(golden_dir, name + ".npz"))
    wd = str(tmp_path)
    write_joint_inputs(wd, g)
    with open(os.path.join(wd, "mixtures.txt"), "w") as f:
        f.write("%d\n" % g["mixtures"].shape[0])
        for r in g["mixtures"]:
            f.write(" ".join("%.17g" % v for v in r) + "\n")
    with open(os.path.join(wd, "cell_weights.txt"), "w") as f:
        f.write("%d %d\n" % g["cell_weights"].shape)
        for r in g["cell_weights"]:
            f.write(" ".join("%.17g" % v for v in r) + "\n")
    out = _sub_run([EXE, "-p", "Parfile.txt"], cwd=wd, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "THE END." in out.stdout and "WAVELET_DOMAIN = F" in out.stdout, out.stdout[-3000:] + out.stderr[-2000:]
    for tag, sfx in (("grav", "grav"), ("magn", "mag")):
        model = read_tokens(os.path.join(wd, "out", "model", sfx + "_final_model_full.txt"), 1)[:, 0]
        ref = g["np1_%s_model_final" % tag]
        assert np.linalg.norm(model - ref) <= 1e-6 * np.linalg.norm(ref), (tag, np.linalg.norm(model - ref) / np.linalg.norm(ref))
    rs = [float(t.split()[0]) for t in out.stdout.split("End of subroutine lsqr_solve_sensit, r =")[1:]]
    assert np.allclose(rs, g["np1_lsqr_r"], rtol=1e-5)
    assert np.isclose(float(out.stdout.split("Clustering mixture_max =")[1].split()[0]), float(g["np1_mixture_max"][0]), rtol=1e-7)


MPIEXEC = "/opt/conda/bin/mpiexec"


@pytest.mark.parametrize("name", ["e2e_joint", "e2e_mag31"])
def test_two_ranks_under_mpiexec_match_the_reference_two_rank_run(tmp_path, golden_dir, name):
    """`mpiexec -n 2 tomofastx_amd -p Parfile`: one process per rank (both on the one GPU of this box), column ranges from the
    reference's nnz-balancing rule, LSQR reductions through the MPI-staged hook, model-update slices gathered per major
    iteration - against the reference's own 2-rank run of the same Parfile."""
    if not os.path.isfile(EXE):
        ref_binaries.missing("Fortran host not built (no amdflang)")
    if not os.path.isfile(MPIEXEC):
        ref_binaries.missing("no mpiexec in this image")
    g = _load_npz(os.path.join(golden_dir, name + ".npz"))
    wd = str(tmp_path)
    if name == "e2e_joint":
        n = g["X1"].size
        nx, ny, nz = int(g["nx"]), int(g["ny"]), int(g["nz"])
        k, j, i = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
        with open(os.path.join(wd, "grid.txt"), "w") as f:
            f.write("%d\n" % n)
            for p in range(n):
                f.write("%.17g %.17g %.17g %.17g %.17g %.17g %d %d %d\n" % (g["X1"][p], g["X2"][p], g["Y1"][p], g["Y2"][p], g["Z1"][p],
                                                                          g["Z2"][p], i.ravel()[p] + 1, j.ravel()[p] + 1, k.ravel()[p] + 1))
        for tag in ("grav", "magn"):
            with open(os.path.join(wd, "data_grid_%s.txt" % tag), "w") as f:
                f.write("%d\n" % g["obs_" + tag].shape[0])
                for o in g["obs_" + tag]:
                    f.write("%.17g %.17g %.17g 0.0\n" % tuple(o))
            with open(os.path.join(wd, "model_true_%s.txt" % tag), "w") as f:
                f.write("%d\n" % n)
                f.write("\n".join("%.17g" % v for v in g["model_true_" + tag]) + "\n")
        cases = [("grav", "grav", 1), ("magn", "mag", 1)]
    else:
        write_case_inputs(wd, g)
        cases = [(None, "mag", int(g["ncm"]))]
    open(os.path.join(wd, "Parfile.txt"), "w").write(str(g["parfile"]))
    out = _sub_run([MPIEXEC, "-n", "2", EXE, "-p", "Parfile.txt"], cwd=wd, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "THE END." in out.stdout and "Number of ranks" in out.stdout, out.stdout[-3000:] + out.stderr[-2000:]
    nel = [int(v) for v in out.stdout.split("nelements_at_cpu =")[1].split()[:2]]
    assert np.all(np.abs(np.array(nel) - g["np2_nelements_at_cpu"]) <= 2)          # a threshold tie may move the cut by a cell
    for tag, sfx, ncm in cases:
        key = "np2_%s_model_final" % tag if tag else "np2_model_final"
        key1 = "np1_%s_model_final" % tag if tag else "np1_model_final"
        model = read_tokens(os.path.join(wd, "out", "model", sfx + "_final_model_full.txt"), ncm)
        ref = g[key].reshape(model.shape)
        self_diff = np.linalg.norm(g[key1].reshape(model.shape) - ref) / np.linalg.norm(ref)
        tol = max(1e-6, 100.0 * self_diff)
        assert np.linalg.norm(model - ref) <= tol * np.linalg.norm(ref), (tag, np.linalg.norm(model - ref) / np.linalg.norm(ref))
    if name == "e2e_mag31":
        # the magnetisation-vector kernel went through the row-parallel build, and the two ranks wrote their row files (one
        # record per model component): the set reads back as the kernel of the reference's own files
        assert "row-parallel" in out.stdout
        import importlib
        back = importlib.import_module("tomofast-x_amd").sensit_io.read_sensit(os.path.join(wd, "out", "SENSIT"), 2)
        assert back["meta"]["nbproc"] == 2 and back["meta"]["nmodel_components"] == 3
        # the fixture keeps the reference's records, one per (datum, model component), with cell columns
        ncells = g["X1"].size
        lp = g["np1_row_ptr"]
        assert np.array_equal(back["rowptr"], lp[::3])
        gcols = np.concatenate([g["np1_cols"][lp[q]:lp[q + 1]].astype(np.int64) + (q % 3) * ncells for q in range(lp.size - 1)])
        assert np.array_equal(back["cols"], gcols)
        assert np.max(np.abs(back["vals"].view(np.int32).astype(np.int64) - g["np1_vals"].view(np.int32).astype(np.int64))) <= 2
        assert np.array_equal(back["nnz_hist"], g["np1_sensit_nnz"])
    # sensit.readFromFiles = 2 on 2 ranks: the depth weight comes from the SENSIT folder (written by a single-rank run: the
    # column-partitioned multi-rank build keeps its kernel on the devices), the kernel is built again
    # (problem_joint_gravmag.F90:189-202) - same models
    first = {sfx: read_tokens(os.path.join(wd, "out", "model", sfx + "_final_model_full.txt"), ncm) for _, sfx, ncm in cases}
    out = _sub_run([EXE, "-p", "Parfile.txt"], cwd=wd, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "THE END." in out.stdout, out.stdout[-3000:] + out.stderr[-2000:]
    par2 = re.sub(r"sensit\.readFromFiles\s*=\s*\d", "sensit.readFromFiles                = 2", str(g["parfile"]))
    assert par2 != str(g["parfile"])
    open(os.path.join(wd, "Parfile.txt"), "w").write(par2)
    out = _sub_run([MPIEXEC, "-n", "2", EXE, "-p", "Parfile.txt"], cwd=wd, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "THE END." in out.stdout, out.stdout[-3000:] + out.stderr[-2000:]
    for _, sfx, ncm in cases:
        again = read_tokens(os.path.join(wd, "out", "model", sfx + "_final_model_full.txt"), ncm)
        assert np.linalg.norm(again - first[sfx]) <= 1e-6 * np.linalg.norm(first[sfx])


@pytest.mark.parametrize("nranks", [2, 3])
def test_config1_with_admm_under_mpiexec(tmp_path, golden_dir, nranks):
    """BASELINE config 1 (60 x 100 LSQR iterations with damping AND the ADMM bound constraints) on 2 and 3 ranks: inside jinv%solve the
    ADMM projection runs on each rank's own cells (admm_method.F90:70-134), its cost is all-reduced, the damping / ADMM right-hand
    sides come from the gathered model - the final model must be the single-rank run's (the reference's own 1- vs 2- vs 4-rank
    scatter on this Parfile is 5e-12)."""
    if not os.path.isfile(EXE):
        ref_binaries.missing("Fortran host not built (no amdflang)")
    if not os.path.isfile(MPIEXEC):
        ref_binaries.missing("no mpiexec in this image")
    g = _load_npz(os.path.join(golden_dir, "mansf.npz"))
    wd = str(tmp_path)
    write_inputs(wd, g)
    out = _sub_run([MPIEXEC, "-n", str(nranks), EXE, "-p", "Parfile.txt"], cwd=wd, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "THE END." in out.stdout, out.stdout[-3000:] + out.stderr[-2000:]
    assert "ADMM cost" in out.stdout
    model = np.loadtxt(os.path.join(wd, "output", "mansf_slice", "model", "grav_final_model_full.txt"), skiprows=1)
    ref = g["model_final"]
    rel = np.linalg.norm(model - ref) / np.linalg.norm(ref)
    assert rel <= 1e-6, rel
    costs = [l.split() for l in open(os.path.join(wd, "output", "mansf_slice", "costs.txt")) if not l.lstrip().startswith("#")]
    assert len(costs) == 61 and abs(float(costs[-1][2]) - 0.22595168071843558) <= 1e-5 * 0.22595168071843558


@pytest.mark.parametrize("name", ["e2e_dgrad", "e2e_xgrad", "e2e_clust"])
def test_spatial_unknowns_two_ranks_under_mpiexec(tmp_path, golden_dir, name):
    """WAVELET_DOMAIN = F on 2 ranks (gradient damping; cross-gradient and clustering on joint runs): spatial unknowns per cell
    range (tfx_lsqr_set_partition), the constraint rows replicated with each rank's own columns - against the reference's own
    2-rank run of the same Parfile."""
    if not os.path.isfile(EXE):
        ref_binaries.missing("Fortran host not built (no amdflang)")
    if not os.path.isfile(MPIEXEC):
        ref_binaries.missing("no mpiexec in this image")
    g = _load_npz(os.path.join(golden_dir, name + ".npz"))
    wd = str(tmp_path)
    if name == "e2e_dgrad":
        write_case_inputs(wd, g)
        open(os.path.join(wd, "Parfile.txt"), "w").write(str(g["parfile"]))
        cases = [(None, "grav")]
    else:
        write_joint_inputs(wd, g)
        cases = [("grav", "grav"), ("magn", "mag")]
        if name == "e2e_clust":
            with open(os.path.join(wd, "mixtures.txt"), "w") as f:
                f.write("%d\n" % g["mixtures"].shape[0])
                for r in g["mixtures"]:
                    f.write(" ".join("%.17g" % v for v in r) + "\n")
            with open(os.path.join(wd, "cell_weights.txt"), "w") as f:
                f.write("%d %d\n" % g["cell_weights"].shape)
                for r in g["cell_weights"]:
                    f.write(" ".join("%.17g" % v for v in r) + "\n")
    out = _sub_run([MPIEXEC, "-n", "2", EXE, "-p", "Parfile.txt"], cwd=wd, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "THE END." in out.stdout and "Number of ranks" in out.stdout and "WAVELET_DOMAIN = F" in out.stdout, \
        out.stdout[-3000:] + out.stderr[-2000:]
    for tag, sfx in cases:
        model = read_tokens(os.path.join(wd, "out", "model", sfx + "_final_model_full.txt"), 1)[:, 0]
        ref = g["np2_%s_model_final" % tag if tag else "np2_model_final"]
        assert np.linalg.norm(model - ref) <= 1e-5 * np.linalg.norm(ref), (tag, np.linalg.norm(model - ref) / np.linalg.norm(ref))
    rs = [float(t.split()[0]) for t in out.stdout.split("End of subroutine lsqr_solve_sensit, r =")[1:]]
    assert np.allclose(rs[:len(g["np2_lsqr_r"])], g["np2_lsqr_r"], rtol=1e-4)


PAR_BIG = """global.outputFolderPath     = out/
modelGrid.size                      = 32 24 12
modelGrid.grav.file                 = grid.txt
forward.data.grav.nData             = {nd}
forward.data.grav.dataGridFile      = data_grid.txt
forward.data.grav.useSyntheticModelForDataValues = 1
forward.data.grav.syntheticModelFile = model_true.txt
forward.depthWeighting.type         = 1
forward.depthWeighting.grav.power   = 2.0d0
sensit.readFromFiles                = 0
forward.matrixCompression.type      = 2
forward.matrixCompression.rate      = 0.05d0
inversion.priorModel.type           = 1
inversion.priorModel.grav.value     = 0.d0
inversion.startingModel.type        = 1
inversion.startingModel.grav.value  = 0.d0
inversion.nMajorIterations          = 2
inversion.nMinorIterations          = 8
inversion.minResidual               = 1.d-13
inversion.modelDamping.grav.weight  = 1.d-7
inversion.joint.grav.problemWeight  = 1.d0
inversion.joint.magn.problemWeight  = 0.d0
"""


def test_row_parallel_build_with_mpi_relayout(tmp_path):
    """3 ranks under mpiexec, 4608 data = 3 row blocks: every rank compresses its row block with all columns, the pieces are
    cut by column range on the GPU and travel over MPI to the owners (read_sensitivity_kernel's relayout without the files).
    Same final model as the single-rank run and as the scheme where every rank builds all rows for its own columns."""
    import importlib
    if not os.path.isfile(EXE) or not os.path.isfile(MPIEXEC):
        ref_binaries.missing("Fortran host / mpiexec not available")
    syn = importlib.import_module("tomofast-x_amd").synthetic
    nx, ny, nz = 32, 24, 12
    grid = syn.grid(nx, ny, nz)
    xs, ys, zs = syn.observations(nx, ny, 72, 64)
    n = nx * ny * nz
    k, j, i = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
    mtrue = np.where((k.ravel() >= nz // 4) & (k.ravel() < nz // 2) & (j.ravel() >= ny // 3) & (j.ravel() < 2 * ny // 3) &
                     (i.ravel() >= nx // 3) & (i.ravel() < 2 * nx // 3), 300.0, 0.0)
    models = {}
    for tag, cmd, env in (("one", [EXE], {}), ("exchange", [MPIEXEC, "-n", "3", EXE], {}),
                          ("redundant", [MPIEXEC, "-n", "3", EXE], {"TFX_BUILD_MODE": "redundant"})):
        wd = os.path.join(str(tmp_path), tag)
        os.makedirs(wd)
        with open(os.path.join(wd, "grid.txt"), "w") as f:
            f.write("%d\n" % n)
            for p in range(n):
                f.write("%.17g %.17g %.17g %.17g %.17g %.17g %d %d %d\n" % (grid[0][p], grid[1][p], grid[2][p], grid[3][p], grid[4][p],
                                                                          grid[5][p], i.ravel()[p] + 1, j.ravel()[p] + 1, k.ravel()[p] + 1))
        with open(os.path.join(wd, "data_grid.txt"), "w") as f:
            f.write("%d\n" % xs.size)
            for o in zip(xs, ys, zs):
                f.write("%.17g %.17g %.17g 0.0\n" % o)
        with open(os.path.join(wd, "model_true.txt"), "w") as f:
            f.write("%d\n" % n)
            f.write("\n".join("%.17g" % v for v in mtrue) + "\n")
        open(os.path.join(wd, "Parfile.txt"), "w").write(PAR_BIG.format(nd=xs.size))
        # (the products are reproducible - fixed summation order / exact integer accumulation - so runs that stop mid-convergence
        # can be compared: nothing run-dependent decides the outcome)
        e = dict(os.environ, TFX_WRITE_SENSIT="0", **env)
        out = _sub_run(cmd + ["-p", "Parfile.txt"], cwd=wd, capture_output=True, text=True, timeout=300, env=e)
        assert out.returncode == 0 and "THE END." in out.stdout, out.stdout[-3000:] + out.stderr[-2000:]
        if tag == "exchange":
            assert "row-parallel" in out.stdout
        if tag == "redundant":
            assert "row-parallel" not in out.stdout
        nnz = int(out.stdout.split("nnz_total =")[1].split()[0])
        models[tag] = (read_tokens(os.path.join(wd, "out", "model", "grav_final_model_full.txt"), 1)[:, 0], nnz)
    ref, nnz_ref = models["one"]
    for tag in ("exchange", "redundant"):
        m, nnz = models[tag]
        assert nnz == nnz_ref
        # 2 x 8 LSQR iterations stop mid-convergence, where the Golub-Kahan recurrence amplifies the last-bit differences of
        # a different summation order (3 ranks instead of 1): typically 1e-6.  A relayout error (a row piece on the wrong rank,
        # shifted columns) changes the model at the 1e-1 level.
        assert np.linalg.norm(m - ref) <= 1e-4 * np.linalg.norm(ref), (tag, np.linalg.norm(m - ref) / np.linalg.norm(ref))
    assert np.linalg.norm(models["exchange"][0] - models["redundant"][0]) <= 1e-4 * np.linalg.norm(ref)


def test_sensit_files_of_a_multi_rank_run(tmp_path):
    """3 ranks under mpiexec (row-parallel build): every rank writes the row file of its own row block from the device row
    store, rank 0 the metadata / counts / weights.  The set reads back as the kernel a single-rank run writes (same bits), and a
    2-rank run with sensit.readFromFiles = 1 on it gives the same model."""
    import importlib
    if not os.path.isfile(EXE) or not os.path.isfile(MPIEXEC):
        ref_binaries.missing("Fortran host / mpiexec not available")
    pkg = importlib.import_module("tomofast-x_amd")
    syn = pkg.synthetic
    nx, ny, nz = 32, 24, 12
    grid = syn.grid(nx, ny, nz)
    xs, ys, zs = syn.observations(nx, ny, 72, 64)
    n = nx * ny * nz
    k, j, i = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
    mtrue = np.where((k.ravel() >= nz // 4) & (k.ravel() < nz // 2) & (j.ravel() >= ny // 3) & (j.ravel() < 2 * ny // 3) &
                     (i.ravel() >= nx // 3) & (i.ravel() < 2 * nx // 3), 300.0, 0.0)
    models, sets = {}, {}
    for tag, cmd, par in (("one", [EXE], PAR_BIG), ("three", [MPIEXEC, "-n", "3", EXE], PAR_BIG),
                          ("reload", [MPIEXEC, "-n", "2", EXE], PAR_BIG.replace("sensit.readFromFiles                = 0", "sensit.readFromFiles                = 1"))):
        wd = os.path.join(str(tmp_path), "three" if tag == "reload" else tag)
        if tag != "reload":
            os.makedirs(wd)
            with open(os.path.join(wd, "grid.txt"), "w") as f:
                f.write("%d\n" % n)
                for p in range(n):
                    f.write("%.17g %.17g %.17g %.17g %.17g %.17g %d %d %d\n" % (grid[0][p], grid[1][p], grid[2][p], grid[3][p], grid[4][p],
                                                                              grid[5][p], i.ravel()[p] + 1, j.ravel()[p] + 1, k.ravel()[p] + 1))
            with open(os.path.join(wd, "data_grid.txt"), "w") as f:
                f.write("%d\n" % xs.size)
                for o in zip(xs, ys, zs):
                    f.write("%.17g %.17g %.17g 0.0\n" % o)
            with open(os.path.join(wd, "model_true.txt"), "w") as f:
                f.write("%d\n" % n)
                f.write("\n".join("%.17g" % v for v in mtrue) + "\n")
        assert "sensit.readFromFiles" in par
        open(os.path.join(wd, "Parfile.txt"), "w").write(par.format(nd=xs.size) + "sensit.folderPath                   = out/SENSIT/\n")
        out = _sub_run(cmd + ["-p", "Parfile.txt"], cwd=wd, capture_output=True, text=True, timeout=300,
                             env=dict(os.environ))
        assert out.returncode == 0 and "THE END." in out.stdout, out.stdout[-3000:] + out.stderr[-2000:]
        models[tag] = read_tokens(os.path.join(wd, "out", "model", "grav_final_model_full.txt"), 1)[:, 0]
        if tag != "reload":
            sets[tag] = pkg.sensit_io.read_sensit(os.path.join(wd, "out", "SENSIT"), 1)
    a, b = sets["one"], sets["three"]
    assert b["meta"]["nbproc"] == 3 and a["meta"]["nbproc"] == 1 and a["meta"]["nnz_total"] == b["meta"]["nnz_total"]
    assert sorted(f for f in os.listdir(os.path.join(str(tmp_path), "three", "out", "SENSIT")) if f.startswith("sensit_grav_3_")) == \
        ["sensit_grav_3_0", "sensit_grav_3_1", "sensit_grav_3_2"]
    assert np.array_equal(a["rowptr"], b["rowptr"]) and np.array_equal(a["cols"], b["cols"]) and a["vals"].tobytes() == b["vals"].tobytes()
    assert np.array_equal(a["nnz_hist"], b["nnz_hist"]) and a["column_weight"].tobytes() == b["column_weight"].tobytes()
    assert abs(a["meta"]["comp_error"] - b["meta"]["comp_error"]) <= 1e-12 * a["meta"]["comp_error"]
    for tag in ("three", "reload"):                   # mid-convergence comparison: see test_row_parallel_build_with_mpi_relayout
        assert np.linalg.norm(models[tag] - models["one"]) <= 1e-4 * np.linalg.norm(models["one"]), tag


def test_gradient_damping_parfile_matches_reference(tmp_path, golden_dir):
    """inversion.dampingGradient.grav.weight /= 0: the host builds the first-difference rows, uploads them as the general
    constraint matrix and solves with WAVELET_DOMAIN = F (spatial unknowns, per-iteration device transform)."""
    if not os.path.isfile(EXE):
        ref_binaries.missing("Fortran host not built (no amdflang)")
    g = _load_npz(os.path.join(golden_dir, "e2e_dgrad.npz"))
    wd = str(tmp_path)
    write_case_inputs(wd, g)
    open(os.path.join(wd, "Parfile.txt"), "w").write(str(g["parfile"]))
    out = _sub_run([EXE, "-p", "Parfile.txt"], cwd=wd, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "THE END." in out.stdout and "WAVELET_DOMAIN = F" in out.stdout, out.stdout[-3000:] + out.stderr[-2000:]
    model = read_tokens(os.path.join(wd, "out", "model", "grav_final_model_full.txt"), 1)[:, 0]
    ref = g["np1_model_final"]
    assert np.linalg.norm(model - ref) <= 1e-5 * np.linalg.norm(ref), np.linalg.norm(model - ref) / np.linalg.norm(ref)
    rs = [float(t.split()[0]) for t in out.stdout.split("End of subroutine lsqr_solve_sensit, r =")[1:]]
    assert np.allclose(rs, g["np1_lsqr_r"], rtol=1e-4)


def test_mindist_depth_weight_parfile_matches_reference(tmp_path, golden_dir):
    """forward.depthWeighting.type = 3 from the Parfile: calculate_depth_weight -> tfx_column_weight_type3, then the usual run."""
    if not os.path.isfile(EXE):
        ref_binaries.missing("Fortran host not built (no amdflang)")
    g = _load_npz(os.path.join(golden_dir, "e2e_dw3.npz"))
    wd = str(tmp_path)
    write_case_inputs(wd, g)
    open(os.path.join(wd, "Parfile.txt"), "w").write(str(g["parfile"]))
    out = _sub_run([EXE, "-p", "Parfile.txt"], cwd=wd, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "THE END." in out.stdout, out.stdout[-3000:] + out.stderr[-2000:]
    assert "Calculating the depth weight, type =" in out.stdout
    model = read_tokens(os.path.join(wd, "out", "model", "grav_final_model_full.txt"), 1)[:, 0]
    ref = g["np1_model_final"]
    assert np.linalg.norm(model - ref) <= 1e-6 * np.linalg.norm(ref), np.linalg.norm(model - ref) / np.linalg.norm(ref)
    w = np.frombuffer(open(os.path.join(wd, "out", "SENSIT", "sensit_grav_weight"), "rb").read(), ">f8", offset=4).astype(np.float64)
    assert np.max(np.abs(w[:ref.size] - g["np1_column_weight"]) / g["np1_column_weight"]) <= 1e-14


def test_lp_norm_damping_parfile_matches_reference(tmp_path, golden_dir):
    if not os.path.isfile(EXE):
        ref_binaries.missing("Fortran host not built (no amdflang)")
    g = _load_npz(os.path.join(golden_dir, "e2e_lp.npz"))
    wd = str(tmp_path)
    write_case_inputs(wd, g)
    open(os.path.join(wd, "Parfile.txt"), "w").write(str(g["parfile"]))
    out = _sub_run([EXE, "-p", "Parfile.txt"], cwd=wd, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "THE END." in out.stdout and "WAVELET_DOMAIN = F" in out.stdout, out.stdout[-3000:] + out.stderr[-2000:]
    model = read_tokens(os.path.join(wd, "out", "model", "grav_final_model_full.txt"), 1)[:, 0]
    ref = g["np1_model_final"]
    assert np.linalg.norm(model - ref) <= 1e-5 * np.linalg.norm(ref), np.linalg.norm(model - ref) / np.linalg.norm(ref)


def test_admm_local_bounds_parfile_matches_reference(tmp_path, golden_dir):
    if not os.path.isfile(EXE):
        ref_binaries.missing("Fortran host not built (no amdflang)")
    g = _load_npz(os.path.join(golden_dir, "e2e_admm_local.npz"))
    wd = str(tmp_path)
    write_case_inputs(wd, g)
    with open(os.path.join(wd, "bounds.txt"), "w") as f:
        f.write("%d 2\n" % g["bounds"].shape[0])
        for bnd, w in zip(g["bounds"], g["bound_weight"]):
            f.write("%.17g %.17g %.17g %.17g %.17g\n" % (bnd[0], bnd[1], bnd[2], bnd[3], w))
    open(os.path.join(wd, "Parfile.txt"), "w").write(str(g["parfile"]))
    out = _sub_run([EXE, "-p", "Parfile.txt"], cwd=wd, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "THE END." in out.stdout and "WAVELET_DOMAIN = F" in out.stdout, out.stdout[-3000:] + out.stderr[-2000:]
    model = read_tokens(os.path.join(wd, "out", "model", "grav_final_model_full.txt"), 1)[:, 0]
    ref = g["np1_model_final"]
    assert np.linalg.norm(model - ref) <= 1e-5 * np.linalg.norm(ref), np.linalg.norm(model - ref) / np.linalg.norm(ref)


def test_data_errors_parfile_matches_reference(tmp_path, golden_dir):
    if not os.path.isfile(EXE):
        ref_binaries.missing("Fortran host not built (no amdflang)")
    g = _load_npz(os.path.join(golden_dir, "e2e_err.npz"))
    wd = str(tmp_path)
    write_case_inputs(wd, g)
    with open(os.path.join(wd, "data_error.txt"), "w") as f:
        f.write("%d\n" % g["data_error"].size)
        f.write("\n".join("%.17g" % e for e in g["data_error"]) + "\n")
    open(os.path.join(wd, "Parfile.txt"), "w").write(str(g["parfile"]))
    out = _sub_run([EXE, "-p", "Parfile.txt"], cwd=wd, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "THE END." in out.stdout, out.stdout[-3000:] + out.stderr[-2000:]
    model = read_tokens(os.path.join(wd, "out", "model", "grav_final_model_full.txt"), 1)[:, 0]
    ref = g["np1_model_final"]
    assert np.linalg.norm(model - ref) <= 1e-6 * np.linalg.norm(ref), np.linalg.norm(model - ref) / np.linalg.norm(ref)
    # the SENSIT file holds the UNSCALED kernel, like the reference's
    import importlib
    sio = importlib.import_module("tomofast-x_amd").sensit_io
    got = sio.read_sensit(os.path.join(wd, "out", "SENSIT"), 1)
    same = np.intersect1d(got["cols"][:got["rowptr"][1]], g["np1_cols"][:g["np1_row_ptr"][1]]).size
    assert same >= 0.99 * g["np1_row_ptr"][1]
    r0 = slice(0, int(min(got["rowptr"][1], g["np1_row_ptr"][1])))
    common, ia, ib = np.intersect1d(got["cols"][:got["rowptr"][1]], g["np1_cols"][:g["np1_row_ptr"][1]], return_indices=True)
    assert np.allclose(got["vals"][ia], g["np1_vals"][ib], rtol=1e-6)


@pytest.mark.parametrize("name", ["e2e_localw", "e2e_localw_lp"])
def test_local_weights_parfile_matches_reference(tmp_path, golden_dir, name):
    if not os.path.isfile(EXE):
        ref_binaries.missing("Fortran host not built (no amdflang)")
    g = _load_npz(os.path.join(golden_dir, name + ".npz"))
    wd = str(tmp_path)
    write_case_inputs(wd, g)
    for fname, key in (("lw_depth.txt", "lw_depth"), ("lw_damp.txt", "lw_damp")):
        with open(os.path.join(wd, fname), "w") as f:
            f.write("%d\n" % g[key].size)
            f.write("\n".join("%.17g" % v for v in g[key]) + "\n")
    open(os.path.join(wd, "Parfile.txt"), "w").write(str(g["parfile"]))
    out = _sub_run([EXE, "-p", "Parfile.txt"], cwd=wd, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "THE END." in out.stdout and "WAVELET_DOMAIN = F" in out.stdout, out.stdout[-3000:] + out.stderr[-2000:]
    model = read_tokens(os.path.join(wd, "out", "model", "grav_final_model_full.txt"), 1)[:, 0]
    ref = g["np1_model_final"]
    # e2e_localw runs 400 iterations on 9 data rows (converged solution, sensitive at the 5e-6 level); e2e_localw_lp is the
    # well-conditioned 5-iteration case (oracle 1-ulp sensitivity 2e-14; the model file holds ~1e-16 relative precision)
    tol = 1e-5 if name == "e2e_localw" else 1e-9
    assert np.linalg.norm(model - ref) <= tol * np.linalg.norm(ref), np.linalg.norm(model - ref) / np.linalg.norm(ref)


def test_parfile_errors_like_the_reference(tmp_path):
    if not os.path.isfile(EXE):
        ref_binaries.missing("Fortran host not built (no amdflang)")
    out = _sub_run([EXE], cwd=str(tmp_path), capture_output=True, text=True, timeout=60)
    assert out.returncode != 0 and "UNKNOWN Parfile" in out.stdout
    open(os.path.join(str(tmp_path), "P.txt"), "w").write("inversion.joint.grav.problemWeight = 1.d0\nfoo.bar = 3\nmodelGrid.size = 2 2 2\n"
                                                           "forward.data.grav.nData = 3\nforward.depthWeighting.type = 4\n")
    out = _sub_run([EXE, "-p", "P.txt"], cwd=str(tmp_path), capture_output=True, text=True, timeout=60)
    assert out.returncode != 0 and "Unknown parameter name: foo.bar" in out.stdout and "Not known depth weight type!" in out.stdout     # weights_gravmag.f90:164
