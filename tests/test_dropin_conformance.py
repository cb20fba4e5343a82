"""Compiler-checked conformance of the drop-in boundary (development container only: needs /root/reference and amdflang).
oracle/dropin_build.sh --check compiles the UNMODIFIED reference callers of the hot path - problem_joint_gravmag.F90,
joint_inverse_problem.F90, model.F90, model_IO.F90, inversion_arrays.f90, wavelet_utils.F90, damping.F90, admm_method.F90,
cross_gradient.F90, clustering.F90, damping_gradient.F90, the five unit-test files and program_tomofastx.F90 - against this
repository's drop-in modules sparse_matrix, lsqr_solver, wavelet_transform, sensitivity_gravmag and weights_gravmag
(tomofast-x_amd/host/dropin/).  A name, argument order, kind or type that differs from the reference's is a compile error here."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not (os.path.isdir("/root/reference/src") and os.path.isfile("/opt/rocm/bin/amdflang")),
                    reason="needs the reference sources and amdflang (development container)")
def test_unmodified_reference_callers_compile_against_the_dropin_modules():
    out = subprocess.run(["bash", os.path.join(ROOT, "oracle", "dropin_build.sh"), "--check"], capture_output=True, text=True, timeout=1200)
    assert out.returncode == 0, (out.stdout[-2000:] + out.stderr[-4000:])
    assert "drop-in conformance: the unmodified reference callers compile against the drop-in modules" in out.stdout
    objs = os.listdir(os.path.join(ROOT, "oracle", "_ref", "dropin", "build"))
    for must in ("problem_joint_gravmag.o", "joint_inverse_problem.o", "model.o", "damping.o", "admm_method.o", "cross_gradient.o", "clustering.o",
                 "damping_gradient.o", "tests_lsqr.o", "tests_sparse_matrix.o", "tests_wavelet_compression.o", "program_tomofastx.o",
                 "dropin_sparse_matrix.o", "dropin_lsqr_solver.o", "dropin_wavelet_transform.o", "dropin_sensitivity_gravmag.o",
                 "dropin_weights_gravmag.o"):
        assert must in objs, must
    # the swapped reference modules must NOT have been compiled into the build
    for never in ("sparse_matrix.o", "lsqr_solver2.o", "wavelet_transform.o", "sensitivity_gravmag.o", "weights_gravmag.o", "gravity_field.o",
                  "magnetic_field.o"):
        assert never not in objs, never


def test_dropin_sources_do_not_reach_the_oracle():
    for f in os.listdir(os.path.join(ROOT, "tomofast-x_amd", "host", "dropin")):
        txt = open(os.path.join(ROOT, "tomofast-x_amd", "host", "dropin", f)).read()
        assert "tfx_oracle" not in txt and "orc_" not in txt
