"""csrc/fastmath.h (the fp64 log / atan2 of the prism kernels): host build against the long double libm.

The header compiles for the host as well as for the device (same arithmetic except the starting estimate of the reciprocal);
tests/fastmath_check.cpp draws arguments the way the prism kernels form them (gravity_field.f90:165-186) and reports the largest
error.  The device build itself is compared with the host libm in tests/test_gpu_parity.py::test_fastmath_device.
"""
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def test_fastmath_host_accuracy(tmp_path):
    exe = str(tmp_path / "fmcheck")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-I", os.path.join(ROOT, "tomofast-x_amd", "csrc"),
                           os.path.join(HERE, "fastmath_check.cpp"), "-o", exe])
    out = subprocess.check_output([exe, "1500000"], text=True)
    sys.stdout.write(out)
    m = re.search(r"log_ulp ([\d.]+).*atan2_ulp ([\d.]+).*atan2_abs_ulp ([\d.]+) special_bad (\d+)", out)
    assert m, out
    log_ulp, atan_ulp, atan_abs, bad = float(m.group(1)), float(m.group(2)), float(m.group(3)), int(m.group(4))
    # log: error against 1 ulp of max(|log x|, 1) (the kernels multiply it by a coordinate: its absolute error is what counts);
    # atan2: ulps of the result; the largest ones sit at results near 1/128 (table node 1 minus a correction of half its size)
    assert log_ulp <= 0.6
    assert atan_ulp <= 1.2          # (rounds 2-4: 1.57 - two reflections with a rounding each and an uncompensated denominator)
    assert atan_abs <= 0.55
    assert bad == 0


def test_tables_are_reproducible(tmp_path):
    """math_tables.h is exactly what tools/gen_math_tables.py writes."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("gen_math_tables", os.path.join(ROOT, "tools", "gen_math_tables.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    path = os.path.join(ROOT, "tomofast-x_amd", "csrc", "math_tables.h")
    before = open(path).read()
    mod.main()
    assert open(path).read() == before
