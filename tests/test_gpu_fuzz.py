"""A short fixed-seed run of every randomised sweep under tools/ (HIP path vs the CPU oracle on random inputs).  The long runs
(hundreds of cases, other seeds) are `bash tools/run_fuzz.sh`."""
import os
import subprocess
import sys

import pytest
import ref_binaries

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("script,args", [("fuzz_matrix.py", ["15", "901"]), ("fuzz_build.py", ["15", "902"]), ("fuzz_build_comp.py", ["12", "903"]),
                                         ("fuzz_lsqr_wavelet.py", ["12", "904"]), ("fuzz_band.py", ["6", "905"]), ("fuzz_misc.py", ["15", "906"]),
                                         ("fuzz_hosts.py", ["6", "907"])])
def test_randomised_sweep(script, args):
    if script == "fuzz_hosts.py" and not os.path.isfile(os.path.join(ROOT, "tomofast-x_amd", "host", "tomofastx_amd")):
        ref_binaries.missing("Fortran host not built (no amdflang)")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", script)] + args, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "OK" in out.stdout.splitlines()[-1], out.stdout[-2000:] + out.stderr[-3000:]
