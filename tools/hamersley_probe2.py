#!/usr/bin/env python3
"""(oracle/_ref/hamersley_xgrad_SENSIT: `KEEP_SENSIT=1 python tests/golden/make_golden.py hamersley_conv` in the development container.)
xgrad Hamersley example, two discriminating runs of the shipping host: (1) ONE major iteration with 100 / 400 / 1600 LSQR iterations (the
reference: r = 0.014977706, 0.0048015005, 0.0047371974 - the residual is still falling fast at iteration 100) and (2) the full run on the
REFERENCE'S OWN kernel files (oracle/_ref/hamersley_xgrad_SENSIT, sensit.readFromFiles = 1: identical matrix bits)."""
import os, re, sys, subprocess, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_dropin as td
import test_gpu_fortran_host as fh
g = fh._load_npz(os.path.join(ROOT, "tests", "golden", "hamersley.npz"))
par = str(g["xgrad_parfile"])
def run(p, env=None):
    wd = tempfile.mkdtemp()
    td._write_hamersley_inputs(wd, g)
    open(os.path.join(wd, "Parfile.txt"), "w").write(p)
    out = subprocess.run([fh.EXE, "-p", "Parfile.txt"], cwd=wd, capture_output=True, text=True, timeout=900, env=dict(os.environ, **(env or {})))
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    r = [float(m.group(1)) for m in re.finditer(r"End of subroutine lsqr_solve_sensit, r =\s*([0-9.eE+-]+)", out.stdout)]
    od = os.path.join(wd, str(g["xgrad_outdir"]))
    return r, {t: fh.read_tokens(os.path.join(od, "model", t + "_final_model_full.txt"), 1)[:, 0] for t in ("grav", "mag")}
ref = {100: 0.01497770582171863, 400: 0.0048015004803665185, 1600: 0.004737197449822196}
for nminor in (100, 400, 1600):
    p = re.sub(r"inversion.nMajorIterations\s*=\s*\d+", "inversion.nMajorIterations          = 1", par)
    p = re.sub(r"inversion.nMinorIterations\s*=\s*\d+", "inversion.nMinorIterations          = %d" % nminor, p)
    r, _ = run(p)
    print("1 x %4d iterations: r = %.12e   reference %.12e   relative difference %.1e" % (nminor, r[0], ref[nminor], abs(r[0] - ref[nminor]) / ref[nminor]), flush=True)
sd = os.path.join(ROOT, "oracle", "_ref", "hamersley_xgrad_SENSIT")
if os.path.isdir(sd):
    p = par.replace("sensit.readFromFiles                = 0", "sensit.readFromFiles                = 1")
    p = re.sub(r"sensit.folderPath\s*=\s*\S+", "sensit.folderPath                   = " + sd + "/", p)
    r, m = run(p)
    r1 = g["xgrad_np1_lsqr_r"]
    print("on the reference's kernel files: r relative differences per major iteration:", " ".join("%.1e" % (abs(a - b) / b) for a, b in zip(r, r1)))
    for t in ("grav", "mag"):
        ref1 = g["xgrad_np1_%s_model_final" % t]
        print("   %s final model rel-L2 vs reference np1: %.2e" % (t, np.linalg.norm(m[t] - ref1) / np.linalg.norm(ref1)))
