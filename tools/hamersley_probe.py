#!/usr/bin/env python3
"""The Hamersley field-data examples through the shipping host under several arithmetic variants of the products (run length of the forward
kernel, adjoint with / without the transposed copy, merged / separate tail launches of the LSQR iteration): per-major-iteration LSQR r against the reference's, final models
against the reference's 1-rank run and against each other - how much of the distance to the reference is the example's own sensitivity.
  python tools/hamersley_probe.py [grav|magn|xgrad]  -> text"""
import os, re, sys, time, subprocess, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_dropin as td
import test_gpu_fortran_host as fh
case = sys.argv[1] if len(sys.argv) > 1 else "xgrad"
g = fh._load_npz(os.path.join(ROOT, "tests", "golden", "hamersley.npz"))
tags = ("grav",) if case == "grav" else ("mag",) if case == "magn" else ("grav", "mag")
variants = [("default", {}), ("fwd_run=1", {"TFX_FWD_RUN": "1"}), ("fwd_run=4", {"TFX_FWD_RUN": "4"}), ("no adjoint copy", {"TFX_ADJ_COPY": "0"}),
            ("separate tail launches", {"TFX_LSQR_MERGE_TAIL": "0"})]
if os.environ.get("PROBE_QUICK"):
    variants = variants[:1]
models, rs = {}, {}
for name, env in variants:
    wd = tempfile.mkdtemp()
    td._write_hamersley_inputs(wd, g)
    open(os.path.join(wd, "Parfile.txt"), "w").write(str(g[case + "_parfile"]))
    t0 = time.time()
    out = subprocess.run([fh.EXE, "-p", "Parfile.txt"], cwd=wd, capture_output=True, text=True, timeout=900, env=dict(os.environ, **env))
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    od = os.path.join(wd, str(g[case + "_outdir"]))
    models[name] = {t: fh.read_tokens(os.path.join(od, "model", t + "_final_model_full.txt"), 1)[:, 0] for t in tags}
    rs[name] = np.array([float(m.group(1)) for m in re.finditer(r"End of subroutine lsqr_solve_sensit, r =\s*([0-9.eE+-]+)", out.stdout)])
    print("%-20s wall %.1f s" % (name, time.time() - t0), flush=True)
ref_r = {k: g["%s_np%d_lsqr_r" % (case, k)] for k in (1, 2)}
print("\nLSQR r per major iteration: reference np1 | relative difference of reference np2 | of each variant")
n = len(ref_r[1])
for i in range(n):
    row = ["%2d %.12e" % (i + 1, ref_r[1][i]), "%9.1e" % (abs(ref_r[2][i] - ref_r[1][i]) / ref_r[1][i])]
    row += ["%9.1e" % (abs(rs[name][i] - ref_r[1][i]) / ref_r[1][i]) for name, _ in variants]
    print("  ".join(row))
def rel(a, b):
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))
print("\nfinal model rel-L2 distances")
for t in tags:
    ref1, ref2 = g["%s_np1_%s_model_final" % (case, t)], g["%s_np2_%s_model_final" % (case, t)]
    print(t, "reference np2 vs np1: %.2e" % rel(ref2, ref1))
    for name, _ in variants:
        print("   %-20s vs reference np1 %.2e   vs reference np2 %.2e   vs default %.2e" % (name, rel(models[name][t], ref1), rel(models[name][t], ref2),
                                                                                         rel(models[name][t], models["default"][t])))
