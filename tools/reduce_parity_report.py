#!/usr/bin/env python3
"""Reduces the measured parity distances the `-m gpu` tests leave in gpurun_out/parity_report/*.jsonl (tests/parity_report.py) to the
table committed as profiles/rNN_parity.json: the newest record of every test, grouped, plus the handful of headline numbers DESIGN.md
section 4 quotes ("End-to-end tolerance by scale").  Optionally folds in the parity legs of a bench line (reference_medium).
  python tools/reduce_parity_report.py [--bench profiles/r06_bench_hamersley_1e7.json] > profiles/r06_parity.json"""
import argparse
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_records(folder):
    recs = {}
    files = sorted(glob.glob(os.path.join(folder, "*.jsonl")))
    for f in files:
        for ln in open(f):
            ln = ln.strip()
            if not ln:
                continue
            try:
                r = json.loads(ln)
            except ValueError:
                continue
            if "test" in r and (r["test"] not in recs or r.get("t", 0) >= recs[r["test"]].get("t", 0)):
                recs[r["test"]] = r
    return recs, files


def pick(recs, name, key):
    r = recs.get(name)
    return None if r is None else r.get(key)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--folder", default=os.path.join(ROOT, "gpurun_out", "parity_report"))
    ap.add_argument("--note", default=None, help="free text stored with the table (e.g. which runs it was reduced from)")
    ap.add_argument("--bench", default=None, help="a bench.py JSON line (file) whose cpu_baseline.reference_medium legs are folded in")
    args = ap.parse_args()
    recs, files = load_records(args.folder)
    if not recs:
        sys.exit("no records under %s" % args.folder)
    groups = {}
    for name in sorted(recs):
        r = dict(recs[name])
        r.pop("t", None)
        r.pop("test", None)
        groups.setdefault(name.split("[")[0], {})[name] = r
    head = {
        "config 1 (8192 cells, 60 x 100 iterations + ADMM), final model rel-L2 vs the reference": {
            k.split("[")[1].rstrip("]"): recs[k].get("model_rel_l2") for k in recs if k.startswith("config1_mansf_end_to_end[")},
        "64x64x32 x 1024 (2 x 100 iterations), final model rel-L2 vs the reference's 1-rank run": {
            k.split("[")[1].rstrip("]"): {"gpu": recs[k].get("model_rel_l2_vs_reference_np1"), "reference_own_rank_scatter": recs[k].get("reference_own_scatter_np2_4_8_vs_np1")}
            for k in recs if k.startswith("medium_scale_end_to_end[")},
        "unconverged 101-iteration solve, distance to the 80-bit trajectory (HIP vs sequential fp64)": {
            k.split("[")[1].rstrip("]"): {"gpu": recs[k].get("model_gpu_vs_ext80"), "sequential_fp64": recs[k].get("model_seq64_vs_ext80"),
                                          "r": [recs[k].get("r_seq64"), recs[k].get("r_gpu"), recs[k].get("r_ext80")]}
            for k in recs if k.startswith("unconverged_lsqr_vs_extended_precision[")},
        "Hamersley field data, final model rel-L2 vs the reference's 1-rank run [its own 1- vs 2-rank]": {
            k.split("[", 1)[1].rstrip("]"): [recs[k].get("model_rel_l2"), recs[k].get("reference_own_1_vs_2_ranks")]
            for k in recs if k.startswith(("hamersley[", "hamersley_xgrad_15x100["))},
        "Hamersley joint example, converged first solve": {
            k.split("[", 1)[1].rstrip("]"): {kk: vv for kk, vv in recs[k].items() if kk not in ("t", "test")}
            for k in recs if k.startswith(("hamersley_xgrad_converged[", "hamersley_xgrad_first_solve["))},
        "full size (the sizes the bench times): rows vs the oracle, LSQR residual vs the products": {
            k.split("[")[1].rstrip("]"): {kk: recs[k].get(kk) for kk in ("rows_checked", "worst_value_distance_over_row_scale", "fp32_identical_fraction", "worst_fp32_ulp",
                                                                          "adjoint_identity", "lsqr_r_rel_err", "lsqr_bits_identical", "lsqr_r5", "lsqr_r10",
                                                                          "lsqr_grad0", "lsqr_grad5", "lsqr_grad10")}
            for k in recs if k.startswith("full_size[")},
        "graviprism_full rows vs the reference (worst |difference| / row maximum)": pick(recs, "graviprism_full_rows_vs_reference", "worst_row_scale_distance"),
    }
    out = {"what": "measured parity distances of the `-m gpu` suite on an MI355X, newest record per test (tests/parity_report.py; tools/reduce_parity_report.py)",
           "records": len(recs), "source_files": [os.path.basename(f) for f in files], "headline_numbers": head, "by_test": groups}
    if args.note:
        out["note"] = args.note
    if args.bench:
        try:
            line = [ln for ln in open(args.bench) if ln.lstrip().startswith("{")][-1]
            b = json.loads(line)
            b = b.get("parsed", b)
            rm = (b.get("cpu_baseline") or {}).get("reference_medium")
            if rm:
                out["headline_numbers"]["128x128x32 x 1024 live in the bench line (converged 3 x 100)"] = {
                    "model_rel_l2": rm.get("model_rel_l2"), "data_cost": rm.get("data_cost"),
                    "on_the_reference_kernel_files": (rm.get("gpu_host_on_the_reference_kernel") or {}).get("model_rel_l2"),
                    "reference_own_scatter": (rm.get("reference_own_scatter_between_rank_counts") or {}).get("model_rel_l2"),
                    "timing_leg_1x101": (rm.get("timing_leg_1x101") or {}).get("model_rel_l2")}
        except Exception as e:      # noqa
            out["bench_fold_in_error"] = repr(e)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
