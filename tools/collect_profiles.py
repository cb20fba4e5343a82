"""Copies the summaries of a `tools/r2_final.sh` / `tools/profile_round.sh` run from gpurun_out/ into profiles/ under the round's
names (run in the repo after the gpurun call returned):  python tools/collect_profiles.py <round number>"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "profile_round")
FIN = os.path.join(ROOT, "gpurun_out", "r2_final")
DST = os.path.join(ROOT, "profiles")


def newest(pattern):
    f = glob.glob(pattern, recursive=True)
    return max(f, key=os.path.getmtime) if f else None


def last_json_line(path, dst):
    line = open(path).read().strip().splitlines()[-1]
    json.loads(line)
    open(dst, "w").write(line + "\n")


def main(rnd):
    tag = "r%02d" % rnd
    last_json_line(os.path.join(SRC, "bench_plain.json"), os.path.join(DST, tag + "_bench_hamersley_1e7.json"))
    last_json_line(os.path.join(SRC, "bench_profiled.json"), os.path.join(DST, tag + "_bench_hamersley_1e7_profiled.json"))
    shutil.copy(newest(os.path.join(SRC, "stats", "**", "*kernel_stats.csv")), os.path.join(DST, tag + "_bench_hamersley_1e7_kernel_stats.csv"))
    shutil.copy(newest(os.path.join(SRC, "stats", "**", "*domain_stats.csv")), os.path.join(DST, tag + "_bench_hamersley_1e7_domain_stats.csv"))
    for sub, name in (("pmc_FETCH_SIZE", "pmc_FETCH_SIZE_k_spmv"), ("pmc_WRITE_SIZE", "pmc_WRITE_SIZE_k_spmv"), ("pmc_SQ1", "pmc_SQ_pass1_k_spmv"),
                      ("pmc_SQ2", "pmc_SQ_pass2_k_spmv")):
        shutil.copy(newest(os.path.join(SRC, sub, "**", "*counter_collection.csv")), os.path.join(DST, "%s_%s.csv" % (tag, name)))
    # build kernels: thousands of dispatches -> sums per kernel and counter
    agg, rows = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(newest(os.path.join(SRC, "pmc_build", "**", "*counter_collection.csv")))):
        k = "prism" if "k_prism_gz_tensor" in r["Kernel_Name"] else "wavelet" if "k_wavelet_axis" in r["Kernel_Name"] else None
        if k:
            agg[(k, r["Counter_Name"])] += float(r["Counter_Value"])
            rows[(k, r["Counter_Name"])] += 1
    with open(os.path.join(DST, tag + "_pmc_SQ_build_kernels_medium.csv"), "w") as f:
        f.write("kernel,counter,dispatch_counter_rows,sum,avg_per_row\n")
        for key in sorted(agg):
            f.write("%s,%s,%d,%g,%g\n" % (key[0], key[1], rows[key], agg[key], agg[key] / rows[key]))
    last_json_line(os.path.join(SRC, "rowgen.json"), os.path.join(DST, tag + "_rowgen.json"))
    shutil.copy(newest(os.path.join(SRC, "rowgen", "**", "*kernel_stats.csv")), os.path.join(DST, tag + "_rowgen_kernel_stats.csv"))
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import pmc_reduce
    pmc_reduce.main(SRC, os.path.join(DST, tag + "_pmc_summary.json"), rnd)
    # the same workload without the transposed copy (TFX_ADJ_COPY=0: the adjoint on the tiles of S, k_spmv_adj)
    if os.path.isfile(os.path.join(SRC, "bench_plain_nocopy.json")):
        last_json_line(os.path.join(SRC, "bench_plain_nocopy.json"), os.path.join(DST, tag + "_bench_hamersley_1e7_nocopy.json"))
        for sub, name in (("pmc0_FETCH_SIZE", "nocopy_pmc_FETCH_SIZE_k_spmv"), ("pmc0_WRITE_SIZE", "nocopy_pmc_WRITE_SIZE_k_spmv"),
                          ("pmc0_SQ1", "nocopy_pmc_SQ_pass1_k_spmv"), ("pmc0_SQ2", "nocopy_pmc_SQ_pass2_k_spmv")):
            f = newest(os.path.join(SRC, sub, "**", "*counter_collection.csv"))
            if f:
                shutil.copy(f, os.path.join(DST, "%s_%s.csv" % (tag, name)))
        pmc_reduce.main(SRC, os.path.join(DST, tag + "_nocopy_pmc_summary.json"), rnd, "pmc0", "bench_plain_nocopy.json")
    if os.path.isdir(FIN):
        for src, dst in (("gpu_tests.log", tag + "_gpu_tests.log"), ("fuzz.log", tag + "_fuzz.log")):
            if os.path.isfile(os.path.join(FIN, src)):
                shutil.copy(os.path.join(FIN, src), os.path.join(DST, dst))
        for w in ("haar_512", "dense_256", "medium", "small"):
            p = os.path.join(FIN, "bench_%s.json" % w)
            if os.path.isfile(p):
                last_json_line(p, os.path.join(DST, "%s_bench_%s.json" % (tag, w)))


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 2)
