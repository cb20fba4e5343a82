"""Reduces the two rocprofv3 PMC passes of tools/profile_round.sh (FETCH_SIZE, WRITE_SIZE on the k_spmv kernels) to HBM bytes per
launch -> profiles/rNN_pmc_summary.json, which bench.py reads for roofline.traffic.  Corrections: MI355X_MICROARCH.md, HBM section
(gfx950: FETCH_SIZE counts half of a wide coalesced streaming read -> x2; both counters are in KiB)."""
import csv
import glob
import json
import os
import sys


def per_kernel(folder, counter):
    out, grids = {}, {}
    # gpurun merges every call's files into gpurun_out/: only the newest pass counts (dispatch ids repeat from run to run)
    files = glob.glob(os.path.join(folder, "**", "*counter_collection.csv"), recursive=True)
    for f in sorted(files, key=os.path.getmtime)[-1:]:
        rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Dispatch_Id"]))
        for row in rows:
            if row.get("Counter_Name") != counter:
                continue
            name = row["Kernel_Name"]
            key = "k_spmv_fwd" if "k_spmv_fwd" in name else "k_spmv_adj" if "k_spmv_adj" in name else None
            if key is None:
                continue
            if key == "k_spmv_fwd":
                # with the transposed copy the adjoint product is the same kernel on S^T: the two work lists have different lengths, so
                # the grid size tells the launches apart; the first k_spmv_fwd launch of a bench run is a forward product on S
                g = row.get("Grid_Size") or row.get("Grid_Size_X") or "?"
                grids.setdefault(g, len(grids))
                if grids[g] > 0:
                    key = "k_spmv_fwd_on_copy"
            disp = row["Dispatch_Id"]
            out.setdefault(key, {}).setdefault(disp, 0.0)
            out[key][disp] += float(row["Counter_Value"])
    return {k: (sum(v.values()) / len(v), len(v)) for k, v in out.items()}


def main(src, dst, rnd, prefix="pmc", plain_name="bench_plain.json"):
    fetch = per_kernel(os.path.join(src, prefix + "_FETCH_SIZE"), "FETCH_SIZE")
    write = per_kernel(os.path.join(src, prefix + "_WRITE_SIZE"), "WRITE_SIZE")
    bench = json.loads(open(os.path.join(src, prefix + "_FETCH_SIZE.json")).read().strip().splitlines()[-1])
    res = {"round": rnd, "workload": bench["config"]["workload"].split(":")[0],
           "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-include-regex k_spmv), "
                     "bench.py --steps 3 --warmup 1 --no-cpu --no-profile (tools/profile_round.sh)",
           "correction": "MI355X_MICROARCH.md HBM: on gfx950 FETCH_SIZE reports exactly half of the bytes of a wide (16 B/lane) "
                         "coalesced streaming read -> doubled; FETCH_SIZE/WRITE_SIZE are in KiB (x1024). WRITE_SIZE is uncalibrated (small here). "
                         "The product kernels read 71 % of their bytes with 16 B/lane loads and 27 % with 12 B/lane (dwordx3) loads; the x2 "
                         "factor is calibrated on this access mix by the device bytes of the matrix (every stored byte is read exactly once "
                         "per launch): 2 x FETCH_SIZE / device bytes is reported as `fetch_over_device_bytes` and should be ~1.",
           "kernels": {}, "nnz": bench["config"]["nnz"]}
    plain = os.path.join(src, plain_name)
    if os.path.isfile(plain):
        roof = json.loads(open(plain).read().strip().splitlines()[-1])["roofline"]
        res["algorithmic_bytes"] = roof.get("algorithmic_bytes_per_launch", roof.get("stored_bytes_per_launch"))
        res["device_bytes_of_the_matrix"] = roof.get("device_bytes_of_the_matrix")
    ncopies = 2.0 if "k_spmv_fwd_on_copy" in fetch else 1.0
    res["copies_of_the_tiles"] = int(ncopies)
    for k in fetch:
        rd = fetch[k][0] * 1024.0 * 2.0
        wr = write.get(k, (0.0, 0))[0] * 1024.0
        res["kernels"][k] = {"FETCH_SIZE_raw_KiB_avg": fetch[k][0], "FETCH_SIZE_launches": fetch[k][1],
                             "WRITE_SIZE_raw_KiB_avg": write.get(k, (0.0, 0))[0], "WRITE_SIZE_launches": write.get(k, (0.0, 0))[1],
                             "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr, "traffic_bytes_per_launch": rd + wr,
                             # (a launch streams ONE copy of the tiles: with the transposed copy the device bytes count both)
                             "fetch_over_device_bytes": (rd / (res["device_bytes_of_the_matrix"] / ncopies)) if res.get("device_bytes_of_the_matrix") else None}
    # every stored byte of the matrix is read exactly once per launch: a reduction that is not within a few per cent of that is a
    # bookkeeping error (e.g. passes of several runs summed), not traffic - do not write it where bench.py would pick it up
    for k, v in res["kernels"].items():
        r = v["fetch_over_device_bytes"]
        if r is not None and not 0.9 < r < 1.2:
            raise SystemExit("pmc_reduce: %s reads %.3f x the device bytes of the matrix - not written" % (k, r))
    json.dump(res, open(dst, "w"), indent=1)
    print(json.dumps(res["kernels"], indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 1, *(sys.argv[4:6]))
