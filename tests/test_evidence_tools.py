"""The reducers behind profiles/: tools/pmc_reduce.py must count only the newest PMC pass of a folder (gpurun merges the files of
every call into gpurun_out/, and dispatch ids repeat from run to run) and must refuse a reduction that is not ~1 x the device
bytes of the matrix (bench.py would otherwise report it as roofline.traffic)."""
import json
import os
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
HEAD = '"Dispatch_Id","Kernel_Name","Counter_Name","Counter_Value"\n'


def _pass(folder, name, counter, kib, launches=3):
    os.makedirs(folder, exist_ok=True)
    with open(os.path.join(folder, name), "w") as f:
        f.write(HEAD)
        for i in range(launches):
            f.write('%d,"void tfx::k_spmv_fwd<16>(...)","%s",%f\n' % (100 + 2 * i, counter, kib))
            f.write('%d,"void tfx::k_spmv_adj<16>(...)","%s",%f\n' % (101 + 2 * i, counter, kib))


def _bench_lines(src, device_bytes):
    line = json.dumps({"config": {"workload": "hamersley_1e7: test", "nnz": 123},
                       "roofline": {"algorithmic_bytes_per_launch": device_bytes, "device_bytes_of_the_matrix": device_bytes}})
    for name in ("pmc_FETCH_SIZE.json", "bench_plain.json"):
        open(os.path.join(src, name), "w").write(line + "\n")


def test_pmc_reduce_counts_only_the_newest_pass_and_refuses_nonsense(tmp_path):
    import pmc_reduce
    src = str(tmp_path)
    device_bytes = 2.0 * 1024.0 * 1000.0            # FETCH_SIZE 1000 KiB x 2 (gfx950 correction) = exactly the device bytes
    _bench_lines(src, device_bytes)
    _pass(os.path.join(src, "pmc_FETCH_SIZE", "runc"), "1_counter_collection.csv", "FETCH_SIZE", 5000.0)     # an older run: same dispatch ids
    _pass(os.path.join(src, "pmc_WRITE_SIZE", "runc"), "1_counter_collection.csv", "WRITE_SIZE", 7.0)
    time.sleep(0.05)
    _pass(os.path.join(src, "pmc_FETCH_SIZE", "runc"), "2_counter_collection.csv", "FETCH_SIZE", 1000.0)
    _pass(os.path.join(src, "pmc_WRITE_SIZE", "runc"), "2_counter_collection.csv", "WRITE_SIZE", 3.0)
    now = time.time()
    for n, t in (("1_counter_collection.csv", now - 100), ("2_counter_collection.csv", now)):
        for sub in ("pmc_FETCH_SIZE", "pmc_WRITE_SIZE"):
            os.utime(os.path.join(src, sub, "runc", n), (t, t))
    dst = os.path.join(src, "summary.json")
    pmc_reduce.main(src, dst, 3)
    d = json.load(open(dst))
    k = d["kernels"]["k_spmv_adj"]
    assert k["FETCH_SIZE_launches"] == 3 and k["FETCH_SIZE_raw_KiB_avg"] == 1000.0 and k["WRITE_SIZE_raw_KiB_avg"] == 3.0
    assert abs(k["fetch_over_device_bytes"] - 1.0) < 1e-12 and d["workload"] == "hamersley_1e7"
    # a pass that reads 5 x the matrix is a bookkeeping error: nothing is written
    os.remove(dst)
    t = time.time() + 10
    os.utime(os.path.join(src, "pmc_FETCH_SIZE", "runc", "1_counter_collection.csv"), (t, t))
    with pytest.raises(SystemExit):
        pmc_reduce.main(src, dst, 3)
    assert not os.path.exists(dst)


def test_pmc_reduce_tells_the_two_forward_launches_apart_by_grid_size(tmp_path):
    """With the transposed copy the adjoint product is k_spmv_fwd on S^T: the launches differ in grid size (the two work lists), the
    first launch of a run is a forward product on S; a launch streams one of the two copies."""
    import pmc_reduce
    src = str(tmp_path)
    one_copy = 2.0 * 1024.0 * 1000.0
    _bench_lines(src, 2.0 * one_copy)                  # the device bytes count both copies
    head = '"Dispatch_Id","Grid_Size","Kernel_Name","Counter_Name","Counter_Value"\n'
    for sub, counter, kib_s, kib_t in (("pmc_FETCH_SIZE", "FETCH_SIZE", 1000.0, 1010.0), ("pmc_WRITE_SIZE", "WRITE_SIZE", 2.0, 5.0)):
        os.makedirs(os.path.join(src, sub, "runc"))
        with open(os.path.join(src, sub, "runc", "1_counter_collection.csv"), "w") as f:
            f.write(head)
            for i in range(4):
                f.write('%d,4194304,"void tfx::k_spmv_fwd<16>(...)","%s",%f\n' % (10 + 2 * i, counter, kib_s))
                f.write('%d,5000192,"void tfx::k_spmv_fwd<16>(...)","%s",%f\n' % (11 + 2 * i, counter, kib_t))
    dst = os.path.join(src, "summary.json")
    pmc_reduce.main(src, dst, 4)
    d = json.load(open(dst))
    assert d["copies_of_the_tiles"] == 2 and set(d["kernels"]) == {"k_spmv_fwd", "k_spmv_fwd_on_copy"}
    assert d["kernels"]["k_spmv_fwd"]["FETCH_SIZE_raw_KiB_avg"] == 1000.0 and d["kernels"]["k_spmv_fwd_on_copy"]["FETCH_SIZE_raw_KiB_avg"] == 1010.0
    assert abs(d["kernels"]["k_spmv_fwd"]["fetch_over_device_bytes"] - 1.0) < 1e-12
    assert d["kernels"]["k_spmv_fwd_on_copy"]["WRITE_SIZE_raw_KiB_avg"] == 5.0


def test_bench_gpus_n_without_a_launcher_starts_its_own_ranks(monkeypatch):
    """`python bench.py --gpus 4` with no WORLD_SIZE in the environment re-runs itself under torch.distributed.run on 127.0.0.1 at a
    free port with the same arguments, and exits with that run's status - it must never die on a missing launcher."""
    import subprocess
    sys.path.insert(0, ROOT)
    import bench
    seen = {}

    def fake_call(cmd, *a, **k):
        seen["cmd"] = cmd
        return 7
    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "5", "--workload", "small"])
    with pytest.raises(SystemExit) as e:
        bench.self_launch(4)
    assert e.value.code == 7
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"] and "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and 1024 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    assert cmd[-6:] == ["--gpus", "4", "--steps", "5", "--workload", "small"] and cmd[-7].endswith("bench.py")
    # main() takes that route exactly when WORLD_SIZE is absent and N > 1
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    called = []
    monkeypatch.setattr(bench, "self_launch", lambda n: called.append(n) or (_ for _ in ()).throw(SystemExit(0)))
    with pytest.raises(SystemExit):
        bench.main()
    assert called == [4]


def test_missing_reference_binary_fails_where_it_was_expected(tmp_path, monkeypatch):
    """tests/ref_binaries.py: a `-m gpu` test that does not find a reference binary / Fortran host skips on a machine that never had
    it and FAILS where build() recorded it (oracle/ref_expected.json) or TFX_EXPECT_REFERENCE_BINARIES=1 says so."""
    import pytest
    import ref_binaries as rb
    monkeypatch.setattr(rb, "MARKER", str(tmp_path / "ref_expected.json"))
    monkeypatch.delenv("TFX_EXPECT_REFERENCE_BINARIES", raising=False)
    with pytest.raises(pytest.skip.Exception):
        rb.missing("not built")                                          # no marker, no variable: skip
    (tmp_path / "ref_expected.json").write_text('{"binaries": ["oracle/_ref/tomofastx"], "mpiexec": true}')
    with pytest.raises(pytest.fail.Exception):
        rb.missing("not built")                                          # build() saw the binaries: a missing one fails
    with pytest.raises(pytest.skip.Exception):
        rb.missing("not built", "oracle/_ref/never_recorded")            # a binary build() did not record is not expected
    with pytest.raises(pytest.fail.Exception):
        rb.missing("not built", "oracle/_ref/tomofastx_not_here", "oracle/_ref/tomofastx") if not os.path.isfile(os.path.join(rb.ROOT, "oracle/_ref/tomofastx")) else rb.missing("not built")
    monkeypatch.setenv("TFX_EXPECT_REFERENCE_BINARIES", "0")
    with pytest.raises(pytest.skip.Exception):
        rb.missing("not built")                                          # explicit opt-out
    monkeypatch.setenv("TFX_EXPECT_REFERENCE_BINARIES", "1")
    (tmp_path / "ref_expected.json").unlink()
    with pytest.raises(pytest.fail.Exception):
        rb.missing("not built")


def test_no_plain_skip_left_in_the_gpu_tests():
    """Every skip site of the `-m gpu` files goes through ref_binaries.missing."""
    import glob
    import os
    here = os.path.dirname(os.path.abspath(__file__))
    for f in sorted(glob.glob(os.path.join(here, "test_gpu_*.py"))):
        txt = open(f).read()
        assert "pytest.skip(" not in txt and "skipif" not in txt and "importorskip" not in txt, f


def test_parity_report_files_and_their_reduction(tmp_path, monkeypatch):
    """tests/parity_report.py writes one file per pytest process (gpurun merges gpurun_out/ file by file: a fixed name is overwritten by
    the next call - it happened in round 6); tools/reduce_parity_report.py keeps the newest record of every test."""
    import subprocess
    import parity_report as pr
    monkeypatch.setattr(pr, "ROOT", str(tmp_path))
    monkeypatch.delenv("TFX_PARITY_REPORT", raising=False)
    pr.report("full_size[config5_hamersley_d4]", rows_checked=13, lsqr_r_rel_err=4e-16)
    pr.report("config1_mansf_end_to_end[python host]", model_rel_l2=1.7e-10)
    folder = os.path.join(str(tmp_path), "gpurun_out", "parity_report")
    files = os.listdir(folder)
    assert len(files) == 1 and files[0].startswith("run_") and str(os.getpid()) in files[0]
    later = {"test": "full_size[config5_hamersley_d4]", "t": time.time() + 100, "rows_checked": 13, "lsqr_r_rel_err": 5e-16}
    open(os.path.join(folder, "run_9999999999_1.jsonl"), "w").write(json.dumps(later) + "\n")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "reduce_parity_report.py"), "--folder", folder, "--note", "n"],
                       capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    out = json.loads(p.stdout)
    assert out["records"] == 2 and out["note"] == "n"
    assert out["by_test"]["full_size"]["full_size[config5_hamersley_d4]"]["lsqr_r_rel_err"] == 5e-16
    assert out["headline_numbers"]["config 1 (8192 cells, 60 x 100 iterations + ADMM), final model rel-L2 vs the reference"] == {"python host": 1.7e-10}
