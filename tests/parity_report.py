"""Measured parity distances, kept: every `-m gpu` test that measures a distance to the oracle / the reference calls report(); the lines
accumulate in gpurun_out/parity_report/run_<start>_<pid>.jsonl (merged back from the GPU box by gpurun) and tools/reduce_parity_report.py
reduces them (newest record per test) to
the table committed as profiles/rNN_parity.json - the numbers DESIGN.md quotes come from there, not from a scrolled-away test log."""
import json
import os
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


_SESSION = "%d_%d" % (int(time.time()), os.getpid())


def report_dir():
    return os.path.join(ROOT, "gpurun_out", "parity_report")


def report_path():
    """One file per pytest process (gpurun merges gpurun_out/ file by file: a fixed name would be overwritten by the next call)."""
    return os.environ.get("TFX_PARITY_REPORT") or os.path.join(report_dir(), "run_%s.jsonl" % _SESSION)


def _plain(v):
    try:
        import numpy as np
        if isinstance(v, np.generic):
            return v.item()
        if isinstance(v, np.ndarray):
            return v.tolist()
    except ImportError:
        pass
    if isinstance(v, (list, tuple)):
        return [_plain(x) for x in v]
    if isinstance(v, dict):
        return {str(k): _plain(x) for k, x in v.items()}
    return v


def report(test, **numbers):
    """Appends {"test": ..., "t": ..., **numbers} to the report; never fails a test (a read-only tree just loses the line)."""
    rec = {"test": test, "t": round(time.time(), 1)}
    rec.update({k: _plain(v) for k, v in numbers.items()})
    try:
        path = report_path()
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "a") as f:
            f.write(json.dumps(rec) + "\n")
    except OSError:
        pass
    return rec
