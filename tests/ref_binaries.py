"""Reference binaries and Fortran hosts the `-m gpu` tests run against: when __graft_entry__.build() recorded them as built
(oracle/ref_expected.json, written in the development container and carried to the GPU box with the tree) or
TFX_EXPECT_REFERENCE_BINARIES=1 is set, a missing one is a FAILURE - a snapshot that lost the git-ignored oracle/_ref/ must not
turn ~80 parity tests into skips and stay green.  TFX_EXPECT_REFERENCE_BINARIES=0 (a machine that never had the reference) skips."""
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MARKER = os.path.join(ROOT, "oracle", "ref_expected.json")

# everything a `-m gpu` test may execute besides libtfx.so / libtfx_oracle.so (paths relative to the repository root)
CANDIDATES = [
    "oracle/_ref/tomofastx", "oracle/_ref/dropin/tomofastx_dropin", "oracle/_ref/gold_prism", "oracle/_ref/gold_lsqr",
    "tomofast-x_amd/host/tomofastx_amd", "tomofast-x_amd/host/tfx_host_demo", "tomofast-x_amd/host/tfx_reference_demo",
]
MPIEXEC = "/opt/conda/bin/mpiexec"


def write_marker():
    """Called by build(): records which of the binaries exist now."""
    present = [p for p in CANDIDATES if os.path.isfile(os.path.join(ROOT, p))]
    rec = {"binaries": present, "mpiexec": os.path.isfile(MPIEXEC)}
    with open(MARKER, "w") as f:
        json.dump(rec, f, indent=1)
    return rec


def expected():
    """-> (set of repository-relative paths that must exist, mpiexec expected?)"""
    env = os.environ.get("TFX_EXPECT_REFERENCE_BINARIES")
    if env == "0":
        return set(), False
    if env == "1":
        return set(CANDIDATES), True
    if os.path.isfile(MARKER):
        rec = json.load(open(MARKER))
        return set(rec.get("binaries", [])), bool(rec.get("mpiexec"))
    return set(), False


def missing(msg, *paths):
    """A test found that `paths` (absolute or repository-relative; none = decide from the message) are absent: fail when they were
    expected, skip otherwise."""
    want, want_mpi = expected()
    rel = [os.path.relpath(p, ROOT) if os.path.isabs(p) else p for p in paths]
    hit = [p for p in rel if p in want and not os.path.isfile(os.path.join(ROOT, p))]
    if MPIEXEC in paths and want_mpi and not os.path.isfile(MPIEXEC):
        hit.append(MPIEXEC)
    if not paths and (want or want_mpi):
        hit = ["(unspecified)"]
    if hit:
        pytest.fail("%s - expected here (oracle/ref_expected.json / TFX_EXPECT_REFERENCE_BINARIES): missing %s. A missing reference "
                    "binary or Fortran host is a FAILURE on a box that should have it, not a skip." % (msg, ", ".join(hit)))
    pytest.skip(msg)
