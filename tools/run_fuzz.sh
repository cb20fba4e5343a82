#!/bin/bash
# All randomised sweeps against the oracle on a GPU box (about two minutes): `gpurun -- 'bash tools/run_fuzz.sh [seed]'`.
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
S=${1:-1}
python $R/tools/fuzz_matrix.py 60 $((S + 1)) | tail -1
python $R/tools/fuzz_build.py 60 $((S + 2)) | tail -1
python $R/tools/fuzz_build_comp.py 40 $((S + 3)) | tail -1
python $R/tools/fuzz_lsqr_wavelet.py 40 $((S + 4)) | tail -3
python $R/tools/fuzz_band.py 25 $((S + 5)) | tail -1
python $R/tools/fuzz_misc.py 40 $((S + 6)) | tail -1
python $R/tools/fuzz_many_rows.py | tail -1
python $R/tools/fuzz_hosts.py 20 $((S + 7)) | tail -1
