#!/bin/bash
# The reference as a HOST of this repository's GPU library: the UNMODIFIED Tomofast-x sources (/root/reference) with exactly five
# modules swapped for the drop-in modules of tomofast-x_amd/host/dropin/ (same module names, same public interfaces, the reference's
# own types as arguments):
#     sparse_matrix  lsqr_solver  wavelet_transform  sensitivity_gravmag  weights_gravmag
# (+ gravity_field / magnetic_field, which only sensitivity_gravmag used).  Everything else - problem_joint_gravmag.F90,
# joint_inverse_problem.F90, model.F90, damping / ADMM / cross-gradient / clustering builders, Parfile reader, I/O, the unit tests -
# is compiled as it lies in /root/reference.  That they COMPILE is the compiler-checked statement that the boundary of
# INTEGRATION.md has the reference's names, argument orders and types; that the result RUNS (oracle/_ref/dropin/tomofastx_dropin,
# tests/test_gpu_dropin.py) is the drop-in claim itself.
#
# Development-container only (/root/reference does not exist on the GPU box); outputs only into oracle/_ref/dropin/ (git-ignored,
# travels to the GPU box as a binary like oracle/_ref/tomofastx).  Nothing of the reference is copied into the repository.
#   oracle/dropin_build.sh            compile + link
#   oracle/dropin_build.sh --check    compile only (no libtfx.so needed): the conformance check
set -euo pipefail
REF=${REF:-/root/reference}
HERE="$(cd "$(dirname "$0")" && pwd)"
ROOT="$(dirname "$HERE")"
OUT="$HERE/_ref/dropin"
B="$OUT/build"
[ -d "$REF/src" ] || { echo "reference not present at $REF - skipping"; exit 0; }
MODE=${1:-link}
mkdir -p "$B"
cd "$B"
FC=${FC:-/opt/rocm/bin/amdflang}
MPI_INC=${MPI_INC:-/opt/conda/include}
HOST="$ROOT/tomofast-x_amd/host"
F="-O2 -fconvert=big-endian -DUSE_FLUSH6 -I$MPI_INC"
S=$REF/src
D=$HOST/dropin

DIRTY=0
compile() {   # source -> object in the build directory, when out of date - or when anything before it in the order was recompiled
  local f=$1 o     # (a module file carries the checksums of the modules it uses: a stale dependant is a compile error, not a warning)
  o=$(basename "${f%.*}").o
  if [ $DIRTY = 1 ] || [ ! -f "$o" ] || [ "$f" -nt "$o" ]; then DIRTY=1; $FC $F -c "$f" -o "$o"; fi
}

printf 'module mpi\n  implicit none\n  include "mpif.h"\nend module mpi\n' > mpi.f90
compile mpi.f90
# this repository's binding layer (its own copies are compiled here so that every .mod of the build comes from one place)
for f in $HOST/tfx_binding.f90 $HOST/tfx_host_mpi.f90 $HOST/tfx_reference_api.f90; do compile "$f"; done
# reference, unmodified: what the swapped modules depend on
for f in $S/global_typedefs.F90 $S/libs/ftnunit.f90 $S/utils/file_utils.F90 $S/utils/mpi_tools.F90 $S/utils/costs.f90 $S/utils/vector.f90 \
         $S/utils/string.f90 $S/utils/noise.f90 $S/utils/paraview.f90 $S/utils/parallel_tools.f90 $S/utils/memory_tools.F90 $S/utils/sort.f90 \
         $S/inversion/parameters_inversion.f90; do compile "$f"; done
# SWAPPED: wavelet_transform, sparse_matrix
compile $D/dropin_wavelet_transform.f90
compile $D/dropin_sparse_matrix.f90
for f in $S/inversion/wavelet_utils.F90 $S/inversion/grid.F90 $S/inversion/model.F90 $S/inversion/model_IO.F90 \
         $S/inversion/inversion_arrays.f90; do compile "$f"; done
# SWAPPED: lsqr_solver
compile $D/dropin_lsqr_solver.f90
for f in $S/inversion/damping.F90 $S/inversion/gradient.F90 $S/inversion/cross_gradient.F90 $S/inversion/admm_method.F90 \
         $S/inversion/clustering.F90 $S/inversion/damping_gradient.F90 $S/inversion/joint_inverse_problem.F90 \
         $S/forward/gravmag/parameters_gravmag.f90 $S/forward/gravmag/grav/parameters_grav.f90 $S/forward/gravmag/mag/parameters_mag.f90 \
         $S/forward/gravmag/data_gravmag.f90; do compile "$f"; done
# SWAPPED: weights_gravmag, sensitivity_gravmag (gravity_field.f90 / magnetic_field.f90 are not compiled at all)
compile $D/dropin_gravmag_convert.f90
compile $D/dropin_weights_gravmag.f90
compile $D/dropin_sensitivity_gravmag.f90
# the flang-runtime accommodation of oracle/ref_build.sh (one close(10) in a build-time copy of the Parfile reader; no arithmetic)
awk '{ if ($0 ~ /Finished reading the parameter file|Finished reading/ && !done) { print "  close(10)"; done=1 } print }' \
    $S/parameters_init.f90 > parameters_init_patched.f90
grep -q 'close(10)' parameters_init_patched.f90 || { echo "patch point not found"; exit 1; }
{ [ $DIRTY = 0 ] && [ -f parameters_init.o ] && [ parameters_init.o -nt $S/parameters_init.f90 ]; } || { DIRTY=1; $FC $F -c parameters_init_patched.f90 -o parameters_init.o; }
# reference, unmodified: the callers of the boundary, the unit tests, the program
for f in $S/problem_joint_gravmag.F90 $S/tests/tests_inversion.f90 $S/tests/tests_lsqr.f90 $S/tests/tests_parallel_tools.f90 \
         $S/tests/tests_sparse_matrix.f90 $S/tests/tests_wavelet_compression.f90 $S/tests/unit_tests.f90 $S/program_tomofastx.F90; do compile "$f"; done
echo "drop-in conformance: the unmodified reference callers compile against the drop-in modules"
[ "$MODE" = "--check" ] && exit 0
[ -f "$ROOT/tomofast-x_amd/libtfx.so" ] || { echo "libtfx.so not built - run make -C tomofast-x_amd/csrc first"; exit 1; }
make -s -C "$HOST" mpilib
OBJ=$(ls *.o | tr '\n' ' ')
$FC $F -o "$OUT/tomofastx_dropin" $OBJ -L"$ROOT/tomofast-x_amd" -ltfx -L"$HOST/mpilib" -lmpifort -lmpi \
    -Wl,-rpath,'$ORIGIN/../../../tomofast-x_amd' -Wl,-rpath,'$ORIGIN/../../../tomofast-x_amd/host/mpilib'
echo "drop-in build OK -> $OUT/tomofastx_dropin"
