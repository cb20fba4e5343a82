#!/bin/bash
# Builds the UNMODIFIED reference (Tomofast-x, /root/reference) into oracle/_ref/ with amdflang + conda MPICH.
# Development-container only: /root/reference does not exist on the GPU box. Nothing from the reference is
# copied into the repository; object files, .mod files and the binary live under oracle/_ref/ (git-ignored).
#
# Two build accommodations (neither touches arithmetic; both are documented in DESIGN.md "Oracle"):
#  1. MPICH's mpi.mod in this image is gfortran-format; flang cannot read it. The MPI standard's other Fortran
#     binding, mpif.h (the real header of the installed MPICH), is wrapped in a module named `mpi`.
#  2. flang's runtime mis-handles re-opening unit 10 while it is still connected (the reference never closes the
#     parfile). A build-time COPY of src/parameters_init.f90 gets one `close(10)`; only needed for full parfile
#     runs (config 1), not for the unit-level golden vectors.
set -euo pipefail
REF=${REF:-/root/reference}
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/_ref"
B="$OUT/build"
[ -d "$REF/src" ] || { echo "reference not present at $REF - skipping"; exit 0; }
mkdir -p "$B"
cd "$B"
FC=${FC:-/opt/rocm/bin/amdflang}
MPI_INC=${MPI_INC:-/opt/conda/include}
MPI_LIB=${MPI_LIB:-/opt/conda/lib}
F="-O3 -fconvert=big-endian -DUSE_FLUSH6 -I$MPI_INC"

printf 'module mpi\n  implicit none\n  include "mpif.h"\nend module mpi\n' > mpi.f90
$FC $F -c mpi.f90

S=$REF/src
LIST="
$S/global_typedefs.F90
$S/libs/ftnunit.f90
$S/utils/file_utils.F90 $S/utils/mpi_tools.F90 $S/utils/costs.f90 $S/utils/vector.f90 $S/utils/string.f90
$S/utils/noise.f90 $S/utils/paraview.f90 $S/utils/parallel_tools.f90 $S/utils/memory_tools.F90
$S/utils/wavelet_transform.F90 $S/utils/sort.f90
$S/inversion/parameters_inversion.f90 $S/inversion/sparse_matrix.f90 $S/inversion/wavelet_utils.F90
$S/inversion/grid.F90 $S/inversion/model.F90 $S/inversion/model_IO.F90 $S/inversion/inversion_arrays.f90
$S/inversion/lsqr_solver2.F90 $S/inversion/damping.F90 $S/inversion/gradient.F90 $S/inversion/cross_gradient.F90
$S/inversion/admm_method.F90 $S/inversion/clustering.F90 $S/inversion/damping_gradient.F90
$S/inversion/joint_inverse_problem.F90
$S/forward/gravmag/parameters_gravmag.f90 $S/forward/gravmag/grav/parameters_grav.f90
$S/forward/gravmag/mag/parameters_mag.f90 $S/forward/gravmag/data_gravmag.f90
$S/forward/gravmag/grav/gravity_field.f90 $S/forward/gravmag/mag/magnetic_field.f90
$S/forward/gravmag/weights_gravmag.f90 $S/forward/gravmag/sensitivity_gravmag.F90
"
for f in $LIST; do
  o=$(basename "${f%.*}").o
  [ "$o" -nt "$f" ] || $FC $F -c "$f" -o "$o"
done
# patched copy (accommodation 2)
awk '{ if ($0 ~ /Finished reading the parameter file|Finished reading/ && !done) { print "  close(10)"; done=1 } print }' \
    $S/parameters_init.f90 > parameters_init_patched.f90
grep -q 'close(10)' parameters_init_patched.f90 || { echo "patch point not found"; exit 1; }
$FC $F -c parameters_init_patched.f90 -o parameters_init.o
for f in $S/problem_joint_gravmag.F90 $S/tests/tests_inversion.f90 $S/tests/tests_lsqr.f90 \
         $S/tests/tests_parallel_tools.f90 $S/tests/tests_sparse_matrix.f90 \
         $S/tests/tests_wavelet_compression.f90 $S/tests/unit_tests.f90; do
  o=$(basename "${f%.*}").o
  [ "$o" -nt "$f" ] || $FC $F -c "$f" -o "$o"
done
$FC $F -c $S/program_tomofastx.F90 -o program_tomofastx.o
LIBOBJ=$(ls *.o | grep -v -e '^program_tomofastx.o$' -e '^drv_' | tr '\n' ' ')
$FC $F -o "$OUT/tomofastx" program_tomofastx.o $LIBOBJ -L$MPI_LIB -lmpifort -lmpi -Wl,-rpath,$MPI_LIB

# Golden-vector drivers (OUR code, oracle/drivers/*.f90) link against the reference's module objects.
for d in "$HERE"/drivers/*.f90; do
  [ -f "$d" ] || continue
  n=$(basename "${d%.f90}")
  $FC $F -c "$d" -o "drv_$n.o"
  $FC $F -o "$OUT/$n" "drv_$n.o" $LIBOBJ -L$MPI_LIB -lmpifort -lmpi -Wl,-rpath,$MPI_LIB
done
echo "reference build OK -> $OUT"
