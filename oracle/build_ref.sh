#!/bin/bash
# oracle/build_ref.sh -- TEST INFRASTRUCTURE.  Builds the *unmodified* reference hot path
# (solver.f advance.f bounds_forcing.f initialize.f parallel_mpi.f, compiled where they lie
# under /root/reference) into oracle/_ref/libpomref_<IM>x<JM>x<KB>[_<IML>x<JML>p<NP>].so
# with AMD flang.  Nothing from the reference is copied into the repository: the pom.h the
# reference expects its user to derive from pom.h_dist (compile-time grid sizes,
# pom.h_dist:22-28) is produced in a throw-away directory under $TMPDIR.
# Usage: build_ref.sh IM JM KB [IM_LOCAL JM_LOCAL N_PROC]
set -euo pipefail
REF=${POM_REFERENCE:-/root/reference}
[ -d "$REF/pom" ] || { echo "build_ref: $REF absent - nothing to do"; exit 0; }
IM=$1; JM=$2; KB=$3; IML=${4:-$IM}; JML=${5:-$JM}; NP=${6:-1}
HERE=$(cd "$(dirname "$0")" && pwd)
OUT=$HERE/_ref; mkdir -p "$OUT"
TAG=${IM}x${JM}x${KB}; [ "$NP" != 1 ] && TAG=${TAG}_${IML}x${JML}p${NP}
LIB=$OUT/libpomref_$TAG.so
FC=${FC:-/opt/rocm/lib/llvm/bin/flang}
MPIINC=${MPIINC:-/opt/conda/include}; MPILIB=${MPILIB:-/opt/conda/lib}
W=$(mktemp -d "${TMPDIR:-/tmp}/pomref.XXXXXX"); trap 'rm -rf "$W"' EXIT
sed -e "s/im_global=282 /im_global=$IM /; s/jm_global=306 /jm_global=$JM /; s/kb=40 /kb=$KB /" \
    -e "s/im_local=142 /im_local=$IML /; s/jm_local=306 /jm_local=$JML /; s/n_proc=2 /n_proc=$NP /" \
    "$REF/pom.h_dist" > "$W/pom.h"
grep -q "im_global=$IM " "$W/pom.h" && grep -q "kb=$KB " "$W/pom.h" && grep -q "n_proc=$NP " "$W/pom.h" \
  || { echo "build_ref: size substitution failed"; exit 1; }
# -ffp-contract=off: x86-64 baseline has no FMA anyway; stated so the arithmetic is unambiguous.
FFLAGS="-O2 -fPIC -ffp-contract=off -mcmodel=medium -w -I$W -I$MPIINC"
for f in solver advance bounds_forcing initialize parallel_mpi; do
  "$FC" -c $FFLAGS "$REF/pom/$f.f" -o "$W/$f.o"
done
gcc -c -O1 -fPIC "$HERE/ref_traps.c" -o "$W/ref_traps.o"
"$FC" -shared -o "$LIB" "$W"/solver.o "$W"/advance.o "$W"/bounds_forcing.o "$W"/initialize.o \
  "$W"/parallel_mpi.o "$W"/ref_traps.o -L"$MPILIB" -Wl,-rpath,"$MPILIB" -lmpifort -lmpi -lm
echo "built $LIB"
