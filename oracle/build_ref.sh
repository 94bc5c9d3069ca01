#!/usr/bin/env bash
# TEST INFRASTRUCTURE ONLY (oracle).
#
# Builds the *true reference* hot path (qgostep/ocinvq/ocqbdy + FFTPACK +
# constr/homsol/eigmod) from the sources where they lie under
# /root/reference/src, together with our C-callable harness
# oracle/ref/qgcm_ref_harness.F90, into oracle/_ref/libqgcm_ref_<cfg>.so.
#
#  * nothing is copied into the repository: objects, .mod files and the
#    per-config parameters_data.F (grid dimensions are compile-time
#    PARAMETERs in the reference, src/parameters_data.F:23-147, so one
#    library is built per grid) go to oracle/_ref/ only, which is
#    git-ignored but travels to the GPU box with the snapshot.
#  * the reference's own Makefile is not used; the compile lines below
#    follow the flag split of src/Makefile (QGOPTS only where it passes
#    them) as recorded in SURVEY.md appendix D.
#  * LAPACK (DGETRF/DGETRS/DGERFS + the eigmod chain) is not vendored by
#    the reference (src/lasubs.f INCLUDEs an absent lapack/ dir); the
#    image's MKL (/opt/conda/lib/libmkl_rt.so) provides it.
#
# usage: build_ref.sh <cfg> <nxta> <nyta> <nxaooc|nxta> <nyaooc> <ndxr> <nlo> <fnot> <beta> <cyclic 0|1> [extra cpp options] [coupled 0|1]
#   extra cpp options: e.g. -Dsb_hflux (mixed-layer boundary variants, src/omlsubs.F:405-422,437-454);
#   the cyclic builds carry -Dnb_hflux as examples/southern_ocean_ocean_only does
#   coupled = 1: a coupled build (no -Docean_only, as examples/double_gyre_coupled/make.config.coupled):
#   adds the atmosphere path qgasubs.F / atisubs.F / atqzbd and oracle/ref/qgcm_ref_atmos.F90
set -euo pipefail

CFG=$1; NXTA=$2; NYTA=$3; NXAOOC=$4; NYAOOC=$5; NDXR=$6; NLO=$7; FNOT=$8; BETA=$9; CYC=${10}; EXTRA=${11:-}; CPL=${12:-0}

REF=${QGCM_REFERENCE:-/root/reference}
SRC=$REF/src
HERE=$(cd "$(dirname "$0")" && pwd)
OUT=$HERE/_ref
WRK=$OUT/$CFG
FC=${FC:-/opt/rocm/bin/amdflang}
MKLDIR=${MKLDIR:-/opt/conda/lib}

if [ ! -d "$SRC" ]; then
  echo "build_ref: $SRC not present (GPU box?) - keeping prebuilt files" >&2
  exit 0
fi

mkdir -p "$WRK"
cd "$WRK"

# per-config dimension module: the example file with its three active
# PARAMETER lines (grid, ocean, rotation) rewritten
sed -e "s|^      PARAMETER ( nxta = .*|      PARAMETER ( nxta = $NXTA, nyta = $NYTA, nla = 3 )|" \
    -e "s|^      PARAMETER ( nxaooc = .*|      PARAMETER ( nxaooc = $NXAOOC, nyaooc = $NYAOOC, ndxr = $NDXR, nlo = $NLO )|" \
    -e "s|^      PARAMETER ( fnot = .*|      PARAMETER ( fnot = $FNOT, beta = $BETA )|" \
    "$REF/examples/double_gyre_ocean_only/parameters_data.F.dg_oo" > parameters_data.F
# -Dsponge_layer_k247 builds: the example dimension files lack the fork's sponge constants ("fork
# breakage"); the two lines of src/parameters_data.F:140,144 are appended, as q-gcm_amd/fortran/dropin/build_dropin.sh does
case "$EXTRA" in *sponge_layer_k247*)
  sed -i -e "s|^      END MODULE parameters|      double precision :: c1_spl, l_spl\n      PARAMETER ( c1_spl = -2.5D-5, l_spl = 4.0D5 )\n      END MODULE parameters|" parameters_data.F ;;
esac

Q="-Docean_only"
if [ "$CPL" = "1" ]; then Q=""; fi
if [ "$CYC" = "1" ]; then Q="$Q -Dcyclic_ocean -Dnb_hflux"; fi
Q="$Q $EXTRA"

FCB="$FC -ffixed-line-length-132 -O2 -fPIC"
FCO="$FCB -fopenmp"

$FCO -c parameters_data.F
ATMOBJ=""
if [ "$CPL" = "1" ]; then
  for f in atconst athomog atstate; do $FCO $Q -c -I"$SRC" "$SRC/${f}_data.F"; done
  ATMOBJ="atconst_data.o athomog_data.o atstate_data.o atisubs.o qgasubs.o qgcm_ref_atmos.o"
fi
if [ "$CPL" != "1" ]; then
  # ocean-only builds: MODULE atconst is USEd by xfosubs.F (below) in every configuration
  $FCO $Q -c -I"$SRC" "$SRC/atconst_data.F"
  ATMOBJ="atconst_data.o xfosubs.o"
fi
for f in occonst ochomog ocstate; do $FCO $Q -c -I"$SRC" "$SRC/${f}_data.F"; done
$FCO -c -I"$SRC" "$SRC/monitor_data.F"
$FCO -c -I"$SRC" "$SRC/intsubs.f"
$FCO -c -I"$SRC" "$SRC/eigmode.f"
( cd "$SRC" && $FCO -c -o "$WRK/fftsubs.o" fftsubs.f ) 2> fftsubs.warn || { cat fftsubs.warn; exit 1; }
if [ "$CPL" = "1" ]; then
  for f in atisubs qgasubs; do $FCO $Q -c -I"$SRC" "$SRC/$f.F"; done
fi
for f in vorsubs qgosubs ocisubs conhoms; do $FCO $Q -c -I"$SRC" "$SRC/$f.F"; done
# ocean mixed layer: intrfac / radiate data modules + omlsubs.F; the latter without
# -fopenmp (flang rejects its REDUCTION(-:...) clause)
$FCO $Q -c -I"$SRC" "$SRC/intrfac_data.F"
$FCO -c -I"$SRC" "$SRC/radiate_data.F"
$FCB $Q -c -I"$SRC" "$SRC/omlsubs.F"
# validity scan
$FCO $Q -c -I"$SRC" "$SRC/valsubs.F"
# ocean-only builds: xforc's wekto / wekpo from the wind stress (src/xfosubs.F:566-683) - the
# whole routine as it stands; in an ocean_only build everything before that section is compiled out
if [ "$CPL" != "1" ]; then $FCO $Q -c -I"$SRC" "$SRC/xfosubs.F"; fi
$FC -O2 -fPIC -fopenmp -cpp $Q -c "$HERE/ref/qgcm_ref_harness.F90"
$FC -O2 -fPIC -fopenmp -cpp $Q -c "$HERE/ref/qgcm_ref_oml.F90"
if [ "$CPL" = "1" ]; then $FC -O2 -fPIC -fopenmp -cpp $Q -c "$HERE/ref/qgcm_ref_atmos.F90"; fi

$FC -shared -fopenmp -o "$OUT/libqgcm_ref_$CFG.so" \
    parameters_data.o occonst_data.o ochomog_data.o ocstate_data.o monitor_data.o \
    intsubs.o eigmode.o fftsubs.o vorsubs.o qgosubs.o ocisubs.o conhoms.o \
    intrfac_data.o radiate_data.o omlsubs.o valsubs.o qgcm_ref_harness.o qgcm_ref_oml.o $ATMOBJ \
    -L"$MKLDIR" -Wl,--no-as-needed -lmkl_gf_lp64 -lmkl_sequential -lmkl_core -Wl,--as-needed -Wl,-rpath,"$MKLDIR" -Wl,-rpath,/opt/rocm/lib/llvm/lib

# the per-config dimension file is edited text of a reference source: it does not stay in a directory that travels
rm -f parameters_data.F
echo "built $OUT/libqgcm_ref_$CFG.so"
