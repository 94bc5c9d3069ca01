#!/usr/bin/env bash
# Builds the ACTUAL drop-in: the reference main program src/q-gcm.F (patched copy, dropin/patch_main.py) and the
# rest of the reference model, compiled from the sources where they lie under /root/reference/src, with
# src/qgosubs.F, src/ocisubs.F (and, in coupled builds, src/qgasubs.F, src/atisubs.F) and src/omlsubs.F REPLACED by
# the shim modules of this directory, linked against q-gcm_amd/lib/libqgcm_hip.so.  Also builds the unmodified
# reference executable next to it (same flags), whose restart dump is the golden vector of the drop-in test.
#
#   build_dropin.sh <cfg> <nxta> <nyta> <nxaooc|nxta> <nyaooc> <ndxr> <nlo> <fnot> <beta> <mode>
#       mode: box | cyclic | coupled      ->  -Docean_only | -Docean_only -Dcyclic_ocean -Dnb_hflux | (coupled) -Dsb_hflux
#             box_spl = box + -Dsponge_layer_k247 (the fork's sponge layer, src/qgosubs.F:203-205)
#
# Nothing is copied into the repository: all outputs (objects, .mod, the two executables) go to
# q-gcm_amd/fortran/_dropin/<cfg>/, which is git-ignored; the patched copy of the main program is deleted after it
# has been compiled.  The reference's own Makefile is not used; compile lines follow its flag split (QGOPTS only
# where src/Makefile passes them).  -Duse_netcdf / -Dqoc_diag are dropped (no netCDF here).
# LAPACK for eigmod / homsol comes from the image's MKL, as in oracle/build_ref.sh.
set -euo pipefail
CFG=$1; NXTA=$2; NYTA=$3; NXAOOC=$4; NYAOOC=$5; NDXR=$6; NLO=$7; FNOT=$8; BETA=$9; MODE=${10}

REF=${QGCM_REFERENCE:-/root/reference}
SRC=$REF/src
HERE=$(cd "$(dirname "$0")" && pwd)
FDIR=$(cd "$HERE/.." && pwd)
OUT=$FDIR/_dropin/$CFG
FC=${FC:-/opt/rocm/bin/amdflang}
MKLDIR=${MKLDIR:-/opt/conda/lib}
LIBDIR=$(cd "$FDIR/../lib" && pwd)

if [ ! -d "$SRC" ]; then
  echo "build_dropin: $SRC not present (GPU box?) - keeping prebuilt files" >&2
  exit 0
fi

case "$MODE" in
  box)     Q="-Docean_only -Dsb_hflux" ;;
  box_spl) Q="-Docean_only -Dsb_hflux -Dsponge_layer_k247" ;;
  cyclic)  Q="-Docean_only -Dcyclic_ocean -Dnb_hflux" ;;
  coupled) Q="-Dsb_hflux" ;;
  *) echo "mode must be box | box_spl | cyclic | coupled" >&2; exit 2 ;;
esac

FCB="$FC -ffixed-line-length-132 -O2"
FCO="$FCB -fopenmp"
LAPACK="-L$MKLDIR -Wl,--no-as-needed -lmkl_gf_lp64 -lmkl_sequential -lmkl_core -Wl,--as-needed -Wl,-rpath,$MKLDIR -Wl,-rpath,/opt/rocm/lib/llvm/lib"

# everything both executables share: data modules, set-up, diagnostics, forcing (reference sources, unchanged)
common() {
  sed -e "s|^      PARAMETER ( nxta = .*|      PARAMETER ( nxta = $NXTA, nyta = $NYTA, nla = 3 )|" \
      -e "s|^      PARAMETER ( nxaooc = .*|      PARAMETER ( nxaooc = $NXAOOC, nyaooc = $NYAOOC, ndxr = $NDXR, nlo = $NLO )|" \
      -e "s|^      PARAMETER ( fnot = .*|      PARAMETER ( fnot = $FNOT, beta = $BETA )|" \
      -e "s|^      END MODULE parameters|      double precision :: c1_spl, l_spl\n      PARAMETER ( c1_spl = -2.5D-5, l_spl = 4.0D5 )\n      END MODULE parameters|" \
      "$REF/examples/double_gyre_ocean_only/parameters_data.F.dg_oo" > parameters_data.F
  # (c1_spl, l_spl: the fork's sponge-layer constants of src/parameters_data.F:140-144, which src/out_param.f:270-272
  #  prints; the example's parameters file predates them - SURVEY.md 8c)
  $FCO -c parameters_data.F
  rm -f parameters_data.F   # edited text of a reference source: it does not stay in a directory that travels
  for f in atconst occonst athomog ochomog atstate ocstate intrfac; do $FCO $Q -c -I"$SRC" "$SRC/${f}_data.F"; done
  $FCO -c -I"$SRC" "$SRC/radiate_data.F"
  $FCO $Q -c -I"$SRC" "$SRC/timinfo_data.F"
  $FCO -c -I"$SRC" "$SRC/monitor_data.F"
  $FCO -c -I"$SRC" "$SRC/intsubs.f"
  $FCO -c -I"$SRC" "$SRC/eigmode.f"
  ( cd "$SRC" && $FCO -c -o "$1/fftsubs.o" fftsubs.f ) 2> fftsubs.warn || { cat fftsubs.warn; exit 1; }
  for f in nc_subs xfosubs; do $FCO $Q -c -I"$SRC" "$SRC/$f.F"; done
  $FCO -c -I"$SRC" "$SRC/radsubs.f"
  $FCO $Q -c -I"$SRC" "$SRC/topsubs.F"
}
COMMON_OBJ="parameters_data.o atconst_data.o occonst_data.o athomog_data.o ochomog_data.o atstate_data.o ocstate_data.o
            intrfac_data.o radiate_data.o timinfo_data.o monitor_data.o intsubs.o eigmode.o fftsubs.o nc_subs.o xfosubs.o
            radsubs.o topsubs.o"
REST="areasubs_diag covaria_diag monitor_diag qocdiag timavge vorsubs valsubs"
REST_OBJ="areasubs_diag.o covaria_diag.o monitor_diag.o qocdiag.o timavge.o vorsubs.o valsubs.o"

# ---- 1. the unmodified reference ------------------------------------------------------------------------------------
mkdir -p "$OUT/ref" && cd "$OUT/ref"
common "$OUT/ref"
for f in $REST; do $FCO $Q -c -I"$SRC" "$SRC/$f.F"; done
$FCB $Q -c -I"$SRC" "$SRC/omlsubs.F"          # no -fopenmp: flang rejects its REDUCTION(-:...) clause
$FCB $Q -c -I"$SRC" "$SRC/amlsubs.F"
for f in qgosubs ocisubs qgasubs atisubs conhoms; do $FCO $Q -c -I"$SRC" "$SRC/$f.F"; done
( cd "$SRC" && $FCO $Q -c -o "$OUT/ref/q-gcm.o" -J"$OUT/ref" -I"$OUT/ref" q-gcm.F )
$FC -fopenmp -o "$OUT/q-gcm_ref" $COMMON_OBJ $REST_OBJ omlsubs.o amlsubs.o qgosubs.o ocisubs.o qgasubs.o atisubs.o conhoms.o q-gcm.o $LAPACK

# ---- 2. the drop-in ------------------------------------------------------------------------------------------------
mkdir -p "$OUT/hip" && cd "$OUT/hip"
common "$OUT/hip"
FH="$FC -O2 -cpp -DQGCM_DROPIN $Q"
$FH -c "$FDIR/ocisubs_data.F90"
$FH -c "$FDIR/qgcm_hip_iface.F90"
$FH -c "$FDIR/qgcm_hip_shim.F90"               # MODULE qgosubs, ocisubs, vorsubs_hip, omlsubs, valsubs_hip [, qgasubs, atisubs, vorsubs_hip_at]
for f in $REST; do $FCO $Q -c -I"$SRC" "$SRC/$f.F"; done
$FCB $Q -c -I"$SRC" "$SRC/amlsubs.F"
SHIM_OBJ="ocisubs_data.o qgcm_hip_iface.o qgcm_hip_shim.o"
if [ "$MODE" != "coupled" ]; then
  # ocean-only builds keep the (empty) reference modules qgasubs / atisubs, which q-gcm.F USEs unconditionally
  for f in qgasubs atisubs; do $FCO $Q -c -I"$SRC" "$SRC/$f.F"; done
  SHIM_OBJ="$SHIM_OBJ qgasubs.o atisubs.o"
fi
$FCO $Q -c -I"$SRC" "$SRC/conhoms.F"           # homsol -> hsbxoc / hscyoc / hscyat of the shim
python3 "$HERE/patch_main.py" "$SRC/q-gcm.F" "$OUT/hip/q-gcm_dropin.F"
# the INCLUDEd parameter lists (in_param.f, out_param.f) are found through -I
$FCO $Q -c -I"$SRC" -o q-gcm.o q-gcm_dropin.F
rm -f q-gcm_dropin.F
$FC -fopenmp -o "$OUT/q-gcm_hip" $COMMON_OBJ $REST_OBJ amlsubs.o $SHIM_OBJ conhoms.o q-gcm.o \
    -L"$LIBDIR" -lqgcm_hip -Wl,-rpath,'$ORIGIN/../../../lib' -Wl,-rpath,/opt/rocm/lib -L/opt/rocm/lib -lamdhip64 $LAPACK
echo "built $OUT/q-gcm_ref and $OUT/q-gcm_hip ($MODE)"
