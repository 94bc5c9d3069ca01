// qgcm_dev.h - shared device/host declarations for the MI355X (gfx950) ocean path.
//
// Data layout in HBM (all fp64):
//   field(i,j,k)  p-grid fields po,pom,qo,qom(nxpo,nypo,nlo), wekpo, entoc,
//                 ddynoc(nxpo,nypo), ochom(nxpo,nypo,nlo-1):
//                 element (i,j,k) 1-based at  (i-1) + ldx*((j-1) + nypo*(k-1)),
//                 ldx = nxpo rounded up to 16 doubles (rows start 128-B aligned).
//   wrk(c,j,m)    spectral/physical work array of the Helmholtz solve, one
//                 128-B aligned row of ldw doubles per (j,m):
//                 box:    c = i-2  for interior i=2..nxpo-1 (nk = nxto-1 sine coeffs)
//                 cyclic: c = i-1  for i=1..nxto            (nk = nxto, own spectral order)
//   Thomas pivot tables (QgThomasParams): small.
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/qgcm_hip.h"

#define QG_MAXL QGCM_HIP_MAXL

// In-kernel phase stamps (development builds only, -DQG_STAMPS: scratch/stamps.py): thread 0 of every workgroup
// records the constant 100 MHz wall clock at phase boundaries; never compiled into the product library.
#ifdef QG_STAMPS
#define QG_NSTAMP 10
#define QG_STAMP_BLOCKS 4096
__device__ long long qg_stamps[4][QG_STAMP_BLOCKS][QG_NSTAMP];
#define QG_STAMP(kern, i)                                                                   \
  do {                                                                                      \
    const unsigned qg_b = blockIdx.x + gridDim.x * blockIdx.y;                              \
    if (threadIdx.x == 0 && qg_b < QG_STAMP_BLOCKS) qg_stamps[kern][qg_b][i] = wall_clock64(); \
  } while (0)
#define QG_STAMP_DRAIN() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#else
#define QG_STAMP(kern, i) do { } while (0)
#define QG_STAMP_DRAIN() do { } while (0)
#endif
#define QG_MAXFAC 24

// 16-byte WRITE-THROUGH store (global_store_dwordx4 ... sc1).  Why: a kernel whose stores all come at its end leaves its
// whole output dirty in the XCDs' L2s, and the end-of-kernel write-back of those lines runs AFTER the last wave, in
// series with everything (in-kernel stamps, NAtl 5 km: 3.4 us between the forward rows and the Thomas sweep, 5.7 us
// between the sweep and the inverse rows, for 22 MB each).  Written through, the bytes leave while other waves still
// compute and the next kernel finds them in the Infinity Cache: forward rows + sweep -2.7 us per step (A/B,
// profiles/r3_*).  Only the 16-byte form pays: 8-byte sc1 stores go out as one fabric write per lane
// (MI355X_MICROARCH.md, "stores of each flavour"; measured: no gain).  Inline asm because HIP has no 16-byte
// agent-scope store; the compiler does not count it in vmcnt, which only makes its own waits more conservative -
// nothing in these kernels reads what they have stored.  p must be 16-byte aligned.  tests: every parity test of the
// Helmholtz path runs through these stores (a missing hazard pad showed as 1e-2 errors in scattered column pairs).
__device__ __forceinline__ void qg_store16_wt(double *p, double a, double b) {
#ifdef QG_WT_PLAIN // (A/B builds: the same pairing with plain cached stores)
  *reinterpret_cast<double2 *>(p) = double2{a, b};
#else
  typedef double qg_v2d __attribute__((ext_vector_type(2)));
  const qg_v2d v = {a, b};
  // (s_nop 1: the hazard pad hipcc would put behind a 16-byte store of its own - without it the next instruction may
  //  overwrite the data registers before the store has read them; cdna_hip_programming.md 5.7, item 1)
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
#endif
}

// Point-per-lane kernels: the even lane of every lane pair stores its own value and its right-hand neighbour's as one
// 16-byte write-through store.  Every lane of the wave must arrive here (no divergent return before it); p = the
// lane's own element, 16-byte aligned on even lanes; `valid`: the lane's element exists (an even lane whose neighbour
// does not exist writes one element of row padding).
__device__ __forceinline__ void qg_pair_store_wt(double *p, double v, bool valid) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, 0xb1, 0xf, 0xf, true); // quad_perm [1,0,3,2]
  hi = __builtin_amdgcn_update_dpp(0, hi, 0xb1, 0xf, 0xf, true);
  const double nb = __hiloint2double(hi, lo);
  if (valid && !(threadIdx.x & 1)) qg_store16_wt(p, v, nb);
}
// scalars that live on the device between kernels (MODULE ochomog state)
struct QgScalars {
  double dpioc[QG_MAXL], dpiocp[QG_MAXL], xon[QG_MAXL];
  double xinhom[QG_MAXL], hclco[QG_MAXL];
  // continuity monitors of the zonally cyclic ocinvq / atinvq (MODULE monitor: ermaso, emfroc / ermasa, emfrat;
  // src/ocisubs.F:268-283, src/atisubs.F:236-248), per interface
  double ermas[QG_MAXL], emfr[QG_MAXL];
  // cyclic
  double ocncs[QG_MAXL], ocncn[QG_MAXL], ocncsp[QG_MAXL], ocncnp[QG_MAXL];
  double enisoc[QG_MAXL], eninoc[QG_MAXL];
  double ajisoc[QG_MAXL], ajinoc[QG_MAXL], ap3soc[QG_MAXL], ap3noc[QG_MAXL], ap5soc[QG_MAXL], ap5noc[QG_MAXL];
  double txisoc, txinoc, bdrins, bdrinn;
  double c1[QG_MAXL], c2[QG_MAXL], c3;
};

// constants of the box / cyclic constraint solve (host-prepared)
struct QgConstr {
  double cdiffo[QG_MAXL * QG_MAXL]; // (nl, nl-1)
  double cdhoc[QG_MAXL * QG_MAXL];  // (nl-1, nl-1)
  double cdhlu[QG_MAXL * QG_MAXL];  // LU of cdhoc
  int ipiv[QG_MAXL];
  double aipcho[QG_MAXL], hc1soc[QG_MAXL], hc2soc[QG_MAXL], hc1noc[QG_MAXL], hc2noc[QG_MAXL];
  double hbsioc, aipbho;
};

// geometry shared by all kernels
struct QgGeom {
  int nx, ny, nl, cyc; // ny = rows of the LOCAL arrays (owned rows + halo rows)
  int nxt;        // nxto = nx-1
  int nk;         // spectral coefficients per row
  int ldx, ldw;   // row pitches (doubles)
  long fstride;   // ldx*ny
  long wstride;   // ldw*ny
  // y-slab view: global row = local row + joff; this handle owns local rows jlo..jhi;
  // jr0..jr1 = owned rows that are interior to the global domain (2..nyg-1).
  int nyg, joff, jlo, jhi, jr0, jr1;
  int atm; // 1: atmospheric channel (qgcm_hip_params.atmos): cyclic kernels with the atmosphere's conventions
};

struct QgTendParams {
  QgGeom g;
  const double *pom, *po, *qo; // read (halo)
  double *qnew;                // in: qom, out: new qo (same buffer)
  const double *wekpo, *entoc, *ddynoc, *yporel;
  double *wrk;
  QgScalars *sc;
  double *bsum; // cyclic boundary line-sum partials
  double adfaco, bcfaco, dxom2, tdto, bdrfac, fnot, beta;
  double fohfac[QG_MAXL], ah2fac[QG_MAXL], ah4fac[QG_MAXL];
  double ctl2m[QG_MAXL * QG_MAXL]; // (k,m) at k + nl*m
  // box ocean inside qgcm_hip_steps: the leapfrog update of the mass-constraint integrals dpioc / dpiocp
  // (src/ocisubs.F:343-347) is done here, by one thread, so that the constraint solve can run redundantly in every
  // workgroup of the fused inverse-transform kernel (k_dst64_unpack<.., CONSTR>) without a launch of its own
  int upd_dpi;
  double gpoc[QG_MAXL];
  // window of tile rows (y-slabs, halo exchange overlapped with the tendency of the next step): trows == 0 -> all tile
  // rows; else this launch does the tile rows trow0 + r*tstride, r = 0..trows-1, and carries no edge / line-sum
  // workgroups unless the host appended them to the grid (they belong to the launch that follows the halo rows)
  int trow0, trows, tstride;
  // the fork's sponge layer (-Dsponge_layer_k247, src/qgosubs.F:203-205): the leapfrog step gains
  // + tdto*c1_spl*r_spl(i,j)*(qom - beta*yporel(j)); rspl = the ramp r_spl(nxpo,nypo) (field layout), tdc1 = tdto*c1_spl.
  // nullptr (every BASELINE configuration): the term is absent
  const double *rspl;
  double tdc1;
};

struct QgDstParams {
  QgGeom g;
  double *wrk;            // in/out rows
  const double2 *twid;    // exp(-2 pi i t / N), t = 0..N-1
  const double *sintab;   // 2 sin(k pi / N), k = 0..N/2
  double *rowsum;         // (ny, nl) row sums (inverse pass only) or nullptr
  int N;                  // complex FFT length (= nxto for the box DST)
  int nfac;
  int fac[QG_MAXFAC];
  int nlayers;            // layers to process (nl, or 1 for helmholtz())
  int layer0;             // first layer (mode) of this launch
  int single;             // generic kernels: one LDS buffer, in-place stages (long rows: two workgroups per CU)
  // generic cyclic inverse rows inside qgcm_hip_steps: one extra workgroup (blockIdx.x == gridDim.x - 1) runs part B of
  // the constraint algebra (k_cyclic.h) instead of a launch of its own; nullptr otherwise
  const struct QgCycConstrParams *cycq;
  // generic box inverse rows: one extra workgroup (blockIdx.x == gridDim.x - 1) runs the box constraint solve
  // (k_constr_box's body, k_misc.h) instead of a launch of its own; nullptr otherwise
  const struct QgConstrParams *boxq;
};

struct QgThomasParams {
  QgGeom g;
  const double *gath; // distributed sweep: all ranks' per-step summaries (rank-major), else nullptr
  const double *cgath; // distributed sweep: all ranks' set-up constants (slabDE of every rank, rank-major)
  double *send;       // distributed sweep: this rank's slab map
  double *slabDE;     // (4, ldw, nl): D, E, SP, SQ of this slab (PHASE 4/5 write, PHASE 1 reads)
  double *ksum;       // (ldw, nl): ftnorm * column sums of the solution per spectral index (see k_thomas.h)
  int rank, nranks;
  long gath_stride;   // doubles between two ranks' step messages in gath (>= TH_MSG*nl*ldw: a cyclic ocean appends its
                      // boundary line sums to the message)
  double *ybnd;       // PHASE 2, cyclic: (2, nl) zonal-mean solution at global rows 2 and nyg-1 (see k_cyclic.h), or nullptr
  double *wrk;
  // Thomas pivots, tabulated by the host (build_pivots): per block of 16 spectral indices the pivots of the local rows
  // r < rcb (rcb = rows until the block's last index is bitwise stationary, >= 1), 16 doubles per row starting at row
  // poff of ptab; binf (ldw, nlayers) = the stationary pivot per spectral index
  const double *binf, *ptab;
  const int *rcb, *poff; // (nblk, nlayers)
  int nblk;
  double aoc, ftnorm;
  int nlayers, layer0;
  // cyclic / atmosphere inside qgcm_hip_steps: device copy of the constraint parameters; one extra workgroup
  // (blockIdx.x == gridDim.x - 1) then runs part A of the constraint algebra (k_cyclic.h), which needs nothing of
  // the sweeps, so that part B can ride in the fused inverse-transform kernel.  nullptr otherwise.
  const struct QgCycConstrParams *cycq;
};

struct QgUnpackParams {
  QgGeom g;
  const double *wrk;
  const double *ochom;
  double *pnew; // old pom buffer, receives the new po
  double *msg_lo, *msg_hi; // y-slab halo messages (k_misc.h layout) written by the fused unpack, or nullptr
  // leapfrog averaging fused into k_dst64_unpack<.., AVG> (src/q-gcm.F:1345-1351: po = 0.5*(po + pom), qo = 0.5*(qo + qom) right
  // after this step): pavg = the po of this step (pom by then), qavg = the qo of this step; the new po and the boundary
  // qo are stored averaged (the boundary PV is formed from the un-averaged po, as the reference's ocqbdy is)
  const double *pavg, *qavg;
  const QgScalars *sc;
  const double *pch1, *pch2, *pbh; // cyclic (ny, nl-1), (ny)
  double ctm2l[QG_MAXL * QG_MAXL]; // (m,k) at m + nl*k
};

struct QgBdyParams {
  QgGeom g;
  const double *po;
  double *qo;
  const double *ddynoc, *yporel;
  double bcfaco_f0; // bccooc*dxom2/(0.5*bccooc+1)/fnot
  double beta;
  double f0A[QG_MAXL * QG_MAXL]; // fnot*amatoc(k,l) at k + nl*l
};

// what the box constraint solve needs, packed for nlo <= 4 (the kernels are instantiated for 2..4 layers): rides as a
// fourth argument of k_dst64_unpack<.., CONSTR> - the full QgConstrParams would push the kernel arguments past 4 KB
struct QgConstrLite {
  QgGeom g;
  const double *ksum, *wcot;
  QgScalars *sc;
  struct {
    double cdiffo[16], cdhoc[16], cdhlu[16];
    int ipiv[4];
  } cs;
  double dxo, dyo;
};

struct QgConstrParams {
  QgGeom g;
  const double *ksum; // (ldw, nl) spectral column sums of the solution (k_thomas)
  const double *wcot; // (ldw) sum over the interior points of sin(k i pi/n): 2cot(k pi/2n)-weights of dsint's synthesis, 0 for even k
  const double *rowsum;
  const double *wrk;
  QgScalars *sc;
  QgConstr cs;
  double dxo, dyo, tdto, fnot;
  double gpoc[QG_MAXL], hoc[QG_MAXL];
  double ctl2m[QG_MAXL * QG_MAXL], ctm2l[QG_MAXL * QG_MAXL];
};
