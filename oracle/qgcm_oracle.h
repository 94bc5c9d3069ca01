/* TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C, fp64) of the jinkakei/q-gcm ocean hot path
 *   qgostep -> ocinvq -> ocqbdy   (src/q-gcm.F:1243-1249)
 * plus the init-time routines that feed it (eigmod, homsol, constr, qcomp,
 * merqcy).  It is the checker for the HIP path: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may call it.
 * Every function cites the reference lines it follows.
 *
 * Pinning: validated against the true reference compiled from
 * /root/reference/src (oracle/build_ref.sh -> oracle/_ref/) and against the
 * golden vectors under tests/golden/ that were generated from that build
 * (tests/golden/make_golden.py).  See tests/test_oracle_vs_golden.py.
 * The mixed layer (qgo_oml) is pinned bitwise by three reference builds (tests/golden/make_golden_oml.py,
 * tests/test_oml_oracle.py), the validity scan (qgo_valids) by the reference's verdicts on 16 crafted states
 * (tests/golden/make_golden_valids.py, tests/test_valids_oracle.py).
 *
 * Arrays are Fortran ordered: element (i,j,k), 1-based, lives at
 * (i-1) + nxpo*((j-1) + nypo*(k-1)).
 */
#ifndef QGCM_ORACLE_H
#define QGCM_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct qgo_ctx qgo_ctx;

/* grid + physical parameters (src/parameters_data.F, src/in_param.f) */
qgo_ctx *qgo_create(int nxpo, int nypo, int nlo, int cyclic,
                    double fnot, double beta, double dxo, double dto,
                    double delek, double bccooc,
                    const double *ah2oc, const double *ah4oc,
                    const double *hoc, const double *gpoc,
                    const double *yporel, const double *ddynoc);
/* The atmospheric channel of a coupled run: qgastep / atinvq / atqzbd
 * (src/qgasubs.F:45-317, src/atisubs.F:60-395, src/vorsubs.F:396-480) behind the same entry points -
 * qgo_qgostep = qgastep, qgo_ocinvq = atinvq, qgo_ocqbdy = atqzbd, qgo_set_p = constr + qcomp + atqzbd + merqcy,
 * qgo_steps averages when mod(nt-1,100) == 0 (src/q-gcm.F:1370), qgo_set_cyc_forcing takes txisat/txinat,
 * enisat/eninat, qgo_set_forcing wekpa/entat/xan, the scalars are dpiat, dpiatp, atmcs, atmcn, atmcsp, atmcnp.
 * Pinned by tests/golden/atm_*.npz from a coupled build of the reference. */
qgo_ctx *qgo_create_atmos(int nxpa, int nypa, int nla, double fnot, double beta, double dxa, double dta,
                          double bccoat, const double *ah4at, const double *hat, const double *gpat,
                          const double *yparel, const double *ddynat);
void qgo_destroy(qgo_ctx *c);
void qgo_set_threads(int nthreads);

/* state */
void qgo_set_p(qgo_ctx *c, const double *po, const double *pom); /* + constr, qcomp, ocqbdy, merqcy */
void qgo_set_state(qgo_ctx *c, const double *po, const double *pom, const double *qo, const double *qom);
void qgo_get_state(qgo_ctx *c, double *po, double *pom, double *qo, double *qom);
void qgo_set_forcing(qgo_ctx *c, const double *wekpo, const double *entoc, const double *xon);
void qgo_set_cyc_forcing(qgo_ctx *c, double txis, double txin, const double *enis, const double *enin);
void qgo_set_sponge(qgo_ctx *c, const double *r_spl, double c1_spl); /* src/qgosubs.F:203-205; NULL = off */
/* scal = dpioc(nlo-1), dpiocp(nlo-1), ocncs, ocncn, ocncsp, ocncnp (nlo each; zero for box) */
void qgo_get_scalars(qgo_ctx *c, double *scal);
void qgo_set_scalars(qgo_ctx *c, const double *scal);
/* diagnostics of the last ocinvq: xinhom(nlo), then hclco(nlo-1) [box] or c1(nlo-1),c2(nlo-1),c3 [cyclic] */
void qgo_get_inv_diag(qgo_ctx *c, double *xinhom, double *coef);
/* cyclic / atmosphere: boundary line sums of the last qgostep - ajis, ajin, ap5s, ap5n (nlo each) */
void qgo_get_bsums(qgo_ctx *c, double *b);

/* constants */
void qgo_get_consts(qgo_ctx *c, double *amatoc, double *ctl2moc, double *ctm2loc,
                    double *rdm2oc, double *bd2oc, double *aoc);
/* box: hom = ochom(nxpo,nypo,nlo-1), aux = aipohs, cdiffo(nlo,nlo-1), cdhoc(nlo-1,nlo-1)
 * cyc: hom = pch1oc(nypo,nlo-1), pch2oc(nypo,nlo-1), pbhoc(nypo);
 *      aux = aipcho, hc1soc, hc2soc, hc1noc, hc2noc (nlo-1 each), hbsioc, aipbho */
void qgo_get_homog(qgo_ctx *c, double *hom, double *aux);

/* the path */
void qgo_qgostep(qgo_ctx *c);
void qgo_ocinvq(qgo_ctx *c);
void qgo_ocqbdy(qgo_ctx *c);
void qgo_lf_average(qgo_ctx *c);
void qgo_steps(qgo_ctx *c, int s0, int n);

/* building blocks */
void qgo_project(qgo_ctx *c, double *wrk);                 /* ocisubs.F:117-139 */
void qgo_helmholtz(qgo_ctx *c, double *wrk, const double *boc); /* hsbxoc / hscyoc */
double qgo_xintp(const double *val, int nx, int ny);        /* intsubs.f:78-133 */
void qgo_dsint(int n, double *x);                           /* x has n+1 elements */
void qgo_rfftf(int n, double *x);                           /* FFTPACK half-complex layout */
void qgo_rfftb(int n, double *x);
void qgo_eigmod(int nl, const double *gpr, const double *h, double fnot,
                double *amat, double *rdm2, double *ctl2m, double *ctm2l);
/* eigmod with case = 'Atmosphere' (no Flierl normalisation, eigmode.f:309): see the note in qgcm_oracle.c */
void qgo_eigmod_atmos(int nl, const double *gpr, const double *h, double fnot,
                      double *amat, double *rdm2, double *ctl2m, double *ctm2l);
/* ocean-only Ekman pumping from wind stress, xfosubs.F:138,566-645 */
void qgo_wekpo_from_tau(int nxpo, int nypo, int cyclic, double dxo, double fnot,
                        const double *tauxo, const double *tauyo, double *wekto, double *wekpo);

/* ocean mixed layer oml / omladf (src/omlsubs.F:47-236, 244-763), SURVEY 8 row f1.
 * T-grid arrays are (nxto,nyto) = (nxpo-1, nypo-1); sb_hflux / nb_hflux are the reference's compile-time
 * boundary options as run-time flags.  qgo_oml_get: scal = xon(1), cfraoc, centoc, enisoc(1), eninoc(1). */
void qgo_oml_init(qgo_ctx *c, double hmoc, double toc1, double toc2, double st2d, double st4d, double ycexp,
                  double rrcpoc, int sb_hflux, double tsbdy, int nb_hflux, double tnbdy);
void qgo_oml_set(qgo_ctx *c, const double *sst, const double *sstm, const double *fnetoc, const double *wekto,
                 const double *tauxo, const double *tauyo);
void qgo_oml_get(qgo_ctx *c, double *sst, double *sstm, double *entoc, double *scal);
void qgo_oml(qgo_ctx *c);
void qgo_steps_oml(qgo_ctx *c, int s0, int n); /* oml, qgostep, ocinvq, ocqbdy (+ averaging incl. sst) */

/* valids, ocean part (src/valsubs.F:272-527): out = min/max of po, qo, sst, wekto, full layer thickness
 * top / intermediate / bottom (14 values), hfbad(1..nlo) in per cent; returns solnok. dtopoc may be NULL (flat). */
int qgo_valids(qgo_ctx *c, const double *dtopoc, double *out);

#ifdef __cplusplus
}
#endif
#endif
