/* TEST INFRASTRUCTURE ONLY - see qgcm_oracle.h.
 *
 * Plain-C fp64 restatement of the reference algorithm; loops keep the
 * reference's association order so that, compiled with -ffp-contract=off,
 * the pointwise parts (qgostep, ocqbdy, projection, unpack) reproduce the
 * reference bit for bit on x86-64.  The row transforms are an
 * algebraically equivalent mixed-radix FFT (the reference calls FFTPACK;
 * agreement there is to rounding, see tests/test_oracle_vs_golden.py).
 */
#include "qgcm_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define PI_ 3.14159265358979324
#define TWOPI_ 6.28318530717958648

struct qgo_ctx {
  int nx, ny, nl, cyclic, nxt; /* nxt = nxto = nx-1 */
  /* atmos = 1: the handle is the atmospheric channel (qgastep / atinvq / atqzbd): the cyclic
   * code path with the atmosphere's conventions - layer 1 is the bottom layer (topography term in layer 1),
   * forcing signs of src/qgasubs.F:128-131, no drag / Del-4th terms, constraint right-hand sides of
   * src/atisubs.F:177-196, dpiat = integral of pa(k)-pa(k+1), averaging every 100 steps.  The member names stay
   * the ocean's (po = pa, ocncs = atmcs, ...). */
  int atmos;
  double fnot, beta, dxo, dyo, dxom2, dto, tdto, delek, bccooc, xlo, ylo;
  double *ah2oc, *ah4oc, *hoc, *gpoc, *yporel, *ddynoc;
  double *r_spl; /* sponge-layer ramp (cpp option sponge_layer_k247, src/occonst_data.F:100-105) or NULL */
  double c1_spl; /* src/parameters_data.F:144 */
  double *amatoc, *ctl2moc, *ctm2loc, *rdm2oc; /* (nl,nl) Fortran order */
  double aoc, *bd2oc;
  /* state */
  double *po, *pom, *qo, *qom, *wekpo, *entoc;
  double *xon, *dpioc, *dpiocp;
  /* box */
  double *ochom, *aipohs, *cdiffo, *cdhoc, *cdhlu;
  int *ipivch;
  /* cyclic */
  double *pch1oc, *pch2oc, *pbhoc, *aipcho, *hc1soc, *hc2soc, *hc1noc, *hc2noc;
  double hbsioc, aipbho;
  double *ocncs, *ocncn, *ocncsp, *ocncnp, *enisoc, *eninoc;
  double *ajisoc, *ajinoc, *ap3soc, *ap3noc, *ap5soc, *ap5noc;
  double txisoc, txinoc, bdrins, bdrinn;
  /* diagnostics */
  double *xinhom, *invcoef;
  /* scratch */
  double *d2p, *d4p, *dqdt, *wrk;
  /* ocean mixed layer (src/omlsubs.F), SURVEY 8 row f1 */
  int oml_on, sb_hflux, nb_hflux;
  double hmoc, toc1, toc2, st2d, st4d, ycexp, rrcpoc, tsbdy, tnbdy;
  double *sst, *sstm, *fnetoc, *wekto, *tauxo, *tauyo; /* T grid (nxto,nyto) x4, p grid x2 */
  double *omrhs, *omd2t, *omxfo;                        /* scratch */
  double cfraoc, centoc;
};

#define IX(i, j) ((size_t)((i)-1) + (size_t)nx * (size_t)((j)-1))
#define IX3(i, j, k) ((size_t)((i)-1) + (size_t)nx * ((size_t)((j)-1) + (size_t)ny * (size_t)((k)-1)))
#define M2(a, r, c) a[((r)-1) + nl * ((c)-1)] /* Fortran (r,c) of an (nl,nl) matrix */

static double *dalloc(size_t n) {
  double *p = (double *)calloc(n ? n : 1, sizeof(double));
  if (!p) { fprintf(stderr, "qgcm_oracle: out of memory\n"); abort(); }
  return p;
}

void qgo_set_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}

/* ------------------------------------------------------------------ */
/* Mixed-radix complex FFT (Stockham autosort, decimation in frequency) */
/* ------------------------------------------------------------------ */
static int factorize(int n, int *fac) {
  int nf = 0;
  static const int tr[4] = {4, 2, 3, 5}; /* same try-order as drfti1.f:16 */
  for (int t = 0; t < 4; ++t)
    while (n % tr[t] == 0) { fac[nf++] = tr[t]; n /= tr[t]; }
  for (int p = 7; n > 1; p += 2)
    while (n % p == 0) { fac[nf++] = p; n /= p; }
  return nf;
}

/* plan: per-stage twiddles and radix DFT matrices, interleaved re/im data */
typedef struct {
  int n, nf, fac[64];
  double *twr, *twi; /* per stage twiddles, concatenated */
  double *dr, *di;   /* per stage radix DFT matrices */
  size_t *two, *dfo;
} cplan;

static cplan *cplan_make(int n, int sign) {
  cplan *P = (cplan *)calloc(1, sizeof(cplan));
  P->n = n;
  P->nf = (n > 1) ? factorize(n, P->fac) : 0;
  size_t ntw = 0, ndf = 0;
  int len = n;
  P->two = (size_t *)calloc(P->nf + 1, sizeof(size_t));
  P->dfo = (size_t *)calloc(P->nf + 1, sizeof(size_t));
  for (int f = 0; f < P->nf; ++f) {
    int R = P->fac[f], m = len / R;
    P->two[f] = ntw; P->dfo[f] = ndf;
    ntw += (size_t)m * R; ndf += (size_t)R * R; len = m;
  }
  P->twr = dalloc(ntw); P->twi = dalloc(ntw); P->dr = dalloc(ndf); P->di = dalloc(ndf);
  len = n;
  for (int f = 0; f < P->nf; ++f) {
    int R = P->fac[f], m = len / R;
    for (int p = 0; p < m; ++p)
      for (int u = 0; u < R; ++u) {
        double ang = sign * TWOPI_ * (double)(((long)p * u) % len) / (double)len;
        P->twr[P->two[f] + (size_t)p * R + u] = cos(ang);
        P->twi[P->two[f] + (size_t)p * R + u] = sin(ang);
      }
    for (int u = 0; u < R; ++u)
      for (int r = 0; r < R; ++r) {
        double ang = sign * TWOPI_ * (double)((r * u) % R) / (double)R;
        P->dr[P->dfo[f] + (size_t)u * R + r] = cos(ang);
        P->di[P->dfo[f] + (size_t)u * R + r] = sin(ang);
      }
    len = m;
  }
  return P;
}

static void cplan_free(cplan *P) {
  if (!P) return;
  free(P->twr); free(P->twi); free(P->dr); free(P->di); free(P->two); free(P->dfo); free(P);
}

static void cplan_exec(const cplan *P, double *x, double *y) {
  int n = P->n;
  if (n <= 1) return;
  double *in = x, *out = y;
  int s = 1, len = n;
  for (int f = 0; f < P->nf; ++f) {
    int R = P->fac[f], m = len / R;
    const double *twr = P->twr + P->two[f], *twi = P->twi + P->two[f];
    const double *dr = P->dr + P->dfo[f], *di = P->di + P->dfo[f];
    for (int p = 0; p < m; ++p) {
      for (int q = 0; q < s; ++q) {
        double ar[32], ai[32];
        for (int r = 0; r < R; ++r) {
          ar[r] = in[2 * (q + s * (p + m * r))];
          ai[r] = in[2 * (q + s * (p + m * r)) + 1];
        }
        for (int u = 0; u < R; ++u) {
          double br = 0.0, bi = 0.0;
          for (int r = 0; r < R; ++r) {
            double c = dr[u * R + r], sn = di[u * R + r];
            br += ar[r] * c - ai[r] * sn;
            bi += ar[r] * sn + ai[r] * c;
          }
          double c = twr[(size_t)p * R + u], sn = twi[(size_t)p * R + u];
          out[2 * (q + s * (R * p + u))] = br * c - bi * sn;
          out[2 * (q + s * (R * p + u)) + 1] = br * sn + bi * c;
        }
      }
    }
    double *t = in; in = out; out = t;
    s *= R; len = m;
  }
  if (in != x) memcpy(x, in, sizeof(double) * 2 * (size_t)n);
}

/* Real transform plans producing / consuming FFTPACK's half-complex layout
 * (fft.doc, drfftf: r(1)=sum, r(2k)=Re X_k, r(2k+1)=Im X_k, r(n)=X_{n/2}). */
typedef struct {
  int n;
  cplan *fwd, *bwd; /* length n/2 (n even) or n (n odd) */
  double *wr, *wi;  /* exp(-2 pi i k / n), k=0..n/2 */
} rplan;

static rplan *rplan_make(int n) {
  rplan *P = (rplan *)calloc(1, sizeof(rplan));
  P->n = n;
  int m = (n % 2 == 0) ? n / 2 : n;
  P->fwd = cplan_make(m, -1);
  P->bwd = cplan_make(m, +1);
  P->wr = dalloc(n / 2 + 1); P->wi = dalloc(n / 2 + 1);
  for (int k = 0; k <= n / 2; ++k) {
    P->wr[k] = cos(-TWOPI_ * (double)k / (double)n);
    P->wi[k] = sin(-TWOPI_ * (double)k / (double)n);
  }
  return P;
}

static void rplan_free(rplan *P) {
  if (!P) return;
  cplan_free(P->fwd); cplan_free(P->bwd); free(P->wr); free(P->wi); free(P);
}

/* work: 4*n doubles */
static void rplan_fwd(const rplan *P, double *x, double *work) {
  int n = P->n;
  if (n == 1) return;
  if (n % 2) {
    double *z = work, *y = work + 2 * n;
    for (int j = 0; j < n; ++j) { z[2 * j] = x[j]; z[2 * j + 1] = 0.0; }
    cplan_exec(P->fwd, z, y);
    x[0] = z[0];
    for (int k = 1; k <= (n - 1) / 2; ++k) { x[2 * k - 1] = z[2 * k]; x[2 * k] = z[2 * k + 1]; }
    return;
  }
  int M = n / 2;
  double *z = work, *y = work + 2 * M;
  memcpy(z, x, sizeof(double) * n); /* z[j] = x[2j] + i x[2j+1] */
  cplan_exec(P->fwd, z, y);
  /* X[k] = Xe[k] + w^k Xo[k] */
  double *o = y; /* reuse as output staging (n+2 doubles <= 2M+... ) */
  for (int k = 0; k <= M; ++k) {
    int k1 = k % M, k2 = (M - k) % M;
    double ar = z[2 * k1], ai = z[2 * k1 + 1], br = z[2 * k2], bi = -z[2 * k2 + 1];
    double er = 0.5 * (ar + br), ei = 0.5 * (ai + bi);
    double dr = 0.5 * (ar - br), di = 0.5 * (ai - bi); /* (Z - conj Z')/2 ; Xo = that / i */
    double xor_ = di, xoi = -dr;
    double c = P->wr[k], s = P->wi[k];
    double Xr = er + (xor_ * c - xoi * s), Xi = ei + (xor_ * s + xoi * c);
    if (k == 0) o[0] = Xr;
    else if (k == M) o[n - 1] = Xr;
    else { o[2 * k - 1] = Xr; o[2 * k] = Xi; }
  }
  memcpy(x, o, sizeof(double) * n);
}

static void rplan_bwd(const rplan *P, double *x, double *work) {
  int n = P->n;
  if (n == 1) return;
  if (n % 2) {
    double *z = work, *y = work + 2 * n;
    z[0] = x[0]; z[1] = 0.0;
    for (int k = 1; k <= (n - 1) / 2; ++k) {
      z[2 * k] = x[2 * k - 1]; z[2 * k + 1] = x[2 * k];
      z[2 * (n - k)] = x[2 * k - 1]; z[2 * (n - k) + 1] = -x[2 * k];
    }
    cplan_exec(P->bwd, z, y);
    for (int j = 0; j < n; ++j) x[j] = z[2 * j];
    return;
  }
  int M = n / 2;
  double *z = work, *y = work + 2 * M;
  for (int k = 0; k < M; ++k) {
    /* X[k], conj X[M-k] */
    double ar, ai, br, bi;
    if (k == 0) { ar = x[0]; ai = 0.0; br = x[n - 1]; bi = 0.0; }
    else { ar = x[2 * k - 1]; ai = x[2 * k]; br = x[2 * (M - k) - 1]; bi = -x[2 * (M - k)]; }
    double er = 0.5 * (ar + br), ei = 0.5 * (ai + bi);
    double dr = 0.5 * (ar - br), di = 0.5 * (ai - bi);
    /* Xo = d * conj(w^k) */
    double c = P->wr[k], s = -P->wi[k];
    double xor_ = dr * c - di * s, xoi = dr * s + di * c;
    /* Z = Xe + i Xo */
    z[2 * k] = er - xoi; z[2 * k + 1] = ei + xor_;
  }
  cplan_exec(P->bwd, z, y);
  for (int j = 0; j < n; ++j) x[j] = 2.0 * z[j];
}

void qgo_rfftf(int n, double *x) {
  rplan *P = rplan_make(n);
  double *w = dalloc(4 * (size_t)n + 8);
  rplan_fwd(P, x, w);
  free(w); rplan_free(P);
}

void qgo_rfftb(int n, double *x) {
  rplan *P = rplan_make(n);
  double *w = dalloc(4 * (size_t)n + 8);
  rplan_bwd(P, x, w);
  free(w); rplan_free(P);
}

/* DST-I exactly as FFTPACK organises it (src/fftpack/newbihar/dsint.f:16-48,
 * dsinti.f:15-25): pre-twiddle into a length n+1 real sequence, real FFT,
 * running-sum post-process.  x has n+1 elements. */
typedef struct { int n; double *ws; rplan *rp; } splan;

static splan *splan_make(int n) {
  splan *S = (splan *)calloc(1, sizeof(splan));
  S->n = n;
  int np1 = n + 1, ns2 = n / 2;
  S->ws = dalloc(ns2 + 1);
  double dt = PI_ / (double)np1;
  for (int k = 1; k <= ns2; ++k) S->ws[k - 1] = 2.0 * sin((double)k * dt);
  S->rp = rplan_make(np1);
  return S;
}

static void splan_free(splan *S) {
  if (!S) return;
  free(S->ws); rplan_free(S->rp); free(S);
}

/* work: 4*(n+1)+8 doubles */
static void splan_exec(const splan *S, double *x, double *work) {
  int n = S->n;
  if (n > 2) {
    int np1 = n + 1, ns2 = n / 2;
    double x1 = x[0];
    x[0] = 0.0;
    for (int k = 1; k <= ns2; ++k) {
      double xkc = x[np1 - k - 1];
      double t1 = x1 - xkc;
      double t2 = S->ws[k - 1] * (x1 + xkc);
      x1 = x[k];
      x[k] = t1 + t2;
      x[np1 - k] = t2 - t1;
    }
    int modn = n % 2;
    if (modn != 0) x[ns2 + 1] = 4.0 * x1;
    rplan_fwd(S->rp, x, work);
    x[0] = 0.5 * x[0];
    for (int i = 3; i <= n; i += 2) {
      double xim1 = x[i - 2];
      x[i - 2] = -x[i - 1];
      x[i - 1] = x[i - 3] + xim1;
    }
    if (modn == 0) x[n - 1] = -x[n];
  } else if (n == 2) {
    const double SQRT3 = 1.73205080756887729;
    double xh = SQRT3 * (x[0] + x[1]);
    x[1] = SQRT3 * (x[0] - x[1]);
    x[0] = xh;
  } else {
    x[0] = x[0] + x[0];
  }
}

void qgo_dsint(int n, double *x) {
  splan *S = splan_make(n);
  double *w = dalloc(4 * (size_t)(n + 1) + 8);
  splan_exec(S, x, w);
  free(w); splan_free(S);
}

/* ------------------------------------------------------------------ */
/* xintp: src/intsubs.f:78-133                                          */
/* ------------------------------------------------------------------ */
double qgo_xintp(const double *valp, int nx, int ny) {
  double sump = 0.0;
  double xxs = 0.5 * valp[IX(1, 1)];
  double xxn = 0.5 * valp[IX(1, ny)];
#pragma omp parallel for schedule(static) reduction(+ : sump)
  for (int j = 2; j <= ny - 1; ++j) {
    double sumi = 0.5 * valp[IX(1, j)];
    for (int i = 2; i <= nx - 1; ++i) sumi = sumi + valp[IX(i, j)];
    sumi = sumi + 0.5 * valp[IX(nx, j)];
    sump = sump + sumi;
  }
  for (int i = 2; i <= nx - 1; ++i) {
    xxs = xxs + valp[IX(i, 1)];
    xxn = xxn + valp[IX(i, ny)];
  }
  xxs = xxs + 0.5 * valp[IX(nx, 1)];
  xxn = xxn + 0.5 * valp[IX(nx, ny)];
  return sump + 0.5 * (xxs + xxn);
}

/* ------------------------------------------------------------------ */
/* eigmod: src/eigmode.f:41-538.  The reference runs a non-symmetric    */
/* LAPACK chain on A; A = H^-1 T with T symmetric, so the same modes    */
/* follow from a symmetric Jacobi solve of H^-1/2 T H^-1/2 (SURVEY      */
/* appendix A).  Normalisation/sign/sort follow eigmode.f:310-428.      */
/* ------------------------------------------------------------------ */
static void eigmod_impl(int atmos, int nl, const double *gpr, const double *h, double fnot,
                double *aaa, double *rdm2, double *ctl2m, double *ctm2l) {
  double *S = dalloc((size_t)nl * nl), *V = dalloc((size_t)nl * nl), *lam = dalloc(nl);
#define A_(r, c) aaa[((r)-1) + nl * ((c)-1)]
#define S_(r, c) S[((r)-1) + nl * ((c)-1)]
#define V_(r, c) V[((r)-1) + nl * ((c)-1)]
  memset(aaa, 0, sizeof(double) * nl * nl);
  /* eigmode.f:131-144 */
  A_(1, 2) = -1.0 / (gpr[0] * h[0]);
  A_(1, 1) = -A_(1, 2);
  for (int k = 2; k <= nl - 1; ++k) {
    A_(k, k - 1) = -1.0 / (gpr[k - 2] * h[k - 1]);
    A_(k, k + 1) = -1.0 / (gpr[k - 1] * h[k - 1]);
    A_(k, k) = -A_(k, k - 1) - A_(k, k + 1);
  }
  A_(nl, nl - 1) = -1.0 / (gpr[nl - 2] * h[nl - 1]);
  A_(nl, nl) = -A_(nl, nl - 1);
  for (int r = 1; r <= nl; ++r)
    for (int c = 1; c <= nl; ++c) {
      S_(r, c) = 0.0;
      V_(r, c) = (r == c) ? 1.0 : 0.0;
    }
  for (int k = 1; k <= nl; ++k) S_(k, k) = A_(k, k);
  for (int k = 1; k <= nl - 1; ++k) {
    double v = -1.0 / (gpr[k - 1] * sqrt(h[k - 1] * h[k]));
    S_(k, k + 1) = v; S_(k + 1, k) = v;
  }
  /* cyclic Jacobi */
  for (int sweep = 0; sweep < 100; ++sweep) {
    double off = 0.0, diag = 0.0;
    for (int r = 1; r <= nl; ++r)
      for (int c = 1; c <= nl; ++c) {
        if (r != c) off += S_(r, c) * S_(r, c);
        else diag += S_(r, c) * S_(r, c);
      }
    if (off <= 1e-60 * diag || off == 0.0) break;
    for (int p = 1; p <= nl - 1; ++p)
      for (int q = p + 1; q <= nl; ++q) {
        if (S_(p, q) == 0.0) continue;
        double theta = (S_(q, q) - S_(p, p)) / (2.0 * S_(p, q));
        double t = ((theta >= 0) ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        double cs = 1.0 / sqrt(t * t + 1.0), sn = t * cs;
        for (int k = 1; k <= nl; ++k) {
          double skp = S_(k, p), skq = S_(k, q);
          S_(k, p) = cs * skp - sn * skq;
          S_(k, q) = sn * skp + cs * skq;
        }
        for (int k = 1; k <= nl; ++k) {
          double spk = S_(p, k), sqk = S_(q, k);
          S_(p, k) = cs * spk - sn * sqk;
          S_(q, k) = sn * spk + cs * sqk;
        }
        for (int k = 1; k <= nl; ++k) {
          double vkp = V_(k, p), vkq = V_(k, q);
          V_(k, p) = cs * vkp - sn * vkq;
          V_(k, q) = sn * vkp + cs * vkq;
        }
      }
  }
  for (int m = 1; m <= nl; ++m) lam[m - 1] = S_(m, m);
  /* sort by |lambda| ascending (eigmode.f:386-402) */
  int idx[64];
  for (int m = 0; m < nl; ++m) idx[m] = m;
  for (int m = 1; m < nl; ++m) {
    int t = idx[m], i = m - 1;
    while (i >= 0 && fabs(lam[idx[i]]) > fabs(lam[t])) { idx[i + 1] = idx[i]; --i; }
    idx[i + 1] = t;
  }
  double htotal = 0.0;
  for (int k = 0; k < nl; ++k) htotal += h[k];
  for (int m = 1; m <= nl; ++m) {
    int im = idx[m - 1] + 1;
    /* right eigenvector R = H^-1/2 v, Flierl-normalised, +ve at k=1 (eigmode.f:310-328) */
    double R[64], dotp = 0.0;
    for (int k = 1; k <= nl; ++k) { R[k - 1] = V_(k, im) / sqrt(h[k - 1]); dotp += h[k - 1] * R[k - 1] * R[k - 1]; }
    double fl = sqrt(htotal / dotp);
    if (R[0] < 0.0) fl = -fl;
    if (atmos) {
      /* case 'Atmosphere': no Flierl normalisation (eigmode.f:309); the right eigenvectors stay as LAPACK's DTREVC
       * leaves them, i.e. scaled so that the component of largest magnitude has magnitude 1 (eigmode.f:283-296). Their
       * sign comes out of the Schur vectors and cannot be restated; positive at k = 1 reproduces the reference for the
       * example configurations, and pa, qa do not depend on scale or sign of a mode (cm2l * cl2m = 1). */
      double mx = 0.0;
      for (int k = 0; k < nl; ++k) mx = fmax(mx, fabs(R[k]));
      fl = ((R[0] < 0.0) ? -1.0 : 1.0) / mx;
    }
    for (int k = 1; k <= nl; ++k) R[k - 1] *= fl;
    /* left eigenvector L = H R (up to scale): cl2m(m,k) = L(k)/(L.R)  (eigmode.f:420-428) */
    double LR = 0.0;
    for (int k = 1; k <= nl; ++k) LR += h[k - 1] * R[k - 1] * R[k - 1];
    for (int k = 1; k <= nl; ++k) {
      ctl2m[(k - 1) + nl * (m - 1)] = h[k - 1] * R[k - 1] / LR; /* ctl2m(k,m) */
      ctm2l[(m - 1) + nl * (k - 1)] = R[k - 1];                 /* ctm2l(m,k) */
    }
    rdm2[m - 1] = (m == 1) ? 0.0 : fnot * fnot * fabs(lam[im - 1]);
  }
  free(S); free(V); free(lam);
#undef A_
#undef S_
#undef V_
}

void qgo_eigmod(int nl, const double *gpr, const double *h, double fnot,
                double *amat, double *rdm2, double *ctl2m, double *ctm2l) {
  eigmod_impl(0, nl, gpr, h, fnot, amat, rdm2, ctl2m, ctm2l);
}
void qgo_eigmod_atmos(int nl, const double *gpr, const double *h, double fnot,
                      double *amat, double *rdm2, double *ctl2m, double *ctm2l) {
  eigmod_impl(1, nl, gpr, h, fnot, amat, rdm2, ctl2m, ctm2l);
}

/* ------------------------------------------------------------------ */
/* Helmholtz solvers: src/ocisubs.F:415-512 (box), 521-618 (cyclic)     */
/* ------------------------------------------------------------------ */
static void thomas_column(int ny, double aoc, double bk, const double *col, size_t stride,
                          double *uvec, double *gam, double ftnorm, double *out) {
  /* ocisubs.F:470-487 ; arrays indexed 2..ny-1 */
  double betinv = 1.0 / bk;
  uvec[2] = col[stride * (2 - 1)] * betinv;
  for (int j = 3; j <= ny - 1; ++j) {
    gam[j] = aoc * betinv;
    betinv = 1.0 / (bk - aoc * gam[j]);
    uvec[j] = (col[stride * (size_t)(j - 1)] - aoc * uvec[j - 1]) * betinv;
  }
  for (int j = ny - 2; j >= 2; --j) uvec[j] = uvec[j] - gam[j + 1] * uvec[j + 1];
  for (int j = 2; j <= ny - 1; ++j) out[stride * (size_t)(j - 1)] = ftnorm * uvec[j];
}

static void hsbxoc(qgo_ctx *c, double *wrk, const double *boc) {
  int nx = c->nx, ny = c->ny, nxt = c->nxt;
  double ftnorm = 0.5 / nxt;
  splan *S = splan_make(nxt - 1);
#pragma omp parallel
  {
    double *work = dalloc(4 * (size_t)(nxt + 1) + 8);
    double *uvec = dalloc(ny + 2), *gam = dalloc(ny + 2);
#pragma omp for schedule(static)
    for (int j = 2; j <= ny - 1; ++j) splan_exec(S, &wrk[IX(2, j)], work);
#pragma omp for schedule(static)
    for (int i = 2; i <= nx - 1; ++i)
      thomas_column(ny, c->aoc, boc[i - 2], &wrk[IX(i, 1)], (size_t)nx, uvec, gam, ftnorm, &wrk[IX(i, 1)]);
#pragma omp for schedule(static)
    for (int j = 2; j <= ny - 1; ++j) {
      splan_exec(S, &wrk[IX(2, j)], work);
      wrk[IX(1, j)] = 0.0;
      wrk[IX(nx, j)] = 0.0;
    }
    free(work); free(uvec); free(gam);
  }
  for (int i = 1; i <= nx; ++i) { wrk[IX(i, 1)] = 0.0; wrk[IX(i, ny)] = 0.0; }
  splan_free(S);
}

static void hscyoc(qgo_ctx *c, double *wrk, const double *boc) {
  int nx = c->nx, ny = c->ny, nxt = c->nxt;
  double ftnorm = 1.0 / nxt;
  rplan *P = rplan_make(nxt);
#pragma omp parallel
  {
    double *work = dalloc(4 * (size_t)nxt + 8);
    double *uvec = dalloc(ny + 2), *gam = dalloc(ny + 2);
#pragma omp for schedule(static)
    for (int j = 2; j <= ny - 1; ++j) rplan_fwd(P, &wrk[IX(1, j)], work);
#pragma omp for schedule(static)
    for (int i = 1; i <= nxt; ++i)
      thomas_column(ny, c->aoc, boc[i - 1], &wrk[IX(i, 1)], (size_t)nx, uvec, gam, ftnorm, &wrk[IX(i, 1)]);
#pragma omp for schedule(static)
    for (int j = 2; j <= ny - 1; ++j) {
      rplan_bwd(P, &wrk[IX(1, j)], work);
      wrk[IX(nx, j)] = wrk[IX(1, j)];
    }
    free(work); free(uvec); free(gam);
  }
  for (int i = 1; i <= nx; ++i) { wrk[IX(i, 1)] = 0.0; wrk[IX(i, ny)] = 0.0; }
  rplan_free(P);
}

void qgo_helmholtz(qgo_ctx *c, double *wrk, const double *boc) {
  if (c->cyclic) hscyoc(c, wrk, boc);
  else hsbxoc(c, wrk, boc);
}

/* ------------------------------------------------------------------ */
/* small dense helpers replacing LAPACK on the (nlo-1)x(nlo-1) system   */
/* DGETRF (conhoms.F:627), DGETRS + DGERFS (ocisubs.F:359-370)          */
/* ------------------------------------------------------------------ */
static int lu_factor(int n, double *a, int *piv) {
  for (int k = 0; k < n; ++k) {
    int p = k;
    double mx = fabs(a[k + n * k]);
    for (int i = k + 1; i < n; ++i)
      if (fabs(a[i + n * k]) > mx) { mx = fabs(a[i + n * k]); p = i; }
    piv[k] = p;
    if (mx == 0.0) return k + 1;
    if (p != k)
      for (int j = 0; j < n; ++j) { double t = a[k + n * j]; a[k + n * j] = a[p + n * j]; a[p + n * j] = t; }
    for (int i = k + 1; i < n; ++i) {
      a[i + n * k] /= a[k + n * k];
      for (int j = k + 1; j < n; ++j) a[i + n * j] -= a[i + n * k] * a[k + n * j];
    }
  }
  return 0;
}

static void lu_solve(int n, const double *lu, const int *piv, double *b) {
  for (int k = 0; k < n; ++k) {
    if (piv[k] != k) { double t = b[k]; b[k] = b[piv[k]]; b[piv[k]] = t; }
  }
  for (int k = 0; k < n; ++k)
    for (int i = k + 1; i < n; ++i) b[i] -= lu[i + n * k] * b[k];
  for (int k = n - 1; k >= 0; --k) {
    b[k] /= lu[k + n * k];
    for (int i = 0; i < k; ++i) b[i] -= lu[i + n * k] * b[k];
  }
}

/* iterative refinement with DGERFS's stopping rule (ITMAX = 5) */
static void lu_refine(int n, const double *a, const double *lu, const int *piv, const double *b, double *x) {
  const double eps = 1.1102230246251565e-16, safmin = 2.2250738585072014e-308;
  double safe1 = (n + 1) * safmin, safe2 = safe1 / eps;
  double lstres = 3.0;
  double r[64], w[64];
  for (int count = 1;; ++count) {
    for (int i = 0; i < n; ++i) {
      double s = b[i];
      for (int j = 0; j < n; ++j) s -= a[i + n * j] * x[j];
      r[i] = s;
      double t = fabs(b[i]);
      for (int j = 0; j < n; ++j) t += fabs(a[i + n * j]) * fabs(x[j]);
      w[i] = t;
    }
    double berr = 0.0;
    for (int i = 0; i < n; ++i) {
      double v = (w[i] > safe2) ? fabs(r[i]) / w[i] : (fabs(r[i]) + safe1) / (w[i] + safe1);
      if (v > berr) berr = v;
    }
    if (berr > eps && 2.0 * berr <= lstres && count <= 5) {
      lu_solve(n, lu, piv, r);
      for (int i = 0; i < n; ++i) x[i] += r[i];
      lstres = berr;
    } else break;
  }
}

/* ------------------------------------------------------------------ */
/* homsol: src/conhoms.F:376-641                                        */
/* ------------------------------------------------------------------ */
static void homsol(qgo_ctx *c) {
  int nx = c->nx, ny = c->ny, nl = c->nl, nxt = c->nxt;
  double *boc = dalloc(nxt);
  if (c->cyclic) {
    /* conhoms.F:384-389 */
    for (int j = 1; j <= ny; ++j) c->pbhoc[j - 1] = (double)(ny - j) / (double)(ny - 1);
    c->hbsioc = c->ylo / c->xlo;
    c->aipbho = 0.5 * c->xlo * c->ylo;
    double *w1 = dalloc((size_t)nx * ny), *w2 = dalloc((size_t)nx * ny);
    for (int m = 1; m <= nl - 1; ++m) {
      for (int i = 0; i < nxt; ++i) boc[i] = c->bd2oc[i] - c->rdm2oc[m];
      double *p1 = c->pch1oc + (size_t)ny * (m - 1), *p2 = c->pch2oc + (size_t)ny * (m - 1);
      /* ypo(nypo)-ypo(j) = (nypo-j)*dyo ; conhoms.F:424-431 */
      for (int j = 1; j <= ny; ++j) {
        double ypoj = c->yporel[j - 1], ypon = c->yporel[ny - 1], ypo1 = c->yporel[0];
        if (c->atmos) { /* conhoms.F:664-665 uses ypa(j) = (j-1)*dya itself, src/q-gcm.F:402 */
          ypoj = (j - 1) * c->dyo; ypon = (ny - 1) * c->dyo; ypo1 = 0 * c->dyo;
        }
        p1[j - 1] = (ypon - ypoj) / c->ylo;
        p2[j - 1] = (ypoj - ypo1) / c->ylo;
        for (int i = 1; i <= nx; ++i) { w1[IX(i, j)] = p1[j - 1]; w2[IX(i, j)] = p2[j - 1]; }
      }
      hscyoc(c, w1, boc);
      hscyoc(c, w2, boc);
      for (int j = 1; j <= ny; ++j) {
        for (int i = 1; i <= nx; ++i) {
          w1[IX(i, j)] = p1[j - 1] + c->rdm2oc[m] * w1[IX(i, j)];
          w2[IX(i, j)] = p2[j - 1] + c->rdm2oc[m] * w2[IX(i, j)];
        }
        p1[j - 1] = w1[IX(1, j)];
        p2[j - 1] = w2[IX(1, j)];
      }
      double a1 = qgo_xintp(w1, nx, ny), a2 = qgo_xintp(w2, nx, ny);
      c->aipcho[m - 1] = 0.5 * (a1 + a2) * c->dxo * c->dyo;
      /* conhoms.F:514-534 */
      double dyo = c->dyo, rd = c->rdm2oc[m];
      double p1ys = (p1[1] - p1[0]) / dyo, p2ys = (p2[1] - p2[0]) / dyo;
      double p1yn = (p1[ny - 1] - p1[ny - 2]) / dyo, p2yn = (p2[ny - 1] - p2[ny - 2]) / dyo;
      p1ys = -p1ys + 0.5 * dyo * rd * p1[0];
      p2ys = -p2ys + 0.5 * dyo * rd * p2[0];
      p1yn = p1yn + 0.5 * dyo * rd * p1[ny - 1];
      p2yn = p2yn + 0.5 * dyo * rd * p2[ny - 1];
      p1ys = c->xlo * p1ys; p2ys = c->xlo * p2ys; p1yn = c->xlo * p1yn; p2yn = c->xlo * p2yn;
      double det = p1ys * p2yn - p2ys * p1yn;
      c->hc1soc[m - 1] = p1ys / det; c->hc2soc[m - 1] = p2ys / det;
      c->hc1noc[m - 1] = p1yn / det; c->hc2noc[m - 1] = p2yn / det;
    }
    free(w1); free(w2);
  } else {
    for (int m = 1; m <= nl - 1; ++m) {
      for (int i = 0; i < nxt; ++i) boc[i] = c->bd2oc[i] - c->rdm2oc[m];
      double *oh = c->ochom + (size_t)nx * ny * (m - 1);
      for (size_t t = 0; t < (size_t)nx * ny; ++t) oh[t] = 1.0;
      hsbxoc(c, oh, boc);
      for (size_t t = 0; t < (size_t)nx * ny; ++t) oh[t] = 1.0 + c->rdm2oc[m] * oh[t];
      c->aipohs[m - 1] = qgo_xintp(oh, nx, ny) * c->dxo * c->dyo;
    }
    /* conhoms.F:602-611 ; cdiffo(m,k) stored (nl, nl-1), cdhoc(k,m) stored (nl-1,nl-1) */
    int n1 = nl - 1;
    for (int k = 1; k <= nl - 1; ++k) {
      for (int m = 1; m <= nl; ++m)
        c->cdiffo[(m - 1) + nl * (k - 1)] = M2(c->ctm2loc, m, k + 1) - M2(c->ctm2loc, m, k);
      for (int m = 1; m <= nl - 1; ++m) {
        double v = (M2(c->ctm2loc, m + 1, k + 1) - M2(c->ctm2loc, m + 1, k)) * c->aipohs[m - 1];
        c->cdhoc[(k - 1) + n1 * (m - 1)] = v;
        c->cdhlu[(k - 1) + n1 * (m - 1)] = v;
      }
    }
    if (lu_factor(n1, c->cdhlu, c->ipivch)) { fprintf(stderr, "qgcm_oracle: singular cdhoc\n"); abort(); }
  }
  free(boc);
}

/* ------------------------------------------------------------------ */
static qgo_ctx *create_impl(int nxpo, int nypo, int nlo, int cyclic, double fnot, double beta,
                    double dxo, double dto, double delek, double bccooc,
                    const double *ah2oc, const double *ah4oc, const double *hoc,
                    const double *gpoc, const double *yporel, const double *ddynoc, int atmos) {
  qgo_ctx *c = (qgo_ctx *)calloc(1, sizeof(qgo_ctx));
  int nx = nxpo, ny = nypo, nl = nlo;
  c->atmos = atmos;
  c->nx = nx; c->ny = ny; c->nl = nl; c->cyclic = cyclic; c->nxt = nx - 1;
  c->fnot = fnot; c->beta = beta; c->dxo = dxo; c->dyo = dxo; c->dxom2 = 1.0 / (dxo * dxo);
  c->dto = dto; c->tdto = 2.0 * dto; c->delek = delek; c->bccooc = bccooc;
  c->xlo = (nx - 1) * dxo; c->ylo = (ny - 1) * c->dyo;
  size_t N = (size_t)nx * ny;
  c->ah2oc = dalloc(nl); c->ah4oc = dalloc(nl); c->hoc = dalloc(nl); c->gpoc = dalloc(nl);
  memcpy(c->ah2oc, ah2oc, sizeof(double) * nl); memcpy(c->ah4oc, ah4oc, sizeof(double) * nl);
  memcpy(c->hoc, hoc, sizeof(double) * nl); memcpy(c->gpoc, gpoc, sizeof(double) * (nl - 1));
  c->yporel = dalloc(ny); memcpy(c->yporel, yporel, sizeof(double) * ny);
  c->ddynoc = dalloc(N);
  if (ddynoc) memcpy(c->ddynoc, ddynoc, sizeof(double) * N);
  c->amatoc = dalloc((size_t)nl * nl); c->ctl2moc = dalloc((size_t)nl * nl);
  c->ctm2loc = dalloc((size_t)nl * nl); c->rdm2oc = dalloc(nl);
  eigmod_impl(atmos, nl, c->gpoc, c->hoc, fnot, c->amatoc, c->rdm2oc, c->ctl2moc, c->ctm2loc);
  /* src/q-gcm.F:932-954 */
  int nxt = c->nxt;
  c->aoc = 1.0 / (c->dyo * c->dyo);
  c->bd2oc = dalloc(nxt);
  if (cyclic) {
    for (int i = 2; i <= nxt / 2; ++i) {
      int i1 = 2 * i - 1;
      c->bd2oc[i1 - 2] = -2.0 * c->aoc + 2.0 * c->dxom2 * (cos((i - 1) * TWOPI_ / nxt) - 1.0);
      c->bd2oc[i1 - 1] = c->bd2oc[i1 - 2];
    }
    c->bd2oc[0] = -2.0 * c->aoc;
    c->bd2oc[nxt - 1] = -2.0 * c->aoc - 4.0 * c->dxom2;
  } else {
    for (int i = 2; i <= nxt; ++i)
      c->bd2oc[i - 2] = -2.0 * c->aoc + 2.0 * c->dxom2 * (cos((i - 1) * PI_ / nxt) - 1.0);
    c->bd2oc[nxt - 1] = 0.0;
  }
  c->po = dalloc(N * nl); c->pom = dalloc(N * nl); c->qo = dalloc(N * nl); c->qom = dalloc(N * nl);
  c->wekpo = dalloc(N); c->entoc = dalloc(N);
  c->xon = dalloc(nl); c->dpioc = dalloc(nl); c->dpiocp = dalloc(nl);
  c->ochom = dalloc(cyclic ? 1 : N * (nl - 1)); c->aipohs = dalloc(nl);
  c->cdiffo = dalloc((size_t)nl * nl); c->cdhoc = dalloc((size_t)nl * nl); c->cdhlu = dalloc((size_t)nl * nl);
  c->ipivch = (int *)calloc(nl, sizeof(int));
  c->pch1oc = dalloc((size_t)ny * nl); c->pch2oc = dalloc((size_t)ny * nl); c->pbhoc = dalloc(ny);
  c->aipcho = dalloc(nl); c->hc1soc = dalloc(nl); c->hc2soc = dalloc(nl); c->hc1noc = dalloc(nl); c->hc2noc = dalloc(nl);
  c->ocncs = dalloc(nl); c->ocncn = dalloc(nl); c->ocncsp = dalloc(nl); c->ocncnp = dalloc(nl);
  c->enisoc = dalloc(nl); c->eninoc = dalloc(nl);
  c->ajisoc = dalloc(nl); c->ajinoc = dalloc(nl); c->ap3soc = dalloc(nl); c->ap3noc = dalloc(nl);
  c->ap5soc = dalloc(nl); c->ap5noc = dalloc(nl);
  c->xinhom = dalloc(nl); c->invcoef = dalloc(2 * nl + 1);
  c->d2p = dalloc(N); c->d4p = dalloc(N); c->dqdt = dalloc(N * nl); c->wrk = dalloc(N * nl);
  homsol(c);
  return c;
}

qgo_ctx *qgo_create(int nxpo, int nypo, int nlo, int cyclic, double fnot, double beta,
                    double dxo, double dto, double delek, double bccooc,
                    const double *ah2oc, const double *ah4oc, const double *hoc,
                    const double *gpoc, const double *yporel, const double *ddynoc) {
  return create_impl(nxpo, nypo, nlo, cyclic, fnot, beta, dxo, dto, delek, bccooc, ah2oc, ah4oc, hoc, gpoc, yporel,
                     ddynoc, 0);
}

/* the atmospheric channel: grid (nxpa, nypa, nla), dxa, dta, bccoat, ah4at, hat, gpat, yparel, ddynat */
qgo_ctx *qgo_create_atmos(int nxpa, int nypa, int nla, double fnot, double beta, double dxa, double dta,
                          double bccoat, const double *ah4at, const double *hat, const double *gpat,
                          const double *yparel, const double *ddynat) {
  double zero[64] = {0};
  return create_impl(nxpa, nypa, nla, 1, fnot, beta, dxa, dta, 0.0, bccoat, zero, ah4at, hat, gpat, yparel, ddynat, 1);
}

void qgo_destroy(qgo_ctx *c) {
  if (!c) return;
  double *ptrs[] = {c->ah2oc, c->ah4oc, c->hoc, c->gpoc, c->yporel, c->ddynoc, c->r_spl, c->amatoc, c->ctl2moc,
                    c->ctm2loc, c->rdm2oc, c->bd2oc, c->po, c->pom, c->qo, c->qom, c->wekpo, c->entoc,
                    c->xon, c->dpioc, c->dpiocp, c->ochom, c->aipohs, c->cdiffo, c->cdhoc, c->cdhlu,
                    c->pch1oc, c->pch2oc, c->pbhoc, c->aipcho, c->hc1soc, c->hc2soc, c->hc1noc, c->hc2noc,
                    c->ocncs, c->ocncn, c->ocncsp, c->ocncnp, c->enisoc, c->eninoc, c->ajisoc, c->ajinoc,
                    c->ap3soc, c->ap3noc, c->ap5soc, c->ap5noc, c->xinhom, c->invcoef, c->d2p, c->d4p,
                    c->dqdt, c->wrk, c->sst, c->sstm, c->fnetoc, c->wekto, c->tauxo, c->tauyo,
                    c->omrhs, c->omd2t, c->omxfo};
  for (size_t i = 0; i < sizeof(ptrs) / sizeof(ptrs[0]); ++i) free(ptrs[i]);
  free(c->ipivch);
  free(c);
}

/* ------------------------------------------------------------------ */
/* qcomp / merqcy: src/vorsubs.F:49-239                                 */
/* ------------------------------------------------------------------ */
static void qcomp(qgo_ctx *c, double *q, const double *p) {
  int nx = c->nx, ny = c->ny, nl = c->nl;
  double dx2fac = c->dxom2 / c->fnot, fnot = c->fnot;
  const double *aaa = c->amatoc;
  for (int k = 1; k <= nl; ++k) {
#pragma omp parallel for schedule(static)
    for (int j = 2; j <= ny - 1; ++j) {
      double betay = c->beta * c->yporel[j - 1];
      for (int i = 2; i <= nx - 1; ++i) {
        double lap = dx2fac * (p[IX3(i, j - 1, k)] + p[IX3(i - 1, j, k)] + p[IX3(i + 1, j, k)] +
                               p[IX3(i, j + 1, k)] - 4.0 * p[IX3(i, j, k)]) + betay;
        double ap;
        if (k == 1) ap = M2(aaa, 1, 1) * p[IX3(i, j, 1)] + M2(aaa, 1, 2) * p[IX3(i, j, 2)];
        else if (k == nl) ap = M2(aaa, nl, nl - 1) * p[IX3(i, j, nl - 1)] + M2(aaa, nl, nl) * p[IX3(i, j, nl)];
        else ap = M2(aaa, k, k - 1) * p[IX3(i, j, k - 1)] + M2(aaa, k, k) * p[IX3(i, j, k)] +
                  M2(aaa, k, k + 1) * p[IX3(i, j, k + 1)];
        q[IX3(i, j, k)] = lap - fnot * ap;
      }
    }
  }
  const int kt = c->atmos ? 1 : nl; /* layer that feels the topography: last argument of qcomp, src/q-gcm.F:719-741 */
  for (int j = 2; j <= ny - 1; ++j)
    for (int i = 2; i <= nx - 1; ++i) q[IX3(i, j, kt)] = q[IX3(i, j, kt)] + c->ddynoc[IX(i, j)];
}

static void merqcy(qgo_ctx *c, double *q, const double *p) {
  int nx = c->nx, ny = c->ny, nl = c->nl;
  double dx2fac = c->dxom2 / c->fnot, fnot = c->fnot;
  const double *aaa = c->amatoc;
  for (int k = 1; k <= nl; ++k)
    for (int j = 2; j <= ny - 1; ++j) {
      double betay = c->beta * c->yporel[j - 1];
      double lap = dx2fac * (p[IX3(1, j - 1, k)] + p[IX3(nx - 1, j, k)] + p[IX3(2, j, k)] +
                             p[IX3(1, j + 1, k)] - 4.0 * p[IX3(1, j, k)]) + betay;
      double ap;
      if (k == 1) ap = M2(aaa, 1, 1) * p[IX3(1, j, 1)] + M2(aaa, 1, 2) * p[IX3(1, j, 2)];
      else if (k == nl) ap = M2(aaa, nl, nl - 1) * p[IX3(1, j, nl - 1)] + M2(aaa, nl, nl) * p[IX3(1, j, nl)];
      else ap = M2(aaa, k, k - 1) * p[IX3(1, j, k - 1)] + M2(aaa, k, k) * p[IX3(1, j, k)] +
                M2(aaa, k, k + 1) * p[IX3(1, j, k + 1)];
      q[IX3(1, j, k)] = lap - fnot * ap;
      q[IX3(nx, j, k)] = q[IX3(1, j, k)];
    }
  const int kt = c->atmos ? 1 : nl;
  for (int j = 2; j <= ny - 1; ++j) {
    q[IX3(1, j, kt)] = q[IX3(1, j, kt)] + c->ddynoc[IX(1, j)];
    q[IX3(nx, j, kt)] = q[IX3(1, j, kt)];
  }
}

/* atqzbd: src/vorsubs.F:396-480.  Zonal boundaries only; topography in layer 1.  The southern value of the top
 * layer reads pa(i,2,nla) where every other layer reads the boundary point pa(i,1,k) (src/vorsubs.F:470) - kept
 * as written, the reference's results depend on it. */
static void atqzbd(qgo_ctx *c, double *qa, const double *pa) {
  int nx = c->nx, ny = c->ny, nl = c->nl;
  double fnot = c->fnot;
  const double *aaa = c->amatoc;
  double zbfaca = c->bccooc * c->dxom2 / (0.5 * c->bccooc + 1.0) / fnot;
  double betays = c->beta * c->yporel[0], betayn = c->beta * c->yporel[ny - 1];
  double f0Ac = fnot * M2(aaa, 1, 1), f0Ap = fnot * M2(aaa, 1, 2), f0Am;
  for (int i = 1; i <= nx; ++i) {
    qa[IX3(i, 1, 1)] = zbfaca * (pa[IX3(i, 2, 1)] - pa[IX3(i, 1, 1)]) - (f0Ac * pa[IX3(i, 1, 1)] + f0Ap * pa[IX3(i, 1, 2)]) +
                       betays + c->ddynoc[IX(i, 1)];
    qa[IX3(i, ny, 1)] = zbfaca * (pa[IX3(i, ny - 1, 1)] - pa[IX3(i, ny, 1)]) -
                        (f0Ac * pa[IX3(i, ny, 1)] + f0Ap * pa[IX3(i, ny, 2)]) + betayn + c->ddynoc[IX(i, ny)];
  }
  for (int k = 2; k <= nl - 1; ++k) {
    f0Am = fnot * M2(aaa, k, k - 1); f0Ac = fnot * M2(aaa, k, k); f0Ap = fnot * M2(aaa, k, k + 1);
    for (int i = 1; i <= nx; ++i) {
      qa[IX3(i, 1, k)] = zbfaca * (pa[IX3(i, 2, k)] - pa[IX3(i, 1, k)]) -
                         (f0Am * pa[IX3(i, 1, k - 1)] + f0Ac * pa[IX3(i, 1, k)] + f0Ap * pa[IX3(i, 1, k + 1)]) + betays;
      qa[IX3(i, ny, k)] = zbfaca * (pa[IX3(i, ny - 1, k)] - pa[IX3(i, ny, k)]) -
                          (f0Am * pa[IX3(i, ny, k - 1)] + f0Ac * pa[IX3(i, ny, k)] + f0Ap * pa[IX3(i, ny, k + 1)]) + betayn;
    }
  }
  f0Am = fnot * M2(aaa, nl, nl - 1); f0Ac = fnot * M2(aaa, nl, nl);
  for (int i = 1; i <= nx; ++i) {
    qa[IX3(i, 1, nl)] = zbfaca * (pa[IX3(i, 2, nl)] - pa[IX3(i, 1, nl)]) -
                        (f0Am * pa[IX3(i, 1, nl - 1)] + f0Ac * pa[IX3(i, 2, nl)]) + betays; /* sic: row 2, vorsubs.F:470 */
    qa[IX3(i, ny, nl)] = zbfaca * (pa[IX3(i, ny - 1, nl)] - pa[IX3(i, ny, nl)]) -
                         (f0Am * pa[IX3(i, ny, nl - 1)] + f0Ac * pa[IX3(i, ny, nl)]) + betayn;
  }
}

/* ocqbdy: src/vorsubs.F:245-388 */
static void ocqbdy(qgo_ctx *c, double *qo, const double *po) {
  if (c->atmos) { atqzbd(c, qo, po); return; }
  int nx = c->nx, ny = c->ny, nl = c->nl;
  double fnot = c->fnot;
  const double *aaa = c->amatoc;
  double bcfaco = c->bccooc * c->dxom2 / (0.5 * c->bccooc + 1.0) / fnot;
  double betays = c->beta * c->yporel[0], betayn = c->beta * c->yporel[ny - 1];
  for (int k = 1; k <= nl; ++k) {
    double f0Am = (k > 1) ? fnot * M2(aaa, k, k - 1) : 0.0;
    double f0Ac = fnot * M2(aaa, k, k);
    double f0Ap = (k < nl) ? fnot * M2(aaa, k, k + 1) : 0.0;
#define APSUM(i, j)                                                                         \
  ((k == 1) ? (f0Ac * po[IX3(i, j, 1)] + f0Ap * po[IX3(i, j, 2)])                           \
            : (k == nl) ? (f0Am * po[IX3(i, j, nl - 1)] + f0Ac * po[IX3(i, j, nl)])         \
                        : (f0Am * po[IX3(i, j, k - 1)] + f0Ac * po[IX3(i, j, k)] + f0Ap * po[IX3(i, j, k + 1)]))
    for (int i = 1; i <= nx; ++i) {
      double qs = bcfaco * (po[IX3(i, 2, k)] - po[IX3(i, 1, k)]) - APSUM(i, 1) + betays;
      double qn = bcfaco * (po[IX3(i, ny - 1, k)] - po[IX3(i, ny, k)]) - APSUM(i, ny) + betayn;
      if (k == nl) { qs = qs + c->ddynoc[IX(i, 1)]; qn = qn + c->ddynoc[IX(i, ny)]; }
      qo[IX3(i, 1, k)] = qs;
      qo[IX3(i, ny, k)] = qn;
    }
    if (!c->cyclic) {
      for (int j = 2; j <= ny - 1; ++j) {
        double betay = c->beta * c->yporel[j - 1];
        double qw = bcfaco * (po[IX3(2, j, k)] - po[IX3(1, j, k)]) - APSUM(1, j) + betay;
        double qe = bcfaco * (po[IX3(nx - 1, j, k)] - po[IX3(nx, j, k)]) - APSUM(nx, j) + betay;
        if (k == nl) { qw = qw + c->ddynoc[IX(1, j)]; qe = qe + c->ddynoc[IX(nx, j)]; }
        qo[IX3(1, j, k)] = qw;
        qo[IX3(nx, j, k)] = qe;
      }
    }
#undef APSUM
  }
}

void qgo_ocqbdy(qgo_ctx *c) { ocqbdy(c, c->qo, c->po); }

/* constr: src/conhoms.F:93-193 */
static void constr(qgo_ctx *c) {
  int nx = c->nx, ny = c->ny, nl = c->nl;
  size_t N = (size_t)nx * ny;
  double *w1 = c->d2p, *w2 = c->d4p;
  for (int k = 1; k <= nl - 1; ++k) {
    if (c->atmos) /* dpiat: pa(k) - pa(k+1), conhoms.F:205-216 */
      for (size_t t = 0; t < N; ++t) {
        w1[t] = c->pom[t + N * (k - 1)] - c->pom[t + N * k];
        w2[t] = c->po[t + N * (k - 1)] - c->po[t + N * k];
      }
    else
    for (size_t t = 0; t < N; ++t) {
      w1[t] = c->pom[t + N * k] - c->pom[t + N * (k - 1)];
      w2[t] = c->po[t + N * k] - c->po[t + N * (k - 1)];
    }
    c->dpiocp[k - 1] = qgo_xintp(w1, nx, ny) * c->dxo * c->dyo;
    c->dpioc[k - 1] = qgo_xintp(w2, nx, ny) * c->dxo * c->dyo;
  }
  if (!c->cyclic) return;
  double opins[64], opinn[64], opinsp[64], opinnp[64];
  const double *po = c->po, *pom = c->pom;
  double dxo = c->dxo, dyo = c->dyo;
  for (int k = 1; k <= nl; ++k) {
    opinsp[k] = 0.5 * pom[IX3(1, 1, k)]; opinnp[k] = 0.5 * pom[IX3(1, ny, k)];
    opins[k] = 0.5 * po[IX3(1, 1, k)]; opinn[k] = 0.5 * po[IX3(1, ny, k)];
    double csp = 0.5 * (pom[IX3(1, 2, k)] - pom[IX3(1, 1, k)]);
    double cnp = 0.5 * (pom[IX3(1, ny, k)] - pom[IX3(1, ny - 1, k)]);
    double cs = 0.5 * (po[IX3(1, 2, k)] - po[IX3(1, 1, k)]);
    double cn = 0.5 * (po[IX3(1, ny, k)] - po[IX3(1, ny - 1, k)]);
    for (int i = 2; i <= nx - 1; ++i) {
      opinsp[k] += pom[IX3(i, 1, k)]; opinnp[k] += pom[IX3(i, ny, k)];
      opins[k] += po[IX3(i, 1, k)]; opinn[k] += po[IX3(i, ny, k)];
      csp += (pom[IX3(i, 2, k)] - pom[IX3(i, 1, k)]);
      cnp += (pom[IX3(i, ny, k)] - pom[IX3(i, ny - 1, k)]);
      cs += (po[IX3(i, 2, k)] - po[IX3(i, 1, k)]);
      cn += (po[IX3(i, ny, k)] - po[IX3(i, ny - 1, k)]);
    }
    opinsp[k] += 0.5 * pom[IX3(nx, 1, k)]; opinnp[k] += 0.5 * pom[IX3(nx, ny, k)];
    opins[k] += 0.5 * po[IX3(nx, 1, k)]; opinn[k] += 0.5 * po[IX3(nx, ny, k)];
    csp += 0.5 * (pom[IX3(nx, 2, k)] - pom[IX3(nx, 1, k)]);
    cnp += 0.5 * (pom[IX3(nx, ny, k)] - pom[IX3(nx, ny - 1, k)]);
    cs += 0.5 * (po[IX3(nx, 2, k)] - po[IX3(nx, 1, k)]);
    cn += 0.5 * (po[IX3(nx, ny, k)] - po[IX3(nx, ny - 1, k)]);
    c->ocncsp[k - 1] = csp * (dxo / dyo); c->ocncnp[k - 1] = cnp * (dxo / dyo);
    c->ocncs[k - 1] = cs * (dxo / dyo); c->ocncn[k - 1] = cn * (dxo / dyo);
    opinsp[k] *= dxo; opinnp[k] *= dxo; opins[k] *= dxo; opinn[k] *= dxo;
  }
  for (int k = 1; k <= nl; ++k) {
    double apsp = 0, apnp = 0, aps = 0, apn = 0;
    for (int j = 1; j <= nl; ++j) {
      apsp += M2(c->amatoc, k, j) * opinsp[j]; apnp += M2(c->amatoc, k, j) * opinnp[j];
      aps += M2(c->amatoc, k, j) * opins[j]; apn += M2(c->amatoc, k, j) * opinn[j];
    }
    double f = 0.5 * dyo * c->fnot * c->fnot;
    c->ocncsp[k - 1] = -c->ocncsp[k - 1] + f * apsp;
    c->ocncnp[k - 1] = c->ocncnp[k - 1] + f * apnp;
    c->ocncs[k - 1] = -c->ocncs[k - 1] + f * aps;
    c->ocncn[k - 1] = c->ocncn[k - 1] + f * apn;
  }
}

void qgo_set_p(qgo_ctx *c, const double *po, const double *pom) {
  size_t n = (size_t)c->nx * c->ny * c->nl;
  memcpy(c->po, po, sizeof(double) * n);
  memcpy(c->pom, pom, sizeof(double) * n);
  constr(c);
  qcomp(c, c->qo, c->po);
  qcomp(c, c->qom, c->pom);
  ocqbdy(c, c->qo, c->po);
  ocqbdy(c, c->qom, c->pom);
  if (c->cyclic) { merqcy(c, c->qo, c->po); merqcy(c, c->qom, c->pom); }
}

void qgo_set_state(qgo_ctx *c, const double *po, const double *pom, const double *qo, const double *qom) {
  size_t n = (size_t)c->nx * c->ny * c->nl;
  memcpy(c->po, po, sizeof(double) * n); memcpy(c->pom, pom, sizeof(double) * n);
  memcpy(c->qo, qo, sizeof(double) * n); memcpy(c->qom, qom, sizeof(double) * n);
}

void qgo_get_state(qgo_ctx *c, double *po, double *pom, double *qo, double *qom) {
  size_t n = (size_t)c->nx * c->ny * c->nl;
  if (po) memcpy(po, c->po, sizeof(double) * n);
  if (pom) memcpy(pom, c->pom, sizeof(double) * n);
  if (qo) memcpy(qo, c->qo, sizeof(double) * n);
  if (qom) memcpy(qom, c->qom, sizeof(double) * n);
}

void qgo_set_forcing(qgo_ctx *c, const double *wekpo, const double *entoc, const double *xon) {
  size_t N = (size_t)c->nx * c->ny;
  if (wekpo) memcpy(c->wekpo, wekpo, sizeof(double) * N);
  if (entoc) memcpy(c->entoc, entoc, sizeof(double) * N);
  if (xon) memcpy(c->xon, xon, sizeof(double) * (c->nl - 1));
}

/* cpp option sponge_layer_k247: ramp r_spl(nxpo,nypo) as the main program sets it (src/q-gcm.F:1154-1168) and c1_spl
   (src/parameters_data.F:144); NULL switches the term off */
void qgo_set_sponge(qgo_ctx *c, const double *r_spl, double c1_spl) {
  free(c->r_spl);
  c->r_spl = NULL;
  c->c1_spl = 0.0;
  if (!r_spl) return;
  size_t N = (size_t)c->nx * c->ny;
  c->r_spl = dalloc(N);
  memcpy(c->r_spl, r_spl, sizeof(double) * N);
  c->c1_spl = c1_spl;
}

void qgo_set_cyc_forcing(qgo_ctx *c, double txis, double txin, const double *enis, const double *enin) {
  c->txisoc = txis; c->txinoc = txin;
  if (enis) memcpy(c->enisoc, enis, sizeof(double) * (c->nl - 1));
  if (enin) memcpy(c->eninoc, enin, sizeof(double) * (c->nl - 1));
}

void qgo_get_scalars(qgo_ctx *c, double *s) {
  int nl = c->nl, o = 2 * (nl - 1);
  memset(s, 0, sizeof(double) * (o + 4 * nl));
  for (int k = 0; k < nl - 1; ++k) { s[k] = c->dpioc[k]; s[nl - 1 + k] = c->dpiocp[k]; }
  if (c->cyclic)
    for (int k = 0; k < nl; ++k) {
      s[o + k] = c->ocncs[k]; s[o + nl + k] = c->ocncn[k];
      s[o + 2 * nl + k] = c->ocncsp[k]; s[o + 3 * nl + k] = c->ocncnp[k];
    }
}

void qgo_set_scalars(qgo_ctx *c, const double *s) {
  int nl = c->nl, o = 2 * (nl - 1);
  for (int k = 0; k < nl - 1; ++k) { c->dpioc[k] = s[k]; c->dpiocp[k] = s[nl - 1 + k]; }
  if (c->cyclic)
    for (int k = 0; k < nl; ++k) {
      c->ocncs[k] = s[o + k]; c->ocncn[k] = s[o + nl + k];
      c->ocncsp[k] = s[o + 2 * nl + k]; c->ocncnp[k] = s[o + 3 * nl + k];
    }
}

/* boundary line sums of the last qgostep (cyclic / atmosphere): ajis, ajin, ap5s, ap5n (nl each) */
void qgo_get_bsums(qgo_ctx *c, double *b) {
  for (int k = 0; k < c->nl; ++k) {
    b[k] = c->ajisoc[k]; b[c->nl + k] = c->ajinoc[k];
    b[2 * c->nl + k] = c->ap5soc[k]; b[3 * c->nl + k] = c->ap5noc[k];
  }
}

void qgo_get_inv_diag(qgo_ctx *c, double *xinhom, double *coef) {
  memcpy(xinhom, c->xinhom, sizeof(double) * c->nl);
  memcpy(coef, c->invcoef, sizeof(double) * (c->cyclic ? 2 * (c->nl - 1) + 1 : c->nl - 1));
}

void qgo_get_consts(qgo_ctx *c, double *amatoc, double *ctl2moc, double *ctm2loc, double *rdm2oc,
                    double *bd2oc, double *aoc) {
  int nl = c->nl;
  memcpy(amatoc, c->amatoc, sizeof(double) * nl * nl);
  memcpy(ctl2moc, c->ctl2moc, sizeof(double) * nl * nl);
  memcpy(ctm2loc, c->ctm2loc, sizeof(double) * nl * nl);
  memcpy(rdm2oc, c->rdm2oc, sizeof(double) * nl);
  memcpy(bd2oc, c->bd2oc, sizeof(double) * c->nxt);
  *aoc = c->aoc;
}

void qgo_get_homog(qgo_ctx *c, double *hom, double *aux) {
  int nl = c->nl, ny = c->ny, n1 = nl - 1;
  if (c->cyclic) {
    memcpy(hom, c->pch1oc, sizeof(double) * ny * n1);
    memcpy(hom + (size_t)ny * n1, c->pch2oc, sizeof(double) * ny * n1);
    memcpy(hom + 2 * (size_t)ny * n1, c->pbhoc, sizeof(double) * ny);
    memcpy(aux, c->aipcho, sizeof(double) * n1);
    memcpy(aux + n1, c->hc1soc, sizeof(double) * n1);
    memcpy(aux + 2 * n1, c->hc2soc, sizeof(double) * n1);
    memcpy(aux + 3 * n1, c->hc1noc, sizeof(double) * n1);
    memcpy(aux + 4 * n1, c->hc2noc, sizeof(double) * n1);
    aux[5 * n1] = c->hbsioc; aux[5 * n1 + 1] = c->aipbho;
  } else {
    memcpy(hom, c->ochom, sizeof(double) * (size_t)c->nx * ny * n1);
    memcpy(aux, c->aipohs, sizeof(double) * n1);
    memcpy(aux + n1, c->cdiffo, sizeof(double) * nl * n1);
    memcpy(aux + n1 + nl * n1, c->cdhoc, sizeof(double) * n1 * n1);
  }
}

/* ------------------------------------------------------------------ */
/* qgostep + ocadif: src/qgosubs.F:45-221, 231-446                      */
/* ------------------------------------------------------------------ */
static void ocadif(qgo_ctx *c, int k, double *dqdt, const double *d2p, double ah2ock, double ah4ock,
                   double bcfaco, const double *p, const double *q, double adfaco) {
  int nx = c->nx, ny = c->ny, cyc = c->cyclic;
  double dxom2 = c->dxom2, fnot = c->fnot;
  double ah2fac = ah2ock / fnot, ah4fac = ah4ock / fnot;
  double *d4p = c->d4p;
  if (cyc) { /* qgosubs.F:279-297 */
    double aj5 = 0.5 * q[IX(1, 1)] * (p[IX(2, 2)] - p[IX(nx - 1, 2)]);
    double aj9 = 0.5 * q[IX(1, 2)] * (p[IX(2, 2)] - p[IX(nx - 1, 2)]);
    for (int i = 2; i <= nx - 1; ++i) {
      aj5 = aj5 + q[IX(i, 1)] * (p[IX(i + 1, 2)] - p[IX(i - 1, 2)]);
      aj9 = aj9 + q[IX(i, 2)] * (p[IX(i + 1, 2)] - p[IX(i - 1, 2)]);
    }
    aj5 = aj5 + 0.5 * q[IX(nx, 1)] * (p[IX(2, 2)] - p[IX(nx - 1, 2)]);
    aj9 = aj9 + 0.5 * q[IX(nx, 2)] * (p[IX(2, 2)] - p[IX(nx - 1, 2)]);
    c->ajisoc[k - 1] = c->dxo * c->dyo * (fnot * adfaco * (aj5 + 2.0 * aj9));
  }
  /* Del-4th(p): qgosubs.F:306-342 */
  for (int i = 1; i <= nx; ++i) {
    d4p[IX(i, 1)] = bcfaco * (d2p[IX(i, 2)] - d2p[IX(i, 1)]);
    d4p[IX(i, ny)] = bcfaco * (d2p[IX(i, ny - 1)] - d2p[IX(i, ny)]);
  }
#pragma omp parallel for schedule(static)
  for (int j = 2; j <= ny - 1; ++j) {
    if (cyc)
      d4p[IX(1, j)] = (d2p[IX(1, j - 1)] + d2p[IX(nx - 1, j)] + d2p[IX(2, j)] + d2p[IX(1, j + 1)] -
                       4.0 * d2p[IX(1, j)]) * dxom2;
    else
      d4p[IX(1, j)] = bcfaco * (d2p[IX(2, j)] - d2p[IX(1, j)]);
    for (int i = 2; i <= nx - 1; ++i)
      d4p[IX(i, j)] = dxom2 * (d2p[IX(i, j - 1)] + d2p[IX(i - 1, j)] + d2p[IX(i + 1, j)] +
                               d2p[IX(i, j + 1)] - 4.0 * d2p[IX(i, j)]);
    if (cyc) d4p[IX(nx, j)] = d4p[IX(1, j)];
    else d4p[IX(nx, j)] = bcfaco * (d2p[IX(nx - 1, j)] - d2p[IX(nx, j)]);
  }
  /* qgosubs.F:349-400 */
#pragma omp parallel for schedule(static)
  for (int j = 2; j <= ny - 1; ++j) {
    if (cyc) {
      double d6p = dxom2 * (d4p[IX(1, j - 1)] + d4p[IX(nx - 1, j)] + d4p[IX(2, j)] + d4p[IX(1, j + 1)] -
                            4.0 * d4p[IX(1, j)]);
      double diffus = ah2fac * d4p[IX(1, j)] - ah4fac * d6p;
      dqdt[IX(1, j)] =
          adfaco * ((q[IX(2, j)] - q[IX(nx - 1, j)]) * (p[IX(1, j + 1)] - p[IX(1, j - 1)]) +
                    (q[IX(1, j - 1)] - q[IX(1, j + 1)]) * (p[IX(2, j)] - p[IX(nx - 1, j)]) +
                    q[IX(2, j)] * (p[IX(2, j + 1)] - p[IX(2, j - 1)]) -
                    q[IX(nx - 1, j)] * (p[IX(nx - 1, j + 1)] - p[IX(nx - 1, j - 1)]) -
                    q[IX(1, j + 1)] * (p[IX(2, j + 1)] - p[IX(nx - 1, j + 1)]) +
                    q[IX(1, j - 1)] * (p[IX(2, j - 1)] - p[IX(nx - 1, j - 1)]) +
                    p[IX(1, j + 1)] * (q[IX(2, j + 1)] - q[IX(nx - 1, j + 1)]) -
                    p[IX(1, j - 1)] * (q[IX(2, j - 1)] - q[IX(nx - 1, j - 1)]) -
                    p[IX(2, j)] * (q[IX(2, j + 1)] - q[IX(2, j - 1)]) +
                    p[IX(nx - 1, j)] * (q[IX(nx - 1, j + 1)] - q[IX(nx - 1, j - 1)])) +
          diffus;
    } else {
      dqdt[IX(1, j)] = 0.0;
    }
    for (int i = 2; i <= nx - 1; ++i) {
      double d6p = dxom2 * (d4p[IX(i, j - 1)] + d4p[IX(i - 1, j)] + d4p[IX(i + 1, j)] + d4p[IX(i, j + 1)] -
                            4.0 * d4p[IX(i, j)]);
      double diffus = ah2fac * d4p[IX(i, j)] - ah4fac * d6p;
      dqdt[IX(i, j)] =
          adfaco * ((q[IX(i + 1, j)] - q[IX(i - 1, j)]) * (p[IX(i, j + 1)] - p[IX(i, j - 1)]) +
                    (q[IX(i, j - 1)] - q[IX(i, j + 1)]) * (p[IX(i + 1, j)] - p[IX(i - 1, j)]) +
                    q[IX(i + 1, j)] * (p[IX(i + 1, j + 1)] - p[IX(i + 1, j - 1)]) -
                    q[IX(i - 1, j)] * (p[IX(i - 1, j + 1)] - p[IX(i - 1, j - 1)]) -
                    q[IX(i, j + 1)] * (p[IX(i + 1, j + 1)] - p[IX(i - 1, j + 1)]) +
                    q[IX(i, j - 1)] * (p[IX(i + 1, j - 1)] - p[IX(i - 1, j - 1)]) +
                    p[IX(i, j + 1)] * (q[IX(i + 1, j + 1)] - q[IX(i - 1, j + 1)]) -
                    p[IX(i, j - 1)] * (q[IX(i + 1, j - 1)] - q[IX(i - 1, j - 1)]) -
                    p[IX(i + 1, j)] * (q[IX(i + 1, j + 1)] - q[IX(i + 1, j - 1)]) +
                    p[IX(i - 1, j)] * (q[IX(i - 1, j + 1)] - q[IX(i - 1, j - 1)])) +
          diffus;
    }
    if (cyc) dqdt[IX(nx, j)] = dqdt[IX(1, j)];
    else dqdt[IX(nx, j)] = 0.0;
  }
  if (cyc) { /* qgosubs.F:404-443 */
    double aj5 = -0.5 * q[IX(1, ny)] * (p[IX(2, ny - 1)] - p[IX(nx - 1, ny - 1)]);
    double aj9 = -0.5 * q[IX(1, ny - 1)] * (p[IX(2, ny - 1)] - p[IX(nx - 1, ny - 1)]);
    for (int i = 2; i <= nx - 1; ++i) {
      aj5 = aj5 - q[IX(i, ny)] * (p[IX(i + 1, ny - 1)] - p[IX(i - 1, ny - 1)]);
      aj9 = aj9 - q[IX(i, ny - 1)] * (p[IX(i + 1, ny - 1)] - p[IX(i - 1, ny - 1)]);
    }
    aj5 = aj5 - 0.5 * q[IX(nx, ny)] * (p[IX(2, ny - 1)] - p[IX(nx - 1, ny - 1)]);
    aj9 = aj9 - 0.5 * q[IX(nx, ny - 1)] * (p[IX(2, ny - 1)] - p[IX(nx - 1, ny - 1)]);
    c->ajinoc[k - 1] = c->dxo * c->dyo * (fnot * adfaco * (aj5 + 2.0 * aj9));
    double ah3s = 0, ah3n = 0, ah5s = 0, ah5n = 0;
    if (c->atmos) { /* atadif: trapezoid over i = 1..nxpa, no Del-4th (ah2) term: src/qgasubs.F:305-314 */
      ah5s = 0.5 * (d4p[IX(1, 2)] - d4p[IX(1, 1)]);
      ah5n = 0.5 * (d4p[IX(1, ny)] - d4p[IX(1, ny - 1)]);
      for (int i = 2; i <= nx - 1; ++i) {
        ah5s = ah5s + (d4p[IX(i, 2)] - d4p[IX(i, 1)]);
        ah5n = ah5n + (d4p[IX(i, ny)] - d4p[IX(i, ny - 1)]);
      }
      ah5s = ah5s + 0.5 * (d4p[IX(nx, 2)] - d4p[IX(nx, 1)]);
      ah5n = ah5n + 0.5 * (d4p[IX(nx, ny)] - d4p[IX(nx, ny - 1)]);
    } else
    for (int i = 1; i <= nx - 1; ++i) {
      ah3s = ah3s + (d2p[IX(i, 2)] - d2p[IX(i, 1)]);
      ah3n = ah3n + (d2p[IX(i, ny)] - d2p[IX(i, ny - 1)]);
      ah5s = ah5s + (d4p[IX(i, 2)] - d4p[IX(i, 1)]);
      ah5n = ah5n + (d4p[IX(i, ny)] - d4p[IX(i, ny - 1)]);
    }
    c->ap3soc[k - 1] = ah2ock * ah3s; c->ap3noc[k - 1] = ah2ock * ah3n;
    c->ap5soc[k - 1] = ah4ock * ah5s; c->ap5noc[k - 1] = ah4ock * ah5n;
  }
}

void qgo_qgostep(qgo_ctx *c) {
  int nx = c->nx, ny = c->ny, nl = c->nl, cyc = c->cyclic;
  size_t N = (size_t)nx * ny;
  double fnot = c->fnot, dxom2 = c->dxom2;
  double adfaco = 1.0 / (12.0 * c->dxo * c->dyo * fnot);
  double bcfaco = c->bccooc * dxom2 / (0.5 * c->bccooc + 1.0);
  double sgn = (fnot >= 0.0) ? 1.0 : -1.0;
  double bdrfac = 0.5 * sgn * c->delek / c->hoc[nl - 1];
  double fohfac[64];
  for (int k = 0; k < nl; ++k) fohfac[k] = fnot / c->hoc[k];
  double *del2p = c->d2p;
  const double *pom = c->pom;
  for (int k = 1; k <= nl; ++k) {
    for (int i = 1; i <= nx; ++i) {
      del2p[IX(i, 1)] = bcfaco * (pom[IX3(i, 2, k)] - pom[IX3(i, 1, k)]);
      del2p[IX(i, ny)] = bcfaco * (pom[IX3(i, ny - 1, k)] - pom[IX3(i, ny, k)]);
    }
#pragma omp parallel for schedule(static)
    for (int j = 2; j <= ny - 1; ++j) {
      if (cyc)
        del2p[IX(1, j)] = (pom[IX3(1, j - 1, k)] + pom[IX3(nx - 1, j, k)] + pom[IX3(2, j, k)] +
                           pom[IX3(1, j + 1, k)] - 4.0 * pom[IX3(1, j, k)]) * dxom2;
      else
        del2p[IX(1, j)] = bcfaco * (pom[IX3(2, j, k)] - pom[IX3(1, j, k)]);
      for (int i = 2; i <= nx - 1; ++i)
        del2p[IX(i, j)] = (pom[IX3(i, j - 1, k)] + pom[IX3(i - 1, j, k)] + pom[IX3(i + 1, j, k)] +
                           pom[IX3(i, j + 1, k)] - 4.0 * pom[IX3(i, j, k)]) * dxom2;
      if (cyc) del2p[IX(nx, j)] = del2p[IX(1, j)];
      else del2p[IX(nx, j)] = bcfaco * (pom[IX3(nx - 1, j, k)] - pom[IX3(nx, j, k)]);
    }
    ocadif(c, k, c->dqdt + N * (k - 1), del2p, c->ah2oc[k - 1], c->ah4oc[k - 1], bcfaco,
           c->po + N * (k - 1), c->qo + N * (k - 1), adfaco);
  }
  if (cyc && !c->atmos) { /* qgosubs.F:150-163 */
    double bds = 0.0, bdn = 0.0;
    for (int i = 1; i <= nx - 1; ++i) {
      bds = bds + (pom[IX3(i, 2, nl)] - pom[IX3(i, 1, nl)]);
      bdn = bdn + (pom[IX3(i, ny, nl)] - pom[IX3(i, ny - 1, nl)]);
    }
    c->bdrins = 0.5 * sgn * c->delek * bds;
    c->bdrinn = 0.5 * sgn * c->delek * bdn;
  }
  /* qgosubs.F:173-219 */
  double tdto = c->tdto;
#pragma omp parallel for schedule(static)
  for (int j = 2; j <= ny - 1; ++j)
    for (int i = 1; i <= nx; ++i) {
      double qdot[64];
      if (c->atmos) { /* src/qgasubs.F:128-134: entrainment and Ekman pumping act from below, no drag */
        qdot[0] = c->dqdt[IX3(i, j, 1)] + fohfac[0] * (c->entoc[IX(i, j)] - c->wekpo[IX(i, j)]);
        qdot[1] = c->dqdt[IX3(i, j, 2)] - fohfac[1] * c->entoc[IX(i, j)];
        for (int k = 3; k <= nl; ++k) qdot[k - 1] = c->dqdt[IX3(i, j, k)];
      } else {
      qdot[0] = c->dqdt[IX3(i, j, 1)] + fohfac[0] * (c->wekpo[IX(i, j)] - c->entoc[IX(i, j)]);
      qdot[1] = c->dqdt[IX3(i, j, 2)] + fohfac[1] * c->entoc[IX(i, j)];
      for (int k = 3; k <= nl; ++k) qdot[k - 1] = c->dqdt[IX3(i, j, k)];
      qdot[nl - 1] = qdot[nl - 1] - bdrfac * del2p[IX(i, j)];
      }
      for (int k = 1; k <= nl; ++k) {
        double qold = c->qo[IX3(i, j, k)];
        c->qo[IX3(i, j, k)] = c->qom[IX3(i, j, k)] + tdto * qdot[k - 1];
        /* sponge layer of the k247 fork, src/qgosubs.F:203-205 (betay = beta*yporel(j), :177) */
        if (c->r_spl)
          c->qo[IX3(i, j, k)] = c->qo[IX3(i, j, k)] +
                                tdto * c->c1_spl * c->r_spl[IX(i, j)] * (c->qom[IX3(i, j, k)] - c->beta * c->yporel[j - 1]);
        c->qom[IX3(i, j, k)] = qold;
      }
    }
  for (int k = 1; k <= nl; ++k)
    for (int i = 1; i <= nx; ++i) {
      c->qom[IX3(i, 1, k)] = c->qo[IX3(i, 1, k)];
      c->qom[IX3(i, ny, k)] = c->qo[IX3(i, ny, k)];
    }
}

/* projection: src/ocisubs.F:117-139 */
void qgo_project(qgo_ctx *c, double *wrk) {
  int nx = c->nx, ny = c->ny, nl = c->nl;
#pragma omp parallel for schedule(static)
  for (int j = 2; j <= ny - 1; ++j) {
    double betay = c->beta * c->yporel[j - 1];
    for (int i = 1; i <= nx; ++i) {
      double ql[64];
      for (int k = 1; k <= nl; ++k) ql[k - 1] = c->qo[IX3(i, j, k)] - betay;
      if (c->atmos) ql[0] = ql[0] - c->ddynoc[IX(i, j)]; /* src/atisubs.F:117 */
      else ql[nl - 1] = ql[nl - 1] - c->ddynoc[IX(i, j)];
      for (int m = 1; m <= nl; ++m) {
        double qm = 0.0;
        for (int k = 1; k <= nl; ++k) qm = qm + M2(c->ctl2moc, k, m) * ql[k - 1];
        wrk[IX3(i, j, m)] = c->fnot * qm;
      }
    }
  }
}

/* ocinvq: src/ocisubs.F:64-407 */
void qgo_ocinvq(qgo_ctx *c) {
  int nx = c->nx, ny = c->ny, nl = c->nl, nxt = c->nxt;
  size_t N = (size_t)nx * ny;
  double *wrk = c->wrk;
  double dxo = c->dxo, dyo = c->dyo, fnot = c->fnot, tdto = c->tdto;
  /* the reference leaves rows 1 and nypo of wrk undefined before the solve;
     the solver zeroes them.  Start from zeros for determinism. */
  memset(wrk, 0, sizeof(double) * N * nl);
  qgo_project(c, wrk);
  double *boc = dalloc(nxt);
  for (int m = 1; m <= nl; ++m) {
    for (int i = 0; i < nxt; ++i) boc[i] = c->bd2oc[i] - c->rdm2oc[m - 1];
    qgo_helmholtz(c, wrk + N * (m - 1), boc);
    c->xinhom[m - 1] = qgo_xintp(wrk + N * (m - 1), nx, ny) * dxo * dyo;
  }
  free(boc);
  if (c->cyclic) {
    double rhss[64], rhsn[64], ocsnew[64], ocnnew[64], clhss[64], clhsn[64], c1[64], c2[64], c3;
    double aipmod[64], aiplay[64];
    const double *hoc = c->hoc;
    double entfac = 0.5 * dyo * fnot * fnot;
    if (c->atmos) { /* src/atisubs.F:177-196: layer 1 is the bottom layer, stress enters with the opposite sign */
      rhss[0] = -(entfac / hoc[0]) * c->enisoc[0] - (fnot / hoc[0]) * c->txisoc + c->ajisoc[0] + c->ap5soc[0];
      rhsn[0] = -(entfac / hoc[0]) * c->eninoc[0] + (fnot / hoc[0]) * c->txinoc + c->ajinoc[0] - c->ap5noc[0];
      for (int k = 2; k <= nl - 1; ++k) {
        rhss[k - 1] = -(entfac / hoc[k - 1]) * (c->enisoc[k - 1] - c->enisoc[k - 2]) + c->ajisoc[k - 1] + c->ap5soc[k - 1];
        rhsn[k - 1] = -(entfac / hoc[k - 1]) * (c->eninoc[k - 1] - c->eninoc[k - 2]) + c->ajinoc[k - 1] - c->ap5noc[k - 1];
      }
      rhss[nl - 1] = (entfac / hoc[nl - 1]) * c->enisoc[nl - 2] + c->ajisoc[nl - 1] + c->ap5soc[nl - 1];
      rhsn[nl - 1] = (entfac / hoc[nl - 1]) * c->eninoc[nl - 2] + c->ajinoc[nl - 1] - c->ap5noc[nl - 1];
    } else {
    /* ocisubs.F:176-193 */
    rhss[0] = (entfac / hoc[0]) * c->enisoc[0] + (fnot / hoc[0]) * c->txisoc + c->ajisoc[0] - c->ap3soc[0] + c->ap5soc[0];
    rhsn[0] = (entfac / hoc[0]) * c->eninoc[0] - (fnot / hoc[0]) * c->txinoc + c->ajinoc[0] + c->ap3noc[0] - c->ap5noc[0];
    for (int k = 2; k <= nl - 1; ++k) {
      rhss[k - 1] = (entfac / hoc[k - 1]) * (c->enisoc[k - 1] - c->enisoc[k - 2]) + c->ajisoc[k - 1] - c->ap3soc[k - 1] + c->ap5soc[k - 1];
      rhsn[k - 1] = (entfac / hoc[k - 1]) * (c->eninoc[k - 1] - c->eninoc[k - 2]) + c->ajinoc[k - 1] + c->ap3noc[k - 1] - c->ap5noc[k - 1];
    }
    rhss[nl - 1] = -(entfac / hoc[nl - 1]) * c->enisoc[nl - 2] + c->ajisoc[nl - 1] - c->ap3soc[nl - 1] + c->ap5soc[nl - 1] + (fnot / hoc[nl - 1]) * c->bdrins;
    rhsn[nl - 1] = -(entfac / hoc[nl - 1]) * c->eninoc[nl - 2] + c->ajinoc[nl - 1] + c->ap3noc[nl - 1] - c->ap5noc[nl - 1] - (fnot / hoc[nl - 1]) * c->bdrinn;
    }
    for (int k = 0; k < nl; ++k) { /* ocisubs.F:199-206 */
      ocsnew[k] = c->ocncsp[k] + tdto * rhss[k];
      ocnnew[k] = c->ocncnp[k] + tdto * rhsn[k];
      c->ocncsp[k] = c->ocncs[k]; c->ocncnp[k] = c->ocncn[k];
      c->ocncs[k] = ocsnew[k]; c->ocncn[k] = ocnnew[k];
    }
    for (int m = 1; m <= nl; ++m) { /* ocisubs.F:212-234 */
      const double *w = wrk + N * (m - 1);
      double ayis = 0.5 * w[IX(1, 2)], ayin = -0.5 * w[IX(1, ny - 1)];
      for (int i = 2; i <= nx - 1; ++i) { ayis = ayis + w[IX(i, 2)]; ayin = ayin - w[IX(i, ny - 1)]; }
      ayis = ayis + 0.5 * w[IX(nx, 2)];
      ayin = ayin - 0.5 * w[IX(nx, ny - 1)];
      ayis = ayis * (dxo / dyo); ayin = ayin * (dxo / dyo);
      clhss[m - 1] = 0.0; clhsn[m - 1] = 0.0;
      for (int k = 1; k <= nl; ++k) {
        clhss[m - 1] = clhss[m - 1] + M2(c->ctl2moc, k, m) * ocsnew[k - 1];
        clhsn[m - 1] = clhsn[m - 1] + M2(c->ctl2moc, k, m) * ocnnew[k - 1];
      }
      clhss[m - 1] = clhss[m - 1] + ayis;
      clhsn[m - 1] = clhsn[m - 1] - ayin;
    }
    c3 = clhss[0] * c->hbsioc;
    for (int m = 1; m <= nl - 1; ++m) {
      c1[m - 1] = c->hc2noc[m - 1] * clhss[m] - c->hc2soc[m - 1] * clhsn[m];
      c2[m - 1] = c->hc1soc[m - 1] * clhsn[m] - c->hc1noc[m - 1] * clhss[m];
    }
    aipmod[0] = c->xinhom[0] + c3 * c->aipbho;
    for (int m = 2; m <= nl; ++m) aipmod[m - 1] = c->xinhom[m - 1] + (c1[m - 2] + c2[m - 2]) * c->aipcho[m - 2];
    for (int k = 1; k <= nl; ++k) {
      double pl = 0.0;
      for (int m = 1; m <= nl; ++m) pl = pl + M2(c->ctm2loc, m, k) * aipmod[m - 1];
      aiplay[k - 1] = pl;
    }
    for (int k = 1; k <= nl - 1; ++k) { /* ocisubs.F:268-294 (monitors omitted) */
      c->dpiocp[k - 1] = c->dpioc[k - 1];
      c->dpioc[k - 1] = c->atmos ? aiplay[k - 1] - aiplay[k] /* atisubs.F:256 */ : aiplay[k] - aiplay[k - 1];
    }
    for (int m = 0; m < nl - 1; ++m) { c->invcoef[m] = c1[m]; c->invcoef[nl - 1 + m] = c2[m]; }
    c->invcoef[2 * (nl - 1)] = c3;
    /* ocisubs.F:300-327 */
#pragma omp parallel for schedule(static)
    for (int j = 1; j <= ny; ++j) {
      double homcor[64], pm[64];
      homcor[0] = c3 * c->pbhoc[j - 1];
      for (int m = 2; m <= nl; ++m)
        homcor[m - 1] = c1[m - 2] * c->pch1oc[(j - 1) + (size_t)ny * (m - 2)] + c2[m - 2] * c->pch2oc[(j - 1) + (size_t)ny * (m - 2)];
      for (int i = 1; i <= nx; ++i) {
        for (int m = 1; m <= nl; ++m) pm[m - 1] = wrk[IX3(i, j, m)] + homcor[m - 1];
        for (int k = 1; k <= nl; ++k) {
          c->pom[IX3(i, j, k)] = c->po[IX3(i, j, k)];
          double pl = 0.0;
          for (int m = 1; m <= nl; ++m) pl = pl + M2(c->ctm2loc, m, k) * pm[m - 1];
          c->po[IX3(i, j, k)] = pl;
        }
      }
    }
  } else {
    /* ocisubs.F:333-370 */
    int n1 = nl - 1;
    double aient[64], rhs[64], hclco[64];
    aient[0] = c->xon[0];
    for (int k = 2; k <= nl - 1; ++k) aient[k - 1] = 0.0;
    for (int k = 1; k <= nl - 1; ++k) {
      double aitmp = c->dpioc[k - 1];
      c->dpioc[k - 1] = c->dpiocp[k - 1] - tdto * c->gpoc[k - 1] * aient[k - 1];
      c->dpiocp[k - 1] = aitmp;
      double rhsum = 0.0;
      for (int m = 1; m <= nl; ++m) rhsum = rhsum + c->cdiffo[(m - 1) + nl * (k - 1)] * c->xinhom[m - 1];
      rhs[k - 1] = c->dpioc[k - 1] - rhsum;
      hclco[k - 1] = rhs[k - 1];
    }
    lu_solve(n1, c->cdhlu, c->ipivch, hclco);
    lu_refine(n1, c->cdhoc, c->cdhlu, c->ipivch, rhs, hclco);
    for (int m = 0; m < n1; ++m) c->invcoef[m] = hclco[m];
    /* ocisubs.F:377-401 */
#pragma omp parallel for schedule(static)
    for (int j = 1; j <= ny; ++j)
      for (int i = 1; i <= nx; ++i) {
        double pm[64];
        pm[0] = wrk[IX3(i, j, 1)];
        for (int m = 2; m <= nl; ++m) pm[m - 1] = wrk[IX3(i, j, m)] + hclco[m - 2] * c->ochom[IX3(i, j, m - 1)];
        for (int k = 1; k <= nl; ++k) {
          c->pom[IX3(i, j, k)] = c->po[IX3(i, j, k)];
          double pl = 0.0;
          for (int m = 1; m <= nl; ++m) pl = pl + M2(c->ctm2loc, m, k) * pm[m - 1];
          c->po[IX3(i, j, k)] = pl;
        }
      }
  }
}

/* src/q-gcm.F:1328-1366 (ocean fields + constraint scalars) */
void qgo_lf_average(qgo_ctx *c) {
  size_t n = (size_t)c->nx * c->ny * c->nl;
#pragma omp parallel for schedule(static)
  for (size_t t = 0; t < n; ++t) {
    c->qo[t] = 0.5 * (c->qo[t] + c->qom[t]);
    c->po[t] = 0.5 * (c->po[t] + c->pom[t]);
  }
  for (int k = 0; k < c->nl - 1; ++k) c->dpioc[k] = 0.5 * (c->dpioc[k] + c->dpiocp[k]);
  if (c->cyclic)
    for (int k = 0; k < c->nl; ++k) {
      c->ocncs[k] = 0.5 * (c->ocncs[k] + c->ocncsp[k]);
      c->ocncn[k] = 0.5 * (c->ocncn[k] + c->ocncnp[k]);
    }
}

/* src/q-gcm.F:1243-1249 + 1328: ocean step s (1-based) is followed by the
 * averaging when mod(s-1,25)==0. */
void qgo_steps(qgo_ctx *c, int s0, int n) {
  for (int s = s0; s < s0 + n; ++s) {
    qgo_qgostep(c);
    qgo_ocinvq(c);
    qgo_ocqbdy(c);
    /* the atmosphere is averaged when mod(nt-1,100) == 0, nt = its own step count (src/q-gcm.F:1370) */
    if ((s - 1) % (c->atmos ? 100 : 25) == 0) qgo_lf_average(c);
  }
}

/* ocean-only Ekman pumping: src/xfosubs.F:138 (hxofac), 566-645 */
void qgo_wekpo_from_tau(int nx, int ny, int cyclic, double dxo, double fnot, const double *tauxo,
                        const double *tauyo, double *wekto, double *wekpo) {
  int nxt = nx - 1, nyt = ny - 1;
  double hxofac = 0.5 * (1.0 / (dxo * fnot));
#define WT(i, j) wekto[((i)-1) + (size_t)nxt * ((j)-1)]
  for (int j = 1; j <= nyt; ++j)
    for (int i = 1; i <= nxt; ++i)
      WT(i, j) = hxofac * (tauyo[IX(i + 1, j + 1)] + tauyo[IX(i + 1, j)] - (tauyo[IX(i, j + 1)] + tauyo[IX(i, j)]) +
                           tauxo[IX(i + 1, j)] + tauxo[IX(i, j)] - (tauxo[IX(i + 1, j + 1)] + tauxo[IX(i, j + 1)]));
  for (int jo = 2; jo <= ny - 1; ++jo) {
    if (cyclic) wekpo[IX(1, jo)] = 0.25 * (WT(nxt, jo - 1) + WT(nxt, jo) + WT(1, jo - 1) + WT(1, jo));
    else wekpo[IX(1, jo)] = 0.5 * (WT(1, jo - 1) + WT(1, jo));
    for (int io = 2; io <= nx - 1; ++io)
      wekpo[IX(io, jo)] = 0.25 * (WT(io - 1, jo - 1) + WT(io - 1, jo) + WT(io, jo - 1) + WT(io, jo));
    if (cyclic) wekpo[IX(nx, jo)] = wekpo[IX(1, jo)];
    else wekpo[IX(nx, jo)] = 0.5 * (WT(nxt, jo - 1) + WT(nxt, jo));
  }
  if (cyclic) {
    wekpo[IX(1, 1)] = 0.5 * (WT(nxt, 1) + WT(1, 1));
    wekpo[IX(1, ny)] = 0.5 * (WT(nxt, nyt) + WT(1, nyt));
  } else {
    wekpo[IX(1, 1)] = WT(1, 1);
    wekpo[IX(1, ny)] = WT(1, nyt);
  }
  for (int io = 2; io <= nx - 1; ++io) {
    wekpo[IX(io, 1)] = 0.5 * (WT(io - 1, 1) + WT(io, 1));
    wekpo[IX(io, ny)] = 0.5 * (WT(io - 1, nyt) + WT(io, nyt));
  }
  if (cyclic) {
    wekpo[IX(nx, 1)] = wekpo[IX(1, 1)];
    wekpo[IX(nx, ny)] = wekpo[IX(1, ny)];
  } else {
    wekpo[IX(nx, 1)] = WT(nxt, 1);
    wekpo[IX(nx, ny)] = WT(nxt, nyt);
  }
#undef WT
}


/* ================================================================== */
/* Ocean mixed layer: oml + omladf, src/omlsubs.F:47-236, 244-763      */
/* (SURVEY 8 row f1).  T-grid fields are (nxto,nyto), element (i,j) at  */
/* (i-1) + nxto*(j-1).                                                  */
/* ================================================================== */
void qgo_oml_init(qgo_ctx *c, double hmoc, double toc1, double toc2, double st2d, double st4d, double ycexp,
                  double rrcpoc, int sb_hflux, double tsbdy, int nb_hflux, double tnbdy) {
  size_t nt = (size_t)c->nxt * (c->ny - 1), np = (size_t)c->nx * c->ny;
  c->oml_on = 1;
  c->hmoc = hmoc; c->toc1 = toc1; c->toc2 = toc2; c->st2d = st2d; c->st4d = st4d; c->ycexp = ycexp;
  c->rrcpoc = rrcpoc; c->sb_hflux = sb_hflux; c->tsbdy = tsbdy; c->nb_hflux = nb_hflux; c->tnbdy = tnbdy;
  if (!c->sst) {
    c->sst = dalloc(nt); c->sstm = dalloc(nt); c->fnetoc = dalloc(nt); c->wekto = dalloc(nt);
    c->tauxo = dalloc(np); c->tauyo = dalloc(np);
    c->omrhs = dalloc(nt); c->omd2t = dalloc(nt); c->omxfo = dalloc(nt);
  }
}

void qgo_oml_set(qgo_ctx *c, const double *sst, const double *sstm, const double *fnetoc, const double *wekto,
                 const double *tauxo, const double *tauyo) {
  size_t nt = (size_t)c->nxt * (c->ny - 1) * sizeof(double), np = (size_t)c->nx * c->ny * sizeof(double);
  if (sst) memcpy(c->sst, sst, nt);
  if (sstm) memcpy(c->sstm, sstm, nt);
  if (fnetoc) memcpy(c->fnetoc, fnetoc, nt);
  if (wekto) memcpy(c->wekto, wekto, nt);
  if (tauxo) memcpy(c->tauxo, tauxo, np);
  if (tauyo) memcpy(c->tauyo, tauyo, np);
}

/* scal = xon(1), cfraoc, centoc, enisoc(1), eninoc(1) */
void qgo_oml_get(qgo_ctx *c, double *sst, double *sstm, double *entoc, double *scal) {
  size_t nt = (size_t)c->nxt * (c->ny - 1) * sizeof(double), np = (size_t)c->nx * c->ny * sizeof(double);
  if (sst) memcpy(sst, c->sst, nt);
  if (sstm) memcpy(sstm, c->sstm, nt);
  if (entoc) memcpy(entoc, c->entoc, np);
  if (scal) {
    scal[0] = c->xon[0]; scal[1] = c->cfraoc; scal[2] = c->centoc;
    scal[3] = c->cyclic ? c->enisoc[0] : 0.0; scal[4] = c->cyclic ? c->eninoc[0] : 0.0;
  }
}

/* del2t(i,j) of the lagged sst with the boundary variants of src/omlsubs.F:297-300 (W), 331-346 (E),
 * 403-422 (S), 437-454 (N), 466-647 (corners); the operand order of every case is the reference's. */
static double oml_del2t(const qgo_ctx *c, int i, int j) {
  const int nxt = c->nxt, nyt = c->ny - 1, cyc = c->cyclic;
  const double *T = c->sstm;
#define TM(ii, jj) T[(size_t)((ii)-1) + (size_t)nxt * ((jj)-1)]
  const int hasW = (i > 1) || cyc, hasE = (i < nxt) || cyc;
  const double w = hasW ? TM(i > 1 ? i - 1 : nxt, j) : 0.0, e = hasE ? TM(i < nxt ? i + 1 : 1, j) : 0.0;
  const double cc = TM(i, j);
  double acc = 0.0, n = 0.0;
  int first = 1;
#define ADD(v) do { acc = first ? (v) : acc + (v); first = 0; n += 1.0; } while (0)
  if (j == 1) { /* W, E, N, tsbdy */
    if (hasW) ADD(w);
    if (hasE) ADD(e);
    ADD(TM(i, 2));
    if (c->sb_hflux) ADD(c->tsbdy);
    return acc - n * cc;
  }
  if (j == nyt) {
    if (cyc && i == nxt && c->nb_hflux) /* :630-631: tnbdy is added after the -4 sstm term */
      return TM(i, j - 1) + w + e - 4.0 * cc + c->tnbdy;
    ADD(TM(i, j - 1)); /* S, W, tnbdy, E */
    if (hasW) ADD(w);
    if (c->nb_hflux) ADD(c->tnbdy);
    if (hasE) ADD(e);
    return acc - n * cc;
  }
  ADD(TM(i, j - 1)); /* S, W, E, N */
  if (hasW) ADD(w);
  if (hasE) ADD(e);
  ADD(TM(i, j + 1));
  return acc - n * cc;
#undef ADD
#undef TM
}

/* omladf, src/omlsubs.F:244-763: advective + diffusive right-hand side in c->omrhs */
static void omladf(qgo_ctx *c) {
  const int nx = c->nx, nxt = c->nxt, nyt = c->ny - 1, cyc = c->cyclic;
  const double rdxof0 = 1.0 / (c->dxo * c->fnot), hdxom1 = 0.5 / c->dxo;
  const double uvgfac = c->ycexp * rdxof0, rhf0hm = 0.5 / (c->fnot * c->hmoc);
  const double d2tfac = c->st2d * c->dxom2, d4tfac = c->st4d * (c->dxom2 * c->dxom2);
  const double *po = c->po, *tx = c->tauxo, *ty = c->tauyo, *S = c->sst;
  double *rhs = c->omrhs, *D = c->omd2t;
#define PO1(ii, jj) po[(size_t)((ii)-1) + (size_t)nx * ((jj)-1)]
#define TX(ii, jj) tx[(size_t)((ii)-1) + (size_t)nx * ((jj)-1)]
#define TY(ii, jj) ty[(size_t)((ii)-1) + (size_t)nx * ((jj)-1)]
#define ST(ii, jj) S[(size_t)((ii)-1) + (size_t)nxt * ((jj)-1)]
#define UF(ii, jj) (-uvgfac * (PO1(ii, (jj) + 1) - PO1(ii, jj)) + rhf0hm * (TY(ii, (jj) + 1) + TY(ii, jj)))
#define VF(ii, jj) (uvgfac * (PO1((ii) + 1, jj) - PO1(ii, jj)) - rhf0hm * (TX((ii) + 1, jj) + TX(ii, jj)))
#pragma omp parallel for schedule(static)
  for (int j = 1; j <= nyt; ++j)
    for (int i = 1; i <= nxt; ++i) {
      double um, tm, up, tp;
      if (i == 1 && !cyc) { um = 0.0; tm = 0.0; }
      else { um = UF(i, j); tm = ST(i > 1 ? i - 1 : nxt, j) + ST(i, j); }
      if (i == nxt && !cyc) { up = 0.0; tp = 0.0; }
      else { up = UF(i + 1, j); tp = ST(i, j) + ST(i < nxt ? i + 1 : 1, j); }
      const double hxadv = hdxom1 * (up * tp - um * tm);
      double hyadv;
      if (j == 1) {
        const double vp = VF(i, 2), tp2 = ST(i, 1) + ST(i, 2);
        if (c->sb_hflux) {
          const double vm = -rhf0hm * (TX(i + 1, 1) + TX(i, 1)), tm2 = ST(i, 1) + c->tsbdy;
          hyadv = hdxom1 * (vp * tp2 - vm * tm2);
        } else hyadv = hdxom1 * (vp * tp2);
      } else if (j == nyt) {
        const double vm = VF(i, nyt), tm2 = ST(i, nyt - 1) + ST(i, nyt);
        if (c->nb_hflux) {
          const double vp = -rhf0hm * (TX(i + 1, nyt + 1) + TX(i, nyt + 1)), tp2 = ST(i, nyt) + c->tnbdy;
          hyadv = hdxom1 * (vp * tp2 - vm * tm2);
        } else hyadv = hdxom1 * (-vm * tm2);
      } else {
        const double vm = VF(i, j), vp = VF(i, j + 1);
        hyadv = hdxom1 * (vp * (ST(i, j + 1) + ST(i, j)) - vm * (ST(i, j) + ST(i, j - 1)));
      }
      rhs[(size_t)(i - 1) + (size_t)nxt * (j - 1)] = -(hxadv + hyadv);
      D[(size_t)(i - 1) + (size_t)nxt * (j - 1)] = oml_del2t(c, i, j);
    }
  /* Del-sqd and Del-4th terms, :733-759; dummy columns :349-357 (box: copy of the edge, cyclic: wrap) */
#define DD(ii, jj) D[(size_t)(((ii) < 1 ? (cyc ? nxt : 1) : ((ii) > nxt ? (cyc ? 1 : nxt) : (ii))) - 1) + (size_t)nxt * ((jj)-1)]
#pragma omp parallel for schedule(static)
  for (int j = 1; j <= nyt; ++j)
    for (int i = 1; i <= nxt; ++i) {
      double *r = &rhs[(size_t)(i - 1) + (size_t)nxt * (j - 1)];
      if (j == 1) *r = *r + d2tfac * DD(i, 1) - d4tfac * (DD(i - 1, 1) + DD(i + 1, 1) + DD(i, 2) - 3.0 * DD(i, 1));
      else if (j == nyt)
        *r = *r + d2tfac * DD(i, nyt) - d4tfac * (DD(i, nyt - 1) + DD(i - 1, nyt) + DD(i + 1, nyt) - 3.0 * DD(i, nyt));
      else
        *r = *r + d2tfac * DD(i, j) - d4tfac * (DD(i, j - 1) + DD(i - 1, j) + DD(i + 1, j) + DD(i, j + 1) - 4.0 * DD(i, j));
    }
#undef DD
#undef PO1
#undef TX
#undef TY
#undef ST
#undef UF
#undef VF
}

/* oml, src/omlsubs.F:47-236 */
void qgo_oml(qgo_ctx *c) {
  const int nx = c->nx, ny = c->ny, nxt = c->nxt, nyt = c->ny - 1, cyc = c->cyclic;
  const double hmoinv = 1.0 / c->hmoc, dtoinv = 1.0 / (c->toc1 - c->toc2), entfac = c->hmoc * dtoinv / c->tdto;
  const double ocnorm = 1.0 / ((double)nxt * (double)nyt);
  double *xfo = c->omxfo, *ent = c->entoc;
  omladf(c);
  double cfrasm = 0.0, centsm = 0.0;
  for (int j = 1; j <= nyt; ++j)
    for (int i = 1; i <= nxt; ++i) {
      const size_t o = (size_t)(i - 1) + (size_t)nxt * (j - 1);
      const double diabat = 0.5 * c->wekto[o] * (c->sstm[o] + c->toc1);
      double sstnew = c->sstm[o] + c->tdto * (c->omrhs[o] + hmoinv * (c->rrcpoc * c->fnetoc[o] + diabat));
      const double xfoent = -(0.5 * dtoinv) * c->wekto[o] * (c->sstm[o] - c->toc1);
      const double dtonew = c->toc1 - sstnew;
      const double coneno = entfac * fmax(0.0, dtonew);
      xfo[o] = xfoent - coneno;
      sstnew = sstnew + fmax(0.0, dtonew);
      cfrasm = cfrasm + (0.5 - copysign(0.5, -dtonew));
      centsm = centsm - coneno;
      c->sstm[o] = c->sst[o];
      c->sst[o] = sstnew;
    }
  double xfosum = 0.0;
  for (int j = 1; j <= nyt; ++j) {
    double xfsi = 0.0;
    for (int i = 1; i <= nxt; ++i) xfsi = xfsi + xfo[(size_t)(i - 1) + (size_t)nxt * (j - 1)];
    xfosum = xfosum + xfsi;
  }
  for (size_t o = 0; o < (size_t)nxt * nyt; ++o) xfo[o] = xfo[o] - xfosum * ocnorm;
#define XF(ii, jj) xfo[(size_t)((ii)-1) + (size_t)nxt * ((jj)-1)]
#define EN(ii, jj) ent[(size_t)((ii)-1) + (size_t)nx * ((jj)-1)]
  for (int j = 2; j <= ny - 1; ++j)
    for (int i = 2; i <= nx - 1; ++i) EN(i, j) = 0.25 * (XF(i - 1, j - 1) + XF(i, j - 1) + XF(i - 1, j) + XF(i, j));
  for (int i = 2; i <= nx - 1; ++i) {
    EN(i, 1) = 0.5 * (XF(i - 1, 1) + XF(i, 1));
    EN(i, ny) = 0.5 * (XF(i - 1, nyt) + XF(i, nyt));
  }
  if (cyc) {
    for (int j = 2; j <= ny - 1; ++j) {
      EN(1, j) = 0.25 * (XF(nxt, j - 1) + XF(1, j - 1) + XF(nxt, j) + XF(1, j));
      EN(nx, j) = EN(1, j);
    }
    EN(1, 1) = 0.5 * (XF(nxt, 1) + XF(1, 1));
    EN(1, ny) = 0.5 * (XF(nxt, nyt) + XF(1, nyt));
    EN(nx, 1) = EN(1, 1);
    EN(nx, ny) = EN(1, ny);
  } else {
    for (int j = 2; j <= ny - 1; ++j) {
      EN(1, j) = 0.5 * (XF(1, j - 1) + XF(1, j));
      EN(nx, j) = 0.5 * (XF(nxt, j - 1) + XF(nxt, j));
    }
    EN(1, 1) = XF(1, 1);
    EN(nx, 1) = XF(nxt, 1);
    EN(1, ny) = XF(1, nyt);
    EN(nx, ny) = XF(nxt, nyt);
  }
  c->cfraoc = cfrasm * ocnorm;
  c->centoc = centsm * c->dxo * c->dyo;
  c->xon[0] = qgo_xintp(ent, nx, ny);
  c->xon[0] = c->xon[0] * c->dxo * c->dyo;
  if (cyc) {
    double ensums = 0.5 * EN(1, 1), ensumn = 0.5 * EN(1, ny);
    for (int i = 2; i <= nx - 1; ++i) {
      ensums = ensums + EN(i, 1);
      ensumn = ensumn + EN(i, ny);
    }
    ensums = ensums + 0.5 * EN(nx, 1);
    ensumn = ensumn + 0.5 * EN(nx, ny);
    c->enisoc[0] = c->dxo * ensums;
    c->eninoc[0] = c->dxo * ensumn;
  }
#undef XF
#undef EN
}

/* src/q-gcm.F:1232-1249 + 1328-1366 with the mixed layer on */
void qgo_steps_oml(qgo_ctx *c, int s0, int n) {
  for (int s = s0; s < s0 + n; ++s) {
    qgo_oml(c);
    qgo_qgostep(c);
    qgo_ocinvq(c);
    qgo_ocqbdy(c);
    if ((s - 1) % 25 == 0) {
      qgo_lf_average(c);
      for (size_t o = 0; o < (size_t)c->nxt * (c->ny - 1); ++o) c->sst[o] = 0.5 * (c->sst[o] + c->sstm[o]);
    }
  }
}


/* ================================================================== */
/* valids, ocean part: src/valsubs.F:272-527.        */
/* out = pocmin,pocmax,qocmin,qocmax,sstmin,sstmax,wekmin,wekmax,       */
/*       hfmint,hfmaxt,hfmini,hfmaxi,hfminb,hfmaxb, hfbad(1..nlo) [%]   */
/* returns solnok.  sst / wekto are scanned only when the mixed layer   */
/* is initialised (otherwise their entries keep +/-bignum).             */
/* ================================================================== */
int qgo_valids(qgo_ctx *c, const double *dtopoc, double *out) {
  const int nx = c->nx, ny = c->ny, nl = c->nl;
  const double bignum = 1.0e30, wtoext = 1.0e-3, sstext = 75.0, pocext = 1.0e4, qocext = 0.05; /* :78-82 */
  const double thkmin = 100.0, critpc = 20.0;                                                    /* :96-97 */
  double mn[7], mx[7];
  for (int q = 0; q < 7; ++q) { mn[q] = bignum; mx[q] = -bignum; }
#define MM(q, v) do { double v_ = (v); if (v_ < mn[q]) mn[q] = v_; if (v_ > mx[q]) mx[q] = v_; } while (0)
  for (size_t t = 0; t < (size_t)nx * ny * nl; ++t) { MM(0, c->po[t]); MM(1, c->qo[t]); }
  if (c->sst)
    for (size_t t = 0; t < (size_t)c->nxt * (ny - 1); ++t) { MM(2, c->sst[t]); MM(3, c->wekto[t]); }
  double rg[64], eta[64], hfbad[64];
  for (int k = 0; k < nl - 1; ++k) rg[k] = 1.0 / c->gpoc[k];
  const size_t N = (size_t)nx * ny;
  for (int pass = 0; pass < 2; ++pass) {
    double hfmina = fmin(mn[4], fmin(mn[5], mn[6]));
    if (pass == 1 && !(hfmina <= thkmin)) break;
    for (int k = 0; k < nl; ++k) hfbad[k] = 0.0;
    for (int j = 1; j <= ny; ++j)
      for (int i = 1; i <= nx; ++i) {
        const size_t o = (size_t)(i - 1) + (size_t)nx * (j - 1);
        const double w = ((i == 1 || i == nx) ? 0.5 : 1.0) * ((j == 1 || j == ny) ? 0.5 : 1.0);
        for (int k = 0; k < nl - 1; ++k) eta[k] = rg[k] * (c->po[o + N * (k + 1)] - c->po[o + N * k]);
        double hf = c->hoc[0] - eta[0];
        if (pass == 0) MM(4, hf); else if (hf < thkmin) hfbad[0] += w;
        for (int k = 1; k < nl - 1; ++k) {
          hf = c->hoc[k] - eta[k] + eta[k - 1];
          if (pass == 0) MM(5, hf); else if (hf < thkmin) hfbad[k] += w;
        }
        hf = c->hoc[nl - 1] + eta[nl - 2] - (dtopoc ? dtopoc[o] : 0.0);
        if (pass == 0) MM(6, hf); else if (hf < thkmin) hfbad[nl - 1] += w;
      }
  }
#undef MM
  const double ocnorm = 1.0 / ((double)c->nxt * (double)(ny - 1));
  for (int k = 0; k < nl; ++k) hfbad[k] = 100.0 * hfbad[k] * ocnorm;
  int ok = 1;
  if (fabs(mn[0]) >= pocext || fabs(mx[0]) >= pocext) ok = 0;
  if (fabs(mn[1]) >= qocext || fabs(mx[1]) >= qocext) ok = 0;
  if (c->sst) {
    if (fabs(mn[2]) >= sstext || fabs(mx[2]) >= sstext) ok = 0;
    if (fabs(mn[3]) >= wtoext || fabs(mx[3]) >= wtoext) ok = 0;
  }
  for (int k = 0; k < nl; ++k)
    if (hfbad[k] > critpc) ok = 0; /* spfail = .false.: percentage criterion, :507-512 */
  for (int q = 0; q < 7; ++q) { out[2 * q] = mn[q]; out[2 * q + 1] = mx[q]; }
  for (int k = 0; k < nl; ++k) out[14 + k] = hfbad[k];
  return ok;
}
