/* ba.c -- oracle (TEST INFRASTRUCTURE ONLY; edge types pinned by oracle/_ref/libsvs_ref_edges.so, the g2o solver unpinned, see svs_oracle.h) for the
 * double-window bundle adjustment SlamGraph::optimize (slam_graph.cpp:312-355).
 *
 * The reference's own code on this path is only the g2o vertex/edge types
 * (g2o_types/anchored_points.cpp:26-58,78-83,148-235 + transformations.h:62-95) and the
 * marshalling (slam_graph.cpp:907-1080, slam_graph-impl.cpp:28-126).  The solver is
 * [3rd-party: g2o, unpinned HEAD of strasdat/g2o c. 2012: BlockSolver_6_3 +
 * OptimizationAlgorithmLevenberg + RobustKernelHuber + LinearSolverCSparse], restated from its
 * published algorithm:
 *   - edge quadratic form: b_i += J_i^T (rho1 Omega) (-e), H_ii += J_i^T (rho1 Omega) J_i,
 *     H_ij += J_i^T (rho1 Omega) J_j for i<j (position in the edge);
 *   - Huber: rho0 = e2 (sqrt(e2)<=delta) else 2 delta sqrt(e2) - delta^2; rho1 = 1 or
 *     delta/sqrt(e2);
 *   - LM: lambda0 = userLambdaInit, nu=2; per trial add lambda to every diagonal entry of
 *     H_pp and H_ll, Schur-solve, x (+), rho = (chi2 - chi2_new)/(sum x(lambda x + b) + 1e-3),
 *     accept iff rho>0 and finite; lambda *= max(1/3, min(2/3, 1-(2 rho-1)^3)) / lambda *= nu,
 *     nu *= 2; at most maxTrialsAfterFailure trials; Terminate if trials exhausted or rho==0;
 *   - Schur: Dinv = (H_ll + lambda I)^-1 (closed-form 3x3), H_schur = H_pp - sum W Dinv W^T,
 *     b_schur = b_p - sum W Dinv b_l, dense Cholesky here instead of CSparse (same solution up
 *     to rounding), x_l = Dinv (b_l - W^T x_p).
 */
#include "svs_oracle.h"
#include "svs_math.h"
#include <stdlib.h>
#include <float.h>

/* anchored_points.cpp:33-50 stereocam_uvu_map */
static inline void stereo_map(const svs_cam *c, const double *y, double *o) {
  o[0] = (y[0] / y[2]) * c->f + c->cx;
  o[1] = (y[1] / y[2]) * c->f + c->cy;
  o[2] = ((y[0] - c->b) / y[2]) * c->f + c->cx;
}
/* maths_utils.h:66-69 */
static inline void invert_depth(const double *x, double *o) {
  double a = x[0] / x[2], b = x[1] / x[2], c = 1. / x[2];
  o[0] = a; o[1] = b; o[2] = c;
}
static void edge_error(const double *psi, const double *T_obs, const double *T_anc,
                       const double *obs, const svs_cam *cam, double *err) {
  double Tai[12], Tca[12], xa[3], y[3], pr[3];
  pose_inv(T_anc, Tai);
  pose_mul(T_obs, Tai, Tca);
  invert_depth(psi, xa);
  pose_act(Tca, xa, y);
  stereo_map(cam, y, pr);
  for (int i = 0; i < 3; ++i) err[i] = obs[i] - pr[i];
}
/* anchored_points.cpp:148-189 computeError + linearizeOplus */
void svs_ref_edge_psi2uvu(const double *psi, const double *T_obs, const double *T_anc,
                          const double *obs, const svs_cam *cam, double *err, double *Jp,
                          double *Jo, double *Ja) {
  double Tai[12], Tca[12], xa[3], y[3], pr[3], R[9];
  pose_inv(T_anc, Tai);
  pose_mul(T_obs, Tai, Tca);
  invert_depth(psi, xa);
  pose_act(Tca, xa, y);
  stereo_map(cam, y, pr);
  for (int i = 0; i < 3; ++i) err[i] = obs[i] - pr[i];
  pose_R(Tca, R);
  /* transformations.h:62-71 d_stereoproj_d_y */
  double f = cam->f, b = cam->b, zsq = y[2] * y[2];
  double Jc[9] = {f / y[2], 0, -(f * y[0]) / zsq, 0, f / y[2], -(f * y[1]) / zsq, f / y[2], 0, -(f * (y[0] - b)) / zsq};
  /* transformations.h:82-95 d_Tinvpsi_d_psi: [r1, r2, -R x] / psi_z */
  double Rx[3], D[9];
  m3_vec(R, xa, Rx);
  double ipz = 1. / psi[2];
  for (int i = 0; i < 3; ++i) { D[3 * i] = R[3 * i] * ipz; D[3 * i + 1] = R[3 * i + 1] * ipz; D[3 * i + 2] = -Rx[i] * ipz; }
  double t[9];
  m3_mul(Jc, D, t);
  for (int i = 0; i < 9; ++i) Jp[i] = -t[i];
  /* transformations.h:73-80 d_expy_d_y = [I, -hat(y)] */
  double hy[9], hx[9];
  hat3(y, hy); hat3(xa, hx);
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) {
      Jo[6 * i + j] = -Jc[3 * i + j];
      double s = 0; for (int k = 0; k < 3; ++k) s += Jc[3 * i + k] * (-hy[3 * k + j]);
      Jo[6 * i + 3 + j] = -s;
    }
  }
  double JR[9];
  m3_mul(Jc, R, JR);
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) {
      Ja[6 * i + j] = JR[3 * i + j];
      double s = 0; for (int k = 0; k < 3; ++k) s += JR[3 * i + k] * (-hx[3 * k + j]);
      Ja[6 * i + 3 + j] = s;
    }
  }
}

/* anchored_points.cpp:207-235 */
static void third(const double *A, const double *d, double *out) {
  double Adj[36], dl[36], t1[36], t2[36];
  se3_adj(A, Adj);
  se3_dlie(d, dl);
  m6_mul(dl, Adj, t1);
  m6_mul(dl, t1, t2);
  for (int i = 0; i < 36; ++i) out[i] = Adj[i] + 0.5 * t1[i] + (1. / 12.) * t2[i];
}
static void edge_se3_error(const double *T21, const double *T1, const double *T2, double *err) {
  double T2i[12], t[12];
  pose_inv(T2, T2i);
  pose_mul(T21, T1, t);
  pose_mul(t, T2i, t);
  se3_log(t, err);
}
void svs_ref_edge_se3(const double *T21, const double *T1, const double *T2, double *err,
                      double *J1, double *J2) {
  edge_se3_error(T21, T1, T2, err);
  double I[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0}, nd[6];
  third(T21, err, J1);
  for (int i = 0; i < 6; ++i) nd[i] = -err[i];
  third(I, nd, J2);
  for (int i = 0; i < 36; ++i) J2[i] = -J2[i];
}

static inline void huber(double e2, double delta, int robust, double *rho0, double *rho1) {
  if (!robust) { *rho0 = e2; *rho1 = 1; return; }
  double dsqr = delta * delta;
  if (e2 <= dsqr) { *rho0 = e2; *rho1 = 1.; }
  else { double s = sqrt(e2); *rho0 = 2 * s * delta - dsqr; *rho1 = delta / s; }
}

double svs_ref_ba_chi2(int P, const double *poses, int L, const double *psi, int E,
                       const svs_ba_edge *edges, int C, const svs_ba_constraint *cons,
                       const svs_cam *cam, const svs_ba_params *prm) {
  (void)P; (void)L;
  double chi = 0;
  for (int e = 0; e < E; ++e) {
    const svs_ba_edge *ed = &edges[e];
    double err[3];
    edge_error(psi + 3 * ed->point, poses + 12 * ed->pose, poses + 12 * ed->anchor, ed->obs, cam, err);
    double e2 = err[0] * err[0] * ed->info[0] + err[1] * err[1] * ed->info[1] + err[2] * err[2] * ed->info[2];
    double r0, r1;
    huber(e2, prm->huber_delta, prm->use_robust, &r0, &r1);
    chi += r0;
  }
  for (int c = 0; c < C; ++c) {
    double err[6], oe[6];
    edge_se3_error(cons[c].T_21, poses + 12 * cons[c].pose1, poses + 12 * cons[c].pose2, err);
    double e2 = 0;
    for (int i = 0; i < 6; ++i) { oe[i] = 0; for (int j = 0; j < 6; ++j) oe[i] += cons[c].info[6 * i + j] * err[j]; e2 += err[i] * oe[i]; }
    chi += e2;
  }
  return chi;
}

/* full (undamped) normal equations, dense blocks:
   Hpp [6P x 6P] full symmetric, bp [6P], Hll [L][9], bl [L][3],
   W stored per edge slot: Wobs[E][18] (6x3, pose=edge.pose), Wanc[E][18] (pose=edge.anchor) */
typedef struct {
  int P, L, E;
  double *Hpp, *bp, *Hll, *bl, *Wobs, *Wanc;
} ba_sys;

static void add_block66(double *H, int n, int bi, int bj, const double *M /*6x6*/, int transpose) {
  for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j)
    H[(size_t)(6 * bi + i) * n + 6 * bj + j] += transpose ? M[6 * j + i] : M[6 * i + j];
}

static void build_system(ba_sys *S, const double *poses, const double *psi,
                         const svs_ba_edge *edges, int C, const svs_ba_constraint *cons,
                         const svs_cam *cam, const svs_ba_params *prm) {
  int P = S->P, L = S->L, E = S->E, n = 6 * P;
  memset(S->Hpp, 0, sizeof(double) * (size_t)n * n);
  memset(S->bp, 0, sizeof(double) * n);
  memset(S->Hll, 0, sizeof(double) * 9 * (size_t)L);
  memset(S->bl, 0, sizeof(double) * 3 * (size_t)L);
  memset(S->Wobs, 0, sizeof(double) * 18 * (size_t)E);
  memset(S->Wanc, 0, sizeof(double) * 18 * (size_t)E);
  for (int e = 0; e < E; ++e) {
    const svs_ba_edge *ed = &edges[e];
    double err[3], Jp[9], Jo[18], Ja[18];
    svs_ref_edge_psi2uvu(psi + 3 * ed->point, poses + 12 * ed->pose, poses + 12 * ed->anchor, ed->obs, cam, err, Jp, Jo, Ja);
    double e2 = err[0] * err[0] * ed->info[0] + err[1] * err[1] * ed->info[1] + err[2] * err[2] * ed->info[2];
    double r0, r1;
    huber(e2, prm->huber_delta, prm->use_robust, &r0, &r1);
    double om[3] = {r1 * ed->info[0], r1 * ed->info[1], r1 * ed->info[2]};
    double wr[3] = {-om[0] * err[0], -om[1] * err[1], -om[2] * err[2]};   /* omega_r = -rho1 Omega e */
    /* point block */
    double *Hl = S->Hll + 9 * (size_t)ed->point, *bl = S->bl + 3 * (size_t)ed->point;
    for (int i = 0; i < 3; ++i) {
      for (int j = 0; j < 3; ++j) { double s = 0; for (int k = 0; k < 3; ++k) s += Jp[3 * k + i] * om[k] * Jp[3 * k + j]; Hl[3 * i + j] += s; }
      double s = 0; for (int k = 0; k < 3; ++k) s += Jp[3 * k + i] * wr[k]; bl[i] += s;
    }
    double Moo[36], Maa[36], Moa[36], Wo[18], Wa[18], bo[6], ba[6];
    for (int i = 0; i < 6; ++i) {
      for (int j = 0; j < 6; ++j) {
        double soo = 0, saa = 0, soa = 0;
        for (int k = 0; k < 3; ++k) {
          soo += Jo[6 * k + i] * om[k] * Jo[6 * k + j];
          saa += Ja[6 * k + i] * om[k] * Ja[6 * k + j];
          soa += Jo[6 * k + i] * om[k] * Ja[6 * k + j];
        }
        Moo[6 * i + j] = soo; Maa[6 * i + j] = saa; Moa[6 * i + j] = soa;
      }
      for (int j = 0; j < 3; ++j) {
        double so = 0, sa = 0;
        for (int k = 0; k < 3; ++k) { so += Jo[6 * k + i] * om[k] * Jp[3 * k + j]; sa += Ja[6 * k + i] * om[k] * Jp[3 * k + j]; }
        Wo[3 * i + j] = so; Wa[3 * i + j] = sa;
      }
      double so = 0, sa = 0;
      for (int k = 0; k < 3; ++k) { so += Jo[6 * k + i] * wr[k]; sa += Ja[6 * k + i] * wr[k]; }
      bo[i] = so; ba[i] = sa;
    }
    if (ed->pose == ed->anchor) {
      /* same vertex in slots 1 and 2; J_anc = -J_obs exactly.
         G2O_LITERAL: diag += Moo + Maa + Moa (the (1,2) pair maps onto the diagonal block and is
         added once) = +M;  b and W contributions cancel.  EXACT: nothing for the pose. */
      if (prm->self_edge_mode == 0) {
        add_block66(S->Hpp, n, ed->pose, ed->pose, Moo, 0);
        add_block66(S->Hpp, n, ed->pose, ed->pose, Maa, 0);
        add_block66(S->Hpp, n, ed->pose, ed->pose, Moa, 0);
        for (int i = 0; i < 6; ++i) S->bp[6 * ed->pose + i] += bo[i] + ba[i];
        for (int i = 0; i < 18; ++i) S->Wobs[18 * (size_t)e + i] = Wo[i] + Wa[i];
      }
      continue;
    }
    add_block66(S->Hpp, n, ed->pose, ed->pose, Moo, 0);
    add_block66(S->Hpp, n, ed->anchor, ed->anchor, Maa, 0);
    add_block66(S->Hpp, n, ed->pose, ed->anchor, Moa, 0);
    add_block66(S->Hpp, n, ed->anchor, ed->pose, Moa, 1);
    for (int i = 0; i < 6; ++i) { S->bp[6 * ed->pose + i] += bo[i]; S->bp[6 * ed->anchor + i] += ba[i]; }
    memcpy(S->Wobs + 18 * (size_t)e, Wo, sizeof Wo);
    memcpy(S->Wanc + 18 * (size_t)e, Wa, sizeof Wa);
  }
  for (int c = 0; c < C; ++c) {
    double err[6], J1[36], J2[36], OJ1[36], OJ2[36], M[36], oe[6];
    const svs_ba_constraint *cc = &cons[c];
    svs_ref_edge_se3(cc->T_21, poses + 12 * cc->pose1, poses + 12 * cc->pose2, err, J1, J2);
    m6_mul(cc->info, J1, OJ1);
    m6_mul(cc->info, J2, OJ2);
    for (int i = 0; i < 6; ++i) { oe[i] = 0; for (int j = 0; j < 6; ++j) oe[i] += cc->info[6 * i + j] * err[j]; }
    /* J1^T O J1, J2^T O J2, J1^T O J2 */
    for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) { double s = 0; for (int k = 0; k < 6; ++k) s += J1[6 * k + i] * OJ1[6 * k + j]; M[6 * i + j] = s; }
    add_block66(S->Hpp, n, cc->pose1, cc->pose1, M, 0);
    for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) { double s = 0; for (int k = 0; k < 6; ++k) s += J2[6 * k + i] * OJ2[6 * k + j]; M[6 * i + j] = s; }
    add_block66(S->Hpp, n, cc->pose2, cc->pose2, M, 0);
    for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) { double s = 0; for (int k = 0; k < 6; ++k) s += J1[6 * k + i] * OJ2[6 * k + j]; M[6 * i + j] = s; }
    add_block66(S->Hpp, n, cc->pose1, cc->pose2, M, 0);
    add_block66(S->Hpp, n, cc->pose2, cc->pose1, M, 1);
    for (int i = 0; i < 6; ++i) {
      double s1 = 0, s2 = 0;
      for (int k = 0; k < 6; ++k) { s1 += J1[6 * k + i] * oe[k]; s2 += J2[6 * k + i] * oe[k]; }
      S->bp[6 * cc->pose1 + i] -= s1; S->bp[6 * cc->pose2 + i] -= s2;
    }
  }
}

/* edges grouped per landmark (CSR) */
static void landmark_csr(int L, int E, const svs_ba_edge *edges, int *start, int *idx) {
  for (int l = 0; l <= L; ++l) start[l] = 0;
  for (int e = 0; e < E; ++e) start[edges[e].point + 1]++;
  for (int l = 0; l < L; ++l) start[l + 1] += start[l];
  int *fill = (int *)calloc((size_t)L + 1, sizeof(int));
  for (int e = 0; e < E; ++e) { int l = edges[e].point; idx[start[l] + fill[l]++] = e; }
  free(fill);
}

/* Schur reduction with damping lambda (BlockSolver::solve, Schur branch).
   Per landmark: gather W per distinct pose, Dinv, outer products. */
static void schur_reduce(const ba_sys *S, const svs_ba_edge *edges, const int *start, const int *idx,
                         double lambda, double *Hred, double *bred, double *Dinv_out) {
  int P = S->P, L = S->L, n = 6 * P;
  memcpy(Hred, S->Hpp, sizeof(double) * (size_t)n * n);
  memcpy(bred, S->bp, sizeof(double) * n);
  for (int i = 0; i < n; ++i) Hred[(size_t)i * n + i] += lambda;
  int *slot_of = (int *)malloc(sizeof(int) * P);
  for (int p = 0; p < P; ++p) slot_of[p] = -1;
  for (int l = 0; l < L; ++l) {
    int ne = start[l + 1] - start[l];
    if (ne == 0) continue;
    double D[9], Di[9];
    memcpy(D, S->Hll + 9 * (size_t)l, sizeof D);
    D[0] += lambda; D[4] += lambda; D[8] += lambda;
    m3_inv(D, Di);
    if (Dinv_out) memcpy(Dinv_out + 9 * (size_t)l, Di, sizeof Di);
    int np = 0;
    int *pl = (int *)malloc(sizeof(int) * 2 * (size_t)ne);
    double *W = (double *)calloc(18 * 2 * (size_t)ne, sizeof(double));
    for (int k = 0; k < ne; ++k) {
      int e = idx[start[l] + k];
      int ps[2] = {edges[e].pose, edges[e].anchor};
      const double *src[2] = {S->Wobs + 18 * (size_t)e, S->Wanc + 18 * (size_t)e};
      for (int q = 0; q < 2; ++q) {
        if (q == 1 && ps[1] == ps[0]) break;  /* self edge: combined into Wobs */
        int s = slot_of[ps[q]];
        if (s < 0) { s = np++; slot_of[ps[q]] = s; pl[s] = ps[q]; }
        for (int i = 0; i < 18; ++i) W[18 * s + i] += src[q][i];
      }
    }
    const double *bl = S->bl + 3 * (size_t)l;
    double Db[3];
    m3_vec(Di, bl, Db);
    for (int a = 0; a < np; ++a) {
      double WD[18];
      for (int i = 0; i < 6; ++i) for (int j = 0; j < 3; ++j) { double s = 0; for (int k = 0; k < 3; ++k) s += W[18 * a + 3 * i + k] * Di[3 * k + j]; WD[3 * i + j] = s; }
      for (int i = 0; i < 6; ++i) { double s = 0; for (int k = 0; k < 3; ++k) s += W[18 * a + 3 * i + k] * Db[k]; bred[6 * pl[a] + i] -= s; }
      for (int c = 0; c < np; ++c) {
        for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) {
          double s = 0; for (int k = 0; k < 3; ++k) s += WD[3 * i + k] * W[18 * c + 3 * j + k];
          Hred[(size_t)(6 * pl[a] + i) * n + 6 * pl[c] + j] -= s;
        }
      }
    }
    for (int a = 0; a < np; ++a) slot_of[pl[a]] = -1;
    free(pl); free(W);
  }
  free(slot_of);
}

/* Cholesky solve A x = b (A symmetric n x n row-major, destroyed); returns 0 ok, 1 not PD.
   Envelope (skyline) form: first[i] = first structurally non-zero column of row i of the lower
   triangle; Cholesky creates no fill left of it, so the inner products start there.  Skipping
   exact zeros changes no sum, i.e. the result is bit-identical to the dense loop; it is what makes
   this a fair stand-in for the reference's sparse solver (LinearSolverCSparse) on the CPU. */
static int chol_solve(int n, double *A, const double *b, double *x) {
  int *first = (int *)malloc(sizeof(int) * (size_t)n);
  for (int i = 0; i < n; ++i) { int j = 0; while (j < i && A[(size_t)i * n + j] == 0.0) ++j; first[i] = j; }
  int rc = 0;
  for (int j = 0; j < n && !rc; ++j) {
    double d = A[(size_t)j * n + j];
    for (int k = first[j]; k < j; ++k) d -= A[(size_t)j * n + k] * A[(size_t)j * n + k];
    if (!(d > 0) || !isfinite(d)) { rc = 1; break; }
    d = sqrt(d);
    A[(size_t)j * n + j] = d;
    for (int i = j + 1; i < n; ++i) {
      if (first[i] > j) continue;
      double s = A[(size_t)i * n + j];
      int k0 = first[i] > first[j] ? first[i] : first[j];
      for (int k = k0; k < j; ++k) s -= A[(size_t)i * n + k] * A[(size_t)j * n + k];
      A[(size_t)i * n + j] = s / d;
    }
  }
  if (!rc) {
    for (int i = 0; i < n; ++i) { double s = b[i]; for (int k = first[i]; k < i; ++k) s -= A[(size_t)i * n + k] * x[k]; x[i] = s / A[(size_t)i * n + i]; }
    for (int i = n - 1; i >= 0; --i) {
      double s = x[i];
      for (int k = i + 1; k < n; ++k) if (first[k] <= i) s -= A[(size_t)k * n + i] * x[k];
      x[i] = s / A[(size_t)i * n + i];
    }
  }
  free(first);
  return rc;
}

static ba_sys sys_alloc(int P, int L, int E) {
  ba_sys S; S.P = P; S.L = L; S.E = E;
  int n = 6 * P;
  S.Hpp = (double *)malloc(sizeof(double) * (size_t)n * n);
  S.bp = (double *)malloc(sizeof(double) * n);
  S.Hll = (double *)malloc(sizeof(double) * 9 * (size_t)(L ? L : 1));
  S.bl = (double *)malloc(sizeof(double) * 3 * (size_t)(L ? L : 1));
  S.Wobs = (double *)malloc(sizeof(double) * 18 * (size_t)(E ? E : 1));
  S.Wanc = (double *)malloc(sizeof(double) * 18 * (size_t)(E ? E : 1));
  return S;
}
static void sys_free(ba_sys *S) { free(S->Hpp); free(S->bp); free(S->Hll); free(S->bl); free(S->Wobs); free(S->Wanc); }

int svs_ref_ba_reduced_system(int P, const double *poses, int L, const double *psi, int E,
                              const svs_ba_edge *edges, int C, const svs_ba_constraint *cons,
                              const svs_cam *cam, const svs_ba_params *prm, double lambda,
                              double *Hred, double *bred) {
  ba_sys S = sys_alloc(P, L, E);
  int *start = (int *)malloc(sizeof(int) * ((size_t)L + 1)), *idx = (int *)malloc(sizeof(int) * (size_t)(E ? E : 1));
  landmark_csr(L, E, edges, start, idx);
  build_system(&S, poses, psi, edges, C, cons, cam, prm);
  schur_reduce(&S, edges, start, idx, lambda, Hred, bred, 0);
  free(start); free(idx); sys_free(&S);
  return 0;
}

/* The same accumulation (build + Schur reduction of one LM trial) spread over `threads` host threads -- context for the
   GPU numbers only (upstream g2o guards its landmark loop with an OpenMP pragma).  Landmarks are dealt to the
   threads in contiguous ranges; every thread builds and reduces the system of its own edges into a private dense
   (6P)^2 matrix, the matrices are summed at the end.  Same result as svs_ref_ba_reduced_system up to summation order. */
#include <pthread.h>
typedef struct {
  int P, L, E, C, t, T;
  const double *poses, *psi; const svs_ba_edge *edges; const svs_ba_constraint *cons; const svs_cam *cam; const svs_ba_params *prm;
  double lambda, *Hred, *bred;
} mt_job;
static void *mt_worker(void *arg) {
  mt_job *J = (mt_job *)arg;
  const int l0 = (int)((long long)J->L * J->t / J->T), l1 = (int)((long long)J->L * (J->t + 1) / J->T), n = 6 * J->P;
  int Es = 0;
  for (int e = 0; e < J->E; ++e) Es += J->edges[e].point >= l0 && J->edges[e].point < l1;
  svs_ba_edge *sub = (svs_ba_edge *)malloc(sizeof(svs_ba_edge) * (size_t)(Es ? Es : 1));
  for (int e = 0, k = 0; e < J->E; ++e) if (J->edges[e].point >= l0 && J->edges[e].point < l1) sub[k++] = J->edges[e];
  ba_sys S = sys_alloc(J->P, J->L, Es);
  int *start = (int *)malloc(sizeof(int) * ((size_t)J->L + 1)), *idx = (int *)malloc(sizeof(int) * (size_t)(Es ? Es : 1));
  landmark_csr(J->L, Es, sub, start, idx);
  build_system(&S, J->poses, J->psi, sub, J->t == 0 ? J->C : 0, J->cons, J->cam, J->prm);      /* constraints once */
  schur_reduce(&S, sub, start, idx, J->lambda, J->Hred, J->bred, 0);
  if (J->t != 0) for (int i = 0; i < n; ++i) J->Hred[(size_t)i * n + i] -= J->lambda;          /* pose damping once */
  free(start); free(idx); free(sub); sys_free(&S);
  return 0;
}
int svs_ref_ba_reduced_system_mt(int threads, int P, const double *poses, int L, const double *psi, int E,
                                 const svs_ba_edge *edges, int C, const svs_ba_constraint *cons,
                                 const svs_cam *cam, const svs_ba_params *prm, double lambda,
                                 double *Hred, double *bred) {
  if (threads < 1) threads = 1;
  if (threads > 64) threads = 64;
  const int n = 6 * P;
  pthread_t th[64];
  mt_job jobs[64];
  double *Hs = (double *)malloc(sizeof(double) * (size_t)n * n * threads), *bs = (double *)malloc(sizeof(double) * (size_t)n * threads);
  for (int t = 0; t < threads; ++t) {
    mt_job j = {P, L, E, C, t, threads, poses, psi, edges, cons, cam, prm, lambda, Hs + (size_t)t * n * n, bs + (size_t)t * n};
    jobs[t] = j;
    pthread_create(&th[t], 0, mt_worker, &jobs[t]);
  }
  for (int t = 0; t < threads; ++t) pthread_join(th[t], 0);
  memcpy(Hred, Hs, sizeof(double) * (size_t)n * n);
  memcpy(bred, bs, sizeof(double) * n);
  for (int t = 1; t < threads; ++t) {
    for (size_t i = 0; i < (size_t)n * n; ++i) Hred[i] += Hs[(size_t)t * n * n + i];
    for (int i = 0; i < n; ++i) bred[i] += bs[(size_t)t * n + i];
  }
  free(Hs); free(bs);
  return 0;
}

int svs_ref_ba_optimize(int P, double *poses, int L, double *psi, int E, const svs_ba_edge *edges,
                        int C, const svs_ba_constraint *cons, const svs_cam *cam,
                        const svs_ba_params *prm, svs_ba_stats *stats) {
  int n = 6 * P;
  ba_sys S = sys_alloc(P, L, E);
  int *start = (int *)malloc(sizeof(int) * ((size_t)L + 1)), *idx = (int *)malloc(sizeof(int) * (size_t)(E ? E : 1));
  landmark_csr(L, E, edges, start, idx);
  double *Hred = (double *)malloc(sizeof(double) * (size_t)n * n), *bred = (double *)malloc(sizeof(double) * n);
  double *xp = (double *)calloc(n, sizeof(double)), *xl = (double *)calloc(3 * (size_t)(L ? L : 1), sizeof(double));
  double *Dinv = (double *)malloc(sizeof(double) * 9 * (size_t)(L ? L : 1));
  double *poses_bak = (double *)malloc(sizeof(double) * 12 * (size_t)P), *psi_bak = (double *)malloc(sizeof(double) * 3 * (size_t)(L ? L : 1));
  double lambda = prm->lambda_init, ni = 2;
  svs_ba_stats st; memset(&st, 0, sizeof st);
  int ok = 1;
  for (int it = 0; it < prm->num_iters && ok; ++it) {
    double currentChi = svs_ref_ba_chi2(P, poses, L, psi, E, edges, C, cons, cam, prm);
    if (it == 0) st.chi2_init = currentChi;
    double tempChi = currentChi;
    build_system(&S, poses, psi, edges, C, cons, cam, prm);
    if (it == 0) { lambda = prm->lambda_init; ni = 2; }
    double rho = 0;
    int qmax = 0;
    do {
      memcpy(poses_bak, poses, sizeof(double) * 12 * (size_t)P);       /* push */
      memcpy(psi_bak, psi, sizeof(double) * 3 * (size_t)L);
      schur_reduce(&S, edges, start, idx, lambda, Hred, bred, Dinv);
      int fail = chol_solve(n, Hred, bred, xp);
      if (!fail) {
        /* x_l = Dinv (b_l - W^T x_p) */
        for (int l = 0; l < L; ++l) {
          double c[3] = {S.bl[3 * l], S.bl[3 * l + 1], S.bl[3 * l + 2]};
          for (int k = start[l]; k < start[l + 1]; ++k) {
            int e = idx[k];
            const double *Wo = S.Wobs + 18 * (size_t)e, *Wa = S.Wanc + 18 * (size_t)e;
            const double *xo = xp + 6 * edges[e].pose, *xa = xp + 6 * edges[e].anchor;
            for (int j = 0; j < 3; ++j) { double s = 0; for (int i = 0; i < 6; ++i) s += Wo[3 * i + j] * xo[i] + Wa[3 * i + j] * xa[i]; c[j] -= s; }
          }
          if (start[l + 1] > start[l]) m3_vec(Dinv + 9 * (size_t)l, c, xl + 3 * (size_t)l);
          else { xl[3 * l] = xl[3 * l + 1] = xl[3 * l + 2] = 0; }
        }
      }
      /* update (oplus): T <- exp(d) T ; psi <- psi + d   (anchored_points.cpp:53-58,78-83) */
      for (int p = 0; p < P; ++p) { double Ex[12]; se3_exp(xp + 6 * p, Ex); pose_mul(Ex, poses + 12 * p, poses + 12 * p); }
      for (int i = 0; i < 3 * L; ++i) psi[i] += xl[i];
      tempChi = svs_ref_ba_chi2(P, poses, L, psi, E, edges, C, cons, cam, prm);
      if (fail) tempChi = DBL_MAX;
      rho = currentChi - tempChi;
      double scale = 0;
      for (int j = 0; j < n; ++j) scale += xp[j] * (lambda * xp[j] + S.bp[j]);
      for (int j = 0; j < 3 * L; ++j) scale += xl[j] * (lambda * xl[j] + S.bl[j]);
      scale += 1e-3;
      rho /= scale;
      ++st.trials;
      if (rho > 0 && isfinite(tempChi)) {
        double alpha = 1. - pow((2 * rho - 1), 3);
        if (alpha > 2. / 3.) alpha = 2. / 3.;
        double sf = alpha > 1. / 3. ? alpha : 1. / 3.;
        lambda *= sf; ni = 2; currentChi = tempChi; ++st.accepted;
      } else {
        lambda *= ni; ni *= 2;
        memcpy(poses, poses_bak, sizeof(double) * 12 * (size_t)P);     /* pop */
        memcpy(psi, psi_bak, sizeof(double) * 3 * (size_t)L);
      }
      ++qmax;
    } while (rho < 0 && qmax < prm->max_trials);
    ++st.iterations;
    st.chi2_final = currentChi;
    if (qmax == prm->max_trials || rho == 0) { ok = 0; st.terminated = 1; }
  }
  st.lambda_final = lambda;
  if (stats) *stats = st;
  free(start); free(idx); free(Hred); free(bred); free(xp); free(xl); free(Dinv); free(poses_bak); free(psi_bak);
  sys_free(&S);
  return 0;
}
