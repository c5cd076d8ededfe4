// Offline assembly kernels.
//
// All of these are HBM-bound streaming kernels: one thread per (subdomain, element) or (subdomain, RT face)
// reads the coefficient samples of one element (at the points of the rules in lrbms_quadrature) and writes a handful of 3x3 blocks.  The connectivity comes from
// the shared subdomain template (a few KB, L1/L2 resident), never from per-subdomain index arrays.
// They run once per discretize() and are outside the timed project+estimate region.
#include "lrbms_dev.h"

struct QVec { double v[8]; };

namespace {

typedef lrbms_quadrature Quad;

// Data of one side of a face seen from element `e`: lambda samples on the face in the parametrisation of the
// element that owns the integration (k runs with OUR parametrisation), kappa grad phi_i . n, basis values.
struct FaceSide {
  double lam[LRBMS_MAXQF];     // lambda at the points of the edge rule
  double kgn[3];               // (kappa grad phi_i) . n  for i = 0..2
  double phi[3][LRBMS_MAXQF];  // phi_i at point k: phi[i][k]
};

__device__ inline void kgrad_dot_n(const Tmpl& t, int e, double nx, double ny, double out[3]) {
  for (int i = 0; i < 3; ++i) {
    double gx = t.grad[(e * 3 + i) * 2 + 0], gy = t.grad[(e * 3 + i) * 2 + 1];
    double kx = t.kappa[0] * gx + t.kappa[1] * gy, ky = t.kappa[2] * gx + t.kappa[3] * gy;
    out[i] = kx * nx + ky * ny;
  }
}

// our side of face f of element e (we run the parametrisation from vertex f+1 to f+2); `smp` points at the face's samples
__device__ inline void load_self_side(const Tmpl& t, const lrbms_edge_rule& r, const double* smp, int e, int f, double nx,
                                      double ny, FaceSide& s) {
  for (int k = 0; k < r.n; ++k) s.lam[k] = smp[k];
  kgrad_dot_n(t, e, nx, ny, s.kgn);
  int a = (f + 1) % 3, b = (f + 2) % 3;
  for (int k = 0; k < r.n; ++k) {
    s.phi[f][k] = 0.0;
    s.phi[a][k] = 1.0 - r.t[k];
    s.phi[b][k] = r.t[k];
  }
}

// the other side: element e2 with local face f2, whose own parametrisation runs the opposite way (the edge rules are
// symmetric about the midpoint: its point n - 1 - k is our point k)
__device__ inline void load_other_side(const Tmpl& t, const lrbms_edge_rule& r, const double* smp2, int e2, int f2, double nx,
                                       double ny, FaceSide& s) {
  for (int k = 0; k < r.n; ++k) s.lam[k] = smp2[r.n - 1 - k];
  kgrad_dot_n(t, e2, nx, ny, s.kgn);
  int a = (f2 + 1) % 3, b = (f2 + 2) % 3;
  for (int k = 0; k < r.n; ++k) {
    s.phi[f2][k] = 0.0;
    s.phi[a][k] = r.t[k];
    s.phi[b][k] = 1.0 - r.t[k];
  }
}

// SWIPDG inner-face blocks seen from the "self" element (weights 1/2 for constant kappa):
//   ss[i][j] += -w (D grad phi_j . n) phi_i - w phi_j (D grad phi_i . n) + sigma phi_j phi_i
//   so[i][j] += -w (D+ grad phi+_j . n) phi_i + w phi+_j (D grad phi_i . n) - sigma phi+_j phi_i
__device__ inline void swipdg_inner(const lrbms_edge_rule& r, const FaceSide& m, const FaceSide& p, double len, double delta,
                                    double ss[9], double so[9]) {
  const double gamma = 0.5 * delta;
  for (int k = 0; k < r.n; ++k) {
    double wq = r.w[k] * len;
    double sigma = 0.5 * (m.lam[k] + p.lam[k]) * SIGMA_INNER * gamma / len;
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) {
        ss[i * 3 + j] += wq * (-0.5 * m.lam[k] * m.kgn[j] * m.phi[i][k] - 0.5 * m.phi[j][k] * m.lam[k] * m.kgn[i] +
                               sigma * m.phi[j][k] * m.phi[i][k]);
        so[i * 3 + j] += wq * (-0.5 * p.lam[k] * p.kgn[j] * m.phi[i][k] + 0.5 * p.phi[j][k] * m.lam[k] * m.kgn[i] -
                               sigma * p.phi[j][k] * m.phi[i][k]);
      }
  }
}

__device__ inline void swipdg_boundary(const lrbms_edge_rule& r, const FaceSide& m, double len, double delta, double ss[9]) {
  for (int k = 0; k < r.n; ++k) {
    double wq = r.w[k] * len;
    double sigma = m.lam[k] * SIGMA_BOUNDARY * delta / len;
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j)
        ss[i * 3 + j] += wq * (-m.lam[k] * m.kgn[j] * m.phi[i][k] - m.phi[j][k] * m.lam[k] * m.kgn[i] +
                               sigma * m.phi[j][k] * m.phi[i][k]);
  }
}

__device__ inline double n_kappa_n(const Tmpl& t, double nx, double ny) {
  return nx * (t.kappa[0] * nx + t.kappa[1] * ny) + ny * (t.kappa[2] * nx + t.kappa[3] * ny);
}

__device__ inline double vol_integral(const Tmpl& t, const lrbms_tri_rule& r, const double* smp, int e) {
  double s = 0.0;
  for (int k = 0; k < r.n; ++k) s += r.w[k] * smp[k];
  return s * t.area[e];
}

__device__ inline void stiffness(const Tmpl& t, int e, double K[9]) {
  for (int i = 0; i < 3; ++i) {
    double gx = t.grad[(e * 3 + i) * 2], gy = t.grad[(e * 3 + i) * 2 + 1];
    for (int j = 0; j < 3; ++j) {
      double hx = t.grad[(e * 3 + j) * 2], hy = t.grad[(e * 3 + j) * 2 + 1];
      K[i * 3 + j] = gx * (t.kappa[0] * hx + t.kappa[1] * hy) + gy * (t.kappa[2] * hx + t.kappa[3] * hy);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// K1-K3.  grid.x over (s, e), grid.y = q.  Faces with a neighbour element inside the subdomain are integrated with the
// inner-face rule (they inherit the operator's over_integrate=2, block_swipdg.py:405), faces of the subdomain boundary
// (coupling or Dirichlet: operators built without over_integrate, :409, :426) with the coupling rule.
__global__ __launch_bounds__(256) void k_assemble_swipdg(Tmpl t, const Quad* __restrict__ qd, int S, int S_ext,
                                                         const int* __restrict__ nbr, const double* __restrict__ lam,
                                                         double* __restrict__ A_diag, double* __restrict__ A_cpl) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)S * t.nT) return;
  const Quad& Q_ = *qd;
  const int LS = Q_.lam_stride, oF = Q_.o_sysf, nfs = Q_.nfs;
  const int q = blockIdx.y;
  const int s = (int)(idx / t.nT), e = (int)(idx % t.nT);
  const double* lam_q = lam + (long)q * S_ext * t.nT * LS;
  const double* lam_e = lam_q + ((long)s * t.nT + e) * LS;

  double blk[4][9];
  double K[9];
  stiffness(t, e, K);
  const double li = vol_integral(t, Q_.system_volume, lam_e + Q_.o_sysv, e);
  for (int i = 0; i < 9; ++i) {
    blk[0][i] = li * K[i];
    blk[1][i] = blk[2][i] = blk[3][i] = 0.0;
  }
  for (int f = 0; f < 3; ++f) {
    const double nx = t.normal[(e * 3 + f) * 2], ny = t.normal[(e * 3 + f) * 2 + 1];
    const double len = t.face_len[e * 3 + f];
    const double delta = n_kappa_n(t, nx, ny);
    FaceSide m, p;
    const int nb = t.nb_elem[e * 3 + f];
    if (nb >= 0) {
      const lrbms_edge_rule& r = Q_.system_inner_face;
      load_self_side(t, r, lam_e + oF + f * nfs, e, f, nx, ny, m);
      const int f2 = t.nb_face[e * 3 + f];
      load_other_side(t, r, lam_q + ((long)s * t.nT + nb) * LS + oF + f2 * nfs, nb, f2, nx, ny, p);
      swipdg_inner(r, m, p, len, delta, blk[0], blk[1 + f]);
    } else {
      const lrbms_edge_rule& r = Q_.system_coupling_face;
      load_self_side(t, r, lam_e + oF + f * nfs, e, f, nx, ny, m);
      const int side = -1 - nb;
      const int s2 = nbr[s * 5 + side_to_slot(side)];
      if (s2 >= 0) {
        const int e2 = t.nb_elem_out[e * 3 + f], f2 = t.nb_face_out[e * 3 + f];
        load_other_side(t, r, lam_q + ((long)s2 * t.nT + e2) * LS + oF + f2 * nfs, e2, f2, nx, ny, p);
        double so[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        swipdg_inner(r, m, p, len, delta, blk[0], so);
        if (t.opt_accumulate_coupling) {
          // the reference allocates its coupling matrices once and assembles every component into them
          // (block_swipdg.py:551-565, :581-583): component q sees the coupling terms of all components q' <= q
          for (int q2 = 0; q2 < q; ++q2) {
            const double* lam_q2 = lam + (long)q2 * S_ext * t.nT * LS;
            FaceSide m2, p2;
            load_self_side(t, r, lam_q2 + ((long)s * t.nT + e) * LS + oF + f * nfs, e, f, nx, ny, m2);
            load_other_side(t, r, lam_q2 + ((long)s2 * t.nT + e2) * LS + oF + f2 * nfs, e2, f2, nx, ny, p2);
            swipdg_inner(r, m2, p2, len, delta, blk[0], so);
          }
        }
        double* out = A_cpl + ((((long)q * S + s) * 4 + side) * t.ncf + t.elem_side_pos[e * 3 + f]) * 9;
        for (int i = 0; i < 9; ++i) out[i] = so[i];
      } else {
        swipdg_boundary(r, m, len, delta, blk[0]);
      }
    }
  }
  double* out = A_diag + (((long)q * S + s) * t.nT + e) * 36;
  for (int b = 0; b < 4; ++b)
    for (int i = 0; i < 9; ++i) out[b * 9 + i] = blk[b][i];
}

// ---------------------------------------------------------------------------------------------------------
// K5 + scalars.  One workgroup per subdomain; fixed-order LDS tree reductions (deterministic).
// b_i = int f phi_i (rule `rhs`), ||f||^2 (rule `f2`), min lambda_hat at the points of rule `ceps`.
__global__ __launch_bounds__(256) void k_assemble_rhs(Tmpl t, const Quad* __restrict__ qd, const double* __restrict__ f_smp,
                                                      const double* __restrict__ lhat, double* __restrict__ b,
                                                      double* __restrict__ f2, double* __restrict__ ceps) {
  __shared__ double red_sum[256];
  __shared__ double red_min[256];
  const Quad& Q_ = *qd;
  const int s = blockIdx.x;
  double acc = 0.0, mn = 1.0e300;
  for (int e = threadIdx.x; e < t.nT; e += blockDim.x) {
    const double* fe = f_smp + ((long)s * t.nT + e) * Q_.f_stride;
    const double* lh = lhat + ((long)s * t.nT + e) * Q_.lhat_stride + Q_.o_hceps;
    double bi[3] = {0, 0, 0}, sq = 0.0;
    for (int k = 0; k < Q_.rhs.n; ++k) {
      double w = Q_.rhs.w[k] * t.area[e];
      for (int i = 0; i < 3; ++i) bi[i] += w * fe[Q_.o_frhs + k] * Q_.rhs.b[k][i];
    }
    for (int k = 0; k < Q_.f2.n; ++k) sq += Q_.f2.w[k] * t.area[e] * fe[Q_.o_ff2 + k] * fe[Q_.o_ff2 + k];
    for (int k = 0; k < Q_.ceps.n; ++k) mn = fmin(mn, lh[k]);
    for (int i = 0; i < 3; ++i) b[(long)s * t.n + 3 * e + i] = bi[i];
    acc += sq;
  }
  red_sum[threadIdx.x] = acc;
  red_min[threadIdx.x] = mn;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) {
      red_sum[threadIdx.x] += red_sum[threadIdx.x + off];
      red_min[threadIdx.x] = fmin(red_min[threadIdx.x], red_min[threadIdx.x + off]);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    f2[s] = red_sum[0];
    ceps[s] = red_min[0] * t.kmin;
  }
}

// ---------------------------------------------------------------------------------------------------------
// K6 + K9.  One thread per (s, e).
__global__ __launch_bounds__(256) void k_assemble_products(Tmpl t, const Quad* __restrict__ qd, int S, int S_ext,
                                                           const int* __restrict__ nbr, int Q, QVec theta_bar,
                                                           const double* __restrict__ lam, const double* __restrict__ lam_df,
                                                           const double* __restrict__ lbar, const double* __restrict__ lhat,
                                                           double* __restrict__ P_diag, double* __restrict__ ebar,
                                                           double* __restrict__ caa, double* __restrict__ Aab,
                                                           double* __restrict__ Bbb) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)S * t.nT) return;
  const Quad& Q_ = *qd;
  const int LS = Q_.lam_stride, LD = Q_.lamdf_stride, LH = Q_.lhat_stride;
  const lrbms_edge_rule& rf = Q_.energy_face;
  const int s = (int)(idx / t.nT), e = (int)(idx % t.nT);
  double K[9];
  stiffness(t, e, K);
  double blk[4][9];
  for (int b = 0; b < 4; ++b)
    for (int i = 0; i < 9; ++i) blk[b][i] = 0.0;

  // ---- energy product: sum_q theta_q(mu_bar) (elliptic_q + penalty_q), over_integrate=0 (block_swipdg.py:655,:660)
  for (int q = 0; q < Q; ++q) {
    const double th = theta_bar.v[q];
    const double* lam_q = lam + (long)q * S_ext * t.nT * LS;
    const double* lam_e = lam_q + ((long)s * t.nT + e) * LS;
    const double li = vol_integral(t, Q_.energy_volume, lam_e + Q_.o_env, e);
    for (int i = 0; i < 9; ++i) blk[0][i] += th * li * K[i];
    for (int f = 0; f < 3; ++f) {
      const double nx = t.normal[(e * 3 + f) * 2], ny = t.normal[(e * 3 + f) * 2 + 1];
      const double len = t.face_len[e * 3 + f];
      const double delta = n_kappa_n(t, nx, ny);
      FaceSide m, p;
      load_self_side(t, rf, lam_e + Q_.o_enf + f * rf.n, e, f, nx, ny, m);
      const int nb = t.nb_elem[e * 3 + f];
      if (nb >= 0) {
        const int f2 = t.nb_face[e * 3 + f];
        load_other_side(t, rf, lam_q + ((long)s * t.nT + nb) * LS + Q_.o_enf + f2 * rf.n, nb, f2, nx, ny, p);
        for (int k = 0; k < rf.n; ++k) {
          double wq = rf.w[k] * len;
          double sigma = 0.5 * (m.lam[k] + p.lam[k]) * SIGMA_INNER * (0.5 * delta) / len;
          for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
              blk[0][i * 3 + j] += th * wq * sigma * m.phi[j][k] * m.phi[i][k];
              blk[1 + f][i * 3 + j] -= th * wq * sigma * p.phi[j][k] * m.phi[i][k];
            }
        }
      } else {  // boundary of the subdomain: all-Dirichlet on the subdomain layer (block_swipdg.py:537-539,:658)
        for (int k = 0; k < rf.n; ++k) {
          double wq = rf.w[k] * len;
          double sigma = m.lam[k] * SIGMA_BOUNDARY * delta / len;
          for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) blk[0][i * 3 + j] += th * wq * sigma * m.phi[j][k] * m.phi[i][k];
        }
      }
    }
  }
  double* pout = P_diag + ((long)s * t.nT + e) * 36;
  for (int b = 0; b < 4; ++b)
    for (int i = 0; i < 9; ++i) pout[b * 9 + i] = blk[b][i];

  // ---- E_ii(lambda_bar) scalar
  ebar[(long)s * t.nT + e] = vol_integral(t, Q_.elliptic_bar, lbar + ((long)s * t.nT + e) * Q_.lbar_stride, e);

  // ---- diffusive-flux products (over_integrate=2, block_swipdg.py:327,:347,:369)
  const double* lh = lhat + ((long)s * t.nT + e) * LH;
  const double area = t.area[e];
  for (int q = 0; q < Q; ++q) {
    const double* lq = lam_df + (((long)q * S + s) * t.nT + e) * LD + Q_.o_aa;
    for (int q2 = 0; q2 < Q; ++q2) {
      const double* lq2 = lam_df + (((long)q2 * S + s) * t.nT + e) * LD + Q_.o_aa;
      double c = 0.0;
      for (int k = 0; k < Q_.df_aa.n; ++k) c += Q_.df_aa.w[k] * lq[k] * lq2[k] / lh[Q_.o_haa + k];
      caa[(((long)q * Q + q2) * S + s) * t.nT + e] = c * area;
    }
  }
  // RT0 basis psi_f(x) = sign_f |e_f| / (2|T|) (x - p_f); sign is +1 on domain-boundary faces
  double coef[3], px[3], py[3];
  for (int f = 0; f < 3; ++f) {
    int sign = t.face_sign[e * 3 + f];
    const int nb = t.nb_elem[e * 3 + f];
    if (nb < 0 && nbr[s * 5 + side_to_slot(-1 - nb)] < 0) sign = 1;
    coef[f] = sign * t.face_len[e * 3 + f] / (2.0 * area);
    px[f] = t.points[(e * 3 + f) * 2];
    py[f] = t.points[(e * 3 + f) * 2 + 1];
  }
  double bb[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int k = 0; k < Q_.df_bb.n; ++k) {
    double x = 0.0, y = 0.0;
    for (int v = 0; v < 3; ++v) {
      x += Q_.df_bb.b[k][v] * px[v];
      y += Q_.df_bb.b[k][v] * py[v];
    }
    double psx[3], psy[3];
    for (int f = 0; f < 3; ++f) {
      psx[f] = coef[f] * (x - px[f]);
      psy[f] = coef[f] * (y - py[f]);
    }
    const double w = Q_.df_bb.w[k] * area;
    for (int f = 0; f < 3; ++f) {
      double kx = t.kinv[0] * psx[f] + t.kinv[1] * psy[f], ky = t.kinv[2] * psx[f] + t.kinv[3] * psy[f];
      for (int g = 0; g < 3; ++g) bb[f * 3 + g] += w / lh[Q_.o_hbb + k] * (kx * psx[g] + ky * psy[g]);
    }
  }
  for (int i = 0; i < 9; ++i) Bbb[((long)s * t.nT + e) * 9 + i] = bb[i];
  for (int q = 0; q < Q; ++q) {
    double ab[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    const double* lq = lam_df + (((long)q * S + s) * t.nT + e) * LD + Q_.o_ab;
    for (int k = 0; k < Q_.df_ab.n; ++k) {
      double x = 0.0, y = 0.0;
      for (int v = 0; v < 3; ++v) {
        x += Q_.df_ab.b[k][v] * px[v];
        y += Q_.df_ab.b[k][v] * py[v];
      }
      const double w = Q_.df_ab.w[k] * area * lq[k] / lh[Q_.o_hab + k];
      for (int i = 0; i < 3; ++i) {
        double gx = t.grad[(e * 3 + i) * 2], gy = t.grad[(e * 3 + i) * 2 + 1];
        for (int f = 0; f < 3; ++f) ab[i * 3 + f] += w * (gx * coef[f] * (x - px[f]) + gy * coef[f] * (y - py[f]));
      }
    }
    for (int i = 0; i < 9; ++i) Aab[(((long)q * S + s) * t.nT + e) * 9 + i] = ab[i];
  }
}

// ---------------------------------------------------------------------------------------------------------
// K8 assembly half: one thread per (s, RT face), grid.y = q.
__global__ __launch_bounds__(256) void k_assemble_flux(Tmpl t, const Quad* __restrict__ qd, int S, int S_ext,
                                                       const int* __restrict__ nbr, const double* __restrict__ lam,
                                                       double* __restrict__ F) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)S * t.nrt) return;
  const Quad& Q_ = *qd;
  const lrbms_edge_rule& rf = Q_.flux_face;
  const int LS = Q_.lam_stride;
  const int q = blockIdx.y;
  const int s = (int)(idx / t.nrt), r = (int)(idx % t.nrt);
  const double* lam_q = lam + (long)q * S_ext * t.nT * LS;
  const int e0 = t.rt_e0[r], f0 = t.rt_f0[r], side = t.rt_side[r];
  int e1 = t.rt_e1[r], f1 = t.rt_f1[r];
  int s1 = s;
  bool boundary = false;
  int sign = t.face_sign[e0 * 3 + f0];
  if (side >= 0) {
    s1 = nbr[s * 5 + side_to_slot(side)];
    if (s1 < 0) {
      boundary = true;
      sign = 1;
    }
  }
  // face normal = sign * outward normal of e0
  const double onx = t.normal[(e0 * 3 + f0) * 2], ony = t.normal[(e0 * 3 + f0) * 2 + 1];
  const double nx = sign * onx, ny = sign * ony;
  const double len = t.face_len[e0 * 3 + f0];
  const double delta = n_kappa_n(t, nx, ny);
  FaceSide m, p;
  load_self_side(t, rf, lam_q + ((long)s * t.nT + e0) * LS + Q_.o_flf + f0 * rf.n, e0, f0, nx, ny, m);
  double c0[3] = {0, 0, 0}, c1[3] = {0, 0, 0};
  if (boundary) {
    for (int k = 0; k < rf.n; ++k) {
      double sigma = m.lam[k] * SIGMA_BOUNDARY * delta / len;
      for (int j = 0; j < 3; ++j) c0[j] += rf.w[k] * (-m.lam[k] * m.kgn[j] + sigma * m.phi[j][k]);
    }
  } else {
    load_other_side(t, rf, lam_q + ((long)s1 * t.nT + e1) * LS + Q_.o_flf + f1 * rf.n, e1, f1, nx, ny, p);
    const double rho = (double)sign;  // +1: e0 is the minus element ( [v] = v^- - v^+ )
    for (int k = 0; k < rf.n; ++k) {
      double sigma = 0.5 * (m.lam[k] + p.lam[k]) * SIGMA_INNER * (0.5 * delta) / len;
      for (int j = 0; j < 3; ++j) {
        c0[j] += rf.w[k] * (-0.5 * m.lam[k] * m.kgn[j] + rho * sigma * m.phi[j][k]);
        c1[j] += rf.w[k] * (-0.5 * p.lam[k] * p.kgn[j] - rho * sigma * p.phi[j][k]);
      }
    }
  }
  double* out = F + (((long)q * S + s) * t.nrt + r) * 6;
  for (int j = 0; j < 3; ++j) {
    out[j] = c0[j];
    out[3 + j] = c1[j];
  }
}

// ---------------------------------------------------------------------------------------------------------
// Neighbourhood (online enrichment) problems treat the outer boundary of N(ii) as a Dirichlet boundary
// (block_swipdg.py:240-247 with the all-Dirichlet local_boundary_info of :794-795).  For a coupling face of
// subdomain s that lies on such a boundary the diagonal block of the inside element changes from the inner-face
// form (already summed into A_diag) to the boundary form:  D_corr = boundary_block - inner_self_block.
// One thread per (s, e), grid.y = q; faces without a neighbouring subdomain keep D_corr = 0 (memset by the launcher).
__global__ __launch_bounds__(256) void k_assemble_dcorr(Tmpl t, const Quad* __restrict__ qd, int S, int S_ext,
                                                        const int* __restrict__ nbr, const double* __restrict__ lam,
                                                        double* __restrict__ D_corr) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)S * t.nT) return;
  const Quad& Q_ = *qd;
  const lrbms_edge_rule& r = Q_.system_coupling_face;   // the rule the block in A_diag was assembled with
  const int LS = Q_.lam_stride, oF = Q_.o_sysf, nfs = Q_.nfs;
  const int q = blockIdx.y;
  const int s = (int)(idx / t.nT), e = (int)(idx % t.nT);
  const double* lam_q = lam + (long)q * S_ext * t.nT * LS;
  const double* lam_e = lam_q + ((long)s * t.nT + e) * LS;
  for (int f = 0; f < 3; ++f) {
    const int nb = t.nb_elem[e * 3 + f];
    if (nb >= 0) continue;
    const int side = -1 - nb;
    const int s2 = nbr[s * 5 + side_to_slot(side)];
    if (s2 < 0) continue;
    const double nx = t.normal[(e * 3 + f) * 2], ny = t.normal[(e * 3 + f) * 2 + 1];
    const double len = t.face_len[e * 3 + f];
    const double delta = n_kappa_n(t, nx, ny);
    FaceSide m, p;
    load_self_side(t, r, lam_e + oF + f * nfs, e, f, nx, ny, m);
    const int e2 = t.nb_elem_out[e * 3 + f], f2 = t.nb_face_out[e * 3 + f];
    load_other_side(t, r, lam_q + ((long)s2 * t.nT + e2) * LS + oF + f2 * nfs, e2, f2, nx, ny, p);
    double ss[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, so[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, bd[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    swipdg_inner(r, m, p, len, delta, ss, so);
    swipdg_boundary(r, m, len, delta, bd);
    double* out = D_corr + ((((long)q * S + s) * 4 + side) * t.ncf + t.elem_side_pos[e * 3 + f]) * 9;
    for (int i = 0; i < 9; ++i) out[i] = bd[i] - ss[i];
  }
}

}  // namespace

int launch_assemble_dcorr(lrbms_ctx* ctx, int Q, const double* lam, double* D_corr, hipStream_t st) {
  const Tmpl& t = ctx->t;
  LRBMS_HIP_CHECK(ctx, hipMemsetAsync(D_corr, 0, sizeof(double) * (size_t)Q * ctx->S * 4 * t.ncf * 9, st));
  long total = (long)ctx->S * t.nT;
  dim3 grid((unsigned)((total + 255) / 256), Q);
  if (!ctx->qdev) return lrbms_fail(ctx, LRBMS_E_STATE, "quadrature not set (lrbms_set_quadrature)");
  hipLaunchKernelGGL(k_assemble_dcorr, grid, dim3(256), 0, st, t, ctx->qdev, ctx->S, ctx->S_ext, ctx->nbr, lam, D_corr);
  LRBMS_LAUNCH_CHECK(ctx);
  return LRBMS_OK;
}

int launch_assemble_swipdg(lrbms_ctx* ctx, int Q, const double* lam, double* A_diag, double* A_cpl, hipStream_t st) {
  const Tmpl& t = ctx->t;
  LRBMS_HIP_CHECK(ctx, hipMemsetAsync(A_cpl, 0, sizeof(double) * (size_t)Q * ctx->S * 4 * t.ncf * 9, st));
  long total = (long)ctx->S * t.nT;
  dim3 grid((unsigned)((total + 255) / 256), Q);
  if (!ctx->qdev) return lrbms_fail(ctx, LRBMS_E_STATE, "quadrature not set (lrbms_set_quadrature)");
  hipLaunchKernelGGL(k_assemble_swipdg, grid, dim3(256), 0, st, t, ctx->qdev, ctx->S, ctx->S_ext, ctx->nbr, lam, A_diag, A_cpl);
  LRBMS_LAUNCH_CHECK(ctx);
  return LRBMS_OK;
}

int launch_assemble_rhs(lrbms_ctx* ctx, const double* f_smp, const double* lhat, double* b, double* f2, double* ceps,
                        hipStream_t st) {
  if (!ctx->qdev) return lrbms_fail(ctx, LRBMS_E_STATE, "quadrature not set (lrbms_set_quadrature)");
  hipLaunchKernelGGL(k_assemble_rhs, dim3(ctx->S), dim3(256), 0, st, ctx->t, ctx->qdev, f_smp, lhat, b, f2, ceps);
  LRBMS_LAUNCH_CHECK(ctx);
  return LRBMS_OK;
}

int launch_assemble_products(lrbms_ctx* ctx, int Q, const double* theta_bar, const double* lam, const double* lam_df,
                             const double* lbar, const double* lhat, double* P_diag, double* ebar, double* caa, double* Aab,
                             double* Bbb, hipStream_t st) {
  if (!ctx->qdev) return lrbms_fail(ctx, LRBMS_E_STATE, "quadrature not set (lrbms_set_quadrature)");
  QVec tb;
  for (int q = 0; q < 8; ++q) tb.v[q] = q < Q ? theta_bar[q] : 0.0;
  long total = (long)ctx->S * ctx->t.nT;
  hipLaunchKernelGGL(k_assemble_products, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, ctx->t, ctx->qdev, ctx->S,
                     ctx->S_ext, ctx->nbr, Q, tb, lam, lam_df, lbar, lhat, P_diag, ebar, caa, Aab, Bbb);
  LRBMS_LAUNCH_CHECK(ctx);
  // the rank-2 factors of the df_ab blocks (k_f1w), an assembled quantity: kept by the context, tagged with the Aab they belong to
  const long need = (long)Q * ctx->S * ctx->t.nT * 6;
  if (ctx->wab_cap < need) {
    if (ctx->wab) LRBMS_HIP_CHECK(ctx, hipFree(ctx->wab));
    ctx->wab = nullptr;
    ctx->wab_cap = 0;
    LRBMS_HIP_CHECK(ctx, hipMalloc(&ctx->wab, sizeof(double) * (size_t)need));
    ctx->wab_cap = need;
  }
  ctx->wab_src = nullptr;
  if (int rc = launch_wab(ctx, Q, Aab, ctx->wab, st)) return rc;
  ctx->wab_src = Aab;
  ctx->wab_Q = Q;
  return LRBMS_OK;
}

int launch_assemble_flux(lrbms_ctx* ctx, int Q, const double* lam, double* F, hipStream_t st) {
  long total = (long)ctx->S * ctx->t.nrt;
  dim3 grid((unsigned)((total + 255) / 256), Q);
  if (!ctx->qdev) return lrbms_fail(ctx, LRBMS_E_STATE, "quadrature not set (lrbms_set_quadrature)");
  hipLaunchKernelGGL(k_assemble_flux, grid, dim3(256), 0, st, ctx->t, ctx->qdev, ctx->S, ctx->S_ext, ctx->nbr, lam, F);
  LRBMS_LAUNCH_CHECK(ctx);
  return LRBMS_OK;
}
