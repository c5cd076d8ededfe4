// C ABI of liblrbms_hip.so (declared in include/lrbms_hip.h): context, mesh upload, argument checks, dispatch.
#include <cmath>
#include <cstring>
#include <map>
#include <mutex>

#include "lrbms_dev.h"

int64_t estimator_work_size(lrbms_ctx* ctx, int Q, int N);
int64_t reduced_solve_work_size(lrbms_ctx* ctx, int N);
int launch_reduced_estimate_batch(lrbms_ctx* ctx, int Q, int N, int nmu, const double* theta, const double* u, const double* G_nc,
                                  const double* r_fd, const double* G_rdd, const double* G_bb, const double* G_ab,
                                  const double* G_aa, const double* Fside, const double* Fnc, const double* f2, const double* ceps,
                                  double hdiam, double* eta_loc, hipStream_t st);
int64_t reduced_solve_batch_work_size(lrbms_ctx* ctx, int N, int nmu);
int launch_reduced_solve_batch(lrbms_ctx* ctx, int Q, int N, int nmu, const double* theta, const double* B_sys,
                               const double* rhs_red, double* work, double* u, double rtol, int max_iter, double* info,
                               hipStream_t st);
int launch_assemble_dcorr(lrbms_ctx* ctx, int Q, const double* lam, double* D_corr, hipStream_t st);
int64_t local_correction_work_size(lrbms_ctx* ctx, int nmark);
int launch_local_correction(lrbms_ctx* ctx, int Q, const double* theta, int nmark, const int32_t* marked, const double* A_diag,
                            const double* A_cpl, const double* D_corr, const double* b, double* work, double* corr, double rtol,
                            int max_iter, double* info, hipStream_t st);
int64_t fom_solve_work_size(lrbms_ctx* ctx);
int launch_fom_solve(lrbms_ctx* ctx, int Q, const double* theta, const double* A_diag, const double* A_cpl, const double* b,
                     double* work, double* x, double rtol, int max_iter, double* info, hipStream_t st);
int launch_fom_implicit_euler(lrbms_ctx* ctx, int Q, const double* theta, double dt, int nt, const double* A_diag, const double* A_cpl,
                              const double* b, double* work, double* U, double rtol, int max_iter, double* info, hipStream_t st);
int launch_mass_inverse_norm2(lrbms_ctx* ctx, int C, const double* Y, double* out, hipStream_t st);
int launch_div_apply(lrbms_ctx* ctx, int C, int mode, const double* Rt, double* out, hipStream_t st);
int launch_div_pairing(lrbms_ctx* ctx, int Q, int L, const double* theta, const double* D, const double* G, double* out, hipStream_t st);
int launch_reduced_reconstruction_terms(lrbms_ctx* ctx, int Q, int N, int L, const double* theta, const double* B_sys,
                                        const double* M_red, const double* rhs_red, const double* G_ud, const double* U,
                                        double* work, double* out, hipStream_t st);
int launch_reduced_implicit_euler(lrbms_ctx* ctx, int Q, int N, const double* theta, double dt, int nt, const double* B_sys,
                                  const double* M_red, const double* rhs_red, double* work, double* U, double rtol,
                                  int max_iter, double* info, hipStream_t st);
int launch_reduced_time_residual(lrbms_ctx* ctx, int Q, int N, int L, const double* theta, const double* B_sys, const double* M_red,
                                 const double* dU, double* work, double* out, hipStream_t st);
void coarse_release(lrbms_ctx* ctx);   // online.hip
int64_t reduced_precond_size(lrbms_ctx* ctx, int N);
int launch_reduced_precond_build(lrbms_ctx* ctx, int Q, int N, const double* theta, const double* B_sys, double* work, double* pc,
                                 hipStream_t st);
int64_t fused_work_size(lrbms_ctx* ctx, int Q, int N);
int64_t fused_fside_size(lrbms_ctx* ctx, int Q, int N);
int64_t fused_fnc_size(lrbms_ctx* ctx, int N);
bool fused_supported(lrbms_ctx* ctx, int Q, int N, bool factored);
int launch_project_estimate_fused(lrbms_ctx* ctx, int Q, int N, const double* V, const double* F, const double* A_diag,
                                  const double* A_cpl, const double* P_diag, const double* b, const double* ebar,
                                  const double* caa, const double* Aab, const double* Bbb, double* work, double* B_sys,
                                  double* rhs_red, double* E_red, double* M_red, double* G_nc, double* r_fd, double* G_rdd,
                                  double* G_bb, double* G_ab, double* G_aa, double* Fside, double* Fnc, int phase, hipStream_t st);

namespace {

template <typename T>
int upload(lrbms_ctx* ctx, const T* host, size_t count, const T** dev) {
  if (!host) return lrbms_fail(ctx, LRBMS_E_INVALID, "mesh_upload: null template array");
  void* p = nullptr;
  LRBMS_HIP_CHECK(ctx, hipMalloc(&p, sizeof(T) * (count ? count : 1)));
  ctx->owned.push_back(p);
  LRBMS_HIP_CHECK(ctx, hipMemcpy(p, host, sizeof(T) * count, hipMemcpyHostToDevice));
  *dev = static_cast<const T*>(p);
  return LRBMS_OK;
}

void free_owned(lrbms_ctx* ctx) {
  for (void* p : ctx->owned) (void)hipFree(p);
  ctx->owned.clear();
  ctx->nbr = nullptr;
  ctx->qdev = nullptr;
  ctx->has_mesh = false;
}

}  // namespace

// ---- the process-wide side streams (lrbms_dev.h)
namespace {
struct SideStream { hipStream_t s = nullptr; int users = 0; };
std::mutex side_mutex;
std::map<int, SideStream> side_streams;     // key: device * 4 + i
}  // namespace

hipStream_t lrbms_side_stream_acquire(int device, int i) {
  if (i < 0 || i >= 3) return nullptr;
  std::lock_guard<std::mutex> lock(side_mutex);
  SideStream& e = side_streams[device * 4 + i];
  if (!e.s) {
    int cur = 0;
    if (hipGetDevice(&cur) != hipSuccess) return nullptr;
    if (cur != device && hipSetDevice(device) != hipSuccess) return nullptr;
    const bool ok = hipStreamCreateWithFlags(&e.s, hipStreamNonBlocking) == hipSuccess;
    if (cur != device) (void)hipSetDevice(cur);
    if (!ok) {
      e.s = nullptr;
      return nullptr;
    }
  }
  ++e.users;
  return e.s;
}

void lrbms_side_stream_release(int device, int i) {
  std::lock_guard<std::mutex> lock(side_mutex);
  auto it = side_streams.find(device * 4 + i);
  if (it == side_streams.end() || it->second.users <= 0) return;
  if (--it->second.users == 0) {
    (void)hipStreamDestroy(it->second.s);      // waits for nothing: the last context synchronised its work before it went
    side_streams.erase(it);
  }
}

extern "C" {

const char* lrbms_version(void) { return "lrbms_hip 0.1.0 (gfx950)"; }

int lrbms_ctx_create(int device, lrbms_ctx** out) {
  if (!out) return LRBMS_E_INVALID;
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count) return LRBMS_E_HIP;
  if (hipSetDevice(device) != hipSuccess) return LRBMS_E_HIP;
  lrbms_ctx* ctx = new (std::nothrow) lrbms_ctx();
  if (!ctx) return LRBMS_E_INVALID;
  ctx->device = device;
  bool ok = hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming) == hipSuccess &&
            hipEventCreateWithFlags(&ctx->ev_prep, hipEventDisableTiming) == hipSuccess;
  for (int i = 0; i < 3 && ok; ++i)
    ok = (ctx->aux[i] = lrbms_side_stream_acquire(device, i)) != nullptr &&
         hipEventCreateWithFlags(&ctx->ev_join[i], hipEventDisableTiming) == hipSuccess;
  if (!ok) return lrbms_ctx_destroy(ctx), LRBMS_E_HIP;     // gives back what was acquired
  *out = ctx;
  return LRBMS_OK;
}

int lrbms_ctx_destroy(lrbms_ctx* ctx) {
  if (!ctx) return LRBMS_E_INVALID;
  (void)hipSetDevice(ctx->device);
  coarse_release(ctx);
  free_owned(ctx);
  if (ctx->ksp_part) (void)hipFree(ctx->ksp_part);
  if (ctx->ksp_ticket) (void)hipFree(ctx->ksp_ticket);
  if (ctx->subset) (void)hipFree(ctx->subset);
  if (ctx->wab) (void)hipFree(ctx->wab);
  for (int i = 0; i < 3; ++i) {
    if (ctx->aux[i]) lrbms_side_stream_release(ctx->device, i);
    if (ctx->ev_join[i]) (void)hipEventDestroy(ctx->ev_join[i]);
  }
  if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
  if (ctx->ev_prep) (void)hipEventDestroy(ctx->ev_prep);
  for (auto& k : ctx->ktimers) {
    if (k.e0) (void)hipEventDestroy(k.e0);
    if (k.e1) (void)hipEventDestroy(k.e1);
  }
  delete ctx;
  return LRBMS_OK;
}

void* lrbms_ctx_aux_stream(lrbms_ctx* ctx, int32_t i) { return (ctx && i >= 0 && i < 3) ? (void*)ctx->aux[i] : nullptr; }

const char* lrbms_last_error(lrbms_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int lrbms_ctx_set_option(lrbms_ctx* ctx, int32_t option, int32_t value) {
  if (!ctx) return LRBMS_E_INVALID;
  int lo = 0, hi = 1;
  if (option == LRBMS_OPT_STREAMS) lo = -1;
  if (option == LRBMS_OPT_F1_KSPLIT) hi = 4;
  if (option == LRBMS_OPT_COARSE) hi = 2;
  if (option == LRBMS_OPT_PREP_LDS) hi = 3;
  if (option == LRBMS_OPT_F1_FORM) hi = 3;
  if (value < lo || value > hi || (option == LRBMS_OPT_F1_KSPLIT && value == 3))
    return lrbms_fail(ctx, LRBMS_E_INVALID, "set_option: value out of range for this option");
  switch (option) {
    case LRBMS_OPT_OSWALD_ZERO_ON_SUBDOMAIN_BOUNDARY: ctx->t.opt_oswald_subdomain = value; break;
    case LRBMS_OPT_ACCUMULATE_COUPLING_ACROSS_Q: ctx->t.opt_accumulate_coupling = value; break;
    case LRBMS_OPT_OSWALD_VERTEX_PATCH:
      // the option changes lrbms_fused_fnc_ld, i.e. the shape of F_nc buffers a caller may already have sized
      if (ctx->pass_ran && ctx->t.opt_oswald_vertex != value)
        return lrbms_fail(ctx, LRBMS_E_STATE, "set_option: LRBMS_OPT_OSWALD_VERTEX_PATCH cannot change after a fused pass has run on this context");
      ctx->t.opt_oswald_vertex = value;
      break;
    case LRBMS_OPT_STREAMS: ctx->opt_streams = value; break;
    case LRBMS_OPT_F1_KSPLIT: ctx->opt_f1_ksplit = value; break;
    case LRBMS_OPT_F1_FORM: ctx->opt_f1_legacy = value; break;
    case LRBMS_OPT_COARSE: ctx->opt_coarse = value; break;
    case LRBMS_OPT_SOLVE_VALU: ctx->opt_solve_valu = value; break;
    case LRBMS_OPT_ESTIMATE_VALU: ctx->opt_estimate_valu = value; break;
    case LRBMS_OPT_PREP_LDS: ctx->opt_prep_lds = value; break;
    default: return lrbms_fail(ctx, LRBMS_E_INVALID, "set_option: unknown option");
  }
  return LRBMS_OK;
}

int lrbms_set_diagonal_neighbours(lrbms_ctx* ctx, const int32_t* nbr_diag) {
  LRBMS_REQUIRE_MESH(ctx);
  if (!nbr_diag) return lrbms_fail(ctx, LRBMS_E_INVALID, "set_diagonal_neighbours: null table");
  for (long i = 0; i < (long)ctx->S * 4; ++i)
    if (nbr_diag[i] < -1 || nbr_diag[i] >= ctx->S_ext)
      return lrbms_fail(ctx, LRBMS_E_INVALID, "set_diagonal_neighbours: index out of range [-1, S_ext)");
  LRBMS_HIP_CHECK(ctx, hipDeviceSynchronize());      // no pass in flight reads the table while it changes
  LRBMS_HIP_CHECK(ctx, hipMemcpy(const_cast<int*>(ctx->t.nbr_diag), nbr_diag, sizeof(int) * (size_t)ctx->S * 4, hipMemcpyHostToDevice));
  ctx->diag_explicit = true;
  return LRBMS_OK;
}

int lrbms_fused_set_subset(lrbms_ctx* ctx, const int32_t* subset, int32_t count) {
  LRBMS_REQUIRE_MESH(ctx);
  if (count < 0 || count > ctx->S || (count > 0 && subset == nullptr))
    return lrbms_fail(ctx, LRBMS_E_INVALID, "fused_set_subset: count must be in [0, S] and the list non-null");
  for (int i = 0; i < count; ++i)
    if (subset[i] < 0 || subset[i] >= ctx->S || (i > 0 && subset[i] <= subset[i - 1]))
      return lrbms_fail(ctx, LRBMS_E_INVALID, "fused_set_subset: the list must be strictly ascending local subdomain indices in [0, S)");
  if (count > ctx->subset_cap) {
    if (ctx->subset) LRBMS_HIP_CHECK(ctx, hipFree(ctx->subset));
    ctx->subset = nullptr;
    ctx->subset_cap = 0;
    LRBMS_HIP_CHECK(ctx, hipMalloc(&ctx->subset, sizeof(int) * (size_t)ctx->S));
    ctx->subset_cap = ctx->S;
  }
  // a synchronous copy: the list a pass still in flight reads must not change under it, and the host array may go away after return
  if (count > 0) {
    LRBMS_HIP_CHECK(ctx, hipDeviceSynchronize());
    LRBMS_HIP_CHECK(ctx, hipMemcpy(ctx->subset, subset, sizeof(int) * (size_t)count, hipMemcpyHostToDevice));
  }
  ctx->subset_n = count;
  return LRBMS_OK;
}

int lrbms_kernel_timing(lrbms_ctx* ctx, int32_t enable) {
  if (!ctx) return LRBMS_E_INVALID;
  ctx->ktime = enable != 0;
  ctx->ktime_n = 0;
  for (auto& k : ctx->ktimers) k.used = false;
  return LRBMS_OK;
}

int lrbms_kernel_timing_read(lrbms_ctx* ctx, char* names, int64_t names_cap, double* ms, int32_t cap, int32_t* count) {
  if (!ctx || !names || !ms || !count || names_cap <= 0) return LRBMS_E_INVALID;
  LRBMS_HIP_CHECK(ctx, hipDeviceSynchronize());
  int n = 0;
  std::string joined;
  for (int i = 0; i < ctx->ktime_n && n < cap; ++i) {
    const auto& k = ctx->ktimers[i];
    if (!k.used) continue;
    float t = 0.f;
    LRBMS_HIP_CHECK(ctx, hipEventElapsedTime(&t, k.e0, k.e1));
    ms[n++] = (double)t;
    if (!joined.empty()) joined += '\n';
    joined += k.name;
  }
  if ((int64_t)joined.size() + 1 > names_cap) return lrbms_fail(ctx, LRBMS_E_INVALID, "kernel_timing_read: names buffer too small");
  memcpy(names, joined.c_str(), joined.size() + 1);
  *count = n;
  ctx->ktime_n = 0;          // the next pass records afresh
  for (auto& k : ctx->ktimers) k.used = false;
  return LRBMS_OK;
}

int lrbms_mesh_upload(lrbms_ctx* ctx, const lrbms_mesh_desc* d, int32_t S, int32_t S_ext, const int32_t* nbr) {
  if (!ctx || !d || !nbr) return LRBMS_E_INVALID;
  if (S <= 0 || S_ext < S) return lrbms_fail(ctx, LRBMS_E_INVALID, "mesh_upload: need 0 < S <= S_ext");
  if (d->kx <= 0 || d->ky <= 0 || d->n_T != 8 * d->kx * d->ky || d->n_rt <= 0 ||
      d->n_vertices != (2 * d->kx + 1) * (2 * d->ky + 1) || d->ncf != 2 * (d->kx > d->ky ? d->kx : d->ky))
    return lrbms_fail(ctx, LRBMS_E_INVALID, "mesh_upload: inconsistent template sizes");
  for (int i = 0; i < S; ++i) {
    for (int k = 0; k < 5; ++k) {
      const int v = nbr[i * 5 + k];
      if (v < -1 || v >= S_ext) return lrbms_fail(ctx, LRBMS_E_INVALID, "mesh_upload: neighbour index out of range");
    }
    if (nbr[i * 5 + 2] != i) return lrbms_fail(ctx, LRBMS_E_INVALID, "mesh_upload: nbr[s][2] must be s");
  }
  // validate the template indices on the host before any kernel may dereference them
  const int nT = d->n_T, n = 3 * nT;
  for (int i = 0; i < 3 * nT; ++i) {
    if (d->nb_elem[i] >= nT || d->nb_elem[i] < -4) return lrbms_fail(ctx, LRBMS_E_INVALID, "mesh_upload: nb_elem out of range");
    if (d->nb_elem[i] < 0 && (d->nb_elem_out[i] < 0 || d->nb_elem_out[i] >= nT || d->elem_side_pos[i] < 0 ||
                              d->elem_side_pos[i] >= d->ncf || d->nb_face_out[i] < 0 || d->nb_face_out[i] > 2))
      return lrbms_fail(ctx, LRBMS_E_INVALID, "mesh_upload: side face pairing out of range");
    if (d->nb_face[i] < 0 || d->nb_face[i] > 2) return lrbms_fail(ctx, LRBMS_E_INVALID, "mesh_upload: nb_face out of range");
    if (d->elem_rt[i] < 0 || d->elem_rt[i] >= d->n_rt) return lrbms_fail(ctx, LRBMS_E_INVALID, "mesh_upload: elem_rt out of range");
    if (d->dof_vertex[i] < 0 || d->dof_vertex[i] >= d->n_vertices || d->vdof_idx[i] < 0 || d->vdof_idx[i] >= n)
      return lrbms_fail(ctx, LRBMS_E_INVALID, "mesh_upload: vertex star out of range");
  }
  if (d->vdof_ptr[0] != 0 || d->vdof_ptr[d->n_vertices] != n) return lrbms_fail(ctx, LRBMS_E_INVALID, "mesh_upload: bad vdof_ptr");
  for (int v = 0; v < d->n_vertices; ++v)
    if (d->vdof_ptr[v + 1] < d->vdof_ptr[v]) return lrbms_fail(ctx, LRBMS_E_INVALID, "mesh_upload: vdof_ptr not monotone");
  for (int r = 0; r < d->n_rt; ++r) {
    if (d->rt_e0[r] < 0 || d->rt_e0[r] >= nT || d->rt_f0[r] < 0 || d->rt_f0[r] > 2 || d->rt_side[r] < -1 || d->rt_side[r] > 3 ||
        d->rt_e1[r] < 0 || d->rt_e1[r] >= nT || d->rt_f1[r] < 0 || d->rt_f1[r] > 2)
      return lrbms_fail(ctx, LRBMS_E_INVALID, "mesh_upload: rt face table out of range");
  }
  for (int sd = 0; sd < 4; ++sd) {
    if (d->side_count[sd] < 0 || d->side_count[sd] > d->ncf) return lrbms_fail(ctx, LRBMS_E_INVALID, "mesh_upload: side_count");
    for (int p = 0; p < d->side_count[sd]; ++p)
      if (d->side_elem[sd * d->ncf + p] < 0 || d->side_elem[sd * d->ncf + p] >= nT || d->side_elem_out[sd * d->ncf + p] < 0 ||
          d->side_elem_out[sd * d->ncf + p] >= nT)
        return lrbms_fail(ctx, LRBMS_E_INVALID, "mesh_upload: side element out of range");
  }

  LRBMS_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  free_owned(ctx);
  Tmpl& t = ctx->t;
  t.kx = d->kx; t.ky = d->ky; t.nT = nT; t.n = n; t.nrt = d->n_rt; t.nv = d->n_vertices;
  t.nvx = 2 * d->kx + 1; t.nvy = 2 * d->ky + 1; t.ncf = d->ncf; t.hx = d->hx; t.hy = d->hy;
  std::memcpy(t.kappa, d->kappa, sizeof(double) * 4);
  const double det = t.kappa[0] * t.kappa[3] - t.kappa[1] * t.kappa[2];
  if (!(det > 0.0)) return lrbms_fail(ctx, LRBMS_E_INVALID, "mesh_upload: kappa must be positive definite");
  t.kinv[0] = t.kappa[3] / det; t.kinv[1] = -t.kappa[1] / det; t.kinv[2] = -t.kappa[2] / det; t.kinv[3] = t.kappa[0] / det;
  const double a = t.kappa[0], b = 0.5 * (t.kappa[1] + t.kappa[2]), c = t.kappa[3];
  t.kmin = 0.5 * (a + c) - std::sqrt(0.25 * (a - c) * (a - c) + b * b);
  int rc;
#define UP(field, count) if ((rc = upload(ctx, d->field, (size_t)(count), &t.field))) return rc
  UP(nb_elem, 3 * nT); UP(nb_face, 3 * nT); UP(nb_elem_out, 3 * nT); UP(nb_face_out, 3 * nT); UP(elem_side_pos, 3 * nT);
  UP(elem_rt, 3 * nT); UP(face_sign, 3 * nT); UP(dof_vertex, n); UP(vdof_ptr, d->n_vertices + 1); UP(vdof_idx, n);
  UP(rt_e0, d->n_rt); UP(rt_f0, d->n_rt); UP(rt_e1, d->n_rt); UP(rt_f1, d->n_rt); UP(rt_side, d->n_rt);
  UP(side_elem, 4 * d->ncf); UP(side_elem_out, 4 * d->ncf); UP(side_count, 4);
  if (d->ntouch <= 0 || !d->touch_elem || !d->touch_count) return lrbms_fail(ctx, LRBMS_E_INVALID, "mesh_upload: touch table missing");
  for (int sd = 0; sd < 4; ++sd) {
    if (d->touch_count[sd] < 0 || d->touch_count[sd] > d->ntouch) return lrbms_fail(ctx, LRBMS_E_INVALID, "mesh_upload: touch_count");
    for (int p = 0; p < d->touch_count[sd]; ++p)
      if (d->touch_elem[sd * d->ntouch + p] < 0 || d->touch_elem[sd * d->ntouch + p] >= nT)
        return lrbms_fail(ctx, LRBMS_E_INVALID, "mesh_upload: touch_elem out of range");
  }
  t.ntouch = d->ntouch;
  UP(touch_elem, 4 * d->ntouch); UP(touch_count, 4);
  UP(grad, 6 * nT); UP(area, nT); UP(normal, 6 * nT); UP(face_len, 3 * nT); UP(points, 6 * nT);
#undef UP
  const int* nbr_dev = nullptr;
  if ((rc = upload(ctx, nbr, (size_t)S * 5, &nbr_dev))) return rc;
  ctx->nbr = const_cast<int*>(nbr_dev);
  ctx->nbr_host.assign(nbr, nbr + (size_t)S * 5);
  {
    // diagonal neighbours as far as nbr determines them: the E / W neighbour of the S / N neighbour, if that one is local
    std::vector<int32_t> dg((size_t)S * 4, -1);
    for (int i = 0; i < S; ++i)
      for (int c = 0; c < 4; ++c) {
        const int sa = nbr[i * 5 + (c < 2 ? 0 : 4)];
        if (sa >= 0 && sa < S) dg[(size_t)i * 4 + c] = nbr[sa * 5 + ((c & 1) ? 3 : 1)];
      }
    const int* dg_dev = nullptr;
    if ((rc = upload(ctx, dg.data(), dg.size(), &dg_dev))) return rc;
    t.nbr_diag = dg_dev;
    ctx->diag_explicit = false;
  }
  ctx->subset_n = 0;
  ctx->wab_src = nullptr;      // (factors of another mesh)
  ctx->pass_ran = false;
  ctx->S = S;
  ctx->S_ext = S_ext;
  if ((rc = build_template_tables(ctx))) return rc;
  ctx->has_mesh = true;
  return LRBMS_OK;
}

#define CHECK_Q_N(ctx, Q, N)                                                                        \
  do {                                                                                              \
    if ((Q) < 1 || (Q) > 8) return lrbms_fail(ctx, LRBMS_E_INVALID, "Q must be in 1..8");            \
    if ((N) < 1) return lrbms_fail(ctx, LRBMS_E_INVALID, "N must be >= 1");                          \
  } while (0)
#define CHECK_PTR(ctx, p) \
  do { if (!(p)) return lrbms_fail(ctx, LRBMS_E_INVALID, "null pointer: " #p); } while (0)

int lrbms_set_quadrature(lrbms_ctx* ctx, const lrbms_quadrature* quad) {
  if (!ctx || !quad) return LRBMS_E_INVALID;
  const lrbms_tri_rule* tri[9] = {&quad->system_volume, &quad->energy_volume, &quad->elliptic_bar, &quad->rhs, &quad->f2,
                                  &quad->ceps, &quad->df_aa, &quad->df_ab, &quad->df_bb};
  const lrbms_edge_rule* edge[4] = {&quad->system_inner_face, &quad->system_coupling_face, &quad->energy_face, &quad->flux_face};
  for (auto r : tri)
    if (r->n < 1 || r->n > LRBMS_MAXQV) return lrbms_fail(ctx, LRBMS_E_INVALID, "set_quadrature: triangle rule size out of range");
  for (auto r : edge) {
    if (r->n < 1 || r->n > LRBMS_MAXQF) return lrbms_fail(ctx, LRBMS_E_INVALID, "set_quadrature: edge rule size out of range");
    for (int k = 0; k < r->n; ++k)   // the neighbour across a face reads the samples in reverse order
      if (fabs(r->t[k] + r->t[r->n - 1 - k] - 1.0) > 1e-14 || fabs(r->w[k] - r->w[r->n - 1 - k]) > 1e-14)
        return lrbms_fail(ctx, LRBMS_E_INVALID, "set_quadrature: edge rules must be symmetric about 1/2");
  }
  const int nfs = quad->system_inner_face.n > quad->system_coupling_face.n ? quad->system_inner_face.n : quad->system_coupling_face.n;
  if (quad->nfs != nfs || quad->o_sysf < quad->o_sysv + quad->system_volume.n || quad->o_enf < quad->o_sysf + 3 * nfs ||
      quad->o_flf < quad->o_enf + 3 * quad->energy_face.n || quad->o_env < quad->o_flf + 3 * quad->flux_face.n ||
      quad->lam_stride < quad->o_env + quad->energy_volume.n || quad->lamdf_stride < quad->o_ab + quad->df_ab.n ||
      quad->o_ab < quad->o_aa + quad->df_aa.n || quad->o_hab < quad->o_haa + quad->df_aa.n ||
      quad->o_hbb < quad->o_hab + quad->df_ab.n || quad->o_hceps < quad->o_hbb + quad->df_bb.n ||
      quad->lhat_stride < quad->o_hceps + quad->ceps.n || quad->o_ff2 < quad->o_frhs + quad->rhs.n ||
      quad->f_stride < quad->o_ff2 + quad->f2.n || quad->lbar_stride < quad->elliptic_bar.n)
    return lrbms_fail(ctx, LRBMS_E_INVALID, "set_quadrature: inconsistent sample record layout");
  LRBMS_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (!ctx->qdev) {
    LRBMS_HIP_CHECK(ctx, hipMalloc(&ctx->qdev, sizeof(lrbms_quadrature)));
    ctx->owned.push_back(ctx->qdev);
  }
  LRBMS_HIP_CHECK(ctx, hipMemcpy(ctx->qdev, quad, sizeof(lrbms_quadrature), hipMemcpyHostToDevice));
  return LRBMS_OK;
}

int lrbms_assemble_swipdg(lrbms_ctx* ctx, int32_t Q, const double* lam, double* A_diag, double* A_cpl, void* stream) {
  LRBMS_REQUIRE_MESH(ctx); CHECK_Q_N(ctx, Q, 1); CHECK_PTR(ctx, lam); CHECK_PTR(ctx, A_diag); CHECK_PTR(ctx, A_cpl);
  return launch_assemble_swipdg(ctx, Q, lam, A_diag, A_cpl, (hipStream_t)stream);
}

int lrbms_assemble_rhs(lrbms_ctx* ctx, const double* f_smp, const double* lhat, double* b, double* f2, double* ceps,
                       void* stream) {
  LRBMS_REQUIRE_MESH(ctx); CHECK_PTR(ctx, f_smp); CHECK_PTR(ctx, lhat); CHECK_PTR(ctx, b); CHECK_PTR(ctx, f2); CHECK_PTR(ctx, ceps);
  return launch_assemble_rhs(ctx, f_smp, lhat, b, f2, ceps, (hipStream_t)stream);
}

int lrbms_assemble_products(lrbms_ctx* ctx, int32_t Q, const double* theta_bar, const double* lam, const double* lam_df,
                            const double* lbar, const double* lhat, double* P_diag, double* ebar, double* caa, double* Aab,
                            double* Bbb, void* stream) {
  LRBMS_REQUIRE_MESH(ctx); CHECK_Q_N(ctx, Q, 1); CHECK_PTR(ctx, theta_bar); CHECK_PTR(ctx, lam); CHECK_PTR(ctx, lam_df); CHECK_PTR(ctx, lbar);
  CHECK_PTR(ctx, lhat); CHECK_PTR(ctx, P_diag); CHECK_PTR(ctx, ebar); CHECK_PTR(ctx, caa); CHECK_PTR(ctx, Aab); CHECK_PTR(ctx, Bbb);
  return launch_assemble_products(ctx, Q, theta_bar, lam, lam_df, lbar, lhat, P_diag, ebar, caa, Aab, Bbb, (hipStream_t)stream);
}

int lrbms_assemble_flux(lrbms_ctx* ctx, int32_t Q, const double* lam, double* F, void* stream) {
  LRBMS_REQUIRE_MESH(ctx); CHECK_Q_N(ctx, Q, 1); CHECK_PTR(ctx, lam); CHECK_PTR(ctx, F);
  return launch_assemble_flux(ctx, Q, lam, F, (hipStream_t)stream);
}

int lrbms_oswald_apply(lrbms_ctx* ctx, int32_t N, const double* V, double* Wt, void* stream) {
  LRBMS_REQUIRE_MESH(ctx); CHECK_Q_N(ctx, 1, N); CHECK_PTR(ctx, V); CHECK_PTR(ctx, Wt);
  return launch_oswald(ctx, N, V, Wt, (hipStream_t)stream);
}

int lrbms_flux_reconstruct(lrbms_ctx* ctx, int32_t Q, int32_t N, const double* F, const double* V, double* Rt, void* stream) {
  LRBMS_REQUIRE_MESH(ctx); CHECK_Q_N(ctx, Q, N); CHECK_PTR(ctx, F); CHECK_PTR(ctx, V); CHECK_PTR(ctx, Rt);
  return launch_flux(ctx, Q, N, F, V, Rt, (hipStream_t)stream);
}

int lrbms_project_system(lrbms_ctx* ctx, int32_t Q, int32_t N, const double* V, const double* A_diag, const double* A_cpl,
                         const double* P_diag, const double* b, double* work, double* B_sys, double* rhs_red, double* E_red,
                         double* M_red, void* stream) {
  LRBMS_REQUIRE_MESH(ctx); CHECK_Q_N(ctx, Q, N); CHECK_PTR(ctx, V); CHECK_PTR(ctx, A_diag); CHECK_PTR(ctx, A_cpl);
  CHECK_PTR(ctx, P_diag); CHECK_PTR(ctx, b); CHECK_PTR(ctx, work); CHECK_PTR(ctx, B_sys); CHECK_PTR(ctx, rhs_red);
  CHECK_PTR(ctx, E_red); CHECK_PTR(ctx, M_red);
  if ((size_t)2 * 3 * ctx->t.ncf * N * sizeof(double) > 64 * 1024)
    return lrbms_fail(ctx, LRBMS_E_INVALID, "project_system: coupling tile exceeds 64 KB of LDS");
  return launch_project_system(ctx, Q, N, V, A_diag, A_cpl, P_diag, b, work, B_sys, rhs_red, E_red, M_red, (hipStream_t)stream);
}

int64_t lrbms_estimator_work_size(lrbms_ctx* ctx, int32_t Q, int32_t N) {
  if (!ctx || !ctx->has_mesh || Q < 1 || N < 1) return -1;
  return estimator_work_size(ctx, Q, N);
}

int lrbms_estimator_grams(lrbms_ctx* ctx, int32_t Q, int32_t N, const double* V, const double* Wt, const double* Rt,
                          const double* ebar, const double* caa, const double* Aab, const double* Bbb, const double* b,
                          double* work, double* G_nc, double* r_fd, double* G_rdd, double* G_bb, double* G_ab, double* G_aa,
                          void* stream) {
  LRBMS_REQUIRE_MESH(ctx); CHECK_Q_N(ctx, Q, N); CHECK_PTR(ctx, V); CHECK_PTR(ctx, Wt); CHECK_PTR(ctx, Rt); CHECK_PTR(ctx, ebar);
  CHECK_PTR(ctx, caa); CHECK_PTR(ctx, Aab); CHECK_PTR(ctx, Bbb); CHECK_PTR(ctx, b); CHECK_PTR(ctx, work); CHECK_PTR(ctx, G_nc);
  CHECK_PTR(ctx, r_fd); CHECK_PTR(ctx, G_rdd); CHECK_PTR(ctx, G_bb); CHECK_PTR(ctx, G_ab); CHECK_PTR(ctx, G_aa);
  return launch_estimator_grams(ctx, Q, N, V, Wt, Rt, ebar, caa, Aab, Bbb, b, work, G_nc, r_fd, G_rdd, G_bb, G_ab, G_aa,
                                (hipStream_t)stream);
}

int32_t lrbms_fused_fnc_ld(lrbms_ctx* ctx, int32_t N) {
  if (!ctx || !ctx->has_mesh || N < 1) return -1;
  const int nvs = ctx->t.nvx > ctx->t.nvy ? ctx->t.nvx : ctx->t.nvy;
  return 2 * N + 4 * nvs + (ctx->t.opt_oswald_vertex ? N : 0);
}

int64_t lrbms_fused_mfma_per_subdomain(lrbms_ctx* ctx, int32_t Q, int32_t N) {
  if (!ctx || !ctx->has_mesh) return -1;
  return f1_mfma_per_subdomain(ctx, Q, N);
}

int lrbms_fused_supported(lrbms_ctx* ctx, int32_t Q, int32_t N) {
  if (!ctx || !ctx->has_mesh || Q < 1 || Q > 8 || N < 1) return 0;
  return fused_supported(ctx, Q, N, false) ? 1 : 0;
}

int lrbms_fused_factored_supported(lrbms_ctx* ctx, int32_t Q, int32_t N) {
  if (!ctx || !ctx->has_mesh || Q < 1 || N < 1) return 0;
  return fused_supported(ctx, Q, N, true) ? 1 : 0;
}

int64_t lrbms_fused_work_size(lrbms_ctx* ctx, int32_t Q, int32_t N) {
  if (!ctx || !ctx->has_mesh || Q < 1 || N < 1) return -1;
  return fused_work_size(ctx, Q, N);
}

int lrbms_project_estimate_fused(lrbms_ctx* ctx, int32_t Q, int32_t N, const double* V, const double* F, const double* A_diag,
                                 const double* A_cpl, const double* P_diag, const double* b, const double* ebar,
                                 const double* caa, const double* Aab, const double* Bbb, double* work, double* B_sys,
                                 double* rhs_red, double* E_red, double* M_red, double* G_nc, double* r_fd, double* G_rdd,
                                 double* G_bb, double* G_ab, double* G_aa, void* stream) {
  LRBMS_REQUIRE_MESH(ctx); CHECK_Q_N(ctx, Q, N); CHECK_PTR(ctx, V); CHECK_PTR(ctx, F); CHECK_PTR(ctx, A_diag); CHECK_PTR(ctx, A_cpl);
  CHECK_PTR(ctx, P_diag); CHECK_PTR(ctx, b); CHECK_PTR(ctx, ebar); CHECK_PTR(ctx, caa); CHECK_PTR(ctx, Aab); CHECK_PTR(ctx, Bbb);
  CHECK_PTR(ctx, work); CHECK_PTR(ctx, B_sys); CHECK_PTR(ctx, rhs_red); CHECK_PTR(ctx, E_red); CHECK_PTR(ctx, M_red);
  CHECK_PTR(ctx, G_nc); CHECK_PTR(ctx, r_fd); CHECK_PTR(ctx, G_rdd); CHECK_PTR(ctx, G_bb); CHECK_PTR(ctx, G_ab); CHECK_PTR(ctx, G_aa);
  return launch_project_estimate_fused(ctx, Q, N, V, F, A_diag, A_cpl, P_diag, b, ebar, caa, Aab, Bbb, work, B_sys, rhs_red, E_red,
                                       M_red, G_nc, r_fd, G_rdd, G_bb, G_ab, G_aa, nullptr, nullptr, 0, (hipStream_t)stream);
}

int lrbms_project_estimate_fused_phase(lrbms_ctx* ctx, int32_t phase, int32_t Q, int32_t N, const double* V, const double* F,
                                       const double* A_diag, const double* A_cpl, const double* P_diag, const double* b,
                                       const double* ebar, const double* caa, const double* Aab, const double* Bbb, double* work,
                                       double* B_sys, double* rhs_red, double* E_red, double* M_red, double* G_nc, double* r_fd,
                                       double* G_rdd, double* G_bb, double* G_ab, double* G_aa, void* stream) {
  LRBMS_REQUIRE_MESH(ctx); CHECK_Q_N(ctx, Q, N); CHECK_PTR(ctx, V); CHECK_PTR(ctx, F); CHECK_PTR(ctx, A_diag); CHECK_PTR(ctx, A_cpl);
  CHECK_PTR(ctx, P_diag); CHECK_PTR(ctx, b); CHECK_PTR(ctx, ebar); CHECK_PTR(ctx, caa); CHECK_PTR(ctx, Aab); CHECK_PTR(ctx, Bbb);
  CHECK_PTR(ctx, work); CHECK_PTR(ctx, B_sys); CHECK_PTR(ctx, rhs_red); CHECK_PTR(ctx, E_red); CHECK_PTR(ctx, M_red);
  CHECK_PTR(ctx, G_nc); CHECK_PTR(ctx, r_fd); CHECK_PTR(ctx, G_rdd); CHECK_PTR(ctx, G_bb); CHECK_PTR(ctx, G_ab); CHECK_PTR(ctx, G_aa);
  return launch_project_estimate_fused(ctx, Q, N, V, F, A_diag, A_cpl, P_diag, b, ebar, caa, Aab, Bbb, work, B_sys, rhs_red, E_red,
                                       M_red, G_nc, r_fd, G_rdd, G_bb, G_ab, G_aa, nullptr, nullptr, phase, (hipStream_t)stream);
}

int64_t lrbms_fside_size(lrbms_ctx* ctx, int32_t Q, int32_t N) {
  if (!ctx || !ctx->has_mesh || Q < 1 || N < 1) return -1;
  return fused_fside_size(ctx, Q, N);
}

int64_t lrbms_fnc_size(lrbms_ctx* ctx, int32_t N) {
  if (!ctx || !ctx->has_mesh || N < 1) return -1;
  return fused_fnc_size(ctx, N);
}

int lrbms_project_estimate_fused_factored(lrbms_ctx* ctx, int32_t phase, int32_t Q, int32_t N, const double* V, const double* F,
                                          const double* A_diag, const double* A_cpl, const double* P_diag, const double* b,
                                          const double* ebar, const double* caa, const double* Aab, const double* Bbb, double* work,
                                          double* B_sys, double* rhs_red, double* E_red, double* M_red, double* G_nc_self,
                                          double* r_fd, double* G_rdd_self, double* G_bb_self, double* G_ab_self, double* G_aa,
                                          double* F_side, double* F_nc, void* stream) {
  LRBMS_REQUIRE_MESH(ctx); CHECK_Q_N(ctx, Q, N); CHECK_PTR(ctx, V); CHECK_PTR(ctx, F); CHECK_PTR(ctx, A_diag); CHECK_PTR(ctx, A_cpl);
  CHECK_PTR(ctx, P_diag); CHECK_PTR(ctx, b); CHECK_PTR(ctx, ebar); CHECK_PTR(ctx, caa); CHECK_PTR(ctx, Aab); CHECK_PTR(ctx, Bbb);
  CHECK_PTR(ctx, work); CHECK_PTR(ctx, B_sys); CHECK_PTR(ctx, rhs_red); CHECK_PTR(ctx, E_red); CHECK_PTR(ctx, M_red);
  CHECK_PTR(ctx, G_nc_self); CHECK_PTR(ctx, r_fd); CHECK_PTR(ctx, G_rdd_self); CHECK_PTR(ctx, G_bb_self); CHECK_PTR(ctx, G_ab_self);
  CHECK_PTR(ctx, G_aa); CHECK_PTR(ctx, F_side); CHECK_PTR(ctx, F_nc);
  return launch_project_estimate_fused(ctx, Q, N, V, F, A_diag, A_cpl, P_diag, b, ebar, caa, Aab, Bbb, work, B_sys, rhs_red, E_red,
                                       M_red, G_nc_self, r_fd, G_rdd_self, G_bb_self, G_ab_self, G_aa, F_side, F_nc, phase,
                                       (hipStream_t)stream);
}

int lrbms_reduced_estimate(lrbms_ctx* ctx, int32_t Q, int32_t N, const double* theta, const double* u, const double* G_nc,
                           const double* r_fd, const double* G_rdd, const double* G_bb, const double* G_ab, const double* G_aa,
                           const double* f2, const double* ceps, double hdiam, double* eta_loc, void* stream) {
  LRBMS_REQUIRE_MESH(ctx); CHECK_Q_N(ctx, Q, N); CHECK_PTR(ctx, theta); CHECK_PTR(ctx, u); CHECK_PTR(ctx, G_nc); CHECK_PTR(ctx, r_fd);
  CHECK_PTR(ctx, G_rdd); CHECK_PTR(ctx, G_bb); CHECK_PTR(ctx, G_ab); CHECK_PTR(ctx, G_aa); CHECK_PTR(ctx, f2); CHECK_PTR(ctx, ceps);
  CHECK_PTR(ctx, eta_loc);
  if ((size_t)(5 * N + 5 * Q * N + 256) * sizeof(double) > 64 * 1024)
    return lrbms_fail(ctx, LRBMS_E_INVALID, "reduced_estimate: coefficient tile exceeds 64 KB of LDS");
  return launch_reduced_estimate(ctx, Q, N, theta, u, G_nc, r_fd, G_rdd, G_bb, G_ab, G_aa, nullptr, nullptr, f2, ceps, hdiam, eta_loc,
                                 (hipStream_t)stream);
}

int lrbms_reduced_estimate_factored(lrbms_ctx* ctx, int32_t Q, int32_t N, const double* theta, const double* u,
                                    const double* G_nc_self, const double* r_fd, const double* G_rdd_self, const double* G_bb_self,
                                    const double* G_ab_self, const double* G_aa, const double* F_side, const double* F_nc,
                                    const double* f2, const double* ceps, double hdiam, double* eta_loc, void* stream) {
  LRBMS_REQUIRE_MESH(ctx); CHECK_Q_N(ctx, Q, N); CHECK_PTR(ctx, theta); CHECK_PTR(ctx, u); CHECK_PTR(ctx, G_nc_self); CHECK_PTR(ctx, r_fd);
  CHECK_PTR(ctx, G_rdd_self); CHECK_PTR(ctx, G_bb_self); CHECK_PTR(ctx, G_ab_self); CHECK_PTR(ctx, G_aa); CHECK_PTR(ctx, F_side);
  CHECK_PTR(ctx, F_nc); CHECK_PTR(ctx, f2); CHECK_PTR(ctx, ceps); CHECK_PTR(ctx, eta_loc);
  const int nvs = ctx->t.nvx > ctx->t.nvy ? ctx->t.nvx : ctx->t.nvy;
  if ((size_t)(5 * N + 5 * Q * N + 256 + 4 * ctx->t.ncf * (3 + Q) + 8 * nvs) * sizeof(double) > 64 * 1024)
    return lrbms_fail(ctx, LRBMS_E_INVALID, "reduced_estimate: coefficient tile exceeds 64 KB of LDS");
  return launch_reduced_estimate(ctx, Q, N, theta, u, G_nc_self, r_fd, G_rdd_self, G_bb_self, G_ab_self, G_aa, F_side, F_nc, f2, ceps,
                                 hdiam, eta_loc, (hipStream_t)stream);
}

int lrbms_reduced_estimate_batch(lrbms_ctx* ctx, int32_t Q, int32_t N, int32_t nmu, const double* theta, const double* u,
                                 const double* G_nc, const double* r_fd, const double* G_rdd, const double* G_bb,
                                 const double* G_ab, const double* G_aa, const double* f2, const double* ceps, double hdiam,
                                 double* eta_loc, void* stream) {
  LRBMS_REQUIRE_MESH(ctx); CHECK_Q_N(ctx, Q, N); CHECK_PTR(ctx, theta); CHECK_PTR(ctx, u); CHECK_PTR(ctx, G_nc); CHECK_PTR(ctx, r_fd);
  CHECK_PTR(ctx, G_rdd); CHECK_PTR(ctx, G_bb); CHECK_PTR(ctx, G_ab); CHECK_PTR(ctx, G_aa); CHECK_PTR(ctx, f2); CHECK_PTR(ctx, ceps);
  CHECK_PTR(ctx, eta_loc);
  return launch_reduced_estimate_batch(ctx, Q, N, nmu, theta, u, G_nc, r_fd, G_rdd, G_bb, G_ab, G_aa, nullptr, nullptr, f2, ceps, hdiam,
                                       eta_loc, (hipStream_t)stream);
}

int lrbms_reduced_estimate_batch_factored(lrbms_ctx* ctx, int32_t Q, int32_t N, int32_t nmu, const double* theta, const double* u,
                                          const double* G_nc_self, const double* r_fd, const double* G_rdd_self,
                                          const double* G_bb_self, const double* G_ab_self, const double* G_aa, const double* F_side,
                                          const double* F_nc, const double* f2, const double* ceps, double hdiam, double* eta_loc,
                                          void* stream) {
  LRBMS_REQUIRE_MESH(ctx); CHECK_Q_N(ctx, Q, N); CHECK_PTR(ctx, theta); CHECK_PTR(ctx, u); CHECK_PTR(ctx, G_nc_self); CHECK_PTR(ctx, r_fd);
  CHECK_PTR(ctx, G_rdd_self); CHECK_PTR(ctx, G_bb_self); CHECK_PTR(ctx, G_ab_self); CHECK_PTR(ctx, G_aa); CHECK_PTR(ctx, F_side);
  CHECK_PTR(ctx, F_nc); CHECK_PTR(ctx, f2); CHECK_PTR(ctx, ceps); CHECK_PTR(ctx, eta_loc);
  return launch_reduced_estimate_batch(ctx, Q, N, nmu, theta, u, G_nc_self, r_fd, G_rdd_self, G_bb_self, G_ab_self, G_aa, F_side, F_nc,
                                       f2, ceps, hdiam, eta_loc, (hipStream_t)stream);
}

int64_t lrbms_reduced_solve_work_size(lrbms_ctx* ctx, int32_t N) {
  if (!ctx || !ctx->has_mesh || N < 1) return -1;
  return reduced_solve_work_size(ctx, N);
}

int lrbms_reduced_solve(lrbms_ctx* ctx, int32_t Q, int32_t N, const double* theta, const double* B_sys, const double* rhs_red,
                        double* work, double* u, double rtol, int32_t max_iter, double* info, void* stream) {
  LRBMS_REQUIRE_MESH(ctx); CHECK_Q_N(ctx, Q, N); CHECK_PTR(ctx, theta); CHECK_PTR(ctx, B_sys); CHECK_PTR(ctx, rhs_red);
  CHECK_PTR(ctx, work); CHECK_PTR(ctx, u);
  return launch_reduced_solve(ctx, Q, N, theta, B_sys, rhs_red, work, u, rtol, max_iter, info, (hipStream_t)stream);
}

int64_t lrbms_reduced_solve_batch_work_size(lrbms_ctx* ctx, int32_t N, int32_t nmu) {
  if (!ctx || !ctx->has_mesh || N < 1 || nmu < 1) return -1;
  return reduced_solve_batch_work_size(ctx, N, nmu);
}

int lrbms_reduced_solve_batch(lrbms_ctx* ctx, int32_t Q, int32_t N, int32_t nmu, const double* theta, const double* B_sys,
                              const double* rhs_red, double* work, double* u, double rtol, int32_t max_iter, double* info,
                              void* stream) {
  LRBMS_REQUIRE_MESH(ctx); CHECK_Q_N(ctx, Q, N); CHECK_PTR(ctx, theta); CHECK_PTR(ctx, B_sys); CHECK_PTR(ctx, rhs_red);
  CHECK_PTR(ctx, work); CHECK_PTR(ctx, u);
  return launch_reduced_solve_batch(ctx, Q, N, nmu, theta, B_sys, rhs_red, work, u, rtol, max_iter, info, (hipStream_t)stream);
}

int64_t lrbms_reduced_precond_size(lrbms_ctx* ctx, int32_t N) {
  if (!ctx || !ctx->has_mesh || N < 1 || N > 64) return -1;
  return reduced_precond_size(ctx, N);
}

int lrbms_reduced_precond_build(lrbms_ctx* ctx, int32_t Q, int32_t N, const double* theta, const double* B_sys, double* work,
                                double* pc, void* stream) {
  LRBMS_REQUIRE_MESH(ctx); CHECK_Q_N(ctx, Q, N); CHECK_PTR(ctx, theta); CHECK_PTR(ctx, B_sys); CHECK_PTR(ctx, work); CHECK_PTR(ctx, pc);
  return launch_reduced_precond_build(ctx, Q, N, theta, B_sys, work, pc, (hipStream_t)stream);
}

int lrbms_reduced_precond_use(lrbms_ctx* ctx, int32_t N, const double* pc) {
  LRBMS_REQUIRE_MESH(ctx);
  ctx->user_pc = pc;
  ctx->user_pc_N = pc ? N : 0;
  return LRBMS_OK;
}

int64_t lrbms_fom_solve_work_size(lrbms_ctx* ctx) {
  if (!ctx || !ctx->has_mesh) return -1;
  return fom_solve_work_size(ctx);
}

int lrbms_fom_solve(lrbms_ctx* ctx, int32_t Q, const double* theta, const double* A_diag, const double* A_cpl, const double* b,
                    double* work, double* x, double rtol, int32_t max_iter, double* info, void* stream) {
  LRBMS_REQUIRE_MESH(ctx); CHECK_Q_N(ctx, Q, 1); CHECK_PTR(ctx, theta); CHECK_PTR(ctx, A_diag); CHECK_PTR(ctx, A_cpl);
  CHECK_PTR(ctx, b); CHECK_PTR(ctx, work); CHECK_PTR(ctx, x);
  return launch_fom_solve(ctx, Q, theta, A_diag, A_cpl, b, work, x, rtol, max_iter, info, (hipStream_t)stream);
}

int lrbms_fom_implicit_euler(lrbms_ctx* ctx, int32_t Q, const double* theta, double dt, int32_t nt, const double* A_diag,
                             const double* A_cpl, const double* b, double* work, double* U, double rtol, int32_t max_iter,
                             double* info, void* stream) {
  LRBMS_REQUIRE_MESH(ctx); CHECK_Q_N(ctx, Q, 1); CHECK_PTR(ctx, theta); CHECK_PTR(ctx, A_diag); CHECK_PTR(ctx, A_cpl);
  CHECK_PTR(ctx, b); CHECK_PTR(ctx, work); CHECK_PTR(ctx, U);
  return launch_fom_implicit_euler(ctx, Q, theta, dt, nt, A_diag, A_cpl, b, work, U, rtol, max_iter, info, (hipStream_t)stream);
}

int lrbms_mass_inverse_norm2(lrbms_ctx* ctx, int32_t L, const double* Y, double* out, void* stream) {
  LRBMS_REQUIRE_MESH(ctx); CHECK_PTR(ctx, Y); CHECK_PTR(ctx, out);
  return launch_mass_inverse_norm2(ctx, L, Y, out, (hipStream_t)stream);
}

int lrbms_div_apply(lrbms_ctx* ctx, int32_t C, int32_t mode, const double* Rt, double* out, void* stream) {
  LRBMS_REQUIRE_MESH(ctx); CHECK_PTR(ctx, Rt); CHECK_PTR(ctx, out);
  return launch_div_apply(ctx, C, mode, Rt, out, (hipStream_t)stream);
}

int lrbms_div_pairing(lrbms_ctx* ctx, int32_t Q, int32_t L, const double* theta, const double* D, const double* G, double* out,
                      void* stream) {
  LRBMS_REQUIRE_MESH(ctx); CHECK_PTR(ctx, theta); CHECK_PTR(ctx, D); CHECK_PTR(ctx, G); CHECK_PTR(ctx, out);
  return launch_div_pairing(ctx, Q, L, theta, D, G, out, (hipStream_t)stream);
}

int lrbms_reduced_reconstruction_terms(lrbms_ctx* ctx, int32_t Q, int32_t N, int32_t L, const double* theta, const double* B_sys,
                                       const double* M_red, const double* rhs_red, const double* G_ud, const double* U,
                                       double* work, double* out, void* stream) {
  LRBMS_REQUIRE_MESH(ctx); CHECK_Q_N(ctx, Q, N); CHECK_PTR(ctx, theta); CHECK_PTR(ctx, B_sys); CHECK_PTR(ctx, M_red);
  CHECK_PTR(ctx, rhs_red); CHECK_PTR(ctx, G_ud); CHECK_PTR(ctx, U); CHECK_PTR(ctx, work); CHECK_PTR(ctx, out);
  return launch_reduced_reconstruction_terms(ctx, Q, N, L, theta, B_sys, M_red, rhs_red, G_ud, U, work, out, (hipStream_t)stream);
}

int lrbms_reduced_implicit_euler(lrbms_ctx* ctx, int32_t Q, int32_t N, const double* theta, double dt, int32_t nt,
                                 const double* B_sys, const double* M_red, const double* rhs_red, double* work, double* U,
                                 double rtol, int32_t max_iter, double* info, void* stream) {
  LRBMS_REQUIRE_MESH(ctx); CHECK_Q_N(ctx, Q, N); CHECK_PTR(ctx, theta); CHECK_PTR(ctx, B_sys); CHECK_PTR(ctx, M_red);
  CHECK_PTR(ctx, rhs_red); CHECK_PTR(ctx, work); CHECK_PTR(ctx, U);
  return launch_reduced_implicit_euler(ctx, Q, N, theta, dt, nt, B_sys, M_red, rhs_red, work, U, rtol, max_iter, info,
                                       (hipStream_t)stream);
}

int64_t lrbms_reduced_time_residual_work_size(lrbms_ctx* ctx, int32_t N) {
  if (!ctx || !ctx->has_mesh || N < 1) return -1;
  return (int64_t)ctx->S * (6 * (int64_t)N * N + N + 1);
}

int lrbms_reduced_time_residual(lrbms_ctx* ctx, int32_t Q, int32_t N, int32_t L, const double* theta, const double* B_sys,
                                const double* M_red, const double* dU, double* work, double* out, void* stream) {
  LRBMS_REQUIRE_MESH(ctx); CHECK_Q_N(ctx, Q, N); CHECK_PTR(ctx, theta); CHECK_PTR(ctx, B_sys); CHECK_PTR(ctx, M_red);
  CHECK_PTR(ctx, dU); CHECK_PTR(ctx, work); CHECK_PTR(ctx, out);
  return launch_reduced_time_residual(ctx, Q, N, L, theta, B_sys, M_red, dU, work, out, (hipStream_t)stream);
}

int lrbms_assemble_dirichlet_correction(lrbms_ctx* ctx, int32_t Q, const double* lam, double* D_corr, void* stream) {
  LRBMS_REQUIRE_MESH(ctx); CHECK_Q_N(ctx, Q, 1); CHECK_PTR(ctx, lam); CHECK_PTR(ctx, D_corr);
  return launch_assemble_dcorr(ctx, Q, lam, D_corr, (hipStream_t)stream);
}

int64_t lrbms_local_correction_work_size(lrbms_ctx* ctx, int32_t nmark) {
  if (!ctx || !ctx->has_mesh || nmark < 1) return -1;
  return local_correction_work_size(ctx, nmark);
}

int lrbms_local_correction_solve(lrbms_ctx* ctx, int32_t Q, const double* theta, int32_t nmark, const int32_t* marked,
                                 const double* A_diag, const double* A_cpl, const double* D_corr, const double* b,
                                 double* work, double* corr, double rtol, int32_t max_iter, double* info, void* stream) {
  LRBMS_REQUIRE_MESH(ctx); CHECK_Q_N(ctx, Q, 1); CHECK_PTR(ctx, theta); CHECK_PTR(ctx, marked); CHECK_PTR(ctx, A_diag);
  CHECK_PTR(ctx, A_cpl); CHECK_PTR(ctx, D_corr); CHECK_PTR(ctx, b); CHECK_PTR(ctx, work); CHECK_PTR(ctx, corr);
  return launch_local_correction(ctx, Q, theta, nmark, marked, A_diag, A_cpl, D_corr, b, work, corr, rtol, max_iter, info,
                                 (hipStream_t)stream);
}

int lrbms_blockell_apply(lrbms_ctx* ctx, int32_t M, const double* A, const double* x, double* y, void* stream) {
  LRBMS_REQUIRE_MESH(ctx); CHECK_Q_N(ctx, 1, M); CHECK_PTR(ctx, A); CHECK_PTR(ctx, x); CHECK_PTR(ctx, y);
  return launch_blockell_apply(ctx, ctx->S, M, A, (long)ctx->t.nT * 36, x, y, (hipStream_t)stream);
}

int lrbms_fom_apply(lrbms_ctx* ctx, int32_t Q, int32_t M, const double* theta, const double* A_diag, const double* A_cpl,
                    const double* x, double* y, void* stream) {
  LRBMS_REQUIRE_MESH(ctx); CHECK_Q_N(ctx, Q, M); CHECK_PTR(ctx, theta); CHECK_PTR(ctx, A_diag); CHECK_PTR(ctx, A_cpl);
  CHECK_PTR(ctx, x); CHECK_PTR(ctx, y);
  return launch_fom_apply(ctx, Q, M, theta, A_diag, A_cpl, x, y, (hipStream_t)stream);
}

int lrbms_gemm_tn(lrbms_ctx* ctx, int32_t batch, int32_t K, int32_t Mx, int32_t My, const double* X, int64_t sx, int32_t ldx,
                  const double* Y, int64_t sy, int32_t ldy, double* G, int64_t sg, int32_t ldg, const double* rowscale,
                  double alpha, void* stream) {
  if (!ctx) return LRBMS_E_INVALID;
  CHECK_PTR(ctx, X); CHECK_PTR(ctx, Y); CHECK_PTR(ctx, G);
  return launch_gemm_tn(ctx, batch, K, Mx, My, X, sx, ldx, Y, sy, ldy, G, sg, ldg, rowscale, alpha, (hipStream_t)stream);
}

}  // extern "C"
