// mi_ctx.cpp -- context object and C-ABI (include/mi_elasticity.h) of the device hot path.
//
// Host logic here is the counterpart of Solid::solve_nonlinear_timestep / assemble_system /
// solve_linear_system (nonlinear_elasticity.cc:410-499, :1044-1087, :1153-1211): it owns the device
// state, orders the kernel launches on one HIP stream and turns HIP failures into status codes.
// There is no CPU fallback: every entry point needs a working HIP device.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "mi_internal.h"

namespace mi_detail
{
  std::string g_create_error;
} // namespace mi_detail
using namespace mi_detail;


namespace mi_detail
{
  int fail(mi_ctx *c, int code, const char *fmt, ...)
  {
    char    buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (c)
      c->err = buf;
    else
      g_create_error = buf;
    return code;
  }

  // ---- profiling stamps: HIP events on the context's stream, resolved after a synchronize
  int tic(mi_ctx *c, int cls)
  {
    if (!c->profiling)
      return -1;
    if (c->stamps_used == c->stamps.size())
      {
        mi_ctx::Stamp s;
        if (hipEventCreate(&s.a) != hipSuccess || hipEventCreate(&s.b) != hipSuccess)
          return -1;
        c->stamps.push_back(s);
      }
    mi_ctx::Stamp &s = c->stamps[c->stamps_used];
    s.cls            = cls;
    hipEventRecord(s.a, c->stream);
    return int(c->stamps_used++);
  }
  void toc(mi_ctx *c, int id)
  {
    if (id >= 0)
      hipEventRecord(c->stamps[size_t(id)].b, c->stream);
  }
  void resolve_stamps(mi_ctx *c)
  {
    for (size_t i = 0; i < c->stamps_used; ++i)
      {
        float ms = 0;
        if (hipEventElapsedTime(&ms, c->stamps[i].a, c->stamps[i].b) == hipSuccess)
          {
            c->timings.ms[c->stamps[i].cls] += ms;
            c->timings.count[c->stamps[i].cls] += 1;
          }
      }
    c->stamps_used = 0;
  }
  int sync(mi_ctx *c)
  {
    HIPCHK(c, hipStreamSynchronize(c->stream));
    resolve_stamps(c);
    return MI_OK;
  }

  mi::AsmParams asm_params(mi_ctx *c)
  {
    mi::AsmParams p{};
    p.conn   = c->d_conn;
    p.cverts = c->d_cverts;
    p.off    = c->d_off;
    p.rowptr = c->d_rowptr;
    p.cmask  = c->d_cmask;
    p.tab1d  = c->d_tab;
    p.u      = c->vec(MI_V_TOTAL_DISPLACEMENT);
    p.du     = c->vec(MI_V_SOLUTION_DELTA);
    p.acc    = c->vec(MI_V_ACCELERATION);
    p.stress = c->vec(MI_V_EXTERNAL_STRESS);
    p.rhs    = c->vec(MI_V_SYSTEM_RHS);
    p.vals   = c->d_vals;
    p.mu     = c->mat.mu;
    p.kappa  = c->kappa;
    p.rho    = c->mat.rho;
    p.alpha1 = c->alpha[1];
    for (int i = 0; i < 3; ++i)
      p.body[i] = c->mat.body_force[i];
    return p;
  }

  mi::NewmarkParams newmark_params(mi_ctx *c)
  {
    mi::NewmarkParams p{};
    p.u      = c->vec(MI_V_TOTAL_DISPLACEMENT);
    p.u_old  = c->vec(MI_V_TOTAL_DISPLACEMENT_OLD);
    p.v      = c->vec(MI_V_VELOCITY);
    p.v_old  = c->vec(MI_V_VELOCITY_OLD);
    p.a      = c->vec(MI_V_ACCELERATION);
    p.a_old  = c->vec(MI_V_ACCELERATION_OLD);
    p.du     = c->vec(MI_V_SOLUTION_DELTA);
    p.alpha1 = c->alpha[1];
    p.alpha2 = c->alpha[2];
    p.alpha3 = c->alpha[3];
    p.alpha4 = c->alpha[4];
    p.alpha5 = c->alpha[5];
    p.alpha6 = c->alpha[6];
    p.n      = c->n;
    return p;
  }

  mi::SpmvParams spmv_params(mi_ctx *c, const double *x, double *y, const double *dotv, double *partials,
                             const int32_t *done)
  {
    mi::SpmvParams p{};
    p.rowptr   = c->d_rowptr;
    p.col      = c->d_col;
    p.vals     = c->d_vals;
    p.x        = x;
    p.y        = y;
    p.dotv     = dotv;
    p.partials = partials;
    p.done     = done;
    p.row0     = 0;
    p.nrows    = c->mesh.nnodes;
    p.rowptr_host_nnzb = c->mesh.nnzb;
    return p;
  }

  mi::SellParams sell_params(mi_ctx *c, const double *x, double *y, const double *dotv, double *partials,
                             const int32_t *done)
  {
    mi::SellParams p{};
    p.perm     = c->d_sell_perm;
    p.len      = c->d_sell_len;
    p.off      = c->d_sell_off;
    p.col      = c->d_sell_col;
    p.vals     = c->active_sell_vals ? c->active_sell_vals : c->d_sell_vals;
    p.x        = x;
    p.y        = y;
    p.dotv     = dotv;
    p.partials = partials;
    p.done     = done;
    p.nslices  = int32_t(c->mesh.sell_nslices);
    p.xcd_remap = c->xcd_remap;
    return p;
  }

  // y = K x (+ optional fused dot partials) with the selected kernel variant
  void enqueue_spmv(mi_ctx *c, const double *x, double *y, const double *dotv, double *partials, const int32_t *done)
  {
    if (c->spmv_variant == 3 || c->active_sell_vals) // linear-model operators exist in sliced-ELL form only
      mi::launch_sell_spmv(c->dim, sell_params(c, x, y, dotv, partials, done), c->grid_spmv, c->stream,
                           c->sell_unroll);
    else
      mi::launch_spmv(c->dim, spmv_params(c, x, y, dotv, partials, done), c->grid_spmv, c->stream, c->spmv_variant,
                      c->maxrow);
  }

  // the enqueue part of assemble_system (no host synchronisation)
  int enqueue_assembly(mi_ctx *c)
  {
    const int64_t dd = int64_t(c->dim) * c->dim;
    HIPCHK(c, hipMemsetAsync(c->d_vals, 0, size_t(c->mesh.nnzb) * dd * sizeof(double), c->stream)); // :1054
    HIPCHK(c, hipMemsetAsync(c->vec(MI_V_SYSTEM_RHS), 0, size_t(c->n) * sizeof(double), c->stream)); // :1055
    mi::AsmParams p  = asm_params(c);
    const int     t0 = tic(c, MI_T_ASSEMBLE_CELLS);
    for (int col = 0; col < c->mesh.ncolours; ++col)
      {
        p.cell_begin = c->mesh.colour_begin[col];
        p.cell_count = int32_t(c->mesh.colour_begin[col + 1] - c->mesh.colour_begin[col]);
        if (mi::launch_assemble_cells(c->dim, c->degree, p, c->stream))
          return fail(c, MI_EINVAL, "no assembly kernel for dim=%d degree=%d", c->dim, c->degree);
        const int fb = int(c->mesh.iface_colour_begin[col]);
        const int fc = int(c->mesh.iface_colour_begin[col + 1]) - fb;
        if (mi::launch_neumann_faces(c->dim, c->degree, p, c->d_faces, fb, fc, c->stream))
          return fail(c, MI_EINVAL, "no face kernel for dim=%d degree=%d", c->dim, c->degree);
      }
    toc(c, t0);
    mi::launch_extract_dinv(c->dim, c->d_vals, c->d_diagpos, c->work(W_DINV), c->mesh.nnodes, c->stream);
    // SpMV-side copy of the tangent in sliced-ELL order
    mi::launch_bsr_to_sell(c->dim, sell_params(c, nullptr, nullptr, nullptr, nullptr, nullptr), c->d_rowptr, c->d_vals,
                           c->d_sell_vals, c->stream);
    HIPCHK(c, hipGetLastError());
    return MI_OK;
  }
  // Jacobi-PCG (deal.II SolverCG semantics: start from x, stop when ||r||_2 <= tolerance) on the active matrix;
  // followed by constraints.distribute (x[constrained] = 0)
  int cg_run(mi_ctx *c, double *x, const double *b, double tol, int64_t max_it, int *its, double *res)
  {
  const int tt = tic(c, MI_T_CG_TOTAL);
  mi::CgParams cg{};
  cg.x        = x;
  cg.r        = c->work(W_R);
  cg.p        = c->work(W_P);
  cg.q        = c->work(W_Q);
  cg.dinv     = c->active_dinv ? c->active_dinv : c->work(W_DINV);
  cg.part_rr  = c->part(0);
  cg.part_rz  = c->part(1);
  cg.part_pq  = c->part(2);
  cg.sc       = c->d_sc;
  cg.flags    = c->d_flags;
  cg.n        = c->n;
  cg.npart    = c->grid_vec;
  cg.npart_pq = c->grid_spmv;

  // r0 = b - A x0, tolerance = rel_tol * ||b||  (:1171-1172)
  {
    const int t = tic(c, MI_T_SPMV);
    enqueue_spmv(c, x, cg.q, nullptr, nullptr, nullptr);
    toc(c, t);
  }
  mi::launch_cg_init_residual(cg, b, c->part(4), c->grid_vec, c->stream);
  mi::launch_cg_set_tolerance(cg, c->part(4), tol, c->stream);

  int32_t *h_flags = reinterpret_cast<int32_t *>(c->h_pinned + 8);
  int64_t  it      = 0;
  bool     done    = false;
  while (!done && it < max_it)
    {
      const int64_t stop = std::min<int64_t>(max_it, it + CG_BATCH);
      for (; it < stop;)
        {
          ++it;
          int t = tic(c, MI_T_CG_VECTOR);
          mi::launch_cg_update_p(cg, int(it), c->grid_vec, c->stream);
          toc(c, t);
          t = tic(c, MI_T_SPMV);
          enqueue_spmv(c, cg.p, cg.q, cg.p, cg.part_pq, cg.flags);
          toc(c, t);
          t = tic(c, MI_T_CG_VECTOR);
          mi::launch_cg_update_xr(cg, int(it), c->grid_vec, c->stream);
          toc(c, t);
        }
      mi::launch_cg_final_check(cg, int(it), c->stream);
      HIPCHK(c, hipGetLastError());
      HIPCHK(c, hipMemcpyAsync(h_flags, c->d_flags, 2 * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
      HIPCHK(c, hipMemcpyAsync(c->h_pinned, c->d_sc, 8 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
      HIPCHK(c, hipStreamSynchronize(c->stream));
      done = h_flags[0] != 0;
    }
  if (max_it <= 0)
    {
      mi::launch_cg_final_check(cg, 0, c->stream);
      HIPCHK(c, hipMemcpyAsync(h_flags, c->d_flags, 2 * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
      HIPCHK(c, hipMemcpyAsync(c->h_pinned, c->d_sc, 8 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
      HIPCHK(c, hipStreamSynchronize(c->stream));
      done = h_flags[0] != 0;
    }
  mi::launch_zero_constrained(c->dim, x, c->d_cmask, c->n, c->stream); // constraints.distribute (:1208)
  toc(c, tt);
  HIPCHK(c, hipGetLastError());
  int rc = sync(c);
  if (rc)
    return rc;
  if (its)
    *its = h_flags[1];
  if (res)
    *res = c->h_pinned[3];
  if (!done)
    return fail(c, MI_ENOCONV_LIN, "CG did not reach tolerance %.3e within %lld iterations (residual %.3e)", std::fabs(tol),
                (long long)max_it, c->h_pinned[3]);
  return MI_OK;
}
} // namespace mi_detail

extern "C" {

const char *mi_last_error(const mi_ctx *ctx)
{
  return ctx ? ctx->err.c_str() : g_create_error.c_str();
}

void mi_ctx_destroy(mi_ctx *c)
{
  if (!c)
    return;
  hipSetDevice(c->device);
  if (c->stream)
    hipStreamSynchronize(c->stream);
  linear_destroy(c);
  for (auto &s : c->stamps)
    {
      hipEventDestroy(s.a);
      hipEventDestroy(s.b);
    }
  void *ptrs[] = {c->d_conn, c->d_rowptr, c->d_col,  c->d_diagpos, c->d_iface_nodes, c->d_faces, c->d_flags,
                  c->d_cverts, c->d_tab, c->d_vals, c->d_vecs, c->d_work, c->d_saved, c->d_part,
                  c->d_sc, c->d_iface_buf, c->d_off, c->d_cmask, c->d_sell_perm, c->d_sell_len, c->d_sell_col,
                  c->d_sell_off, c->d_sell_vals};
  for (void *p : ptrs)
    if (p)
      hipFree(p);
  if (c->h_pinned)
    hipHostFree(c->h_pinned);
  if (c->stream)
    hipStreamDestroy(c->stream);
  delete c;
}

int mi_ctx_create(const mi_mesh_desc *md, const mi_material_desc *mat, const mi_newmark_desc *nm, int device_id,
                  const mi_comm_desc *comm, mi_ctx **out)
{
  if (!md || !mat || !nm || !out)
    return fail(nullptr, MI_EINVAL, "null argument");
  *out = nullptr;
  if (comm && comm->size > 1)
    return fail(nullptr, MI_EINVAL, "domain decomposition is not available in this build (comm size %d)", comm->size);
  if (!(mat->nu > -1.0 && mat->nu < 0.5) || !(mat->mu > 0.0) || mat->rho < 0.0)
    return fail(nullptr, MI_EINVAL, "material out of range (mu>0, -1<nu<0.5, rho>=0)");
  if (!(nm->beta > 0.0) || !(nm->delta_t > 0.0))
    return fail(nullptr, MI_EINVAL, "Newmark beta and time step must be positive");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(nullptr, MI_EHIP, "no HIP device available (the hot path has no CPU fallback)");
  if (device_id < 0 || device_id >= ndev)
    return fail(nullptr, MI_EINVAL, "device %d out of range (%d devices)", device_id, ndev);

  mi_ctx *c = new mi_ctx;
  c->device = device_id;
  c->dim    = md->dim;
  c->degree = md->degree;
  c->mat    = *mat;
  c->nm     = *nm;
  auto bail = [&](int code) {
    g_create_error = c->err;
    mi_ctx_destroy(c);
    return code;
  };
  try
    {
      c->mesh.build(md->dim, md->degree, md->reps, md->lo, md->hi, md->face_role, md->vertex_perturbation);
      c->tab.build(md->degree, md->degree + 2); // qf_cell(p+2), qf_face(p+2): nonlinear_elasticity.cc:74-75
    }
  catch (const std::exception &e)
    {
      c->err = e.what();
      return bail(MI_EINVAL);
    }
  if (md->dim == 3 && md->degree > 2)
    {
      c->err = "3D elements above degree 2 are not supported by the device kernels";
      return bail(MI_EINVAL);
    }
  c->n     = c->mesh.ndofs;
  c->kappa = (2.0 * mat->mu * (1.0 + mat->nu)) / (3.0 * (1.0 - 2.0 * mat->nu)); // neo_hook_material.h:20
  // nonlinear_elasticity.h:242-250
  c->alpha[1] = 1. / (nm->beta * std::pow(nm->delta_t, 2));
  c->alpha[2] = 1. / (nm->beta * nm->delta_t);
  c->alpha[3] = (1 - (2 * nm->beta)) / (2 * nm->beta);
  c->alpha[4] = nm->gamma / (nm->beta * nm->delta_t);
  c->alpha[5] = 1 - (nm->gamma / nm->beta);
  c->alpha[6] = (1 - (nm->gamma / (2 * nm->beta))) * nm->delta_t;

#define CREATE_CHK(call)                  \
  do                                      \
    {                                     \
      int rc_ = (call);                   \
      if (rc_ != MI_OK)                   \
        return bail(rc_);                 \
    }                                     \
  while (0)
#define CREATE_HIP(call)                                                                         \
  do                                                                                             \
    {                                                                                            \
      hipError_t e_ = (call);                                                                    \
      if (e_ != hipSuccess)                                                                      \
        {                                                                                        \
          fail(c, MI_EHIP, "%s failed: %s", #call, hipGetErrorString(e_));                       \
          return bail(MI_EHIP);                                                                  \
        }                                                                                        \
    }                                                                                            \
  while (0)

  CREATE_HIP(hipSetDevice(device_id));
  CREATE_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  const mi::HostMesh &m = c->mesh;
  CREATE_CHK(upload(c, &c->d_conn, m.conn));
  CREATE_CHK(upload(c, &c->d_cverts, m.cverts));
  CREATE_CHK(upload(c, &c->d_off, m.off));
  CREATE_CHK(upload(c, &c->d_rowptr, m.rowptr));
  CREATE_CHK(upload(c, &c->d_col, m.colidx));
  CREATE_CHK(upload(c, &c->d_diagpos, m.diagpos));
  CREATE_CHK(upload(c, &c->d_cmask, m.cmask));
  CREATE_CHK(upload(c, &c->d_iface_nodes, m.iface_nodes));
  {
    std::vector<int32_t> f;
    for (const auto &x : m.iface_faces)
      {
        f.push_back(x.cell);
        f.push_back(x.face);
      }
    CREATE_CHK(upload(c, &c->d_faces, f));
  }
  CREATE_CHK(upload(c, &c->d_tab, c->tab.packed()));
  const size_t dd = size_t(c->dim) * c->dim;
  CREATE_HIP(hipMalloc((void **)&c->d_vals, (size_t(m.nnzb) * dd + 2) * sizeof(double))); // +2: 16-byte row reads
  CREATE_CHK(upload(c, &c->d_sell_perm, m.sell_perm));
  CREATE_CHK(upload(c, &c->d_sell_len, m.sell_len));
  CREATE_CHK(upload(c, &c->d_sell_off, m.sell_off));
  CREATE_HIP(hipMalloc((void **)&c->d_sell_col, size_t(m.sell_nblk64) * 64 * sizeof(int32_t)));
  CREATE_HIP(hipMalloc((void **)&c->d_sell_vals, size_t(m.sell_nblk64) * 64 * dd * sizeof(double)));
  CREATE_HIP(hipMalloc((void **)&c->d_vecs, size_t(MI_V_COUNT) * size_t(c->n) * sizeof(double)));
  CREATE_HIP(hipMalloc((void **)&c->d_work, size_t(W_COUNT) * size_t(c->n) * sizeof(double)));
  CREATE_HIP(hipMalloc((void **)&c->d_saved, size_t(6) * size_t(c->n) * sizeof(double)));
  CREATE_HIP(hipMalloc((void **)&c->d_part, size_t(8) * MAX_PART * sizeof(double)));
  CREATE_HIP(hipMalloc((void **)&c->d_sc, 16 * sizeof(double)));
  CREATE_HIP(hipMalloc((void **)&c->d_flags, 4 * sizeof(int32_t)));
  const size_t nif = std::max<size_t>(m.iface_nodes.size(), 1) * size_t(c->dim);
  CREATE_HIP(hipMalloc((void **)&c->d_iface_buf, nif * sizeof(double)));
  c->h_pinned_doubles = nif + 64;
  CREATE_HIP(hipHostMalloc((void **)&c->h_pinned, c->h_pinned_doubles * sizeof(double), hipHostMallocDefault));
  CREATE_HIP(hipMemsetAsync(c->d_vals, 0, size_t(m.nnzb) * dd * sizeof(double), c->stream));
  CREATE_HIP(hipMemsetAsync(c->d_vecs, 0, size_t(MI_V_COUNT) * size_t(c->n) * sizeof(double), c->stream));
  CREATE_HIP(hipMemsetAsync(c->d_work, 0, size_t(W_COUNT) * size_t(c->n) * sizeof(double), c->stream));
  CREATE_HIP(hipMemsetAsync(c->d_sc, 0, 16 * sizeof(double), c->stream));
  CREATE_HIP(hipMemsetAsync(c->d_flags, 0, 4 * sizeof(int32_t), c->stream));
  CREATE_HIP(hipMemsetAsync(c->d_sell_vals, 0, size_t(m.sell_nblk64) * 64 * dd * sizeof(double), c->stream));
  mi::launch_sell_build_cols(sell_params(c, nullptr, nullptr, nullptr, nullptr, nullptr), c->d_rowptr, c->d_col,
                             c->d_sell_col, c->stream);
  CREATE_HIP(hipGetLastError());
  CREATE_HIP(hipStreamSynchronize(c->stream));

  // launch geometry: vector kernels and SpMV use fixed grids so that reduction partials are deterministic
  c->grid_vec = int(std::min<int64_t>(1024, (c->n + 255) / 256));
  // SpMV: one wavefront per 64-row slice (measured best), grid-stride above MAX_PART workgroups
  c->grid_spmv = int(std::max<int64_t>(1, std::min<int64_t>(MAX_PART, (m.sell_nslices + 3) / 4)));
  for (int64_t nd = 0; nd < m.nnodes; ++nd)
    c->maxrow = std::max(c->maxrow, int(m.rowptr[size_t(nd) + 1] - m.rowptr[size_t(nd)]));
  if (const char *v = getenv("MI_SPMV_VARIANT"))
    c->spmv_variant = atoi(v);
  *out = c;
  return MI_OK;
#undef CREATE_CHK
#undef CREATE_HIP
}

int64_t mi_n_dofs(const mi_ctx *c)
{
  return c->n;
}
int64_t mi_n_nodes(const mi_ctx *c)
{
  return c->mesh.nnodes;
}
int64_t mi_n_cells(const mi_ctx *c)
{
  return c->mesh.ncells;
}
int64_t mi_nnz(const mi_ctx *c)
{
  return c->mesh.nnzb * c->dim * c->dim;
}
int mi_n_colours(const mi_ctx *c)
{
  return c->mesh.ncolours;
}
int mi_get_node_coords(const mi_ctx *c, double *xyz)
{
  std::memcpy(xyz, c->mesh.node_xyz.data(), c->mesh.node_xyz.size() * sizeof(double));
  return MI_OK;
}
int mi_get_constrained(const mi_ctx *c, uint8_t *flags)
{
  for (int64_t nd = 0; nd < c->mesh.nnodes; ++nd)
    for (int k = 0; k < c->dim; ++k)
      flags[nd * c->dim + k] = (c->mesh.cmask[size_t(nd)] >> k) & 1;
  return MI_OK;
}
int mi_n_interface_nodes(const mi_ctx *c)
{
  return int(c->mesh.iface_nodes.size());
}
int mi_get_interface_nodes(const mi_ctx *c, int32_t *node_ids, double *coords)
{
  const int dim = c->dim;
  for (size_t i = 0; i < c->mesh.iface_nodes.size(); ++i)
    {
      const int32_t nd = c->mesh.iface_nodes[i];
      if (node_ids)
        node_ids[i] = nd;
      if (coords)
        for (int k = 0; k < dim; ++k)
          coords[i * dim + k] = c->mesh.node_xyz[size_t(nd) * dim + k];
    }
  return MI_OK;
}

int mi_set_interface_traction(mi_ctx *c, int n, const double *vals)
{
  if (n != int(c->mesh.iface_nodes.size()))
    return fail(c, MI_EINVAL, "expected %d interface nodes, got %d", int(c->mesh.iface_nodes.size()), n);
  if (n == 0)
    return MI_OK;
  HIPCHK(c, hipSetDevice(c->device));
  c->h_iface.assign(vals, vals + size_t(n) * c->dim);
  std::memcpy(c->h_pinned + 64, vals, size_t(n) * c->dim * sizeof(double));
  HIPCHK(c, hipMemcpyAsync(c->d_iface_buf, c->h_pinned + 64, size_t(n) * c->dim * sizeof(double),
                           hipMemcpyHostToDevice, c->stream));
  mi::launch_scatter_nodes(c->dim, c->vec(MI_V_EXTERNAL_STRESS), c->d_iface_nodes, n, c->d_iface_buf, c->stream);
  HIPCHK(c, hipGetLastError());
  return sync(c); // the pinned staging buffer is reused by the next call
}

int mi_get_interface_displacement(mi_ctx *c, int n, double *vals)
{
  if (n != int(c->mesh.iface_nodes.size()))
    return fail(c, MI_EINVAL, "expected %d interface nodes, got %d", int(c->mesh.iface_nodes.size()), n);
  if (n == 0)
    return MI_OK;
  HIPCHK(c, hipSetDevice(c->device));
  mi::launch_gather_nodes(c->dim, c->vec(MI_V_TOTAL_DISPLACEMENT), c->d_iface_nodes, n, c->d_iface_buf, c->stream);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipMemcpyAsync(c->h_pinned + 64, c->d_iface_buf, size_t(n) * c->dim * sizeof(double),
                           hipMemcpyDeviceToHost, c->stream));
  int rc = sync(c);
  if (rc)
    return rc;
  std::memcpy(vals, c->h_pinned + 64, size_t(n) * c->dim * sizeof(double));
  return MI_OK;
}

int mi_newton_begin_step(mi_ctx *c)
{
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipMemsetAsync(c->vec(MI_V_SOLUTION_DELTA), 0, size_t(c->n) * sizeof(double), c->stream));
  HIPCHK(c, hipMemsetAsync(c->vec(MI_V_NEWTON_UPDATE), 0, size_t(c->n) * sizeof(double), c->stream));
  return MI_OK;
}

int mi_update_acceleration(mi_ctx *c)
{
  HIPCHK(c, hipSetDevice(c->device));
  const int t = tic(c, MI_T_NEWMARK);
  mi::launch_newmark_acceleration(newmark_params(c), c->stream);
  toc(c, t);
  HIPCHK(c, hipGetLastError());
  return MI_OK;
}

int mi_assemble(mi_ctx *c, double *res_norm)
{
  HIPCHK(c, hipSetDevice(c->device));
  const int t = tic(c, MI_T_ASSEMBLE_TOTAL);
  int       rc = enqueue_assembly(c);
  if (rc)
    return rc;
  mi::launch_masked_norm(c->dim, c->vec(MI_V_SYSTEM_RHS), c->d_cmask, c->n, c->part(3), c->grid_vec, c->d_sc + 8,
                         c->stream);
  toc(c, t);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipMemcpyAsync(c->h_pinned, c->d_sc + 8, sizeof(double), hipMemcpyDeviceToHost, c->stream));
  rc = sync(c);
  if (rc)
    return rc;
  if (res_norm)
    *res_norm = c->h_pinned[0];
  return MI_OK;
}

int mi_cg_solve(mi_ctx *c, double rel_tol, int64_t max_it, int *its, double *res)
{
  HIPCHK(c, hipSetDevice(c->device));
  if (rel_tol < 0)
    return fail(c, MI_EINVAL, "negative tolerance");
  c->active_sell_vals = nullptr; // tangent
  c->active_dinv      = nullptr;
  // warm start: SolverCG starts from the passed vector (:1184-1187)
  return cg_run(c, c->vec(MI_V_NEWTON_UPDATE), c->vec(MI_V_SYSTEM_RHS), rel_tol, max_it, its, res);
}

int mi_apply_newton_update(mi_ctx *c, double *upd_norm)
{
  HIPCHK(c, hipSetDevice(c->device));
  mi::launch_masked_norm(c->dim, c->vec(MI_V_NEWTON_UPDATE), c->d_cmask, c->n, c->part(3), c->grid_vec, c->d_sc + 9,
                         c->stream);
  mi::launch_vec_add(c->vec(MI_V_SOLUTION_DELTA), c->vec(MI_V_NEWTON_UPDATE), c->n, c->stream); // :487
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipMemcpyAsync(c->h_pinned, c->d_sc + 9, sizeof(double), hipMemcpyDeviceToHost, c->stream));
  int rc = sync(c);
  if (rc)
    return rc;
  if (upd_norm)
    *upd_norm = c->h_pinned[0];
  return MI_OK;
}

int mi_newmark_finish_step(mi_ctx *c)
{
  HIPCHK(c, hipSetDevice(c->device));
  const int t = tic(c, MI_T_NEWMARK);
  mi::launch_newmark_finish(newmark_params(c), c->stream);
  toc(c, t);
  HIPCHK(c, hipGetLastError());
  return MI_OK;
}

// solve_nonlinear_timestep (:410-499) followed by :139-144
int mi_newmark_step(mi_ctx *c, const mi_solver_desc *s, mi_step_info *info)
{
  if (!s || !info)
    return fail(c, MI_EINVAL, "null argument");
  const auto t_begin = std::chrono::steady_clock::now();
  std::memset(info, 0, sizeof(*info));
  int rc = mi_newton_begin_step(c);
  if (rc)
    return rc;
  // Errors() default/reset value is 1.0 (nonlinear_elasticity.h:293-315)
  double error_residual = 1.0, error_residual_0 = 1.0, error_residual_norm = 1.0;
  double error_update = 1.0, error_update_0 = 1.0, error_update_norm = 1.0;
  int    newton_iteration = 0;
  for (; newton_iteration < s->max_iterations_NR; ++newton_iteration) // :436
    {
      if ((rc = mi_update_acceleration(c))) // :444
        return rc;
      if ((rc = mi_assemble(c, &error_residual))) // :446-449
        return rc;
      info->assemblies++;
      if (newton_iteration == 0)
        error_residual_0 = error_residual;
      error_residual_norm = error_residual;
      if (error_residual_0 != 0.0)
        error_residual_norm /= error_residual_0;
      if (newton_iteration > 0 && ((error_update_norm <= s->tol_u || error_update <= 1e-15) &&
                                   (error_residual_norm <= s->tol_f || error_residual <= 5e-9))) // :459-463
        {
          info->converged = 1;
          break;
        }
      int    its = 0;
      double res = 0;
      rc         = mi_cg_solve(c, s->tol_lin, int64_t(double(c->n) * s->max_iterations_lin), &its, &res); // :472
      if (info->newton_iterations < 16)
        {
          info->lin_its[info->newton_iterations] = its;
          info->lin_res[info->newton_iterations] = res;
        }
      info->lin_its_total += its;
      info->newton_iterations++;
      if (rc)
        return rc;
      if ((rc = mi_apply_newton_update(c, &error_update))) // :476-487
        return rc;
      if (newton_iteration == 0)
        error_update_0 = error_update;
      error_update_norm = error_update;
      if (error_update_0 != 0.0)
        error_update_norm /= error_update_0;
    }
  info->res_norm = error_residual_norm;
  info->res_abs  = error_residual;
  info->upd_norm = error_update_norm;
  info->upd_abs  = error_update;
  if (!(newton_iteration < s->max_iterations_NR)) // :497
    return fail(c, MI_ENOCONV_NR, "No convergence in nonlinear solver!");
  if ((rc = mi_newmark_finish_step(c)))
    return rc;
  if ((rc = sync(c)))
    return rc;
  c->timings.ms[MI_T_STEP] +=
    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
  c->timings.count[MI_T_STEP] += 1;
  return MI_OK;
}

int mi_state_save(mi_ctx *c)
{
  HIPCHK(c, hipSetDevice(c->device));
  // the six state vectors are the first six of the vector block (nonlinear_elasticity.cc:370-375)
  HIPCHK(c, hipMemcpyAsync(c->d_saved, c->d_vecs, size_t(6) * size_t(c->n) * sizeof(double),
                           hipMemcpyDeviceToDevice, c->stream));
  c->have_saved = true;
  return MI_OK;
}
int mi_state_restore(mi_ctx *c)
{
  if (!c->have_saved)
    return fail(c, MI_EINVAL, "state_variables are not the same as previously saved.");
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipMemcpyAsync(c->d_vecs, c->d_saved, size_t(6) * size_t(c->n) * sizeof(double),
                           hipMemcpyDeviceToDevice, c->stream));
  return MI_OK;
}

struct mi_snapshot
{
  double *d = nullptr;
};

int mi_snapshot_create(mi_ctx *c, mi_snapshot **out)
{
  if (!out)
    return fail(c, MI_EINVAL, "null argument");
  HIPCHK(c, hipSetDevice(c->device));
  mi_snapshot *s = new mi_snapshot;
  hipError_t   e = hipMalloc((void **)&s->d, size_t(c->n) * sizeof(double));
  if (e != hipSuccess)
    {
      delete s;
      return fail(c, MI_EHIP, "hipMalloc of a snapshot failed: %s", hipGetErrorString(e));
    }
  *out = s;
  return MI_OK;
}
void mi_snapshot_destroy(mi_ctx *c, mi_snapshot *s)
{
  if (!s)
    return;
  hipSetDevice(c->device);
  hipStreamSynchronize(c->stream);
  hipFree(s->d);
  delete s;
}
int mi_snapshot_store(mi_ctx *c, mi_snapshot *s, int which)
{
  if (!s || which < 0 || which >= MI_V_COUNT)
    return fail(c, MI_EINVAL, "bad snapshot or vector id");
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipMemcpyAsync(s->d, c->vec(which), size_t(c->n) * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
  return MI_OK;
}
int mi_snapshot_load(mi_ctx *c, const mi_snapshot *s, int which)
{
  if (!s || which < 0 || which >= MI_V_COUNT)
    return fail(c, MI_EINVAL, "bad snapshot or vector id");
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipMemcpyAsync(c->vec(which), s->d, size_t(c->n) * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
  return MI_OK;
}

int mi_vec_get(mi_ctx *c, int which, double *host, int64_t n)
{
  if (which < 0 || which >= MI_V_COUNT || n != c->n)
    return fail(c, MI_EINVAL, "bad vector id or length");
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipMemcpy(host, c->vec(which), size_t(n) * sizeof(double), hipMemcpyDeviceToHost));
  return MI_OK;
}
int mi_vec_set(mi_ctx *c, int which, const double *host, int64_t n)
{
  if (which < 0 || which >= MI_V_COUNT || n != c->n)
    return fail(c, MI_EINVAL, "bad vector id or length");
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipMemcpy(c->vec(which), host, size_t(n) * sizeof(double), hipMemcpyHostToDevice));
  return MI_OK;
}

int mi_matrix_get_csr(mi_ctx *c, int64_t *rowptr, int32_t *col, double *val)
{
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  const int           D = c->dim, DD = D * D;
  const mi::HostMesh &m = c->mesh;
  std::vector<double> bv(size_t(m.nnzb) * DD);
  HIPCHK(c, hipMemcpy(bv.data(), c->d_vals, bv.size() * sizeof(double), hipMemcpyDeviceToHost));
  int64_t k = 0;
  for (int64_t nd = 0; nd < m.nnodes; ++nd)
    for (int i = 0; i < D; ++i)
      {
        rowptr[nd * D + i] = k;
        for (int32_t b = m.rowptr[size_t(nd)]; b < m.rowptr[size_t(nd) + 1]; ++b)
          for (int j = 0; j < D; ++j)
            {
              col[k] = m.colidx[size_t(b)] * D + j;
              val[k] = bv[size_t(b) * DD + i * D + j];
              ++k;
            }
      }
  rowptr[m.nnodes * D] = k;
  return MI_OK;
}

int mi_spmv(mi_ctx *c, const double *x_host, double *y_host)
{
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipMemcpy(c->work(W_P), x_host, size_t(c->n) * sizeof(double), hipMemcpyHostToDevice));
  enqueue_spmv(c, c->work(W_P), c->work(W_Q), nullptr, nullptr, nullptr);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipMemcpy(y_host, c->work(W_Q), size_t(c->n) * sizeof(double), hipMemcpyDeviceToHost));
  return MI_OK;
}

int mi_set_tuning(mi_ctx *c, const char *key, int value)
{
  const std::string k(key ? key : "");
  if (k == "spmv_variant" && (value == 1 || value == 3 || (value >= 11 && value <= 14)))
    c->spmv_variant = value;
  else if (k == "xcd_remap" && (value == 0 || value == 1))
    c->xcd_remap = value;
  else if (k == "sell_unroll" && value >= -2 && value <= 4 && value != 0)
    c->sell_unroll = value;
  else if (k == "spmv_grid" && value >= 1 && value <= MAX_PART)
    c->grid_spmv = value;
  else
    return fail(c, MI_EINVAL, "unknown tuning key '%s' or value %d out of range", k.c_str(), value);
  return MI_OK;
}

int mi_set_profiling(mi_ctx *c, int enable)
{
  c->profiling = enable != 0;
  return MI_OK;
}
int mi_reset_timings(mi_ctx *c)
{
  int rc = sync(c);
  std::memset(&c->timings, 0, sizeof(c->timings));
  return rc;
}
int mi_get_timings(mi_ctx *c, mi_timings *out)
{
  int rc = sync(c);
  *out   = c->timings;
  return rc;
}

int mi_bench_spmv(mi_ctx *c, int reps, double *ms_per_launch)
{
  HIPCHK(c, hipSetDevice(c->device));
  hipEvent_t a, b;
  HIPCHK(c, hipEventCreate(&a));
  HIPCHK(c, hipEventCreate(&b));
  enqueue_spmv(c, c->work(W_P), c->work(W_Q), c->work(W_P), c->part(2), nullptr); // warm-up
  HIPCHK(c, hipEventRecord(a, c->stream));
  for (int i = 0; i < reps; ++i)
    enqueue_spmv(c, c->work(W_P), c->work(W_Q), c->work(W_P), c->part(2), nullptr);
  HIPCHK(c, hipEventRecord(b, c->stream));
  HIPCHK(c, hipEventSynchronize(b));
  float ms = 0;
  HIPCHK(c, hipEventElapsedTime(&ms, a, b));
  hipEventDestroy(a);
  hipEventDestroy(b);
  *ms_per_launch = double(ms) / std::max(1, reps);
  return MI_OK;
}

int mi_bench_assemble(mi_ctx *c, int reps, double *ms_per_assembly)
{
  HIPCHK(c, hipSetDevice(c->device));
  hipEvent_t a, b;
  HIPCHK(c, hipEventCreate(&a));
  HIPCHK(c, hipEventCreate(&b));
  int rc = enqueue_assembly(c); // warm-up
  if (rc)
    return rc;
  HIPCHK(c, hipEventRecord(a, c->stream));
  for (int i = 0; i < reps; ++i)
    if ((rc = enqueue_assembly(c)))
      return rc;
  HIPCHK(c, hipEventRecord(b, c->stream));
  HIPCHK(c, hipEventSynchronize(b));
  float ms = 0;
  HIPCHK(c, hipEventElapsedTime(&ms, a, b));
  hipEventDestroy(a);
  hipEventDestroy(b);
  *ms_per_assembly = double(ms) / std::max(1, reps);
  return sync(c);
}

} // extern "C"
