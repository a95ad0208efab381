// mi_linear.cpp -- the linear elastodynamics model (ElastoDynamics, source/linear_elasticity/linear_elasticity.cc)
// on the device context.
//
// Stiffness K, mass M and the stepping matrix are constant: they are assembled ONCE, on the device
// (assemble_linear_cells, colour by colour like the tangent; linear_elasticity.cc:248-374) straight into the device
// layout of the tangent (x-line-interleaved block rows, mi_mesh.hpp); only the consistent-load operator of the interface
// (interface-sized) is put together on the host.  The per-step path (assemble_rhs :378-454, solve :525-575, update_displacement :579-586) runs on the
// device: fused vector kernels, 2 SpMVs (M v - K (theta(1-theta)dt^2 v + dt d)) and the warm-started PCG.
#include <cmath>
#include <cstring>
#include <map>

#include "mi_internal.h"

namespace mi_detail
{
  struct LinearModel
  {
    double  theta = 0.5;
    double *d_K = nullptr, *d_M = nullptr, *d_A = nullptr, *d_dinvA = nullptr, *d_body = nullptr;
    bool    body_force_enabled = false;
    bool    factored = false; // the banded Cholesky factor of the (constant) system matrix is in the context's band
    // consistent-load operator on the interface nodes (scalar CSR over interface slots), :458-521
    std::vector<int32_t> B_rowptr, B_col;
    std::vector<double>  B_val;
  };

  void linear_destroy(mi_ctx *c)
  {
    if (!c->linear)
      return;
    for (double *p : {c->linear->d_K, c->linear->d_M, c->linear->d_A, c->linear->d_dinvA, c->linear->d_body})
      if (p)
        hipFree(p);
    delete c->linear;
    c->linear = nullptr;
  }

  namespace
  {
    // Jm[i][j] = dX_i/dxi_j of the d-linear cell map
    void jacobian(int dim, const double *verts, const double *xi, double Jm[3][3])
    {
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
          Jm[i][j] = (i == j && i >= dim) ? 1.0 : 0.0;
      for (int v = 0; v < (1 << dim); ++v)
        for (int j = 0; j < dim; ++j)
          {
            double g = ((v >> j) & 1) ? 1.0 : -1.0;
            for (int d = 0; d < dim; ++d)
              if (d != j)
                g *= ((v >> d) & 1) ? xi[d] : 1.0 - xi[d];
            for (int i = 0; i < dim; ++i)
              Jm[i][j] += verts[v * dim + i] * g;
          }
    }
  } // namespace
} // namespace mi_detail

using namespace mi_detail;

extern "C" {

// K, M, stepping matrix, load operator and body force of ONE slab: the launches run over the slab's local cells
// (ghost layer included), which completes every owned row exactly as the tangent's assembly does
static int linear_setup_member(mi_ctx *c, double theta)
{
  linear_destroy(c);
  c->linear      = new LinearModel;
  LinearModel &L = *c->linear;
  L.theta        = theta;

  const mi::HostMesh &m   = c->mesh;
  const int           dim = c->dim, npc = m.npc, DD = dim * dim;
  const double        mu = c->mat.mu, nu = c->mat.nu, rho = c->mat.rho;
  const double        lambda = 2 * mu * nu / (1 - 2 * nu); // parameters.cc:189
  const double        dt     = c->nm.delta_t;

  mi::Tables1D t;
  t.build(c->degree, c->degree + 1); // quad_order = p+1 (linear_elasticity.cc:61)
  int nqf = 1;
  for (int d = 0; d < dim - 1; ++d)
    nqf *= t.nq1;

  double bn = 0;
  for (int d = 0; d < 3; ++d)
    bn += c->mat.body_force[d] * c->mat.body_force[d];
  L.body_force_enabled = std::sqrt(bn) > 1e-15; // :62

  // K, M, A = M + theta^2 dt^2 K with zero boundary values applied, and the body-force vector: on the device
  const size_t nval = std::max<size_t>(1, size_t(m.nvalblocks()) * DD);
  double      *d_tab = nullptr;
  {
    const std::vector<double> tab = t.packed();
    HIPCHK(c, hipMalloc((void **)&d_tab, tab.size() * sizeof(double)));
    HIPCHK(c, hipMemcpy(d_tab, tab.data(), tab.size() * sizeof(double), hipMemcpyHostToDevice));
  }
  for (double **pp : {&L.d_K, &L.d_M, &L.d_A})
    {
      HIPCHK(c, hipMalloc((void **)pp, nval * sizeof(double)));
      HIPCHK(c, hipMemsetAsync(*pp, 0, nval * sizeof(double), c->stream));
    }
  if (L.body_force_enabled)
    {
      HIPCHK(c, hipMalloc((void **)&L.d_body, size_t(c->n) * sizeof(double)));
      HIPCHK(c, hipMemsetAsync(L.d_body, 0, size_t(c->n) * sizeof(double), c->stream));
    }
  {
    mi::LinAsmParams p{};
    p.conn    = c->d_conn;
    p.cverts  = c->d_cverts;
    p.off     = c->d_off;
    p.rowinfo = reinterpret_cast<const int2 *>(c->d_rowinfo);
    p.cmask   = c->d_cmask;
    p.tab     = d_tab;
    p.np1     = t.np1;
    p.nq1     = t.nq1;
    p.lambda  = lambda;
    p.mu      = mu;
    p.rho     = rho;
    p.ctheta  = dt * dt * theta * theta;
    for (int d = 0; d < 3; ++d)
      p.body[d] = c->mat.body_force[d];
    p.K       = L.d_K;
    p.M       = L.d_M;
    p.A       = L.d_A;
    p.bodyvec = L.d_body;
    for (int col = 0; col < m.ncolours; ++col)
      {
        p.cell_begin = m.colour_begin[size_t(col)];
        p.cell_count = int32_t(m.colour_begin[size_t(col) + 1] - m.colour_begin[size_t(col)]);
        mi::launch_assemble_linear(dim, p, c->stream);
      }
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    hipFree(d_tab);
  }

  // consistent-load operator B_ab = int_interface N_a N_b dA (assemble_consistent_loading :458-521, no pull-back)
  {
    std::map<int32_t, int32_t> slot;
    for (size_t i = 0; i < m.iface_nodes.size(); ++i)
      slot[m.iface_nodes[i]] = int32_t(i);
    std::vector<std::map<int32_t, double>> rows(m.iface_nodes.size());
    const int np1 = t.np1, nq1 = t.nq1;
    for (const auto &fc : m.iface_faces)
      for (int f = 0; f < 2 * dim; ++f)
        {
          if (!((fc.face >> f) & 1))
            continue;
          const int     nd = f / 2, side = f & 1;
          const double *verts = &m.cverts[size_t(fc.cell) * m.nv * dim];
          int           tang[2] = {0, 0}, nt = 0;
          for (int d = 0; d < dim; ++d)
            if (d != nd)
              tang[nt++] = d;
          for (int fq = 0; fq < nqf; ++fq)
            {
              const int f1 = fq % nq1, f2 = fq / nq1;
              double    xi[3] = {0, 0, 0}, w = t.qw[f1];
              xi[nd]          = side ? 1.0 : 0.0;
              xi[tang[0]]     = t.qx[f1];
              if (dim == 3)
                {
                  xi[tang[1]] = t.qx[f2];
                  w *= t.qw[f2];
                }
              double Jm[3][3];
              jacobian(dim, verts, xi, Jm);
              double area;
              if (dim == 2)
                area = std::hypot(Jm[0][tang[0]], Jm[1][tang[0]]);
              else
                {
                  const double a0 = Jm[0][tang[0]], a1 = Jm[1][tang[0]], a2 = Jm[2][tang[0]];
                  const double b0 = Jm[0][tang[1]], b1 = Jm[1][tang[1]], b2 = Jm[2][tang[1]];
                  area = std::sqrt((a1 * b2 - a2 * b1) * (a1 * b2 - a2 * b1) + (a2 * b0 - a0 * b2) * (a2 * b0 - a0 * b2) +
                                   (a0 * b1 - a1 * b0) * (a0 * b1 - a1 * b0));
                }
              const double JxW = area * w;
              // shape values on the face: product of 1D values along the tangential axes
              std::vector<std::pair<int32_t, double>> nodes;
              for (int a = 0; a < npc; ++a)
                {
                  const int ai[3] = {a % np1, (a / np1) % np1, dim == 3 ? a / (np1 * np1) : 0};
                  if (ai[nd] != (side ? c->degree : 0))
                    continue;
                  double n = t.N[size_t(f1) * np1 + ai[tang[0]]];
                  if (dim == 3)
                    n *= t.N[size_t(f2) * np1 + ai[tang[1]]];
                  nodes.push_back({m.conn[size_t(fc.cell) * npc + a], n});
                }
              for (const auto &na : nodes)
                for (const auto &nb : nodes)
                  rows[size_t(slot[na.first])][slot[nb.first]] += na.second * nb.second * JxW;
            }
        }
    L.B_rowptr.assign(1, 0);
    for (const auto &r : rows)
      {
        for (const auto &kv : r)
          {
            L.B_col.push_back(kv.first);
            L.B_val.push_back(kv.second);
          }
        L.B_rowptr.push_back(int32_t(L.B_col.size()));
      }
  }

  // Jacobi diagonal of A
  HIPCHK(c, hipMalloc((void **)&L.d_dinvA, size_t(c->n) * sizeof(double)));
  mi::launch_extract_dinv(c->dim, L.d_A, c->d_diagpos, L.d_dinvA, m.nnodes, c->stream);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipMemsetAsync(c->d_vecs, 0, size_t(MI_V_COUNT) * size_t(c->n) * sizeof(double), c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return MI_OK;
}

int mi_linear_setup(mi_ctx *c, double theta)
{
  HIPCHK(c, hipSetDevice(c->device));
  if (!(theta >= 0.0 && theta <= 1.0))
    return fail(c, MI_EINVAL, "theta must be in [0,1]");
  for (mi_ctx *m : c->team->members)
    {
      const int rc = linear_setup_member(m, theta);
      if (rc)
        return m == c ? rc : fail(c, rc, "%s", m->err.c_str());
    }
  return MI_OK;
}

int mi_linear_step(mi_ctx *c, int data_consistent, double abs_tol, int64_t max_it, int *its, double *res)
{
  if (!c->linear)
    return fail(c, MI_EINVAL, "mi_linear_setup has not been called");
  if (!(abs_tol > 0))
    return fail(c, MI_EINVAL, "absolute tolerance must be positive");
  HIPCHK(c, hipSetDevice(c->device));
  Team      &T  = *c->team;
  mi_ctx    *c0 = T.members[0];
  const int  dim = c->dim;
  std::vector<mi::LinearParams> ps;
  for (mi_ctx *m : T.members)
    {
      LinearModel &L   = *m->linear;
      const int    nif = int(m->mesh.iface_nodes.size());
      // load at the interface dofs: face integral of the traction ("Stress", :383-384) or the nodal forces as
      // they come ("Force", :385-388); interface sized, computed on the host from the last coupling data
      // (c0->h_iface holds the GLOBAL interface array, iface_slot maps the slab's nodes into it)
      double *load = m->vec(MI_V_NEWTON_UPDATE);
      if (nif > 0)
        {
          if (c0->h_iface.size() != T.iface_global.size() * size_t(dim))
            c0->h_iface.assign(T.iface_global.size() * size_t(dim), 0.0);
          auto coupling = [&](int i, int k) { return c0->h_iface[size_t(m->iface_slot[size_t(i)]) * dim + k]; };
          double *stage = m->h_pinned + 64;
          for (int i = 0; i < nif; ++i)
            for (int k = 0; k < dim; ++k)
              {
                double s = 0;
                if (data_consistent)
                  for (int32_t e = L.B_rowptr[size_t(i)]; e < L.B_rowptr[size_t(i) + 1]; ++e)
                    s += L.B_val[size_t(e)] * coupling(L.B_col[size_t(e)], k);
                else
                  s = coupling(i, k);
                stage[i * dim + k] = s;
              }
          HIPCHK(m, hipMemcpyAsync(m->d_iface_buf, stage, size_t(nif) * dim * sizeof(double), hipMemcpyHostToDevice,
                                   m->stream));
          mi::launch_scatter_nodes(dim, load, m->d_iface_nodes, nif, m->d_iface_buf, m->stream);
        }
      mi::LinearParams p{};
      p.load  = load;
      p.body  = L.body_force_enabled ? L.d_body : nullptr;
      p.f_old = m->vec(MI_L_OLD_STRESS);
      p.v     = m->vec(MI_L_VELOCITY);
      p.d     = m->vec(MI_L_DISPLACEMENT);
      p.v_old = m->vec(MI_L_OLD_VELOCITY);
      p.d_old = m->vec(MI_L_OLD_DISPLACEMENT);
      p.rhs   = m->vec(MI_L_SYSTEM_RHS);
      p.w     = m->vec(MI_V_SOLUTION_DELTA);
      p.theta = L.theta;
      p.dt    = m->nm.delta_t;
      p.n     = m->n; // pointwise on all local dofs: the ghost copies stay consistent
      mi::launch_linear_rhs_prepare(p, m->stream);
      ps.push_back(p);
    }
  // M v_old and K w  (:411-420 with the two K products merged); ghost planes of both operands are consistent
  auto self = [](mi_ctx *m) { return m; };
  int  rc;
  for (mi_ctx *m : T.members)
    m->active_sell_vals = m->linear->d_M;
  int t = tic(c0, MI_T_SPMV);
  rc    = team_spmv(
    T, self, [](mi_ctx *m) { return m->vec(MI_L_OLD_VELOCITY); }, [](mi_ctx *m) { return m->work(W_R); }, nullptr);
  toc(c0, t);
  for (mi_ctx *m : T.members)
    m->active_sell_vals = m->linear->d_K;
  t = tic(c0, MI_T_SPMV);
  if (!rc)
    rc = team_spmv(
      T, self, [](mi_ctx *m) { return m->vec(MI_V_SOLUTION_DELTA); }, [](mi_ctx *m) { return m->work(W_Q); }, nullptr);
  toc(c0, t);
  for (size_t k = 0; k < T.members.size() && !rc; ++k)
    {
      mi_ctx *m = T.members[k];
      mi::launch_linear_rhs_finish(dim, ps[k], m->work(W_R), m->work(W_Q), m->d_cmask, m->stream);
    }
  HIPCHK(c, hipGetLastError());
  // solve (:531-551): absolute tolerance, start vector = previous velocity
  for (mi_ctx *m : T.members)
    {
      m->active_sell_vals = m->linear->d_A;
      m->active_dinv      = m->linear->d_dinvA;
    }
  bool direct = false;
  if (!rc && c0->solver_direct && T.size == 1 && direct_prepare(c0) == MI_OK)
    {
      // "Solver type = Direct" (:553-559): the system matrix is constant, so it is factorised ONCE and a step is one
      // forward and one backward substitution
      direct          = true;
      LinearModel &Lm = *c0->linear;
      rc = direct_factor_solve(c0, Lm.d_A, c0->vec(MI_L_SYSTEM_RHS), c0->vec(MI_L_VELOCITY), !Lm.factored, true);
      Lm.factored = Lm.factored || rc == MI_OK;
      if (its)
        *its = 1;
      if (res)
        *res = 0.0;
    }
  if (!rc && !direct)
    rc = cg_run(c0, MI_L_VELOCITY, MI_L_SYSTEM_RHS, -abs_tol, max_it, its, res);
  for (mi_ctx *m : T.members)
    {
      m->active_sell_vals = nullptr;
      m->active_dinv      = nullptr;
    }
  if (rc)
    return rc;
  for (size_t k = 0; k < T.members.size(); ++k)
    mi::launch_linear_update_displacement(ps[k], T.members[k]->stream); // :579-586
  HIPCHK(c, hipGetLastError());
  return sync(c0);
}

int mi_linear_matrix_get_csr(mi_ctx *c, int which, int64_t *rowptr, int32_t *col, double *val)
{
  if (!c->linear || which < 0 || which > 2)
    return fail(c, MI_EINVAL, "linear model not set up or bad matrix id");
  if (team_size(c) != 1)
    return fail(c, MI_EINVAL, "matrix export is only available on an undecomposed mesh");
  const int           D = c->dim, DD = D * D;
  const mi::HostMesh &m = c->mesh;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  std::vector<double> bv(size_t(m.nvalblocks()) * DD);
  HIPCHK(c, hipMemcpy(bv.data(), which == 0 ? c->linear->d_K : which == 1 ? c->linear->d_M : c->linear->d_A,
                      bv.size() * sizeof(double), hipMemcpyDeviceToHost));
  int64_t k = 0;
  for (int64_t nd = 0; nd < m.nnodes; ++nd)
    for (int i = 0; i < D; ++i)
      {
        rowptr[nd * D + i] = k;
        for (int32_t b = m.rowptr[size_t(nd)]; b < m.rowptr[size_t(nd) + 1]; ++b)
          for (int j = 0; j < D; ++j)
            {
              col[k] = m.colidx[size_t(b)] * D + j;
              val[k] = bv[size_t(m.valpos(nd, int(b - m.rowptr[size_t(nd)]))) * DD + i * D + j];
              ++k;
            }
      }
  rowptr[m.nnodes * D] = k;
  return MI_OK;
}

} // extern "C"
