// mi_linear.cpp -- the linear elastodynamics model (ElastoDynamics, source/linear_elasticity/linear_elasticity.cc)
// on the device context.
//
// Stiffness K and mass M are constant: they are assembled ONCE on the host (linear_elasticity.cc:248-374, the
// reference does the same on the CPU) into the block pattern and uploaded in the device layout of the tangent
// (slice-interleaved block rows, mi_mesh.hpp).  The per-step path (assemble_rhs :378-454, solve :525-575, update_displacement :579-586) runs on the
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
    std::vector<double>  hK, hM, hA; // host block-CSR copies (parity tests)
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
    struct CellGeom
    {
      std::vector<double> G;   // [nq][npc][dim] real-space gradients
      std::vector<double> N;   // [nq][npc]
      std::vector<double> JxW; // [nq]
    };

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
    double det3(const double A[3][3])
    {
      return A[0][0] * (A[1][1] * A[2][2] - A[1][2] * A[2][1]) - A[0][1] * (A[1][0] * A[2][2] - A[1][2] * A[2][0]) +
             A[0][2] * (A[1][0] * A[2][1] - A[1][1] * A[2][0]);
    }
    void inv3(const double A[3][3], double B[3][3])
    {
      const double r = 1.0 / det3(A);
      B[0][0]        = (A[1][1] * A[2][2] - A[1][2] * A[2][1]) * r;
      B[0][1]        = (A[0][2] * A[2][1] - A[0][1] * A[2][2]) * r;
      B[0][2]        = (A[0][1] * A[1][2] - A[0][2] * A[1][1]) * r;
      B[1][0]        = (A[1][2] * A[2][0] - A[1][0] * A[2][2]) * r;
      B[1][1]        = (A[0][0] * A[2][2] - A[0][2] * A[2][0]) * r;
      B[1][2]        = (A[0][2] * A[1][0] - A[0][0] * A[1][2]) * r;
      B[2][0]        = (A[1][0] * A[2][1] - A[1][1] * A[2][0]) * r;
      B[2][1]        = (A[0][1] * A[2][0] - A[0][0] * A[2][1]) * r;
      B[2][2]        = (A[0][0] * A[1][1] - A[0][1] * A[1][0]) * r;
    }

    void cell_geometry(const mi::Tables1D &t, int dim, const double *verts, CellGeom &g)
    {
      const int np1 = t.np1, nq1 = t.nq1;
      int       npc = 1, nq = 1;
      for (int d = 0; d < dim; ++d)
        {
          npc *= np1;
          nq *= nq1;
        }
      g.G.assign(size_t(nq) * npc * dim, 0.0);
      g.N.assign(size_t(nq) * npc, 0.0);
      g.JxW.assign(nq, 0.0);
      for (int q = 0; q < nq; ++q)
        {
          const int qi[3] = {q % nq1, (q / nq1) % nq1, dim == 3 ? q / (nq1 * nq1) : 0};
          double    xi[3] = {0, 0, 0}, w = 1.0;
          for (int d = 0; d < dim; ++d)
            {
              xi[d] = t.qx[qi[d]];
              w *= t.qw[qi[d]];
            }
          double Jm[3][3], Ji[3][3];
          jacobian(dim, verts, xi, Jm);
          inv3(Jm, Ji);
          g.JxW[q] = det3(Jm) * w;
          for (int a = 0; a < npc; ++a)
            {
              const int ai[3] = {a % np1, (a / np1) % np1, dim == 3 ? a / (np1 * np1) : 0};
              double    n = 1.0, dn[3] = {0, 0, 0};
              for (int d = 0; d < dim; ++d)
                n *= t.N[size_t(qi[d]) * np1 + ai[d]];
              for (int k = 0; k < dim; ++k)
                {
                  double v = 1.0;
                  for (int d = 0; d < dim; ++d)
                    v *= (d == k) ? t.dN[size_t(qi[d]) * np1 + ai[d]] : t.N[size_t(qi[d]) * np1 + ai[d]];
                  dn[k] = v;
                }
              g.N[size_t(q) * npc + a] = n;
              for (int i = 0; i < dim; ++i)
                {
                  double s = 0;
                  for (int j = 0; j < dim; ++j)
                    s += dn[j] * Ji[j][i];
                  g.G[(size_t(q) * npc + a) * dim + i] = s;
                }
            }
        }
    }

    // host block-CSR values -> the device layout of the tangent (slice-interleaved block rows, owned rows only)
    int to_device_sell(mi_ctx *c, const std::vector<double> &bsr, double **d_sell)
    {
      const mi::HostMesh &m  = c->mesh;
      const size_t        dd = size_t(c->dim) * c->dim;
      std::vector<double> v(std::max<size_t>(1, size_t(m.nvalblocks()) * dd), 0.0);
      for (int64_t nd = 0; nd < m.nnodes; ++nd)
        if (m.rowinfo[2 * size_t(nd)] >= 0)
          for (int32_t b = m.rowptr[size_t(nd)]; b < m.rowptr[size_t(nd) + 1]; ++b)
            std::memcpy(&v[size_t(m.valpos(nd, int(b - m.rowptr[size_t(nd)]))) * dd], &bsr[size_t(b) * dd], dd * sizeof(double));
      HIPCHK(c, hipMalloc((void **)d_sell, v.size() * sizeof(double)));
      HIPCHK(c, hipMemcpy(*d_sell, v.data(), v.size() * sizeof(double), hipMemcpyHostToDevice));
      return MI_OK;
    }
  } // namespace
} // namespace mi_detail

using namespace mi_detail;

extern "C" {

// K, M, stepping matrix, load operator and body force of ONE slab: the host loops run over the slab's local cells
// (ghost layer included), which completes every owned row exactly as the device assembly does
static int linear_setup_member(mi_ctx *c, double theta)
{
  linear_destroy(c);
  c->linear      = new LinearModel;
  LinearModel &L = *c->linear;
  L.theta        = theta;

  const mi::HostMesh &m   = c->mesh;
  const int           dim = c->dim, npc = m.npc, dpc = npc * dim, DD = dim * dim;
  const double        mu = c->mat.mu, nu = c->mat.nu, rho = c->mat.rho;
  const double        lambda = 2 * mu * nu / (1 - 2 * nu); // parameters.cc:189
  const double        dt     = c->nm.delta_t;

  mi::Tables1D t;
  t.build(c->degree, c->degree + 1); // quad_order = p+1 (linear_elasticity.cc:61)
  int nq = 1, nqf = 1;
  for (int d = 0; d < dim; ++d)
    {
      nq *= t.nq1;
      if (d < dim - 1)
        nqf *= t.nq1;
    }

  double bn = 0;
  for (int d = 0; d < 3; ++d)
    bn += c->mat.body_force[d] * c->mat.body_force[d];
  L.body_force_enabled = std::sqrt(bn) > 1e-15; // :62

  L.hK.assign(size_t(m.nnzb) * DD, 0.0);
  L.hM.assign(size_t(m.nnzb) * DD, 0.0);
  std::vector<double> body(size_t(c->n), 0.0);
  CellGeom            g;
  std::vector<double> Ke(size_t(dpc) * dpc), Me(size_t(npc) * npc);
  for (int64_t cell = 0; cell < m.ncells; ++cell)
    {
      cell_geometry(t, dim, &m.cverts[size_t(cell) * m.nv * dim], g);
      std::fill(Ke.begin(), Ke.end(), 0.0);
      std::fill(Me.begin(), Me.end(), 0.0);
      for (int q = 0; q < nq; ++q)
        for (int a = 0; a < npc; ++a)
          {
            const double *ga = &g.G[(size_t(q) * npc + a) * dim];
            for (int b = 0; b < npc; ++b)
              {
                const double *gb = &g.G[(size_t(q) * npc + b) * dim];
                double        gg = 0;
                for (int k = 0; k < dim; ++k)
                  gg += ga[k] * gb[k];
                // :301-320  lambda d_ci N_i d_cj N_j + mu d_cj N_i d_ci N_j + delta mu grad N_i . grad N_j
                for (int ci = 0; ci < dim; ++ci)
                  for (int cj = 0; cj < dim; ++cj)
                    Ke[size_t(a * dim + ci) * dpc + b * dim + cj] +=
                      (ga[ci] * gb[cj] * lambda + ga[cj] * gb[ci] * mu + (ci == cj ? gg * mu : 0.0)) * g.JxW[q];
                // create_mass_matrix with coefficient rho (:341-345)
                Me[size_t(a) * npc + b] += rho * g.N[size_t(q) * npc + a] * g.N[size_t(q) * npc + b] * g.JxW[q];
              }
            if (L.body_force_enabled) // create_right_hand_side with rho*b (:358-373)
              for (int ci = 0; ci < dim; ++ci)
                body[size_t(m.conn[size_t(cell) * npc + a]) * dim + ci] +=
                  rho * c->mat.body_force[ci] * g.N[size_t(q) * npc + a] * g.JxW[q];
          }
      const uint16_t *off = &m.off[size_t(cell) * npc * npc];
      for (int a = 0; a < npc; ++a)
        {
          const int32_t A = m.conn[size_t(cell) * npc + a];
          for (int b = 0; b < npc; ++b)
            {
              // off = g << 4 | kx (bit 15: first-touch flag): slot k = g * wx + kx of the block row of A (mi_mesh.hpp)
              const size_t blk = size_t(m.rowptr[size_t(A)]) + ((off[a * npc + b] >> 4) & 0x7ff) * m.rowwx[size_t(A)] + (off[a * npc + b] & 15);
              for (int ci = 0; ci < dim; ++ci)
                {
                  for (int cj = 0; cj < dim; ++cj)
                    L.hK[blk * DD + ci * dim + cj] += Ke[size_t(a * dim + ci) * dpc + b * dim + cj];
                  L.hM[blk * DD + ci * dim + ci] += Me[size_t(a) * npc + b];
                }
            }
        }
    }
  // stepping matrix M + theta^2 dt^2 K (:348-353) with zero boundary values applied (:426-451,
  // MatrixTools::apply_boundary_values: row and column eliminated, diagonal kept)
  L.hA.resize(L.hK.size());
  for (size_t k = 0; k < L.hA.size(); ++k)
    L.hA[k] = L.hK[k] * (dt * dt * theta * theta) + L.hM[k];
  for (int64_t A = 0; A < m.nnodes; ++A)
    for (int32_t blk = m.rowptr[size_t(A)]; blk < m.rowptr[size_t(A) + 1]; ++blk)
      {
        const int32_t B = m.colidx[size_t(blk)];
        for (int i = 0; i < dim; ++i)
          for (int j = 0; j < dim; ++j)
            if ((((m.cmask[size_t(A)] >> i) | (m.cmask[size_t(B)] >> j)) & 1) && !(A == B && i == j))
              L.hA[size_t(blk) * DD + i * dim + j] = 0.0;
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

  // device copies: sliced-ELL K, M, A; Jacobi diagonal of A; body-force vector
  int rc;
  if ((rc = to_device_sell(c, L.hK, &L.d_K)) || (rc = to_device_sell(c, L.hM, &L.d_M)) ||
      (rc = to_device_sell(c, L.hA, &L.d_A)))
    return rc;
  HIPCHK(c, hipMalloc((void **)&L.d_dinvA, size_t(c->n) * sizeof(double)));
  mi::launch_extract_dinv(c->dim, L.d_A, c->d_diagpos, L.d_dinvA, m.nnodes, c->stream);
  HIPCHK(c, hipGetLastError());
  if (L.body_force_enabled)
    {
      HIPCHK(c, hipMalloc((void **)&L.d_body, size_t(c->n) * sizeof(double)));
      HIPCHK(c, hipMemcpy(L.d_body, body.data(), body.size() * sizeof(double), hipMemcpyHostToDevice));
    }
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
  const std::vector<double> &bv = which == 0 ? c->linear->hK : which == 1 ? c->linear->hM : c->linear->hA;
  const int                  D = c->dim, DD = D * D;
  const mi::HostMesh        &m = c->mesh;
  int64_t                    k = 0;
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

} // extern "C"
