// mi_mg.cpp -- geometric multigrid V-cycle used as the CG preconditioner of a slab (the
// reference preconditions with SSOR, which is sequential; Jacobi needs ~450 iterations per solve at 5 M DoFs).
//
// Everything is built from pieces that already exist on the device:
//   * levels are ordinary device contexts on coarser lattices of the same box: first p-coarsening to Q1 on the
//     same cells, then index-space coarsening of the cells by 2 down to a single cell;
//   * coarse operators are RE-ASSEMBLED by the same element kernel at the current state (displacement
//     interpolated to the level), so no sparse triple products are needed and the Newmark mass term, the
//     Dirichlet rows and the material are treated exactly as on the fine level;
//   * smoothers are Chebyshev-Jacobi polynomials on the level's sliced-ELL SpMV, with the largest eigenvalue of
//     D^-1 A from a device-side power iteration; the coarsest level (one cell) is "solved" by a longer polynomial;
//   * transfers are tensor-product linear interpolation in lattice index space and its transpose.
// The cycle is symmetric (same polynomial before and after the coarse correction), so it is a valid CG
// preconditioner.  On a decomposed mesh every slab runs the cycle on its own local box (owned residual in, owned
// correction out): a block preconditioner without any extra communication.
#include <cmath>
#include <cstring>

#include "mi_internal.h"

namespace mi_detail
{
  struct MgTransfer
  {
    mi::LatticeParams prolong{}, restrict_{}, state{};
    std::vector<void *> dev; // device tables to free
  };

  struct MgLevel
  {
    mi_ctx *ctx  = nullptr; // level 0: the slab itself (not owned)
    Team   *team = nullptr; // levels >= 1 own a private team (shared stream)
    double *ws   = nullptr; // workspace: b, x, d, q, ev  (5 local vectors; ev = running eigenvector estimate)
    bool    ev_ready = false;
    double  lmax = 0.0;     // estimate of the largest eigenvalue of D^-1 A
    MgTransfer to_coarse;   // to level l+1
    double *b() const { return ws; }
    double *x() const { return ws + ctx->n; }
    double *d() const { return ws + 2 * ctx->n; }
    double *q() const { return ws + 3 * ctx->n; }
    double *ev() const { return ws + 4 * ctx->n; }
  };

  struct Multigrid
  {
    std::vector<MgLevel> levels;
    int    nu            = 2;    // Chebyshev degree of the pre- and post-smoother
    double smooth_ratio  = 20.0; // smoother targets [lmax/ratio, lmax]
    int    coarse_degree = 12;   // polynomial degree on the coarsest level
    double coarse_ratio  = 60.0;
    int    power_its     = 15;   // first estimate
    int    power_its_update = 4; // refresh, continuing from the previous eigenvector
    double lmax_safety   = 1.15;
  };

  bool mg_active(const mi_ctx *c)
  {
    return c->precond == 1 && c->mg && c->mg->levels.size() > 1 && !c->active_sell_vals;
  }

  namespace
  {
    template <typename T>
    int to_device(mi_ctx *c, MgTransfer &t, const std::vector<T> &h, const T **out)
    {
      T *d = nullptr;
      HIPCHK(c, hipMalloc((void **)&d, std::max<size_t>(1, h.size()) * sizeof(T)));
      if (!h.empty())
        HIPCHK(c, hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
      t.dev.push_back(d);
      *out = d;
      return MI_OK;
    }

    // 1D linear interpolation table: for every target index t of a lattice with nt points, the source index
    // i0 and the weight w of i0+1 on a lattice with ns points covering the same interval
    void interp_table(int nt, int ns, std::vector<int32_t> &i0, std::vector<double> &w)
    {
      i0.resize(size_t(nt));
      w.resize(size_t(nt));
      for (int t = 0; t < nt; ++t)
        {
          const double s = (nt > 1) ? double(t) * double(ns - 1) / double(nt - 1) : 0.0;
          int          i = int(std::floor(s + 1e-12));
          if (i > ns - 2)
            i = std::max(0, ns - 2);
          double wt = s - i;
          if (std::fabs(wt) < 1e-12)
            wt = 0.0;
          if (std::fabs(wt - 1.0) < 1e-12)
            wt = 1.0;
          if (ns == 1)
            {
              i  = 0;
              wt = 0.0;
            }
          i0[size_t(t)] = i;
          w[size_t(t)]  = wt;
        }
    }

    int build_transfer(mi_ctx *fine, mi_ctx *coarse, MgTransfer &t)
    {
      const int dim = fine->dim;
      int       rc;
      t.prolong.n_tgt  = fine->mesh.nnodes;
      t.restrict_.n_tgt = coarse->mesh.nnodes;
      t.state.n_tgt    = coarse->mesh.nnodes;
      for (int d = 0; d < 3; ++d)
        {
          const int nf = d < dim ? fine->mesh.nn[d] : 1, nc = d < dim ? coarse->mesh.nn[d] : 1;
          t.prolong.nt[d] = nf;
          t.prolong.ns[d] = nc;
          t.restrict_.nt[d] = nc;
          t.restrict_.ns[d] = nf;
          t.state.nt[d] = nc;
          t.state.ns[d] = nf;
          if (d >= dim)
            continue;
          std::vector<int32_t> i0;
          std::vector<double>  w;
          interp_table(nf, nc, i0, w); // prolongation: fine target <- coarse source
          if ((rc = to_device(fine, t, i0, &t.prolong.i0[d])) || (rc = to_device(fine, t, w, &t.prolong.w[d])))
            return rc;
          // restriction = transpose: for every coarse index the fine indices whose stencil touches it
          std::vector<std::vector<std::pair<int32_t, double>>> lists((size_t)nc);
          for (int f = 0; f < nf; ++f)
            {
              if (1.0 - w[size_t(f)] != 0.0)
                lists[size_t(i0[size_t(f)])].push_back({f, 1.0 - w[size_t(f)]});
              if (w[size_t(f)] != 0.0)
                lists[size_t(i0[size_t(f)] + 1)].push_back({f, w[size_t(f)]});
            }
          std::vector<int32_t> rs(1, 0), ri;
          std::vector<double>  rw;
          for (const auto &l : lists)
            {
              for (const auto &e : l)
                {
                  ri.push_back(e.first);
                  rw.push_back(e.second);
                }
              rs.push_back(int32_t(ri.size()));
            }
          if ((rc = to_device(fine, t, rs, &t.restrict_.rstart[d])) || (rc = to_device(fine, t, ri, &t.restrict_.ri[d])) ||
              (rc = to_device(fine, t, rw, &t.restrict_.rw[d])))
            return rc;
          interp_table(nc, nf, i0, w); // state transfer: coarse target <- fine source
          if ((rc = to_device(fine, t, i0, &t.state.i0[d])) || (rc = to_device(fine, t, w, &t.state.w[d])))
            return rc;
        }
      return MI_OK;
    }

    // power iteration for lambda_max(D^-1 A) of a level
    int estimate_lmax(Multigrid &mg, MgLevel &L)
    {
      mi_ctx       *c = L.ctx;
      const int64_t n = c->n;
      double       *v = L.ev(), *q = L.q(), *w = L.d();
      int           its = mg.power_its_update;
      if (!L.ev_ready)
        {
          // deterministic start vector with all frequencies: v_i = 1 + 0.5 sin(i) on unconstrained dofs; later
          // updates continue from the previous estimate of the dominant eigenvector
          std::vector<double> h((size_t)n, 0.0);
          double              nrm = 0;
          for (int64_t i = 0; i < n; ++i)
            {
              h[size_t(i)] =
                ((c->mesh.cmask[size_t(i / c->dim)] >> int(i % c->dim)) & 1) ? 0.0 : 1.0 + 0.5 * std::sin(double(i));
              nrm += h[size_t(i)] * h[size_t(i)];
            }
          nrm = std::sqrt(nrm);
          for (double &x : h)
            x /= (nrm > 0 ? nrm : 1.0);
          HIPCHK(c, hipMemcpyAsync(v, h.data(), size_t(n) * sizeof(double), hipMemcpyHostToDevice, c->stream));
          HIPCHK(c, hipStreamSynchronize(c->stream)); // h goes out of scope
          L.ev_ready = true;
          its        = mg.power_its;
        }
      double lam = 0.0;
      for (int it = 0; it < its; ++it)
        {
          enqueue_spmv(c, v, q, nullptr, nullptr, nullptr);
          mi::launch_vec_scale_mul(w, q, c->work(W_DINV), 1.0, n, c->stream); // w = D^-1 A v
          mi::launch_masked_norm(c->dim, w, c->d_cmask, n, c->part(5), c->grid_vec, c->d_sc + 14, c->stream);
          HIPCHK(c, hipMemcpyAsync(c->h_pinned, c->d_sc + 14, sizeof(double), hipMemcpyDeviceToHost, c->stream));
          HIPCHK(c, hipStreamSynchronize(c->stream));
          lam = std::sqrt(c->h_pinned[0]); // |D^-1 A v| with |v| = 1
          if (!(lam > 0.0) || !std::isfinite(lam))
            return fail(c, MI_EINVAL, "multigrid: power iteration broke down on a level with %lld dofs", (long long)n);
          mi::launch_vec_scale_mul(v, w, nullptr, 1.0 / lam, n, c->stream);
        }
      HIPCHK(c, hipGetLastError());
      L.lmax = lam * mg.lmax_safety;
      return MI_OK;
    }
  } // namespace

  void mg_destroy(mi_ctx *c)
  {
    if (!c->mg)
      return;
    for (size_t l = 0; l < c->mg->levels.size(); ++l)
      {
        MgLevel &L = c->mg->levels[l];
        for (void *p : L.to_coarse.dev)
          hipFree(p);
        if (L.ws)
          hipFree(L.ws);
        if (l > 0 && L.team)
          destroy_team(L.team); // destroys the level context; the stream is shared and stays
      }
    delete c->mg;
    c->mg = nullptr;
  }

  int mg_setup(mi_ctx *c)
  {
    mg_destroy(c);
    Multigrid *mg = new Multigrid;
    c->mg         = mg;
    if (const char *e = getenv("MI_MG_NU"))
      mg->nu = std::max(1, atoi(e));
    if (const char *e = getenv("MI_MG_RATIO"))
      mg->smooth_ratio = std::max(2.0, atof(e));
    MgLevel L0;
    L0.ctx = c;
    mg->levels.push_back(L0);
    // level hierarchy over the slab's local box
    int       p = c->degree, reps[3] = {c->mesh.reps[0], c->mesh.reps[1], c->mesh.reps[2]};
    const int dim = c->dim;
    for (int guard = 0; guard < 24; ++guard)
      {
        if (p > 1)
          p = 1;
        else
          {
            bool changed = false;
            for (int d = 0; d < dim; ++d)
              if (reps[d] > 1)
                {
                  reps[d] = (reps[d] + 1) / 2;
                  changed = true;
                }
            if (!changed)
              break;
          }
        mi_mesh_desc md{};
        md.dim    = dim;
        md.degree = p;
        for (int d = 0; d < 3; ++d)
          {
            md.reps[d] = d < dim ? reps[d] : 1;
            md.lo[d]   = c->slab.local_lo[d];
            md.hi[d]   = c->slab.local_hi[d];
          }
        for (int f = 0; f < 6; ++f)
          md.face_role[f] = c->slab.local_face_role[f];
        Team *T        = new Team;
        T->size        = 1;
        T->device      = c->device;
        T->dim         = dim;
        T->stream      = c->stream;
        T->owns_stream = false;
        T->iface_global = mi::global_interface_nodes(dim, p, md.reps, md.face_role);
        mi_ctx   *lc   = nullptr;
        const int rc   = create_member(*T, &md, &c->mat, &c->nm, 0, &lc);
        T->members.push_back(lc);
        if (rc != MI_OK)
          {
            c->err = lc ? lc->err : "multigrid level creation failed";
            destroy_team(T);
            return rc;
          }
        lc->precond = 0; // levels are never preconditioned themselves
        MgLevel L;
        L.ctx  = lc;
        L.team = T;
        mg->levels.push_back(L);
      }
    for (size_t l = 0; l < mg->levels.size(); ++l)
      {
        MgLevel &L = mg->levels[l];
        HIPCHK(c, hipMalloc((void **)&L.ws, size_t(5) * size_t(L.ctx->n) * sizeof(double)));
        HIPCHK(c, hipMemsetAsync(L.ws, 0, size_t(5) * size_t(L.ctx->n) * sizeof(double), c->stream));
        if (l + 1 < mg->levels.size())
          {
            const int rc = build_transfer(L.ctx, mg->levels[l + 1].ctx, L.to_coarse);
            if (rc)
              return rc;
          }
      }
    c->mg_stale = true;
    return MI_OK;
  }

  // coarse operators for the current state of the slab: u_total interpolated down the hierarchy, every level
  // re-assembled, eigenvalue estimates refreshed
  int mg_update(mi_ctx *c)
  {
    if (!c->mg || c->mg->levels.size() < 2)
      return MI_OK;
    Multigrid &mg = *c->mg;
    int        rc;
    // level 0: u_total = u + du into the workspace, then down
    MgLevel &L0 = mg.levels[0];
    HIPCHK(c, hipMemcpyAsync(L0.x(), c->vec(MI_V_TOTAL_DISPLACEMENT), size_t(c->n) * sizeof(double),
                             hipMemcpyDeviceToDevice, c->stream));
    mi::launch_vec_add(L0.x(), c->vec(MI_V_SOLUTION_DELTA), c->n, c->stream);
    const double *src = L0.x();
    for (size_t l = 0; l + 1 < mg.levels.size(); ++l)
      {
        MgLevel &F = mg.levels[l], &C = mg.levels[l + 1];
        mi::launch_lattice_interp(c->dim, false, F.to_coarse.state, C.ctx->vec(MI_V_TOTAL_DISPLACEMENT), src,
                                  C.ctx->d_cmask, c->stream);
        if ((rc = enqueue_assembly(C.ctx)))
          {
            c->err = C.ctx->err;
            return rc;
          }
        src = C.ctx->vec(MI_V_TOTAL_DISPLACEMENT);
      }
    HIPCHK(c, hipGetLastError());
    for (MgLevel &L : mg.levels)
      if ((rc = estimate_lmax(mg, L)))
        {
          c->err = L.ctx->err;
          return rc;
        }
    c->mg_stale = false;
    return MI_OK;
  }

  namespace
  {
    // k Chebyshev-Jacobi steps on level L for A x = b over [lmax/ratio, lmax]; zero_start: x = 0 on entry
    void chebyshev(MgLevel &L, int k, double ratio, bool zero_start)
    {
      mi_ctx      *c = L.ctx;
      const double b = L.lmax, a = L.lmax / ratio;
      const double theta = 0.5 * (b + a), delta = 0.5 * (b - a), sigma = theta / delta;
      double       rho_old = 1.0 / sigma;
      for (int j = 0; j < k; ++j)
        {
          const bool first = (j == 0);
          if (!(first && zero_start))
            enqueue_spmv(c, L.x(), L.q(), nullptr, nullptr, nullptr);
          double c1, c2;
          if (first)
            {
              c1 = 0.0;
              c2 = 1.0 / theta;
            }
          else
            {
              const double rho = 1.0 / (2.0 * sigma - rho_old);
              c1               = rho * rho_old;
              c2               = 2.0 * rho / delta;
              rho_old          = rho;
            }
          mi::launch_cheb_step(L.x(), L.d(), L.b(), (first && zero_start) ? nullptr : L.q(), c->work(W_DINV), c1, c2, c->n,
                               c->stream);
        }
    }

    void vcycle(Multigrid &mg, size_t l)
    {
      MgLevel &L = mg.levels[l];
      mi_ctx  *c = L.ctx;
      if (l + 1 == mg.levels.size())
        {
          chebyshev(L, mg.coarse_degree, mg.coarse_ratio, true);
          return;
        }
      MgLevel &C = mg.levels[l + 1];
      chebyshev(L, mg.nu, mg.smooth_ratio, true);
      enqueue_spmv(c, L.x(), L.q(), nullptr, nullptr, nullptr);
      mi::launch_vec_residual(L.q(), L.b(), L.q(), c->n, c->stream); // q = b - A x
      mi::launch_lattice_restrict(c->dim, L.to_coarse.restrict_, C.b(), L.q(), C.ctx->d_cmask, c->stream);
      vcycle(mg, l + 1);
      mi::launch_lattice_interp(c->dim, true, L.to_coarse.prolong, L.x(), C.x(), c->d_cmask, c->stream);
      chebyshev(L, mg.nu, mg.smooth_ratio, false);
    }
  } // namespace

  int mg_apply(mi_ctx *c, const double *r, double *z)
  {
    Multigrid &mg = *c->mg;
    MgLevel   &L0 = mg.levels[0];
    mi::launch_copy_owned(L0.b(), r, c->n, c->own0, c->own_n, c->stream);
    vcycle(mg, 0);
    HIPCHK(c, hipMemcpyAsync(z, L0.x(), size_t(c->n) * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipGetLastError());
    return MI_OK;
  }
} // namespace mi_detail
