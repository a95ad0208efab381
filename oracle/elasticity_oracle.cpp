/*
 * elasticity_oracle.cpp -- CPU restatement of the hot path of precice/dealii-adapter.
 *
 * TEST INFRASTRUCTURE ONLY (see elasticity_oracle.h).  PARITY UNPINNED: no reference
 * golden vectors exist and deal.II/preCICE are not available; every function cites the
 * reference lines it follows and the deal.II conventions it re-derives are tagged [DEAL.II].
 *
 * The element tangent is computed "as written" in the reference: full 4th-order tensor Jc,
 * symmetric gradients and the i/j loop over the lower triangle.  The product computes a
 * closed-form node-pair block instead, so agreement of the two is a genuine cross-check.
 */
#include "elasticity_oracle.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace
{
  using std::vector;
  int g_threads = 1;

  double now()
  {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
  }

  // ------------------------------------------------------------------ 1D rules / bases
  // [DEAL.II] QGauss<1>(n): n-point Gauss-Legendre mapped to [0,1].
  void gauss_01(int n, double *x, double *w)
  {
    for (int i = 0; i < n; ++i)
      {
        double z = std::cos(M_PI * (i + 0.75) / (n + 0.5));
        double pp = 0;
        for (int it = 0; it < 100; ++it)
          {
            double p1 = 1.0, p2 = 0.0;
            for (int j = 0; j < n; ++j)
              {
                double p3 = p2;
                p2        = p1;
                p1        = ((2.0 * j + 1.0) * z * p2 - j * p3) / (j + 1.0);
              }
            pp        = n * (z * p1 - p2) / (z * z - 1.0);
            double dz = p1 / pp;
            z -= dz;
            if (std::fabs(dz) < 1e-16)
              break;
          }
        // ascending order on [0,1]
        x[n - 1 - i] = 0.5 * (z + 1.0);
        w[n - 1 - i] = 1.0 / ((1.0 - z * z) * pp * pp);
      }
    // symmetrise
    for (int i = 0; i < n / 2; ++i)
      {
        double xm    = 0.5 * (x[i] + (1.0 - x[n - 1 - i]));
        x[i]         = xm;
        x[n - 1 - i] = 1.0 - xm;
        double wm    = 0.5 * (w[i] + w[n - 1 - i]);
        w[i] = w[n - 1 - i] = wm;
      }
    if (n % 2)
      x[n / 2] = 0.5;
  }

  // Legendre P_n and derivative
  void legendre(int n, double z, double &p, double &dp)
  {
    double p1 = 1.0, p2 = 0.0;
    for (int j = 0; j < n; ++j)
      {
        double p3 = p2;
        p2        = p1;
        p1        = ((2.0 * j + 1.0) * z * p2 - j * p3) / (j + 1.0);
      }
    p  = p1;
    dp = n * (z * p1 - p2) / (z * z - 1.0);
  }

  // [DEAL.II] FE_Q(p) unit support points: equidistant for p<=2, Gauss-Lobatto for p>=3.
  void feq_support_1d(int p, double *x)
  {
    if (p <= 2)
      {
        for (int i = 0; i <= p; ++i)
          x[i] = double(i) / p;
        return;
      }
    const int n = p + 1; // GL points: roots of P'_{n-1} plus endpoints
    x[0]        = 0.0;
    x[n - 1]    = 1.0;
    for (int i = 1; i < n - 1; ++i)
      {
        double z = -std::cos(M_PI * i / (n - 1)); // Chebyshev-Lobatto guess
        for (int it = 0; it < 100; ++it)
          {
            // Newton on q(z) = P'_{n-1}(z); q' from Legendre ODE
            double P, dP;
            legendre(n - 1, z, P, dP);
            double d2P = (2.0 * z * dP - (n - 1) * n * P) / (1.0 - z * z);
            double dz  = dP / d2P;
            z -= dz;
            if (std::fabs(dz) < 1e-16)
              break;
          }
        x[i] = 0.5 * (z + 1.0);
      }
    for (int i = 0; i < n / 2; ++i)
      {
        double xm    = 0.5 * (x[i] + (1.0 - x[n - 1 - i]));
        x[i]         = xm;
        x[n - 1 - i] = 1.0 - xm;
      }
    if (n % 2)
      x[n / 2] = 0.5;
  }

  void lagrange_1d(const double *nodes, int np, double x, double *N, double *dN)
  {
    for (int a = 0; a < np; ++a)
      {
        double v = 1.0;
        for (int m = 0; m < np; ++m)
          if (m != a)
            v *= (x - nodes[m]) / (nodes[a] - nodes[m]);
        N[a]     = v;
        double d = 0.0;
        for (int k = 0; k < np; ++k)
          if (k != a)
            {
              double t = 1.0 / (nodes[a] - nodes[k]);
              for (int m = 0; m < np; ++m)
                if (m != a && m != k)
                  t *= (x - nodes[m]) / (nodes[a] - nodes[m]);
              d += t;
            }
        dN[a] = d;
      }
  }

  // ------------------------------------------------------------------ small tensor helpers
  double det3(const double F[3][3], int dim)
  {
    if (dim == 2)
      return F[0][0] * F[1][1] - F[0][1] * F[1][0];
    return F[0][0] * (F[1][1] * F[2][2] - F[1][2] * F[2][1]) - F[0][1] * (F[1][0] * F[2][2] - F[1][2] * F[2][0]) +
           F[0][2] * (F[1][0] * F[2][1] - F[1][1] * F[2][0]);
  }
  void inv3(const double F[3][3], int dim, double Fi[3][3])
  {
    const double d = det3(F, dim);
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j)
        Fi[i][j] = 0;
    if (dim == 2)
      {
        Fi[0][0] = F[1][1] / d;
        Fi[0][1] = -F[0][1] / d;
        Fi[1][0] = -F[1][0] / d;
        Fi[1][1] = F[0][0] / d;
        return;
      }
    Fi[0][0] = (F[1][1] * F[2][2] - F[1][2] * F[2][1]) / d;
    Fi[0][1] = (F[0][2] * F[2][1] - F[0][1] * F[2][2]) / d;
    Fi[0][2] = (F[0][1] * F[1][2] - F[0][2] * F[1][1]) / d;
    Fi[1][0] = (F[1][2] * F[2][0] - F[1][0] * F[2][2]) / d;
    Fi[1][1] = (F[0][0] * F[2][2] - F[0][2] * F[2][0]) / d;
    Fi[1][2] = (F[0][2] * F[1][0] - F[0][0] * F[1][2]) / d;
    Fi[2][0] = (F[1][0] * F[2][1] - F[1][1] * F[2][0]) / d;
    Fi[2][1] = (F[0][1] * F[2][0] - F[0][0] * F[2][1]) / d;
    Fi[2][2] = (F[0][0] * F[1][1] - F[0][1] * F[1][0]) / d;
  }

  // ------------------------------------------------------------------ material
  // compressible_neo_hook_material.h:17-138.  b_bar symmetric dim x dim.
  struct Material
  {
    double kappa, c_1, rho;
    int    dim;
    Material(int dim, double mu, double nu, double rho)
      : kappa((2.0 * mu * (1.0 + nu)) / (3.0 * (1.0 - 2.0 * nu))) // :20
      , c_1(mu / 2.0)                                             // :21
      , rho(rho)
      , dim(dim)
    {}
    double dPsi_vol_dJ(double J) const { return (kappa / 2.0) * (J - 1.0 / J); }           // :74-78
    double d2Psi_vol_dJ2(double J) const { return (kappa / 2.0) * (1.0 + 1.0 / (J * J)); } // :100-104
    double Psi(double J, const double b[3][3]) const                                        // :30-35,62-72
    {
      double tr = 0;
      for (int i = 0; i < dim; ++i)
        tr += b[i][i];
      return (kappa / 4.0) * (J * J - 1.0 - 2.0 * std::log(J)) + c_1 * (tr - dim);
    }
    void tau_bar(const double b[3][3], double tb[3][3]) const // :94-98
    {
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
          tb[i][j] = (i < dim && j < dim) ? 2.0 * c_1 * b[i][j] : 0.0;
    }
    void tau_iso(const double b[3][3], double ti[3][3]) const // :87-92  dev_P : tau_bar
    {
      double tb[3][3];
      tau_bar(b, tb);
      double tr = 0;
      for (int i = 0; i < dim; ++i)
        tr += tb[i][i];
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
          ti[i][j] = (i < dim && j < dim) ? tb[i][j] - (i == j ? tr / dim : 0.0) : 0.0;
    }
    void tau(double J, const double b[3][3], double t[3][3]) const // :37-42, 80-85
    {
      tau_iso(b, t);
      const double tv = dPsi_vol_dJ(J) * J;
      for (int i = 0; i < dim; ++i)
        t[i][i] += tv;
    }
    // :44-49, 106-138.  [DEAL.II] StandardTensors: S_ijkl = (d_ik d_jl + d_il d_jk)/2, IxI = d_ij d_kl,
    // dev_P = S - IxI/dim.
    void Jc(double J, const double b[3][3], double C[3][3][3][3]) const
    {
      double tb[3][3], ti[3][3];
      tau_bar(b, tb);
      tau_iso(b, ti);
      double trb = 0;
      for (int i = 0; i < dim; ++i)
        trb += tb[i][i];
      const double p   = dPsi_vol_dJ(J);
      const double d2  = d2Psi_vol_dJ2(J);
      const double cII = J * (p + J * d2);
      const double cS  = J * (-2.0 * p);
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
          for (int k = 0; k < 3; ++k)
            for (int l = 0; l < 3; ++l)
              {
                if (i >= dim || j >= dim || k >= dim || l >= dim)
                  {
                    C[i][j][k][l] = 0;
                    continue;
                  }
                const double IxI  = (i == j) * (k == l);
                const double S    = 0.5 * ((i == k) * (j == l) + (i == l) * (j == k));
                const double devP = S - IxI / dim;
                const double vol  = cII * IxI + cS * S;                                          // :106-114
                const double iso  = (2.0 / dim) * trb * devP -                                   // :126-127
                                   (2.0 / dim) * (ti[i][j] * (k == l) + (i == j) * ti[k][l]);    // :128
                C[i][j][k][l] = vol + iso; // c_bar == 0, :133-138
              }
    }
  };

  // ------------------------------------------------------------------ FE tables
  struct FETables
  {
    int            dim, p, np1, npc, nq1, nq, nqf;
    vector<double> nodes1, qx, qw;   // 1D
    vector<double> N1, dN1;          // [nq1][np1]
    vector<double> N, dN;            // cell: [nq][npc], [nq][npc][dim]  (unit-cell gradients)
    vector<double> W;                // [nq]
    // face tables for each of 2*dim faces: values of all cell shape fns at face QPs
    vector<double> Nf[6];            // [nqf][npc]
    vector<double> dNf[6];           // [nqf][npc][dim]   (needed only for --correct-face-F)
    vector<double> Wf;               // [nqf]
    vector<double> xif[6];           // [nqf][dim] unit-cell coords of face QPs

    void build(int dim_, int p_, int nq1_)
    {
      dim = dim_;
      p   = p_;
      np1 = p + 1;
      nq1 = nq1_;
      npc = 1;
      nq  = 1;
      nqf = 1;
      for (int d = 0; d < dim; ++d)
        {
          npc *= np1;
          nq *= nq1;
          if (d < dim - 1)
            nqf *= nq1;
        }
      nodes1.resize(np1);
      feq_support_1d(p, nodes1.data());
      qx.resize(nq1);
      qw.resize(nq1);
      gauss_01(nq1, qx.data(), qw.data());
      N1.resize(nq1 * np1);
      dN1.resize(nq1 * np1);
      for (int q = 0; q < nq1; ++q)
        lagrange_1d(nodes1.data(), np1, qx[q], &N1[q * np1], &dN1[q * np1]);
      N.resize(size_t(nq) * npc);
      dN.resize(size_t(nq) * npc * dim);
      W.resize(nq);
      // [DEAL.II] tensor-product quadrature, x fastest
      for (int q = 0; q < nq; ++q)
        {
          int    qi[3] = {q % nq1, (q / nq1) % nq1, dim == 3 ? q / (nq1 * nq1) : 0};
          double xi[3] = {qx[qi[0]], qx[qi[1]], dim == 3 ? qx[qi[2]] : 0.0};
          W[q]         = qw[qi[0]] * qw[qi[1]] * (dim == 3 ? qw[qi[2]] : 1.0);
          eval(xi, &N[size_t(q) * npc], &dN[size_t(q) * npc * dim]);
        }
      // faces.  [DEAL.II, recalled] face-local axes: x-normal faces (y,z); y-normal faces (z,x) in 3D
      // and (x) in 2D; z-normal faces (x,y).  Face QP index f = f1 + nq1*f2, f1 fastest.
      Wf.resize(nqf);
      for (int f = 0; f < 2 * dim; ++f)
        {
          const int nd  = f / 2;
          const double side = (f % 2) ? 1.0 : 0.0;
          int       ax[2];
          if (dim == 2)
            {
              ax[0] = (nd == 0) ? 1 : 0;
              ax[1] = -1;
            }
          else
            {
              if (nd == 0)
                {
                  ax[0] = 1;
                  ax[1] = 2;
                }
              else if (nd == 1)
                {
                  ax[0] = 2;
                  ax[1] = 0;
                }
              else
                {
                  ax[0] = 0;
                  ax[1] = 1;
                }
            }
          Nf[f].resize(size_t(nqf) * npc);
          dNf[f].resize(size_t(nqf) * npc * dim);
          xif[f].resize(size_t(nqf) * dim);
          for (int fq = 0; fq < nqf; ++fq)
            {
              int    f1 = fq % nq1, f2 = fq / nq1;
              double xi[3] = {0, 0, 0};
              xi[nd]       = side;
              xi[ax[0]]    = qx[f1];
              double w     = qw[f1];
              if (dim == 3)
                {
                  xi[ax[1]] = qx[f2];
                  w *= qw[f2];
                }
              Wf[fq] = w;
              for (int d = 0; d < dim; ++d)
                xif[f][fq * dim + d] = xi[d];
              eval(xi, &Nf[f][size_t(fq) * npc], &dNf[f][size_t(fq) * npc * dim]);
            }
        }
    }
    // all cell shape functions (lexicographic, x fastest) and unit gradients at unit point xi
    void eval(const double *xi, double *Nout, double *dNout) const
    {
      double n1[3][8], d1[3][8];
      for (int d = 0; d < dim; ++d)
        lagrange_1d(nodes1.data(), np1, xi[d], n1[d], d1[d]);
      for (int a = 0; a < npc; ++a)
        {
          int ai[3] = {a % np1, (a / np1) % np1, dim == 3 ? a / (np1 * np1) : 0};
          double v  = 1.0;
          for (int d = 0; d < dim; ++d)
            v *= n1[d][ai[d]];
          Nout[a] = v;
          for (int k = 0; k < dim; ++k)
            {
              double g = 1.0;
              for (int d = 0; d < dim; ++d)
                g *= (d == k) ? d1[d][ai[d]] : n1[d][ai[d]];
              dNout[a * dim + k] = g;
            }
        }
    }
  };

  // [DEAL.II] MappingQ1: d-linear in the 2^dim vertices (lexicographic vertex order).
  // Jm[i][j] = dX_i/dxi_j
  void q1_jacobian(int dim, const double *verts, const double *xi, double Jm[3][3], double *X)
  {
    const int nv = 1 << dim;
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j)
        Jm[i][j] = 0;
    double Xl[3] = {0, 0, 0};
    for (int v = 0; v < nv; ++v)
      {
        int    vb[3] = {v & 1, (v >> 1) & 1, (v >> 2) & 1};
        double phi   = 1.0, dphi[3];
        for (int d = 0; d < dim; ++d)
          phi *= vb[d] ? xi[d] : 1.0 - xi[d];
        for (int k = 0; k < dim; ++k)
          {
            double g = 1.0;
            for (int d = 0; d < dim; ++d)
              g *= (d == k) ? (vb[d] ? 1.0 : -1.0) : (vb[d] ? xi[d] : 1.0 - xi[d]);
            dphi[k] = g;
          }
        for (int i = 0; i < dim; ++i)
          {
            Xl[i] += verts[v * dim + i] * phi;
            for (int j = 0; j < dim; ++j)
              Jm[i][j] += verts[v * dim + i] * dphi[j];
          }
      }
    if (X)
      for (int i = 0; i < dim; ++i)
        X[i] = Xl[i];
  }

  // ------------------------------------------------------------------ cell assembly (as written)
  struct CellScratch
  {
    vector<double> G;     // [nq][npc][dim] real-space (reference-configuration) gradients
    vector<double> JxW;   // [nq]
    vector<double> gradu; // [nq][3][3]  solution_grads_u_total
    vector<double> acc;   // [nq][3]     local_acceleration
  };

  // FEValues::reinit + get_function_gradients/values, nonlinear_elasticity.cc:891-906
  void cell_kinematics(const FETables &fe, const double *verts, const double *u, const double *a, CellScratch &s)
  {
    const int dim = fe.dim, npc = fe.npc, nq = fe.nq;
    s.G.assign(size_t(nq) * npc * dim, 0.0);
    s.JxW.assign(nq, 0.0);
    s.gradu.assign(size_t(nq) * 9, 0.0);
    s.acc.assign(size_t(nq) * 3, 0.0);
    for (int q = 0; q < nq; ++q)
      {
        int    qi[3] = {q % fe.nq1, (q / fe.nq1) % fe.nq1, dim == 3 ? q / (fe.nq1 * fe.nq1) : 0};
        double xi[3] = {fe.qx[qi[0]], fe.qx[qi[1]], dim == 3 ? fe.qx[qi[2]] : 0.0};
        double Jm[3][3], Ji[3][3];
        q1_jacobian(dim, verts, xi, Jm, nullptr);
        inv3(Jm, dim, Ji);
        s.JxW[q] = det3(Jm, dim) * fe.W[q];
        for (int k = 0; k < npc; ++k)
          for (int i = 0; i < dim; ++i)
            {
              double g = 0;
              for (int j = 0; j < dim; ++j)
                g += fe.dN[(size_t(q) * npc + k) * dim + j] * Ji[j][i];
              s.G[(size_t(q) * npc + k) * dim + i] = g;
            }
        for (int k = 0; k < npc; ++k)
          for (int c = 0; c < dim; ++c)
            {
              for (int j = 0; j < dim; ++j)
                s.gradu[q * 9 + c * 3 + j] += u[k * dim + c] * s.G[(size_t(q) * npc + k) * dim + j];
              if (a)
                s.acc[q * 3 + c] += a[k * dim + c] * fe.N[size_t(q) * npc + k];
            }
      }
  }

  // assemble_system_tangent_residual_one_cell, nonlinear_elasticity.cc:872-1036
  void cell_tangent_residual(const FETables &fe, const Material &mat, double alpha_1, const double body_force[3],
                             const double *verts, const double *u, const double *a, CellScratch &s, double *Ke,
                             double *re)
  {
    const int dim = fe.dim, npc = fe.npc, nq = fe.nq, dpc = npc * dim;
    std::fill(Ke, Ke + size_t(dpc) * dpc, 0.0); // :889 data.reset()
    std::fill(re, re + dpc, 0.0);
    cell_kinematics(fe, verts, u, a, s); // :891-906
    const double rho = mat.rho;          // :909

    vector<double> grad_Nx(size_t(dpc) * 9), symm_grad_Nx(size_t(dpc) * 9), shape_value(size_t(dpc) * 3);
    for (int q = 0; q < nq; ++q) // :915
      {
        double F[3][3], Fb[3][3], bb[3][3], Fi[3][3];
        for (int i = 0; i < 3; ++i)
          for (int j = 0; j < 3; ++j)
            F[i][j] = (i == j) + s.gradu[q * 9 + i * 3 + j]; // :927-928  Kinematics::F = I + Grad u
        const double det_F = det3(F, dim);                   // :929
        const double sc    = std::pow(det_F, -1.0 / dim);    // :930-931  F_iso = J^{-1/dim} F  [DEAL.II]
        for (int i = 0; i < 3; ++i)
          for (int j = 0; j < 3; ++j)
            Fb[i][j] = sc * F[i][j];
        for (int i = 0; i < 3; ++i) // :932-933  b = symmetrize(F F^T)  [DEAL.II]
          for (int j = 0; j < 3; ++j)
            {
              double v = 0;
              for (int k = 0; k < dim; ++k)
                v += Fb[i][k] * Fb[j][k];
              bb[i][j] = v;
            }
        inv3(F, dim, Fi); // :934

        // :939-955
        std::fill(grad_Nx.begin(), grad_Nx.end(), 0.0);
        std::fill(symm_grad_Nx.begin(), symm_grad_Nx.end(), 0.0);
        std::fill(shape_value.begin(), shape_value.end(), 0.0);
        for (int k = 0; k < dpc; ++k)
          {
            const int a_ = k / dim, c = k % dim;
            // fe_values[u_fe].gradient(k,q) = e_c (x) Grad N_a ; times F_inv
            for (int j = 0; j < dim; ++j)
              {
                double v = 0;
                for (int m = 0; m < dim; ++m)
                  v += s.G[(size_t(q) * npc + a_) * dim + m] * Fi[m][j];
                grad_Nx[k * 9 + c * 3 + j] = v;
              }
            for (int i = 0; i < dim; ++i)
              for (int j = 0; j < dim; ++j)
                symm_grad_Nx[k * 9 + i * 3 + j] = 0.5 * (grad_Nx[k * 9 + i * 3 + j] + grad_Nx[k * 9 + j * 3 + i]);
            shape_value[k * 3 + c] = fe.N[size_t(q) * npc + a_];
          }

        double tau[3][3], Jc[3][3][3][3];
        mat.tau(det_F, bb, tau); // :958-959
        mat.Jc(det_F, bb, Jc);   // :960-961
        const double JxW = s.JxW[q];

        vector<double> JcB(size_t(dpc) * 9); // Jc : symm_grad_Nx[j], hoisted (pure re-association of :1012)
        for (int j = 0; j < dpc; ++j)
          for (int i1 = 0; i1 < dim; ++i1)
            for (int i2 = 0; i2 < dim; ++i2)
              {
                double v = 0;
                for (int k = 0; k < dim; ++k)
                  for (int l = 0; l < dim; ++l)
                    v += Jc[i1][i2][k][l] * symm_grad_Nx[j * 9 + k * 3 + l];
                JcB[j * 9 + i1 * 3 + i2] = v;
              }

        for (int i = 0; i < dpc; ++i) // :973
          {
            const int component_i = i % dim;
            // :984-988
            double sgt = 0;
            for (int k = 0; k < dim; ++k)
              for (int l = 0; l < dim; ++l)
                sgt += symm_grad_Nx[i * 9 + k * 3 + l] * tau[k][l];
            const double Ni = fe.N[size_t(q) * npc + i / dim];
            re[i] -= (sgt - body_force[component_i] * rho * Ni) * JxW;
            // :993-995  (Tensor<1> dot product: non-zero only for equal components)
            for (int j = 0; j < dpc; ++j)
              {
                double dot = 0;
                for (int c = 0; c < dim; ++c)
                  dot += shape_value[i * 3 + c] * shape_value[j * 3 + c];
                re[i] -= dot * rho * s.acc[q * 3 + component_i] * JxW;
              }
            for (int j = 0; j <= i; ++j) // :1001
              {
                const int component_j = j % dim;
                double    v           = 0;
                for (int k = 0; k < dim; ++k)
                  for (int l = 0; l < dim; ++l)
                    v += symm_grad_Nx[i * 9 + k * 3 + l] * JcB[j * 9 + k * 3 + l];
                Ke[size_t(i) * dpc + j] += v * JxW; // :1011-1012
                if (component_i == component_j)     // :1015-1023
                  {
                    double geo = 0;
                    for (int k = 0; k < dim; ++k)
                      for (int l = 0; l < dim; ++l)
                        geo += grad_Nx[i * 9 + component_i * 3 + k] * tau[k][l] * grad_Nx[j * 9 + component_j * 3 + l];
                    Ke[size_t(i) * dpc + j] +=
                      (geo + shape_value[i * 3 + component_i] * rho * alpha_1 * shape_value[j * 3 + component_j]) * JxW;
                  }
              }
          }
      }
    for (int i = 0; i < dpc; ++i) // :1033-1035
      for (int j = i + 1; j < dpc; ++j)
        Ke[size_t(i) * dpc + j] = Ke[size_t(j) * dpc + i];
  }

  // face geometry under MappingQ1: JxW_face and outward unit normal at unit point xi of face f
  void face_geometry(int dim, const double *verts, int f, const double *xi, double w, double &JxW, double n[3])
  {
    double Jm[3][3];
    q1_jacobian(dim, verts, xi, Jm, nullptr);
    const int    nd  = f / 2;
    const double sgn = (f % 2) ? 1.0 : -1.0;
    double       cr[3] = {0, 0, 0};
    if (dim == 2)
      {
        const int t  = (nd == 0) ? 1 : 0; // tangential axis
        double    tx = Jm[0][t], ty = Jm[1][t];
        // rotate tangent by -90deg: (ty,-tx); orientation fixed below by sign of det
        cr[0] = ty;
        cr[1] = -tx;
        // make it point along +xi_nd: check against column nd
        double dotv = cr[0] * Jm[0][nd] + cr[1] * Jm[1][nd];
        if (dotv < 0)
          {
            cr[0] = -cr[0];
            cr[1] = -cr[1];
          }
      }
    else
      {
        const int t1 = (nd + 1) % 3, t2 = (nd + 2) % 3;
        double    a[3] = {Jm[0][t1], Jm[1][t1], Jm[2][t1]}, b[3] = {Jm[0][t2], Jm[1][t2], Jm[2][t2]};
        cr[0] = a[1] * b[2] - a[2] * b[1];
        cr[1] = a[2] * b[0] - a[0] * b[2];
        cr[2] = a[0] * b[1] - a[1] * b[0];
        double dotv = cr[0] * Jm[0][nd] + cr[1] * Jm[1][nd] + cr[2] * Jm[2][nd];
        if (dotv < 0)
          for (int i = 0; i < 3; ++i)
            cr[i] = -cr[i];
      }
    const double len = std::sqrt(cr[0] * cr[0] + cr[1] * cr[1] + cr[2] * cr[2]);
    JxW             = len * w;
    for (int i = 0; i < 3; ++i)
      n[i] = sgn * cr[i] / len;
  }

  // assemble_neumann_contribution_one_cell, nonlinear_elasticity.cc:791-859 (one face)
  void cell_neumann_face(const FETables &fe, int f, bool correct_face_F, const double *verts, const double *u,
                         const double *stress, const CellScratch &s, double *re)
  {
    const int dim = fe.dim, npc = fe.npc;
    for (int fq = 0; fq < fe.nqf; ++fq) // :818
      {
        double JxW, nrm[3];
        face_geometry(dim, verts, f, &fe.xif[f][fq * dim], fe.Wf[fq], JxW, nrm);
        double local_stress[3] = {0, 0, 0}; // :815-816
        for (int k = 0; k < npc; ++k)
          for (int c = 0; c < dim; ++c)
            local_stress[c] += fe.Nf[f][size_t(fq) * npc + k] * stress[k * dim + c];
        double F[3][3];
        if (!correct_face_F)
          {
            // :825-827 QUIRK: cell-QP gradient array indexed by the face-QP counter
            for (int i = 0; i < 3; ++i)
              for (int j = 0; j < 3; ++j)
                F[i][j] = (i == j) + s.gradu[fq * 9 + i * 3 + j];
          }
        else
          {
            double Jm[3][3], Ji[3][3];
            q1_jacobian(dim, verts, &fe.xif[f][fq * dim], Jm, nullptr);
            inv3(Jm, dim, Ji);
            for (int i = 0; i < 3; ++i)
              for (int j = 0; j < 3; ++j)
                F[i][j] = (i == j);
            for (int k = 0; k < npc; ++k)
              for (int j = 0; j < dim; ++j)
                {
                  double g = 0;
                  for (int m = 0; m < dim; ++m)
                    g += fe.dNf[f][(size_t(fq) * npc + k) * dim + m] * Ji[m][j];
                  for (int c = 0; c < dim; ++c)
                    F[c][j] += u[k * dim + c] * g;
                }
          }
        double Fi[3][3];
        inv3(F, dim, Fi);
        const double J = det3(F, dim);
        double       ns[3] = {0, 0, 0}; // :831-833  n* = det F F^{-T} N
        for (int i = 0; i < dim; ++i)
          for (int j = 0; j < dim; ++j)
            ns[i] += J * Fi[j][i] * nrm[j];
        const double nn = std::sqrt(ns[0] * ns[0] + ns[1] * ns[1] + ns[2] * ns[2]);
        for (int i = 0; i < npc * dim; ++i) // :839-856
          re[i] += fe.Nf[f][size_t(fq) * npc + i / dim] * (local_stress[i % dim] * nn) * JxW;
      }
  }

  // ------------------------------------------------------------------ mesh + dofs
  struct Mesh
  {
    int            dim, p, np1;
    int            reps[3], nn[3], nv[3];
    int            ncells, nnodes, nverts, npc;
    vector<double> vx;       // vertex coords
    vector<int>    conn;     // [ncells][npc]
    vector<double> xyz;      // node coords (support points under Q1 mapping)
    vector<int>    cellface; // [ncells][2*dim] role or 0
    vector<unsigned char> constrained; // per dof
    vector<int>    iface_nodes;

    void cell_verts(int c, double *out) const
    {
      int ci[3] = {c % reps[0], (c / reps[0]) % reps[1], dim == 3 ? c / (reps[0] * reps[1]) : 0};
      for (int v = 0; v < (1 << dim); ++v)
        {
          int vi[3] = {ci[0] + (v & 1), ci[1] + ((v >> 1) & 1), ci[2] + ((v >> 2) & 1)};
          int id    = vi[0] + nv[0] * (vi[1] + nv[1] * (dim == 3 ? vi[2] : 0));
          for (int d = 0; d < dim; ++d)
            out[v * dim + d] = vx[size_t(id) * dim + d];
        }
    }

    // make_grid (nonlinear_elasticity.cc:171-301) for a box with given roles; system numbering node-major
    void build(const orc_desc &d, const double *perturb, const double *nodes1)
    {
      dim = d.dim;
      p   = d.degree;
      np1 = p + 1;
      for (int i = 0; i < 3; ++i)
        {
          reps[i] = i < dim ? d.reps[i] : 1;
          nn[i]   = i < dim ? p * reps[i] + 1 : 1;
          nv[i]   = i < dim ? reps[i] + 1 : 1;
        }
      ncells = reps[0] * reps[1] * reps[2];
      nnodes = nn[0] * nn[1] * nn[2];
      nverts = nv[0] * nv[1] * nv[2];
      npc    = 1;
      for (int i = 0; i < dim; ++i)
        npc *= np1;
      vx.resize(size_t(nverts) * dim);
      for (int k = 0; k < nv[2]; ++k)
        for (int j = 0; j < nv[1]; ++j)
          for (int i = 0; i < nv[0]; ++i)
            {
              int id    = i + nv[0] * (j + nv[1] * k);
              int ii[3] = {i, j, k};
              for (int dd = 0; dd < dim; ++dd)
                vx[size_t(id) * dim + dd] = d.lo[dd] + (d.hi[dd] - d.lo[dd]) * ii[dd] / reps[dd] +
                                            (perturb ? perturb[size_t(id) * dim + dd] : 0.0);
            }
      conn.resize(size_t(ncells) * npc);
      xyz.assign(size_t(nnodes) * dim, 0.0);
      cellface.assign(size_t(ncells) * 2 * dim, 0);
      for (int c = 0; c < ncells; ++c)
        {
          int    ci[3] = {c % reps[0], (c / reps[0]) % reps[1], dim == 3 ? c / (reps[0] * reps[1]) : 0};
          double verts[24];
          cell_verts(c, verts);
          for (int a = 0; a < npc; ++a)
            {
              int ai[3] = {a % np1, (a / np1) % np1, dim == 3 ? a / (np1 * np1) : 0};
              int gi[3] = {ci[0] * p + ai[0], ci[1] * p + ai[1], ci[2] * p + ai[2]};
              int node  = gi[0] + nn[0] * (gi[1] + nn[1] * gi[2]);
              conn[size_t(c) * npc + a] = node;
              double xi[3] = {nodes1[ai[0]], nodes1[ai[1]], dim == 3 ? nodes1[ai[2]] : 0.0};
              double Jm[3][3], X[3];
              q1_jacobian(dim, verts, xi, Jm, X);
              for (int dd = 0; dd < dim; ++dd)
                xyz[size_t(node) * dim + dd] = X[dd];
            }
          for (int f = 0; f < 2 * dim; ++f)
            {
              const int nd = f / 2;
              const bool at_bdry = (f % 2 == 0) ? (ci[nd] == 0) : (ci[nd] == reps[nd] - 1);
              if (at_bdry)
                cellface[size_t(c) * 2 * dim + f] = d.face_role[f];
            }
        }
      // make_constraints, nonlinear_elasticity.cc:1094-1150
      constrained.assign(size_t(nnodes) * dim, 0);
      vector<unsigned char> on_iface(nnodes, 0);
      for (int c = 0; c < ncells; ++c)
        for (int f = 0; f < 2 * dim; ++f)
          {
            const int role = cellface[size_t(c) * 2 * dim + f];
            if (!role)
              continue;
            const int nd = f / 2, side = f % 2;
            for (int a = 0; a < npc; ++a)
              {
                int ai[3] = {a % np1, (a / np1) % np1, dim == 3 ? a / (np1 * np1) : 0};
                if (ai[nd] != (side ? p : 0))
                  continue;
                const int node = conn[size_t(c) * npc + a];
                if (role == ORC_FACE_CLAMPED)
                  for (int cc = 0; cc < dim; ++cc)
                    constrained[size_t(node) * dim + cc] = 1;
                else if (role == ORC_FACE_ZCLAMP && dim == 3)
                  constrained[size_t(node) * dim + 2] = 1;
                else if (role == ORC_FACE_INTERFACE)
                  on_iface[node] = 1;
              }
          }
      // adapter.h:250-260,313-321: x-component dofs on the interface in ascending index order
      for (int n = 0; n < nnodes; ++n)
        if (on_iface[n])
          iface_nodes.push_back(n);
    }
  };

  // scalar CSR, all components couple (nonlinear_elasticity.cc:339-345)
  struct CSR
  {
    int            n = 0;
    vector<int>    rowptr, col, diag;
    vector<double> val;
    void           build(const Mesh &m)
    {
      const int dim = m.dim;
      n             = m.nnodes * dim;
      // node adjacency
      vector<vector<int>> adj(m.nnodes);
      for (int c = 0; c < m.ncells; ++c)
        for (int a = 0; a < m.npc; ++a)
          {
            auto &v = adj[m.conn[size_t(c) * m.npc + a]];
            for (int b = 0; b < m.npc; ++b)
              v.push_back(m.conn[size_t(c) * m.npc + b]);
          }
      rowptr.assign(n + 1, 0);
      for (int nd = 0; nd < m.nnodes; ++nd)
        {
          auto &v = adj[nd];
          std::sort(v.begin(), v.end());
          v.erase(std::unique(v.begin(), v.end()), v.end());
          for (int c = 0; c < dim; ++c)
            rowptr[nd * dim + c + 1] = int(v.size()) * dim;
        }
      for (int i = 0; i < n; ++i)
        rowptr[i + 1] += rowptr[i];
      col.resize(rowptr[n]);
      diag.resize(n);
      for (int nd = 0; nd < m.nnodes; ++nd)
        for (int c = 0; c < dim; ++c)
          {
            int k = rowptr[nd * dim + c];
            for (int nb : adj[nd])
              for (int cc = 0; cc < dim; ++cc)
                {
                  if (nb == nd && cc == c)
                    diag[nd * dim + c] = k;
                  col[k++] = nb * dim + cc;
                }
          }
      val.assign(col.size(), 0.0);
    }
    int find(int i, int j) const
    {
      const int *b = &col[rowptr[i]], *e = &col[rowptr[i + 1]];
      const int *it = std::lower_bound(b, e, j);
      return int(it - col.data());
    }
    void vmult(const double *x, double *y) const
    {
#pragma omp parallel for schedule(static) num_threads(g_threads)
      for (int i = 0; i < n; ++i)
        {
          double s = 0;
          for (int k = rowptr[i]; k < rowptr[i + 1]; ++k)
            s += val[k] * x[col[k]];
          y[i] = s;
        }
    }
  };

  double l2norm(const double *x, int n)
  {
    double s = 0;
    for (int i = 0; i < n; ++i)
      s += x[i] * x[i];
    return std::sqrt(s);
  }
  double dot(const double *x, const double *y, int n)
  {
    double s = 0;
    for (int i = 0; i < n; ++i)
      s += x[i] * y[i];
    return s;
  }

  // [DEAL.II] SparseMatrix::precondition_SSOR: (D/w + U)^-1 ((2-w)/w D) (D/w + L)^-1
  void precondition_ssor(const CSR &A, double om, const double *src, double *dst)
  {
    const int n = A.n;
    for (int i = 0; i < n; ++i)
      {
        double s = src[i];
        for (int k = A.rowptr[i]; k < A.diag[i]; ++k)
          s -= A.val[k] * dst[A.col[k]];
        dst[i] = s * om / A.val[A.diag[i]];
      }
    for (int i = 0; i < n; ++i)
      dst[i] *= (2.0 - om) * A.val[A.diag[i]] / om;
    for (int i = n - 1; i >= 0; --i)
      {
        double s = dst[i];
        for (int k = A.diag[i] + 1; k < A.rowptr[i + 1]; ++k)
          s -= A.val[k] * dst[A.col[k]];
        dst[i] = s * om / A.val[A.diag[i]];
      }
  }

  // [DEAL.II] SolverCG with SolverControl(max_it, tol): stop when ||r||_2 <= tol, start from x.
  // precond: 0 SSOR(om), 1 Jacobi.  returns 0 ok, 1 no convergence.
  int solver_cg(const CSR &A, double *x, const double *b, int precond, double om, int max_it, double tol, int *its,
                double *res_out)
  {
    const int      n = A.n;
    vector<double> g(n), d(n), h(n);
    A.vmult(x, g.data());
    for (int i = 0; i < n; ++i)
      g[i] = b[i] - g[i];
    double res = l2norm(g.data(), n);
    int    it  = 0;
    if (res <= tol)
      {
        *its     = 0;
        *res_out = res;
        return 0;
      }
    auto prec = [&](const double *src, double *dst) {
      if (precond == 0)
        precondition_ssor(A, om, src, dst);
      else
        for (int i = 0; i < n; ++i)
          dst[i] = src[i] / A.val[A.diag[i]];
    };
    prec(g.data(), h.data());
    d         = h;
    double gh = dot(g.data(), h.data(), n);
    while (true)
      {
        ++it;
        A.vmult(d.data(), h.data());
        const double alpha = gh / dot(d.data(), h.data(), n);
        for (int i = 0; i < n; ++i)
          {
            x[i] += alpha * d[i];
            g[i] -= alpha * h[i];
          }
        res = l2norm(g.data(), n);
        if (res <= tol)
          break;
        if (it >= max_it)
          {
            *its     = it;
            *res_out = res;
            return 1;
          }
        prec(g.data(), h.data());
        const double gh_new = dot(g.data(), h.data(), n);
        const double beta   = gh_new / gh;
        gh                  = gh_new;
        for (int i = 0; i < n; ++i)
          d[i] = h[i] + beta * d[i];
      }
    *its     = it;
    *res_out = res;
    return 0;
  }

  // stand-in for SparseDirectUMFPACK (nonlinear_elasticity.cc:1192-1200): banded LU without pivoting
  int solver_direct(const CSR &A, double *x, const double *b)
  {
    const int n  = A.n;
    int       bw = 0;
    for (int i = 0; i < n; ++i)
      for (int k = A.rowptr[i]; k < A.rowptr[i + 1]; ++k)
        bw = std::max(bw, std::abs(A.col[k] - i));
    const size_t   ld = 2 * size_t(bw) + 1;
    vector<double> M(size_t(n) * ld, 0.0);
    auto           at = [&](int i, int j) -> double & { return M[size_t(i) * ld + (j - i + bw)]; };
    for (int i = 0; i < n; ++i)
      for (int k = A.rowptr[i]; k < A.rowptr[i + 1]; ++k)
        at(i, A.col[k]) = A.val[k];
    vector<double> y(b, b + n);
    for (int k = 0; k < n; ++k)
      {
        const double piv = at(k, k);
        if (piv == 0.0)
          return 1;
        const int iend = std::min(n - 1, k + bw);
        for (int i = k + 1; i <= iend; ++i)
          {
            const double l = at(i, k) / piv;
            if (l == 0.0)
              continue;
            for (int j = k + 1; j <= iend; ++j)
              at(i, j) -= l * at(k, j);
            y[i] -= l * y[k];
          }
      }
    for (int i = n - 1; i >= 0; --i)
      {
        double    s    = y[i];
        const int jend = std::min(n - 1, i + bw);
        for (int j = i + 1; j <= jend; ++j)
          s -= at(i, j) * x[j];
        x[i] = s / at(i, i);
      }
    return 0;
  }
} // namespace

// ====================================================================== nonlinear problem
struct orc_problem
{
  orc_desc       d;
  FETables       fe;
  Mesh           mesh;
  CSR            K;
  Material       mat;
  int            ndofs;
  vector<double> v[ORC_V_COUNT];
  double         alpha_1, alpha_2, alpha_3, alpha_4, alpha_5, alpha_6;
  orc_problem(const orc_desc &dd)
    : d(dd)
    , mat(dd.dim, dd.mu, dd.nu, dd.rho)
  {}
};

extern "C" {

void orc_set_threads(int n)
{
  g_threads = n < 1 ? 1 : n;
}
void orc_gauss_01(int n, double *x, double *w)
{
  gauss_01(n, x, w);
}
void orc_feq_support_1d(int p, double *x)
{
  feq_support_1d(p, x);
}
void orc_lagrange_1d(int p, double x, double *N, double *dN)
{
  vector<double> nodes(p + 1);
  feq_support_1d(p, nodes.data());
  lagrange_1d(nodes.data(), p + 1, x, N, dN);
}

double orc_material(int dim, double mu, double nu, const double *Fin, double *tau, double *Jc)
{
  Material m(dim, mu, nu, 0.0);
  double   F[3][3], Fb[3][3], bb[3][3];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      F[i][j] = (i < dim && j < dim) ? Fin[i * 3 + j] : double(i == j);
  const double J  = det3(F, dim);
  const double sc = std::pow(J, -1.0 / dim);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      Fb[i][j] = sc * F[i][j];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      {
        double v = 0;
        for (int k = 0; k < dim; ++k)
          v += Fb[i][k] * Fb[j][k];
        bb[i][j] = v;
      }
  double t[3][3], C[3][3][3][3];
  m.tau(J, bb, t);
  m.Jc(J, bb, C);
  std::memcpy(tau, t, sizeof(t));
  std::memcpy(Jc, C, sizeof(C));
  return m.Psi(J, bb);
}

void orc_cell_tangent_residual(const orc_desc *d, const double *verts, const double *u, const double *acc, double *Ke,
                               double *re)
{
  FETables fe;
  fe.build(d->dim, d->degree, d->degree + 2); // qf_cell(p+2), nonlinear_elasticity.cc:74
  Material    mat(d->dim, d->mu, d->nu, d->rho);
  CellScratch s;
  const double alpha_1 = 1.0 / (d->beta * d->delta_t * d->delta_t);
  cell_tangent_residual(fe, mat, alpha_1, d->body_force, verts, u, acc, s, Ke, re);
}

orc_problem *orc_create(const orc_desc *d, const double *perturb)
{
  orc_problem *p = new orc_problem(*d);
  p->fe.build(d->dim, d->degree, d->degree + 2); // :74-75
  p->mesh.build(*d, perturb, p->fe.nodes1.data());
  p->K.build(p->mesh);
  p->ndofs = p->mesh.nnodes * d->dim;
  for (auto &x : p->v)
    x.assign(p->ndofs, 0.0);
  // nonlinear_elasticity.h:242-250
  p->alpha_1 = 1. / (d->beta * std::pow(d->delta_t, 2));
  p->alpha_2 = 1. / (d->beta * d->delta_t);
  p->alpha_3 = (1 - (2 * d->beta)) / (2 * d->beta);
  p->alpha_4 = d->gamma / (d->beta * d->delta_t);
  p->alpha_5 = 1 - (d->gamma / d->beta);
  p->alpha_6 = (1 - (d->gamma / (2 * d->beta))) * d->delta_t;
  return p;
}
void orc_destroy(orc_problem *p)
{
  delete p;
}
int orc_n_dofs(const orc_problem *p)
{
  return p->ndofs;
}
int orc_n_nodes(const orc_problem *p)
{
  return p->mesh.nnodes;
}
int orc_n_cells(const orc_problem *p)
{
  return p->mesh.ncells;
}
long orc_nnz(const orc_problem *p)
{
  return long(p->K.col.size());
}
const double *orc_node_coords(const orc_problem *p)
{
  return p->mesh.xyz.data();
}
const int *orc_csr_rowptr(const orc_problem *p)
{
  return p->K.rowptr.data();
}
const int *orc_csr_col(const orc_problem *p)
{
  return p->K.col.data();
}
const double *orc_csr_val(const orc_problem *p)
{
  return p->K.val.data();
}
const unsigned char *orc_constrained(const orc_problem *p)
{
  return p->mesh.constrained.data();
}
int orc_n_interface_nodes(const orc_problem *p)
{
  return int(p->mesh.iface_nodes.size());
}
const int *orc_interface_nodes(const orc_problem *p)
{
  return p->mesh.iface_nodes.data();
}
double *orc_vec(orc_problem *p, int which)
{
  return p->v[which].data();
}

// update_acceleration, nonlinear_elasticity.cc:592-599
void orc_update_acceleration(orc_problem *p)
{
  const double *du = p->v[ORC_V_SOLUTION_DELTA].data(), *vo = p->v[ORC_V_VELOCITY_OLD].data(),
               *ao = p->v[ORC_V_ACCELERATION_OLD].data();
  double *a = p->v[ORC_V_ACCELERATION].data();
  for (int i = 0; i < p->ndofs; ++i)
    {
      a[i] = p->alpha_1 * du[i];
      a[i] += -p->alpha_2 * vo[i] + -p->alpha_3 * ao[i];
    }
}

// assemble_system (:1044-1087) + copy_local_to_global_ASM (:760-774).
// Cells are computed in parallel batches and scattered serially in cell order (= WorkStream).
void orc_assemble(orc_problem *p)
{
  const Mesh &m   = p->mesh;
  const int   dim = m.dim, npc = m.npc, dpc = npc * dim;
  std::fill(p->K.val.begin(), p->K.val.end(), 0.0); // :1054-1055
  double *rhs = p->v[ORC_V_SYSTEM_RHS].data();
  std::fill(rhs, rhs + p->ndofs, 0.0);
  vector<double> solution_total(p->ndofs); // :1062, :580-588
  for (int i = 0; i < p->ndofs; ++i)
    solution_total[i] = p->v[ORC_V_TOTAL_DISPLACEMENT][i] + p->v[ORC_V_SOLUTION_DELTA][i];
  const double *acc    = p->v[ORC_V_ACCELERATION].data();
  const double *stress = p->v[ORC_V_EXTERNAL_STRESS].data();

  const int      batch = std::max(1, g_threads) * 8;
  vector<double> Kb(size_t(batch) * dpc * dpc), rb(size_t(batch) * dpc);
  for (int c0 = 0; c0 < m.ncells; c0 += batch)
    {
      const int nb = std::min(batch, m.ncells - c0);
#pragma omp parallel num_threads(g_threads)
      {
        CellScratch    s;
        vector<double> ue(dpc), ae(dpc), se(dpc);
#pragma omp for schedule(dynamic, 1)
        for (int b = 0; b < nb; ++b)
          {
            const int c = c0 + b;
            double    verts[24];
            m.cell_verts(c, verts);
            for (int a = 0; a < npc; ++a)
              for (int cc = 0; cc < dim; ++cc)
                {
                  const int g      = m.conn[size_t(c) * npc + a] * dim + cc;
                  ue[a * dim + cc] = solution_total[g];
                  ae[a * dim + cc] = acc[g];
                  se[a * dim + cc] = stress[g];
                }
            double *Ke = &Kb[size_t(b) * dpc * dpc], *re = &rb[size_t(b) * dpc];
            cell_tangent_residual(p->fe, p->mat, p->alpha_1, p->d.body_force, verts, ue.data(), ae.data(), s, Ke, re);
            for (int f = 0; f < 2 * dim; ++f) // :804-805
              if (m.cellface[size_t(c) * 2 * dim + f] == ORC_FACE_INTERFACE)
                cell_neumann_face(p->fe, f, p->d.correct_face_F != 0, verts, ue.data(), se.data(), s, re);
          }
      }
      // [DEAL.II] AffineConstraints::distribute_local_to_global, homogeneous constraints:
      // constrained rows/cols dropped; diagonal of a constrained dof += |K_e(i,i)| (cell mean |diag| if 0).
      for (int b = 0; b < nb; ++b)
        {
          const int     c  = c0 + b;
          const double *Ke = &Kb[size_t(b) * dpc * dpc], *re = &rb[size_t(b) * dpc];
          double        avg = 0;
          for (int i = 0; i < dpc; ++i)
            avg += std::fabs(Ke[size_t(i) * dpc + i]);
          avg /= dpc;
          for (int i = 0; i < dpc; ++i)
            {
              const int gi = m.conn[size_t(c) * npc + i / dim] * dim + i % dim;
              if (m.constrained[gi])
                {
                  const double dg = std::fabs(Ke[size_t(i) * dpc + i]);
                  p->K.val[p->K.diag[gi]] += (dg != 0.0 ? dg : avg);
                  continue;
                }
              rhs[gi] += re[i];
              for (int j = 0; j < dpc; ++j)
                {
                  const int gj = m.conn[size_t(c) * npc + j / dim] * dim + j % dim;
                  if (m.constrained[gj])
                    continue;
                  p->K.val[p->K.find(gi, gj)] += Ke[size_t(i) * dpc + j];
                }
            }
        }
    }
}

// get_error_residual, :549-560
double orc_residual_norm(const orc_problem *p)
{
  double s = 0;
  for (int i = 0; i < p->ndofs; ++i)
    if (!p->mesh.constrained[i])
      s += p->v[ORC_V_SYSTEM_RHS][i] * p->v[ORC_V_SYSTEM_RHS][i];
  return std::sqrt(s);
}

void orc_spmv(const orc_problem *p, const double *x, double *y)
{
  p->K.vmult(x, y);
}

// solve_linear_system, :1153-1211
int orc_solve_linear(orc_problem *p, int solver, double tol_lin, double max_it_mult, int *its, double *res)
{
  double       *x   = p->v[ORC_V_NEWTON_UPDATE].data();
  const double *b   = p->v[ORC_V_SYSTEM_RHS].data();
  int           rc  = 0;
  if (solver == ORC_SOLVER_DIRECT)
    {
      rc   = solver_direct(p->K, x, b); // :1194-1196
      *its = 1;
      *res = 0.0;
    }
  else
    {
      const int    solver_its = int(p->ndofs * max_it_mult);          // :1169-1170
      const double tol_sol    = tol_lin * l2norm(b, p->ndofs);        // :1171-1172
      rc = solver_cg(p->K, x, b, solver == ORC_SOLVER_CG_SSOR ? 0 : 1, 0.65, solver_its, tol_sol, its, res); // :1180-1187
    }
  for (int i = 0; i < p->ndofs; ++i) // constraints.distribute, :1208
    if (p->mesh.constrained[i])
      x[i] = 0.0;
  return rc;
}

// solve_nonlinear_timestep (:410-499) wrapped by the per-step lines of run() (:121, :138-144)
int orc_newmark_step(orc_problem *p, int solver, double tol_lin, double max_it_mult, int max_it_nr, double tol_f,
                     double tol_u, orc_step_info *info)
{
  const int n  = p->ndofs;
  double   *du = p->v[ORC_V_SOLUTION_DELTA].data();
  double   *nu = p->v[ORC_V_NEWTON_UPDATE].data();
  std::fill(du, du + n, 0.0); // :121
  std::fill(nu, nu + n, 0.0); // :419
  // Errors default/reset to u = 1.0 (nonlinear_elasticity.h:293-315)
  double error_residual = 1.0, error_residual_0 = 1.0, error_residual_norm = 1.0;
  double error_update = 1.0, error_update_0 = 1.0, error_update_norm = 1.0;
  std::memset(info, 0, sizeof(*info));
  int newton_iteration = 0;
  for (; newton_iteration < max_it_nr; ++newton_iteration) // :436
    {
      orc_update_acceleration(p); // :444
      double t0 = now();
      orc_assemble(p); // :446
      info->t_assemble += now() - t0;
      info->assemblies++;
      error_residual = orc_residual_norm(p); // :449
      if (newton_iteration == 0)
        error_residual_0 = error_residual;
      error_residual_norm = error_residual;
      if (error_residual_0 != 0.0)
        error_residual_norm /= error_residual_0;
      if (newton_iteration > 0 && ((error_update_norm <= tol_u || error_update <= 1e-15) &&
                                   (error_residual_norm <= tol_f || error_residual <= 5e-9))) // :459-463
        {
          info->converged = 1;
          break;
        }
      int    its = 0;
      double res = 0;
      t0         = now();
      int rc     = orc_solve_linear(p, solver, tol_lin, max_it_mult, &its, &res); // :472
      info->t_solve += now() - t0;
      if (rc)
        return 2; // SolverControl::NoConvergence
      info->lin_its_total += its;
      info->newton_iterations++;
      double s = 0; // get_error_update :564-576
      for (int i = 0; i < n; ++i)
        if (!p->mesh.constrained[i])
          s += nu[i] * nu[i];
      error_update = std::sqrt(s);
      if (newton_iteration == 0)
        error_update_0 = error_update;
      error_update_norm = error_update;
      if (error_update_0 != 0.0)
        error_update_norm /= error_update_0;
      for (int i = 0; i < n; ++i) // :487
        du[i] += nu[i];
    }
  info->res_norm = error_residual_norm;
  info->res_abs  = error_residual;
  info->upd_norm = error_update_norm;
  info->upd_abs  = error_update;
  if (!(newton_iteration < max_it_nr)) // :497
    return 1;
  // run(): :139-144
  double *u = p->v[ORC_V_TOTAL_DISPLACEMENT].data(), *uo = p->v[ORC_V_TOTAL_DISPLACEMENT_OLD].data();
  double *v = p->v[ORC_V_VELOCITY].data(), *vo = p->v[ORC_V_VELOCITY_OLD].data();
  double *a = p->v[ORC_V_ACCELERATION].data(), *ao = p->v[ORC_V_ACCELERATION_OLD].data();
  for (int i = 0; i < n; ++i)
    u[i] += du[i];
  orc_update_acceleration(p); // :142
  for (int i = 0; i < n; ++i) // update_velocity :603-610
    {
      v[i] = p->alpha_4 * du[i];
      v[i] += p->alpha_5 * vo[i] + p->alpha_6 * ao[i];
    }
  for (int i = 0; i < n; ++i) // update_old_variables :614-622
    {
      uo[i] = u[i];
      vo[i] = v[i];
      ao[i] = a[i];
    }
  return 0;
}

// ====================================================================== linear model
} // extern "C"

struct orc_linear
{
  orc_desc       d;
  FETables       fe;
  Mesh           mesh;
  CSR            pat; // pattern; val unused
  vector<double> Kv, Mv, Sv, Av; // stiffness, mass, stepping, system
  vector<double> v[ORC_L_COUNT];
  vector<double> body_force_vector;
  bool           body_force_enabled;
  int            ndofs;
  orc_linear(const orc_desc &dd)
    : d(dd)
  {}
};

extern "C" {

// ElastoDynamics: make_grid :79-188, setup_system :192-244, assemble_system :248-374
orc_linear *orc_linear_create(const orc_desc *d)
{
  orc_linear *p = new orc_linear(*d);
  p->fe.build(d->dim, d->degree, d->degree + 1); // quad_order = p+1, linear_elasticity.cc:61
  p->mesh.build(*d, nullptr, p->fe.nodes1.data());
  p->pat.build(p->mesh);
  const Mesh &m = p->mesh;
  const int   dim = m.dim, npc = m.npc, dpc = npc * dim;
  p->ndofs = m.nnodes * dim;
  const size_t nnz = p->pat.col.size();
  p->Kv.assign(nnz, 0.0);
  p->Mv.assign(nnz, 0.0);
  for (auto &x : p->v)
    x.assign(p->ndofs, 0.0);
  p->body_force_vector.assign(p->ndofs, 0.0);
  double bn = 0;
  for (int i = 0; i < 3; ++i)
    bn += d->body_force[i] * d->body_force[i];
  p->body_force_enabled = std::sqrt(bn) > 1e-15; // :62
  const double lambda = 2 * d->mu * d->nu / (1 - 2 * d->nu); // parameters.cc:189
  CellScratch    s;
  vector<double> zero(dpc, 0.0), Ke(size_t(dpc) * dpc), Me(size_t(dpc) * dpc), be(dpc);
  for (int c = 0; c < m.ncells; ++c)
    {
      double verts[24];
      m.cell_verts(c, verts);
      cell_kinematics(p->fe, verts, zero.data(), nullptr, s);
      std::fill(Ke.begin(), Ke.end(), 0.0);
      std::fill(Me.begin(), Me.end(), 0.0);
      std::fill(be.begin(), be.end(), 0.0);
      for (int i = 0; i < dpc; ++i) // :287-321
        {
          const int ci = i % dim, ai = i / dim;
          for (int j = 0; j < dpc; ++j)
            {
              const int cj = j % dim, aj = j / dim;
              for (int q = 0; q < p->fe.nq; ++q)
                {
                  const double *gi = &s.G[(size_t(q) * npc + ai) * dim], *gj = &s.G[(size_t(q) * npc + aj) * dim];
                  double        v  = gi[ci] * gj[cj] * lambda + gi[cj] * gj[ci] * d->mu;
                  if (ci == cj)
                    {
                      double gg = 0;
                      for (int k = 0; k < dim; ++k)
                        gg += gi[k] * gj[k];
                      v += gg * d->mu;
                    }
                  Ke[size_t(i) * dpc + j] += v * s.JxW[q];
                  // [DEAL.II] MatrixCreator::create_mass_matrix with coefficient rho, :341-345
                  if (ci == cj)
                    Me[size_t(i) * dpc + j] +=
                      d->rho * p->fe.N[size_t(q) * npc + ai] * p->fe.N[size_t(q) * npc + aj] * s.JxW[q];
                }
            }
          // [DEAL.II] VectorTools::create_right_hand_side with constant rho*b, :358-373
          for (int q = 0; q < p->fe.nq; ++q)
            be[i] += d->rho * d->body_force[ci] * p->fe.N[size_t(q) * npc + ai] * s.JxW[q];
        }
      for (int i = 0; i < dpc; ++i) // :325-333
        {
          const int gi = m.conn[size_t(c) * npc + i / dim] * dim + i % dim;
          if (p->body_force_enabled)
            p->body_force_vector[gi] += be[i];
          for (int j = 0; j < dpc; ++j)
            {
              const int gj = m.conn[size_t(c) * npc + j / dim] * dim + j % dim;
              const int k  = p->pat.find(gi, gj);
              p->Kv[k] += Ke[size_t(i) * dpc + j];
              p->Mv[k] += Me[size_t(i) * dpc + j];
            }
        }
    }
  // stepping = M + theta^2 dt^2 K, :348-353
  p->Sv.resize(nnz);
  for (size_t k = 0; k < nnz; ++k)
    p->Sv[k] = p->Kv[k] * (d->delta_t * d->delta_t * d->theta * d->theta) + p->Mv[k];
  p->Av = p->Sv;
  return p;
}
void orc_linear_destroy(orc_linear *p)
{
  delete p;
}
int orc_linear_n_dofs(const orc_linear *p)
{
  return p->ndofs;
}
long orc_linear_nnz(const orc_linear *p)
{
  return long(p->pat.col.size());
}
const int *orc_linear_rowptr(const orc_linear *p)
{
  return p->pat.rowptr.data();
}
const int *orc_linear_col(const orc_linear *p)
{
  return p->pat.col.data();
}
const double *orc_linear_matrix(const orc_linear *p, int which)
{
  return which == 0 ? p->Kv.data() : which == 1 ? p->Mv.data() : which == 2 ? p->Sv.data() : p->Av.data();
}
const double *orc_linear_node_coords(const orc_linear *p)
{
  return p->mesh.xyz.data();
}
int orc_linear_n_interface_nodes(const orc_linear *p)
{
  return int(p->mesh.iface_nodes.size());
}
const int *orc_linear_interface_nodes(const orc_linear *p)
{
  return p->mesh.iface_nodes.data();
}
const unsigned char *orc_linear_constrained(const orc_linear *p)
{
  return p->mesh.constrained.data();
}
double *orc_linear_vec(orc_linear *p, int which)
{
  return p->v[which].data();
}

// one pass of the time loop body: assemble_rhs :378-454, solve :525-575, update_displacement :579-586
int orc_linear_step(orc_linear *p, int solver, int data_consistent, int *its, double *res)
{
  return orc_linear_step_tol(p, solver, data_consistent, 1e-10 /* :542 */, its, res);
}
// the same with the absolute CG tolerance of :542 as an argument (fixtures at tolerances tighter than the reference's)
int orc_linear_step_tol(orc_linear *p, int solver, int data_consistent, double abs_tol, int *its, double *res)
{
  const Mesh &m   = p->mesh;
  const int   dim = m.dim, npc = m.npc, n = p->ndofs;
  double     *rhs = p->v[ORC_L_SYSTEM_RHS].data();
  double     *stress = p->v[ORC_L_STRESS].data(), *old_stress = p->v[ORC_L_OLD_STRESS].data();
  double     *vel = p->v[ORC_L_VELOCITY].data(), *old_vel = p->v[ORC_L_OLD_VELOCITY].data();
  double     *dis = p->v[ORC_L_DISPLACEMENT].data(), *old_dis = p->v[ORC_L_OLD_DISPLACEMENT].data();
  const double dt = p->d.delta_t, theta = p->d.theta;
  if (data_consistent) // assemble_consistent_loading :458-521 (no pull-back)
    {
      std::fill(rhs, rhs + n, 0.0);
      for (int c = 0; c < m.ncells; ++c)
        for (int f = 0; f < 2 * dim; ++f)
          if (m.cellface[size_t(c) * 2 * dim + f] == ORC_FACE_INTERFACE)
            {
              double verts[24];
              m.cell_verts(c, verts);
              for (int fq = 0; fq < p->fe.nqf; ++fq)
                {
                  double JxW, nrm[3];
                  face_geometry(dim, verts, f, &p->fe.xif[f][fq * dim], p->fe.Wf[fq], JxW, nrm);
                  double ls[3] = {0, 0, 0};
                  for (int k = 0; k < npc; ++k)
                    for (int cc = 0; cc < dim; ++cc)
                      ls[cc] += p->fe.Nf[f][size_t(fq) * npc + k] * stress[m.conn[size_t(c) * npc + k] * dim + cc];
                  for (int k = 0; k < npc; ++k)
                    for (int cc = 0; cc < dim; ++cc)
                      rhs[m.conn[size_t(c) * npc + k] * dim + cc] += p->fe.Nf[f][size_t(fq) * npc + k] * ls[cc] * JxW;
                }
            }
    }
  else
    std::copy(stress, stress + n, rhs); // :387-388
  std::copy(vel, vel + n, old_vel);     // :390-391
  std::copy(dis, dis + n, old_dis);
  if (p->body_force_enabled) // :394-395
    for (int i = 0; i < n; ++i)
      rhs[i] += p->body_force_vector[i];
  vector<double> tmp(rhs, rhs + n); // :402-409
  for (int i = 0; i < n; ++i)
    {
      rhs[i] *= dt * theta;
      rhs[i] += dt * (1 - theta) * old_stress[i];
      old_stress[i] = tmp[i];
    }
  CSR &A = p->pat;
  A.val  = p->Mv; // :411-412
  A.vmult(old_vel, tmp.data());
  for (int i = 0; i < n; ++i)
    rhs[i] += tmp[i];
  A.val = p->Kv; // :414-417
  A.vmult(old_vel, tmp.data());
  for (int i = 0; i < n; ++i)
    rhs[i] += -theta * dt * dt * (1 - theta) * tmp[i];
  A.vmult(old_dis, tmp.data()); // :419-420
  for (int i = 0; i < n; ++i)
    rhs[i] += -dt * tmp[i];
  // :426-451  system_matrix = stepping_matrix; [DEAL.II] MatrixTools::apply_boundary_values with zero values:
  // row and column eliminated, diagonal kept, rhs_i = 0, solution_i = 0
  A.val = p->Sv;
  for (int i = 0; i < n; ++i)
    if (m.constrained[i])
      {
        for (int k = A.rowptr[i]; k < A.rowptr[i + 1]; ++k)
          if (k != A.diag[i])
            {
              A.val[k]                   = 0.0;
              A.val[A.find(A.col[k], i)] = 0.0; // symmetric pattern
            }
        rhs[i] = 0.0;
        vel[i] = 0.0;
      }
  p->Av  = A.val;
  int rc = 0;
  *its   = 1;
  *res   = 0.0;
  if (solver == ORC_SOLVER_DIRECT) // :553-559
    rc = solver_direct(A, vel, rhs);
  else // :531-551  abs tol 1e-10, SSOR 1.2, start from previous velocity
    rc = solver_cg(A, vel, rhs, solver == ORC_SOLVER_CG_SSOR ? 0 : 1, 1.2, n /* multiplier 1 */, abs_tol, its, res);
  for (int i = 0; i < n; ++i) // update_displacement :579-586
    {
      dis[i] += dt * theta * vel[i];
      dis[i] += dt * (1 - theta) * old_vel[i];
    }
  return rc;
}

// ====================================================================== Adapter::Time
void orc_time_init(orc_time *t, double time_end, double delta_t) // time_handler.h:24-29
{
  t->timestep     = 0;
  t->time_current = 0.0;
  t->time_end     = time_end;
  t->delta_t      = delta_t;
}
void orc_time_increment(orc_time *t) // :72-77
{
  t->time_current += t->delta_t;
  ++t->timestep;
}
void orc_time_set_absolute(orc_time *t, double new_time) // :63-70 (double -> unsigned truncation)
{
  double factor   = std::pow(10, 10);
  t->timestep     = (unsigned int)(std::round((new_time / t->delta_t) * factor) / factor);
  t->time_current = new_time;
}

} // extern "C"
