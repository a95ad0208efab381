// Sanitizer driver (CPU, test infrastructure): the oracle (oracle/elasticity_oracle.cpp) built with
// -fsanitize=address,undefined and run through everything the tests and bench.py's cpu_baseline leg call: known-answer
// pieces, a 3D Q2 and a 2D Q3 Newmark step with each linear solver, the linear model, the time helper.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "elasticity_oracle.h"

static int g_fail = 0;
#define CHECK(x)                                                          \
  do                                                                      \
    {                                                                     \
      if (!(x))                                                           \
        {                                                                 \
          std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #x);      \
          ++g_fail;                                                       \
        }                                                                 \
    }                                                                     \
  while (0)

static orc_desc desc(int dim, int p, int rx, int ry, int rz)
{
  orc_desc d;
  std::memset(&d, 0, sizeof(d));
  d.dim     = dim;
  d.degree  = p;
  d.reps[0] = rx;
  d.reps[1] = ry;
  d.reps[2] = rz;
  for (int k = 0; k < 3; ++k)
    d.hi[k] = 0.1 * d.reps[k];
  const int roles[6] = {ORC_FACE_CLAMPED, ORC_FACE_INTERFACE, ORC_FACE_INTERFACE, ORC_FACE_INTERFACE,
                        dim == 3 ? ORC_FACE_ZCLAMP : 0, dim == 3 ? ORC_FACE_INTERFACE : 0};
  for (int f = 0; f < 6; ++f)
    d.face_role[f] = roles[f];
  d.mu            = 0.5e6;
  d.nu            = 0.4;
  d.rho           = 1000.0;
  d.body_force[1] = -9.81;
  d.beta          = 0.25;
  d.gamma         = 0.5;
  d.delta_t       = 0.005;
  d.theta         = 0.5;
  return d;
}

int main()
{
  orc_set_threads(2);
  { // known-answer pieces
    double x[5], w[5], N[5], dN[5], s = 0.0;
    orc_gauss_01(4, x, w);
    for (int i = 0; i < 4; ++i)
      s += w[i];
    CHECK(std::abs(s - 1.0) < 1e-14);
    orc_feq_support_1d(4, x);
    orc_lagrange_1d(4, 0.3, N, dN);
    s = 0.0;
    for (int i = 0; i < 5; ++i)
      s += N[i];
    CHECK(std::abs(s - 1.0) < 1e-13);
    const double F[9] = {1.02, 0.01, 0.0, -0.02, 0.97, 0.03, 0.0, 0.01, 1.05};
    double       tau[9], Jc[81]; // full 3 x 3 and 3 x 3 x 3 x 3 tensors
    CHECK(orc_material(3, 0.5e6, 0.4, F, tau, Jc) > 0.0);
  }
  const int shapes[2][5] = {{3, 2, 2, 2, 3}, {2, 3, 4, 3, 1}};
  for (const auto &sh : shapes)
    for (int solver : {ORC_SOLVER_CG_SSOR, ORC_SOLVER_CG_JACOBI, ORC_SOLVER_DIRECT})
      {
        const orc_desc d = desc(sh[0], sh[1], sh[2], sh[3], sh[4]);
        orc_problem   *P = orc_create(&d, nullptr);
        CHECK(P != nullptr);
        const int n = orc_n_dofs(P), ni = orc_n_interface_nodes(P);
        CHECK(n > 0 && ni > 0 && orc_nnz(P) > 0);
        double *t = orc_vec(P, ORC_V_EXTERNAL_STRESS);
        for (int k = 0; k < ni; ++k)
          t[orc_interface_nodes(P)[k] * d.dim + 1] = -50.0;
        orc_step_info info;
        for (int step = 0; step < 2; ++step)
          CHECK(orc_newmark_step(P, solver, 1e-10, 2.0, 10, 1e-9, 1e-9, &info) == 0 && info.converged);
        std::vector<double> x(size_t(n), 1.0), y(size_t(n), 0.0);
        orc_spmv(P, x.data(), y.data());
        CHECK(std::isfinite(orc_residual_norm(P)) && std::isfinite(y[0]));
        orc_destroy(P);
      }
  { // linear model, both read-data kinds
    const orc_desc d = desc(2, 2, 5, 2, 1);
    for (int consistent = 0; consistent < 2; ++consistent)
      {
        orc_linear *L = orc_linear_create(&d);
        CHECK(L != nullptr && orc_linear_n_dofs(L) > 0 && orc_linear_nnz(L) > 0);
        double *t = orc_linear_vec(L, ORC_L_STRESS);
        for (int k = 0; k < orc_linear_n_interface_nodes(L); ++k)
          t[orc_linear_interface_nodes(L)[k] * 2 + 1] = -3.0;
        int    its = 0;
        double res = 0.0;
        for (int step = 0; step < 2; ++step)
          CHECK(orc_linear_step(L, step ? ORC_SOLVER_DIRECT : ORC_SOLVER_CG_SSOR, consistent, &its, &res) == 0);
        orc_linear_destroy(L);
      }
  }
  std::printf(g_fail ? "ORACLE SANITIZER RUN FAILED (%d)\n" : "ORACLE SANITIZER RUN OK\n", g_fail);
  return g_fail ? 1 : 0;
}
