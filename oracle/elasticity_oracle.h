/*
 * elasticity_oracle.h -- C interface of the CPU oracle.
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load this library.  The product
 * (libmi_elasticity.so) never links, loads or calls it.
 *
 * PARITY UNPINNED: the reference (precice/dealii-adapter) ships no tests or
 * golden vectors and its arithmetic lives in deal.II 9.5 / preCICE 3.0, which
 * are neither vendored nor installed.  This oracle is a restatement of the
 * reference loops (file:line cited at each function in the .cpp) validated by
 * analytic known-answer tests (tests/test_oracle_*.py), not by reference output.
 */
#ifndef ELASTICITY_ORACLE_H
#define ELASTICITY_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* face roles for the 2*dim faces of the box, order x-,x+,y-,y+,z-,z+
 * (= deal.II colorize ids 0..5, nonlinear_elasticity.cc:237-241) */
enum { ORC_FACE_CLAMPED = 1, ORC_FACE_INTERFACE = 7, ORC_FACE_ZCLAMP = 8 };

typedef struct
{
  int    dim;          /* 2 or 3                                          */
  int    degree;       /* FE_Q degree p                                   */
  int    reps[3];      /* subdivided_hyper_rectangle repetitions          */
  double lo[3], hi[3]; /* box corners                                     */
  int    face_role[6]; /* ORC_FACE_* for each box side                    */
  double mu, nu, rho;  /* material (parameters.cc:35-45)                  */
  double body_force[3];
  double beta, gamma, delta_t; /* Newmark (nonlinear_elasticity.h:242-250) */
  double theta;                /* linear model (linear_elasticity.cc)      */
  int    correct_face_F;       /* 0 = reproduce face pull-back quirk       */
} orc_desc;

/* ---- stand-alone pieces (known-answer tests) ---- */
void orc_gauss_01(int n, double *x, double *w);
void orc_feq_support_1d(int p, double *x);
void orc_lagrange_1d(int p, double x, double *N, double *dN);
/* F row-major 3x3 (only dim x dim used); tau 3x3; Jc 3x3x3x3; returns Psi */
double orc_material(int dim, double mu, double nu, const double *F, double *tau, double *Jc);
/* one cell, as written in the reference: verts[2^dim][dim], u/acc[npc*dim]
 * local dof = dim*a + c, a lexicographic.  Ke[dpc*dpc] row-major, re[dpc] */
void orc_cell_tangent_residual(const orc_desc *d, const double *verts, const double *u, const double *acc,
                               double *Ke, double *re);

/* ---- full problem ---- */
typedef struct orc_problem orc_problem;
orc_problem *orc_create(const orc_desc *d, const double *vertex_perturbation /* nverts*dim or NULL */);
void         orc_destroy(orc_problem *p);
int          orc_n_dofs(const orc_problem *p);
int          orc_n_nodes(const orc_problem *p);
int          orc_n_cells(const orc_problem *p);
long         orc_nnz(const orc_problem *p);
const double *orc_node_coords(const orc_problem *p); /* nnodes*dim */
const int    *orc_csr_rowptr(const orc_problem *p);
const int    *orc_csr_col(const orc_problem *p);
const double *orc_csr_val(const orc_problem *p);
const unsigned char *orc_constrained(const orc_problem *p); /* ndofs flags */
int          orc_n_interface_nodes(const orc_problem *p);
const int   *orc_interface_nodes(const orc_problem *p); /* ascending x-dof order, adapter.h:313-321 */

/* state vectors, nonlinear_elasticity.h:273-287 */
enum
{
  ORC_V_TOTAL_DISPLACEMENT = 0,
  ORC_V_TOTAL_DISPLACEMENT_OLD,
  ORC_V_VELOCITY,
  ORC_V_VELOCITY_OLD,
  ORC_V_ACCELERATION,
  ORC_V_ACCELERATION_OLD,
  ORC_V_EXTERNAL_STRESS,
  ORC_V_SOLUTION_DELTA,
  ORC_V_NEWTON_UPDATE,
  ORC_V_SYSTEM_RHS,
  ORC_V_COUNT
};
double *orc_vec(orc_problem *p, int which);

void   orc_set_threads(int n);
void   orc_update_acceleration(orc_problem *p);
void   orc_assemble(orc_problem *p);            /* K, rhs from current state */
double orc_residual_norm(const orc_problem *p); /* get_error_residual        */
enum { ORC_SOLVER_CG_SSOR = 0, ORC_SOLVER_CG_JACOBI = 1, ORC_SOLVER_DIRECT = 2 };
/* solve K * newton_update = rhs (warm start), then distribute constraints */
int orc_solve_linear(orc_problem *p, int solver, double tol_lin, double max_it_mult, int *its, double *res);
/* generic y = K x on the current matrix */
void orc_spmv(const orc_problem *p, const double *x, double *y);

typedef struct
{
  int    newton_iterations; /* number of linear solves                */
  int    assemblies;
  int    lin_its_total;
  int    converged;
  double res_norm, res_abs, upd_norm, upd_abs;
  double t_assemble, t_solve;
} orc_step_info;
/* solve_nonlinear_timestep + Newmark updates (nonlinear_elasticity.cc:121,138-144) */
int orc_newmark_step(orc_problem *p, int solver, double tol_lin, double max_it_mult, int max_it_nr, double tol_f,
                     double tol_u, orc_step_info *info);

/* ---- linear model (linear_elasticity.cc) ---- */
typedef struct orc_linear orc_linear;
orc_linear *orc_linear_create(const orc_desc *d);
void        orc_linear_destroy(orc_linear *p);
int         orc_linear_n_dofs(const orc_linear *p);
long        orc_linear_nnz(const orc_linear *p);
const int    *orc_linear_rowptr(const orc_linear *p);
const int    *orc_linear_col(const orc_linear *p);
const double *orc_linear_matrix(const orc_linear *p, int which); /* 0 K, 1 M, 2 stepping, 3 system */
const double *orc_linear_node_coords(const orc_linear *p);
int         orc_linear_n_interface_nodes(const orc_linear *p);
const int  *orc_linear_interface_nodes(const orc_linear *p);
const unsigned char *orc_linear_constrained(const orc_linear *p);
enum
{
  ORC_L_DISPLACEMENT = 0,
  ORC_L_OLD_DISPLACEMENT,
  ORC_L_VELOCITY,
  ORC_L_OLD_VELOCITY,
  ORC_L_STRESS,
  ORC_L_OLD_STRESS,
  ORC_L_SYSTEM_RHS,
  ORC_L_COUNT
};
double *orc_linear_vec(orc_linear *p, int which);
/* assemble_rhs + solve + update_displacement; data_consistent: 1 Stress, 0 Force */
int orc_linear_step(orc_linear *p, int solver, int data_consistent, int *its, double *res);
/* the same with the absolute CG tolerance (1e-10 at linear_elasticity.cc:542) as an argument */
int orc_linear_step_tol(orc_linear *p, int solver, int data_consistent, double abs_tol, int *its, double *res);

/* ---- Adapter::Time (time_handler.h:21-84) ---- */
typedef struct
{
  unsigned int timestep;
  double       time_current, time_end, delta_t;
} orc_time;
void orc_time_init(orc_time *t, double time_end, double delta_t);
void orc_time_increment(orc_time *t);
void orc_time_set_absolute(orc_time *t, double new_time);

#ifdef __cplusplus
}
#endif
#endif
