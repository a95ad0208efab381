#include "nonlinear_elasticity.h"

#include <algorithm>
#include <cmath>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <sstream>

#include <mi/vtk_output.h>

namespace Nonlinear_Elasticity
{
  template <int dim>
  Solid<dim>::Solid(const std::string &parameter_file)
    : parameters(parameter_file)
    , degree(parameters.poly_degree)
    , timer(std::cout)
    , time(parameters.end_time, parameters.delta_t)
    , adapter(parameters, boundary_interface_id)
  {
    // nonlinear_elasticity.cc:83-87
    if (!parameters.data_consistent)
      throw std::runtime_error(
        "The neo-Hookean solid doesn't support 'Force' data reading. Please switch to 'Stress' "
        "data on the Fluid side or use the linear model of the solid solver");
  }

  template <int dim>
  Solid<dim>::~Solid()
  {
    // device-side companion of the TimerOutput table (HIP events on the solver's stream), MI_PROFILE=1
    if (device && std::getenv("MI_PROFILE"))
      {
        mi_timings t{};
        if (mi_get_timings(device->ctx(), &t) == MI_OK)
          {
            static const char *names[MI_T_COUNT] = {"assemble cells", "assemble total", "SpMV (CG)", "CG vector kernels",
                                                    "CG total",       "Newmark",        "step (host wall)", "diagonal blocks",
                                                    "residual pass",  "SpMV (smoother)", "matrix-free launch"};
            std::cout << "\nDevice timings (ms, launches):" << std::endl;
            for (int i = 0; i < MI_T_COUNT; ++i)
              std::cout << "  " << std::left << std::setw(20) << names[i] << std::right << std::setw(12) << std::fixed
                        << std::setprecision(3) << t.ms[i] << std::setw(10) << t.count[i] << std::endl;
          }
      }
  }

  // time loop of nonlinear_elasticity.cc:99-167
  template <int dim>
  void Solid<dim>::run()
  {
    make_grid();
    system_setup();
    output_results();

    adapter.initialize(DoFSource{device.get()}, total_displacement);

    while (adapter.precice.isCouplingOngoing())
      {
        adapter.save_current_state_if_required(state_variables, time);

        time.increment();

        if (!(std::abs(time.get_delta_t() - adapter.precice.getMaxTimeStepSize()) < 1e-10))
          throw std::runtime_error("This solver supports only constant time-step sizes."
                                   "Configured time step size in deal.II parameter file: " +
                                   std::to_string(time.get_delta_t()) + ". Time-window size from preCICE: " +
                                   std::to_string(adapter.precice.getMaxTimeStepSize()) + ".");

        adapter.read_data(time.get_delta_t(), external_stress);

        // Newton-Raphson; solution_delta lives on the device and is zeroed inside
        solve_nonlinear_timestep();
        // total_displacement += solution_delta; update_acceleration/velocity/old_variables (:139-144)
        device->check(mi_newmark_finish_step(device->ctx()), "mi_newmark_finish_step");
        log_step_json();

        timer.enter_subsection("Advance adapter");
        adapter.advance(total_displacement, time.get_delta_t());
        timer.leave_subsection("Advance adapter");

        adapter.reload_old_state_if_required(state_variables, time);

        if (adapter.precice.isTimeWindowComplete() && parameters.output_interval > 0 &&
            time.get_timestep() % parameters.output_interval == 0)
          output_results();
      }
    adapter.precice.finalize();
  }

  // geometry and boundary roles of nonlinear_elasticity.cc:171-301 (+ the additive "Block" scenario)
  template <int dim>
  void Solid<dim>::make_grid()
  {
    if (!((dim == 2 && parameters.body_force[2] == 0) || dim == 3))
      throw std::runtime_error(
        "Setting body forces in z-direction for a two dimensional simulation has no effect");

    mesh_desc        = mi_mesh_desc{};
    mesh_desc.dim    = dim;
    mesh_desc.degree = int(degree);
    // colorize ids: 0/1 = x-/x+, 2/3 = y-/y+, 4/5 = z-/z+
    unsigned int id_flap_long_bottom, id_flap_long_top, id_flap_short_bottom, id_flap_short_top;
    bool         relabel_by_ids = true;
    if (parameters.scenario == "FSI3")
      {
        const int    reps[3] = {18, 3, 1};
        const double lo[3] = {0.24899, 0.19, -0.005}, hi[3] = {0.6, 0.21, 0.005};
        for (int d = 0; d < 3; ++d)
          {
            mesh_desc.reps[d] = reps[d];
            mesh_desc.lo[d]   = lo[d];
            mesh_desc.hi[d]   = hi[d];
          }
        id_flap_long_bottom  = 2;
        id_flap_long_top     = 3;
        id_flap_short_bottom = 0;
        id_flap_short_top    = 1;
      }
    else if (parameters.scenario == "PF")
      {
        const int    reps[3] = {3, 18, 1};
        const double x0      = parameters.flap_location;
        const double lo[3] = {x0 - 0.05, 0, 0}, hi[3] = {x0 + 0.05, 1, 0.3};
        for (int d = 0; d < 3; ++d)
          {
            mesh_desc.reps[d] = reps[d];
            mesh_desc.lo[d]   = lo[d];
            mesh_desc.hi[d]   = hi[d];
          }
        id_flap_long_bottom  = 0;
        id_flap_long_top     = 1;
        id_flap_short_bottom = 2;
        id_flap_short_top    = 3;
      }
    else // Block: clamped x-, coupling interface everywhere else
      {
        for (int d = 0; d < 3; ++d)
          {
            mesh_desc.reps[d] = parameters.repetitions[d];
            mesh_desc.lo[d]   = parameters.lower[d];
            mesh_desc.hi[d]   = parameters.upper[d];
          }
        relabel_by_ids = false;
        id_flap_long_bottom = id_flap_long_top = id_flap_short_bottom = id_flap_short_top = 0;
        mesh_desc.face_role[0] = MI_FACE_CLAMPED;
        for (int f = 1; f < 6; ++f)
          mesh_desc.face_role[f] = MI_FACE_INTERFACE;
      }
    if (relabel_by_ids)
      for (unsigned int f = 0; f < 6; ++f) // :265-285
        {
          if (f == id_flap_short_bottom)
            mesh_desc.face_role[f] = int(clamped_boundary_id);
          else if (f == id_flap_long_bottom || f == id_flap_long_top || f == id_flap_short_top)
            mesh_desc.face_role[f] = int(boundary_interface_id);
          else if (f == 4 || f == 5)
            mesh_desc.face_role[f] = int(out_of_plane_clamped_mesh_id);
          else
            throw std::runtime_error("Unknown boundary id, did you set a boundary condition?");
        }
    // :287-296
    if (clamped_boundary_id == boundary_interface_id || boundary_interface_id == out_of_plane_clamped_mesh_id)
      throw std::runtime_error("Boundary IDs must not be the same, for different boundary types.");
    if (boundary_interface_id != adapter.deal_boundary_interface_id)
      throw std::runtime_error("Wrong interface ID in the Adapter.");

    vol_reference = 1.0;
    for (int d = 0; d < dim; ++d)
      vol_reference *= mesh_desc.hi[d] - mesh_desc.lo[d];
    vol_current = vol_reference; // never updated afterwards, as in the reference (:298-299, :540)
    std::cout << "Grid:\n\t Reference volume: " << vol_reference << std::endl;
  }

  // system_setup of :305-380: DoFs, sparsity, vectors -> one device context
  template <int dim>
  void Solid<dim>::system_setup()
  {
    timer.enter_subsection("Setup system");
    mi_material_desc mat{};
    mat.mu  = parameters.mu;
    mat.nu  = parameters.nu;
    mat.rho = parameters.rho;
    for (int d = 0; d < 3; ++d)
      mat.body_force[d] = parameters.body_force[d];
    mi_newmark_desc nm{parameters.beta, parameters.gamma, parameters.delta_t};
    int             dev_id = 0;
    if (const char *e = std::getenv("MI_DEVICE"))
      dev_id = std::atoi(e);
    device = std::make_unique<mi::Device>(mesh_desc, mat, nm, dev_id);
    if (std::getenv("MI_PROFILE"))
      mi_set_profiling(device->ctx(), 1);
    // a time-stepping run: the j-th linear solve of a step starts from the solution of the j-th solve of the previous
    // step (the reference starts from its previous Newton update, :419 / :472 -- the same stopping rule either way,
    // fewer iterations this way); MI_CG_WARM_START=0|1 selects zero / the reference's start vector instead
    // (switches of this PROGRAM, handed on as tuning keys: the library itself reads no environment variable)
    {
      const char *e = std::getenv("MI_CG_WARM_START");
      device->check(mi_set_tuning(device->ctx(), "cg_warm_start", e ? std::max(0, std::min(3, std::atoi(e))) : 2), "mi_set_tuning");
      if (const char *f = std::getenv("MI_CORRECT_FACE_F")) // "--correct-face-F" (SURVEY section 9); default: the reference's quirk
        device->check(mi_set_tuning(device->ctx(), "correct_face_F", std::atoi(f) != 0), "mi_set_tuning");
      if (const char *f = std::getenv("MI_FINE_LEVEL")) // 1: the fine level matrix-free end to end (3D Q2 meshes)
        {
          if (std::atoi(f) != 0 && mi_set_tuning(device->ctx(), "fine_level", 1) != MI_OK)
            std::cout << "MI_FINE_LEVEL ignored: " << mi_last_error(device->ctx()) << std::endl;
          else if (std::atoi(f) != 0)
            device->check(mi_set_tuning(device->ctx(), "mf_diag_lag", 1), "mi_set_tuning");
        }
    }

    std::cout << "Triangulation:"
              << "\n\t Number of active cells: " << mi_n_cells(device->ctx())
              << "\n\t Polynomial degree: " << parameters.poly_degree
              << "\n\t Number of degrees of freedom: " << mi_n_dofs(device->ctx()) << std::endl;

    total_displacement.bind(*device, MI_V_TOTAL_DISPLACEMENT);
    total_displacement_old.bind(*device, MI_V_TOTAL_DISPLACEMENT_OLD);
    velocity.bind(*device, MI_V_VELOCITY);
    velocity_old.bind(*device, MI_V_VELOCITY_OLD);
    acceleration.bind(*device, MI_V_ACCELERATION);
    acceleration_old.bind(*device, MI_V_ACCELERATION_OLD);
    external_stress.bind(*device, MI_V_EXTERNAL_STRESS);
    // :370-375
    state_variables = {&total_displacement, &total_displacement_old, &velocity,
                       &velocity_old,       &acceleration,           &acceleration_old};
    timer.leave_subsection();
  }

  // Newton-Raphson loop of :410-499
  template <int dim>
  void Solid<dim>::solve_nonlinear_timestep()
  {
    std::cout << std::endl
              << "Timestep " << time.get_timestep() << " @ " << std::fixed << time.current() << "s" << std::endl;

    mi_ctx *ctx = device->ctx();
    device->check(mi_newton_begin_step(ctx), "mi_newton_begin_step"); // solution_delta = 0 (:121), newton_update (:419)

    error_residual.reset();
    error_residual_0.reset();
    error_residual_norm.reset();
    error_update.reset();
    error_update_0.reset();
    error_update_norm.reset();

    print_conv_header();

    // "Direct" (UMFPACK, :1192-1200): banded Cholesky on the device (mi_direct_solve) for the sizes of the reference's
    // geometries; beyond them (MI_EINVAL) the PCG runs to a tolerance at which the answer is solver independent
    const bool   direct  = parameters.type_lin == "Direct";
    const double tol_lin = direct ? 1e-12 : parameters.tol_lin;
    const double it_mult = direct ? std::max(10.0, parameters.max_iterations_lin) : parameters.max_iterations_lin;

    last_newton_iterations = last_lin_iterations = 0;
    unsigned int newton_iteration = 0;
    for (; newton_iteration < parameters.max_iterations_NR; ++newton_iteration)
      {
        std::cout << " " << std::setw(2) << newton_iteration << " " << std::flush;
        std::cout << " CST " << std::flush; // make_constraints: the Dirichlet set is fixed, built once on the device

        device->check(mi_update_acceleration(ctx), "mi_update_acceleration"); // :444

        // the convergence test below needs the update criterion AND the residual criterion.  Only when the first holds
        // (known before the assembly) can this be the last assembly of the step, whose tangent is never multiplied:
        // then the residual alone is formed (same numbers) and the tangent follows only if the test fails.
        const bool update_ok =
          newton_iteration > 0 && (error_update_norm.u <= parameters.tol_u || error_update.u <= 1e-15);
        timer.enter_subsection("Assemble linear system");
        std::cout << " ASM " << std::flush;
        if (update_ok)
          device->check(mi_assemble_residual(ctx, &error_residual.u), "mi_assemble_residual");
        else
          device->check(mi_assemble(ctx, &error_residual.u), "mi_assemble"); // :446-449
        timer.leave_subsection();

        if (newton_iteration == 0)
          error_residual_0 = error_residual;
        error_residual_norm = error_residual;
        error_residual_norm.normalise(error_residual_0);

        // :459-463
        if (newton_iteration > 0 && ((error_update_norm.u <= parameters.tol_u || error_update.u <= 1e-15) &&
                                     (error_residual_norm.u <= parameters.tol_f || error_residual.u <= 5e-9)))
          {
            std::cout << " CONVERGED! " << std::endl;
            print_conv_footer();
            break;
          }
        if (update_ok) // not converged after all: the tangent of this state is needed for the next solve
          {
            timer.enter_subsection("Assemble linear system");
            device->check(mi_assemble(ctx, &error_residual.u), "mi_assemble");
            timer.leave_subsection();
          }

        timer.enter_subsection("Linear solver");
        std::cout << " SLV " << std::flush;
        int       lin_it  = 0;
        double    lin_res = 0.0;
        int rc = MI_EINVAL;
        if (direct && use_device_direct)
          {
            lin_it = 1; // :1198-1199
            rc     = mi_direct_solve(ctx, &lin_res);
            if (rc == MI_EINVAL)
              use_device_direct = false; // too large for the device factorisation: iterate from now on
          }
        if (rc == MI_EINVAL)
          rc = mi_cg_solve(ctx, tol_lin, static_cast<int64_t>(double(mi_n_dofs(ctx)) * it_mult), &lin_it, &lin_res);
        timer.leave_subsection();
        device->check(rc, "mi_cg_solve"); // SolverControl::NoConvergence
        ++last_newton_iterations;
        last_lin_iterations += static_cast<unsigned int>(lin_it);

        device->check(mi_apply_newton_update(ctx, &error_update.u), "mi_apply_newton_update"); // :476-487
        if (newton_iteration == 0)
          error_update_0 = error_update;
        error_update_norm = error_update;
        error_update_norm.normalise(error_update_0);

        std::cout << " | " << std::fixed << std::setprecision(3) << std::setw(7) << std::scientific << lin_it << "  "
                  << lin_res << "  " << error_residual_norm.u << "  " << error_residual.u << "  "
                  << "  " << error_update_norm.u << "  " << error_update.u << "  " << std::endl;
      }
    if (!(newton_iteration < parameters.max_iterations_NR))
      throw std::runtime_error("No convergence in nonlinear solver!"); // :497-498
  }

  template <int dim>
  void Solid<dim>::print_conv_header()
  {
    static const unsigned int l_width = 87;
    for (unsigned int i = 0; i < l_width; ++i)
      std::cout << "_";
    std::cout << std::endl;
    std::cout << "    SOLVER STEP    "
              << " |  LIN_IT   LIN_RES    RES_NORM   "
              << "RES_ABS      U_NORM    "
              << " U_ABS " << std::endl;
    for (unsigned int i = 0; i < l_width; ++i)
      std::cout << "_";
    std::cout << std::endl;
  }

  template <int dim>
  void Solid<dim>::print_conv_footer()
  {
    error_residual.normalise(error_residual_0);
    error_update.normalise(error_update_0);
    static const unsigned int l_width = 87;
    for (unsigned int i = 0; i < l_width; ++i)
      std::cout << "_";
    std::cout << std::endl;
    std::cout << "Relative errors:" << std::endl
              << "Displacement:\t" << error_update.u << std::endl
              << "Residual: \t" << error_residual.u << std::endl
              << "v / V_0:\t" << vol_current << " / " << vol_reference << std::endl;
  }

  template <int dim>
  void Solid<dim>::log_step_json() const
  {
    if (mi::host_rank() > 0)
      return;
    std::ofstream out(parameters.output_folder + "/steps.jsonl", std::ios::app);
    if (!out)
      return;
    out << std::setprecision(17) << "{\"timestep\": " << time.get_timestep() << ", \"time\": " << time.current()
        << ", \"newton_iterations\": " << last_newton_iterations << ", \"linear_iterations\": " << last_lin_iterations
        << ", \"residual_abs\": " << error_residual.u << ", \"update_abs\": " << error_update.u
        << ", \"n_dofs\": " << mi_n_dofs(device->ctx()) << "}\n";
  }

  // :1215-1254: solution-XXX.vtk with index timestep / output_interval
  template <int dim>
  void Solid<dim>::output_results() const
  {
    timer.enter_subsection("Output results");
    const unsigned int interval = parameters.output_interval > 0 ? parameters.output_interval : 1;
    std::ostringstream name;
    name << "solution-" << std::setw(3) << std::setfill('0') << time.get_timestep() / interval << ".vtk";
    // all ranks: the global views behind the output are gathered by team collectives; rank 0 writes the file
    mi::write_vtk(*device, dim, int(degree), mesh_desc.reps, parameters.output_folder + "/" + name.str(), mi::host_rank() == 0,
                  !std::getenv("MI_VTK_LINEAR_CELLS")); // higher-order cells as the reference (:1222-1225); the switch: linear sub-cells
    std::cout << "\t Output written to " << name.str() << " \n" << std::endl;
    timer.leave_subsection("Output results");
  }

  template class Solid<DIM>;
} // namespace Nonlinear_Elasticity
