#include "linear_elasticity.h"

#include <cmath>
#include <iomanip>
#include <iostream>
#include <sstream>

#include <mi/vtk_output.h>

namespace Linear_Elasticity
{
  template <int dim>
  ElastoDynamics<dim>::ElastoDynamics(const std::string &parameter_file)
    : parameters(parameter_file)
    , timer(std::cout)
    , time(parameters.end_time, parameters.delta_t)
    , adapter(parameters, interface_boundary_id)
  {}

  template <int dim>
  ElastoDynamics<dim>::~ElastoDynamics() = default;

  // linear_elasticity.cc:79-188
  template <int dim>
  void ElastoDynamics<dim>::make_grid()
  {
    mesh_desc        = mi_mesh_desc{};
    mesh_desc.dim    = dim;
    mesh_desc.degree = int(parameters.poly_degree);
    unsigned int id_flap_long_bottom = 0, id_flap_long_top = 0, id_flap_short_bottom = 0, id_flap_short_top = 0;
    auto         set_box = [&](const int reps[3], const double lo[3], const double hi[3]) {
      for (int d = 0; d < 3; ++d)
        {
          mesh_desc.reps[d] = reps[d];
          mesh_desc.lo[d]   = lo[d];
          mesh_desc.hi[d]   = hi[d];
        }
    };
    if (parameters.scenario == "FSI3")
      {
        const int    reps[3] = {18, 3, 1};
        const double lo[3] = {0.24899, 0.19, -0.005}, hi[3] = {0.6, 0.21, 0.005};
        set_box(reps, lo, hi);
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
        set_box(reps, lo, hi);
        id_flap_long_bottom  = 0;
        id_flap_long_top     = 1;
        id_flap_short_bottom = 2;
        id_flap_short_top    = 3;
      }
    else // Block
      {
        set_box(parameters.repetitions, parameters.lower, parameters.upper);
        mesh_desc.face_role[0] = MI_FACE_CLAMPED;
        for (int f = 1; f < 6; ++f)
          mesh_desc.face_role[f] = MI_FACE_INTERFACE;
        return;
      }
    if (clamped_mesh_id == interface_boundary_id || out_of_plane_clamped_mesh_id == interface_boundary_id)
      throw std::runtime_error("The interface_id cannot be the same as the clamped one");
    if (interface_boundary_id != adapter.deal_boundary_interface_id)
      throw std::runtime_error("Wrong interface ID in the Adapter specified");
    for (unsigned int f = 0; f < 6; ++f) // :171-187
      {
        if (f == id_flap_short_top || f == id_flap_long_bottom || f == id_flap_long_top)
          mesh_desc.face_role[f] = MI_FACE_INTERFACE;
        else if (f == id_flap_short_bottom)
          mesh_desc.face_role[f] = MI_FACE_CLAMPED;
        else if (f == 4 || f == 5)
          mesh_desc.face_role[f] = MI_FACE_ZCLAMP;
      }
  }

  // setup_system :192-244 + assemble_system :248-374
  template <int dim>
  void ElastoDynamics<dim>::setup_system()
  {
    mi_material_desc mat{};
    mat.mu  = parameters.mu;
    mat.nu  = parameters.nu;
    mat.rho = parameters.rho;
    for (int d = 0; d < 3; ++d)
      mat.body_force[d] = parameters.body_force[d];
    mi_newmark_desc nm{0.25, 0.5, parameters.delta_t}; // Newmark constants are unused by the linear model
    int             dev_id = 0;
    if (const char *e = std::getenv("MI_DEVICE"))
      dev_id = std::atoi(e);
    device = std::make_unique<mi::Device>(mesh_desc, mat, nm, dev_id);
    device->check(mi_linear_setup(device->ctx(), parameters.theta), "mi_linear_setup");

    std::cout << "Triangulation:"
              << "\n\t Number of active cells: " << mi_n_cells(device->ctx())
              << "\n\t Polynomial degree: " << parameters.poly_degree
              << "\n\t Number of degrees of freedom: " << mi_n_dofs(device->ctx()) << std::endl;

    displacement.bind(*device, MI_L_DISPLACEMENT);
    old_displacement.bind(*device, MI_L_OLD_DISPLACEMENT);
    velocity.bind(*device, MI_L_VELOCITY);
    old_velocity.bind(*device, MI_L_OLD_VELOCITY);
    stress.bind(*device, MI_L_STRESS);
    old_stress.bind(*device, MI_L_OLD_STRESS);
    // :238-239
    state_variables = {&old_velocity, &velocity, &old_displacement, &displacement, &old_stress};
  }

  template <int dim>
  void ElastoDynamics<dim>::output_results() const
  {
    timer.enter_subsection("Output results");
    const unsigned int interval = parameters.output_interval > 0 ? parameters.output_interval : 1;
    std::ostringstream name;
    name << "solution-" << std::setw(3) << std::setfill('0') << time.get_timestep() / interval << ".vtk";
    // all ranks: the global views behind the output are gathered by team collectives; rank 0 writes the file
    mi::write_vtk(*device, dim, int(parameters.poly_degree), mesh_desc.reps, parameters.output_folder + "/" + name.str(), mi::host_rank() == 0,
                  !std::getenv("MI_VTK_LINEAR_CELLS")); // higher-order cells as the reference (:1222-1225); the switch: linear sub-cells
    timer.leave_subsection("Output results");
  }

  // :634-716
  template <int dim>
  void ElastoDynamics<dim>::run()
  {
    make_grid();
    setup_system();
    output_results();

    adapter.initialize(DoFSource{device.get()}, displacement);

    while (adapter.precice.isCouplingOngoing())
      {
        adapter.save_current_state_if_required(state_variables, time);
        time.increment();

        std::cout << "  Time = " << time.current() << " at timestep " << time.get_timestep() << std::endl;

        if (!(std::abs(time.get_delta_t() - adapter.precice.getMaxTimeStepSize()) < 1e-10))
          throw std::runtime_error("This solver supports only constant time-step sizes."
                                   "Configured time step size in deal.II parameter file: " +
                                   std::to_string(time.get_delta_t()) + ". Time-window size from preCICE: " +
                                   std::to_string(adapter.precice.getMaxTimeStepSize()) + ".");

        adapter.read_data(time.get_delta_t(), stress);

        // assemble_rhs (:378-454) + solve (:525-575) + update_displacement (:579-586)
        timer.enter_subsection("Solve system");
        int       lin_it  = 1;
        double    lin_res = 0.0;
        const bool direct = parameters.type_lin == "Direct";
        std::cout << (direct ? "\t Direct solver: " : "\t CG solver: ") << std::endl;
        // "Direct" (UMFPACK, :553-559): the constant system matrix is factorised once on the device (banded Cholesky,
        // tuning "solver_type" 1) and a step is two substitutions; where the system is too large for that, the device
        // PCG at a tolerance four orders tighter serves it (the library falls back by itself)
        if (direct && !solver_type_set)
          {
            device->check(mi_set_tuning(device->ctx(), "solver_type", 1), "mi_set_tuning");
            solver_type_set = true;
          }
        device->check(mi_linear_step(device->ctx(), parameters.data_consistent ? 1 : 0, direct ? 1e-14 : 1e-10,
                                     static_cast<int64_t>(double(mi_n_dofs(device->ctx())) *
                                                          std::max(1.0, parameters.max_iterations_lin)),
                                     &lin_it, &lin_res),
                      "mi_linear_step");
        timer.leave_subsection("Solve system");
        std::cout << "\t     No of iterations:\t" << lin_it << "\n \t     Final residual:\t" << lin_res << std::endl;

        timer.enter_subsection("Advance adapter");
        adapter.advance(displacement, time.get_delta_t());
        timer.leave_subsection("Advance adapter");

        adapter.reload_old_state_if_required(state_variables, time);

        if (adapter.precice.isTimeWindowComplete() && parameters.output_interval > 0 &&
            time.get_timestep() % parameters.output_interval == 0)
          output_results();
      }
    adapter.precice.finalize();
  }

  template class ElastoDynamics<DIM>;
} // namespace Linear_Elasticity
