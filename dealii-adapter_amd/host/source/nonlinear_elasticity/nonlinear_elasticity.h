// Nonlinear_Elasticity::Solid<dim> -- finite-strain neo-Hookean elastodynamics (Newmark + Newton-Raphson),
// host driver of the device hot path.  Counterpart of the reference's
// source/nonlinear_elasticity/include/nonlinear_elasticity.h:131-331 / nonlinear_elasticity.cc:
// same constructor/run() surface, same time loop, Newton logic, convergence table and timer sections; the
// work inside assemble_system / solve_linear_system / update_* happens in HIP kernels behind the mi_* C-ABI.
#pragma once
#include <functional>
#include <memory>
#include <string>
#include <vector>

#include <adapter/adapter.h>
#include <adapter/parameters.h>
#include <adapter/time_handler.h>
#include <mi/device_vector.h>
#include <mi/timer_output.h>

namespace Nonlinear_Elasticity
{
  template <int dim>
  class Solid
  {
  public:
    Solid(const std::string &parameter_file);
    virtual ~Solid();
    void run();

    // role of DoFHandler<dim> in Adapter::initialize: the coupling vertices of the device mesh
    struct DoFSource
    {
      const mi::Device *dev;
      int               n_interface_nodes() const { return mi_n_interface_nodes(dev->ctx()); }
      void              interface_nodes(int *ids, double *xyz) const
      {
        dev->check(mi_get_interface_nodes(dev->ctx(), ids, xyz), "mi_get_interface_nodes");
      }
      // values of rank 0 to every rank of the decomposition (Adapter::RankZeroParticipant)
      std::function<void(double *, int)> broadcaster() const
      {
        const mi::Device *d = dev;
        return [d](double *v, int n) { d->check(mi_comm_broadcast(d->ctx(), v, n), "mi_comm_broadcast"); };
      }
    };

  private:
    void make_grid();
    void system_setup();
    void solve_nonlinear_timestep();
    void output_results() const;
    static void print_conv_header();
    void        print_conv_footer();

    const Parameters::AllParameters parameters;
    double                          vol_reference = 0.0, vol_current = 0.0;
    const unsigned int              degree;
    // ids of nonlinear_elasticity.cc:78 and nonlinear_elasticity.h:256-257
    const unsigned int boundary_interface_id        = 7;
    const unsigned int clamped_boundary_id          = 1;
    const unsigned int out_of_plane_clamped_mesh_id = 8;

    mi_mesh_desc                mesh_desc{};
    std::unique_ptr<mi::Device> device;
    mi::Vector total_displacement, total_displacement_old, velocity, velocity_old, acceleration, acceleration_old,
      external_stress;
    std::vector<mi::Vector *> state_variables;

    mutable mi::TimerOutput timer;
    Adapter::Time           time;
    Adapter::Adapter<dim, mi::Vector, Parameters::AllParameters> adapter;

    struct Errors // nonlinear_elasticity.h:293-315
    {
      double u = 1.0;
      void   reset() { u = 1.0; }
      void   normalise(const Errors &val)
      {
        if (val.u != 0.0)
          u /= val.u;
      }
    };
    Errors error_residual, error_residual_0, error_residual_norm, error_update, error_update_0, error_update_norm;

    // machine-readable companion of the Newton table: one JSON line per solved step in <Output folder>/steps.jsonl
    unsigned int last_newton_iterations = 0, last_lin_iterations = 0;
    bool         use_device_direct = true; // "Solver type = Direct": mi_direct_solve until it reports "too large"
    void         log_step_json() const;
  };
} // namespace Nonlinear_Elasticity
