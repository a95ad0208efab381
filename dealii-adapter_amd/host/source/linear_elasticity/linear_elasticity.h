// Linear_Elasticity::ElastoDynamics<dim> -- linear elastodynamics with the one-step-theta scheme, host driver.
// Counterpart of the reference's source/linear_elasticity (linear_elasticity.h:55-150, .cc:634-716): stiffness
// and mass are assembled once (mi_linear_setup), every step is rhs assembly (3 matrix-vector products folded
// into 2), a warm-started CG on M + theta^2 dt^2 K with the Dirichlet rows eliminated, and the displacement
// update -- all on the device (mi_linear_step).
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

namespace Linear_Elasticity
{
  template <int dim>
  class ElastoDynamics
  {
  public:
    ElastoDynamics(const std::string &parameter_file);
    ~ElastoDynamics();
    void run();

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
    void setup_system();
    void output_results() const;

    const Parameters::AllParameters parameters;
    // linear_elasticity.cc:57, :157-158
    const unsigned int interface_boundary_id        = 6;
    const unsigned int clamped_mesh_id              = 0;
    const unsigned int out_of_plane_clamped_mesh_id = 4;

    mi_mesh_desc                mesh_desc{};
    std::unique_ptr<mi::Device> device;
    bool                        solver_type_set = false; // "Solver type = Direct" passed on to the library
    mi::Vector                  old_velocity, velocity, old_displacement, displacement, old_stress, stress;
    std::vector<mi::Vector *>   state_variables;

    mutable mi::TimerOutput timer;
    Adapter::Time           time;
    Adapter::Adapter<dim, mi::Vector, Parameters::AllParameters> adapter;
  };
} // namespace Linear_Elasticity
