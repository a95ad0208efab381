// Adapter -- couples the device-resident solid solver to preCICE (or the replay participant).
//
// Same public surface and call order as the reference's include/adapter/adapter.h:26-209 (constructor,
// initialize, read_data, advance, save_current_state_if_required, reload_old_state_if_required, public
// members `precice` and `deal_boundary_interface_id`).  What changes is where the data lives: VectorType is a
// handle to a vector in HBM, so the per-component IndexSet walk of format_deal_to_precice /
// format_precice_to_deal (:389-443) becomes one gather/scatter kernel plus an interface-sized copy.
//
// Requirements on the template arguments:
//   DoFSource (argument of initialize, role of DoFHandler<dim>): n_interface_nodes(), interface_nodes(ids, xyz)
//   VectorType: gather_interface / scatter_interface, copy-assignable, default-constructible
//   ParameterClass: participant_name, config_file, mesh_name, read_data_name, write_data_name
#pragma once
#include <iostream>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <vector>

#include "../mi/rank_identity.h"
#include "precice_participant.h"
#include "rank_zero_participant.h"
#include "time_handler.h"

namespace Adapter
{
  template <int dim, typename VectorType, typename ParameterClass>
  class Adapter
  {
  public:
    Adapter(const ParameterClass &parameters, const unsigned int deal_boundary_interface_id)
      : precice(parameters.participant_name, parameters.config_file, mi::host_rank(), mi::host_world_size())
      , deal_boundary_interface_id(deal_boundary_interface_id)
      , mesh_name(parameters.mesh_name)
      , read_data_name(parameters.read_data_name)
      , write_data_name(parameters.write_data_name)
    {}

    // adapter.h:229-342
    template <typename DoFSource>
    void initialize(const DoFSource &dof_handler, const VectorType &deal_to_precice)
    {
      // several ranks: what rank 0 reads from the coupling library reaches the others through the solver's communicator
      bind_broadcast(dof_handler, 0);
      if (dim != precice.getMeshDimensions(mesh_name))
        throw std::runtime_error("The dimension of your solver needs to be consistent with the dimension "
                                 "specified in your precice-config file. In case you run one of the tutorials, "
                                 "the dimension can be specified via make DIM=dim .");
      static_assert(dim > 1, "not implemented");

      // coupling vertices = support points of the x-component dofs on the interface, ascending (:250-321)
      n_interface_nodes = dof_handler.n_interface_nodes();
      std::cout << "\t Number of coupling nodes:     " << n_interface_nodes << std::endl;

      std::vector<double> interface_nodes_positions(dim * n_interface_nodes);
      std::vector<int>    node_ids(n_interface_nodes);
      write_data_buffer.resize(dim * n_interface_nodes);
      read_data_buffer.resize(dim * n_interface_nodes);
      interface_nodes_ids.resize(n_interface_nodes);
      dof_handler.interface_nodes(node_ids.data(), interface_nodes_positions.data());

      precice.setMeshVertices(mesh_name, interface_nodes_positions, interface_nodes_ids);

      if (precice.requiresInitialData())
        {
          format_deal_to_precice(deal_to_precice);
          precice.writeData(mesh_name, write_data_name, interface_nodes_ids, write_data_buffer);
        }
      precice.initialize();
    }

    // adapter.h:346-361
    void read_data(double relative_read_time, VectorType &precice_to_deal)
    {
      precice.readData(mesh_name, read_data_name, interface_nodes_ids, relative_read_time, read_data_buffer);
      format_precice_to_deal(precice_to_deal);
    }

    // adapter.h:365-385
    void advance(const VectorType &deal_to_precice, const double computed_timestep_length)
    {
      format_deal_to_precice(deal_to_precice);
      precice.writeData(mesh_name, write_data_name, interface_nodes_ids, write_data_buffer);
      precice.advance(computed_timestep_length);
    }

    // adapter.h:447-464
    void save_current_state_if_required(const std::vector<VectorType *> &state_variables, Time &time_class)
    {
      if (precice.requiresWritingCheckpoint())
        {
          old_state_data.resize(state_variables.size());
          for (unsigned i = 0; i < state_variables.size(); ++i)
            old_state_data[i] = *(state_variables[i]);
          old_time_value = time_class.current();
        }
    }

    // adapter.h:468-489
    void reload_old_state_if_required(std::vector<VectorType *> &state_variables, Time &time_class)
    {
      if (precice.requiresReadingCheckpoint())
        {
          if (state_variables.size() != old_state_data.size())
            throw std::runtime_error("state_variables are not the same as previously saved.");
          for (unsigned i = 0; i < state_variables.size(); ++i)
            *(state_variables[i]) = old_state_data[i];
          time_class.set_absolute_time(old_time_value);
        }
    }

    // adapter.h:136 `precice::Participant precice;` -- here the participant of rank 0 behind the same calls
    RankZeroParticipant precice;
    const unsigned int  deal_boundary_interface_id;

  private:
    const std::string mesh_name;
    const std::string read_data_name;
    const std::string write_data_name;

    // one preCICE-facing process, as in the reference (:152-154: this_mpi_process = 0, n_mpi_processes = 1), also when the
    // mesh is spread over several GPUs: RankZeroParticipant
    template <typename DoFSource>
    auto bind_broadcast(const DoFSource &d, int) -> decltype(d.broadcaster(), void())
    {
      precice.bind(d.broadcaster());
    }
    template <typename DoFSource>
    void bind_broadcast(const DoFSource &, long) // a DoF source without a communicator (single-rank tests)
    {}

    int                 n_interface_nodes = 0;
    std::vector<int>    interface_nodes_ids;
    std::vector<double> read_data_buffer;
    std::vector<double> write_data_buffer;

    std::vector<VectorType> old_state_data;
    double                  old_time_value = 0;

    // [x0,y0,(z0),x1,...] <- device vector (:389-417)
    void format_deal_to_precice(const VectorType &deal_to_precice)
    {
      deal_to_precice.gather_interface(write_data_buffer, n_interface_nodes);
    }
    // device vector <- [x0,y0,(z0),x1,...] (:421-443)
    void format_precice_to_deal(VectorType &precice_to_deal) const
    {
      precice_to_deal.scatter_interface(read_data_buffer, n_interface_nodes);
    }
  };
} // namespace Adapter
