#include "parameters.h"

#include <stdexcept>

namespace Parameters
{
  void Time::add_output_parameters(prm::Handler &prm)
  {
    prm.enter_subsection("Time");
    prm.add_parameter("End time", end_time, "End time", prm::Double());
    prm.add_parameter("Time step size", delta_t, "Time step size", prm::Double());
    prm.add_parameter("Output interval", output_interval, "Write results every x timesteps", prm::Integer(0));
    prm.add_parameter("Output folder", output_folder, "Output folder", prm::Anything());
    prm.leave_subsection();
  }

  void System::add_output_parameters(prm::Handler &prm)
  {
    prm.enter_subsection("System properties");
    prm.add_parameter("Shear modulus", mu, "Shear modulus", prm::Double());
    prm.add_parameter("Poisson's ratio", nu, "Poisson's ratio", prm::Double(-1.0, 0.5));
    prm.add_parameter("rho", rho, "Density", prm::Double(0.0));
    prm.add_parameter("body forces", body_force, "Body forces x,y,z", prm::ListOfDoubles(3, 3));
    prm.leave_subsection();
  }

  void Solver::add_output_parameters(prm::Handler &prm)
  {
    prm.enter_subsection("Solver");
    prm.add_parameter("Model", model, "Structural model: linear or neo-Hookean", prm::Selection("linear|neo-Hookean"));
    prm.add_parameter("Solver type", type_lin, "Linear solver: CG or Direct", prm::Selection("CG|Direct"));
    prm.add_parameter("Residual", tol_lin, "CG residual (multiplied by residual norm)", prm::Double(0.0));
    prm.add_parameter("Max iteration multiplier", max_iterations_lin, "CG iterations (multiples of the system size)",
                      prm::Double(0.0));
    prm.add_parameter("Max iterations Newton-Raphson", max_iterations_NR, "Newton-Raphson iterations allowed",
                      prm::Integer(0));
    prm.add_parameter("Tolerance force", tol_f, "Force residual tolerance", prm::Double(0.0));
    prm.add_parameter("Tolerance displacement", tol_u, "Displacement error tolerance", prm::Double(0.0));
    prm.leave_subsection();
  }

  void Discretization::add_output_parameters(prm::Handler &prm)
  {
    prm.enter_subsection("Discretization");
    prm.add_parameter("Polynomial degree", poly_degree, "Polynomial degree of the FE system", prm::Integer(0));
    prm.add_parameter("theta", theta, "Time integration scheme (linear model)", prm::Double(0, 1));
    prm.add_parameter("beta", beta, "Newmark beta", prm::Double(0, 0.5));
    prm.add_parameter("gamma", gamma, "Newmark gamma", prm::Double(0, 1));
    prm.leave_subsection();
  }

  void PreciceAdapterConfiguration::add_output_parameters(prm::Handler &prm)
  {
    prm.enter_subsection("precice configuration");
    prm.add_parameter("Scenario", scenario, "Cases: FSI3, PF (perpendicular flap) or Block (synthetic box)",
                      prm::Selection("FSI3|PF|Block"));
    prm.add_parameter("precice config-file", config_file, "Name of the precice configuration file", prm::Anything());
    prm.add_parameter("Participant name", participant_name, "Name of the participant", prm::Anything());
    prm.add_parameter("Mesh name", mesh_name, "Name of the coupling mesh", prm::Anything());
    prm.add_parameter("Read data name", read_data_name, "Name of the read data", prm::Anything());
    prm.add_parameter("Write data name", write_data_name, "Name of the write data", prm::Anything());
    prm.add_parameter("Flap location", flap_location, "PF x-location", prm::Double(-3, 3));
    prm.leave_subsection();
  }

  void Block::add_output_parameters(prm::Handler &prm)
  {
    prm.enter_subsection("Block");
    prm.add_parameter("Repetitions", repetitions, "Cells per direction nx,ny,nz", prm::ListOfDoubles(2, 3));
    prm.add_parameter("Lower corner", lower, "x,y,z", prm::ListOfDoubles(3, 3));
    prm.add_parameter("Upper corner", upper, "x,y,z", prm::ListOfDoubles(3, 3));
    prm.leave_subsection();
  }

  AllParameters::AllParameters(const std::string &input_file)
  {
    prm::Handler prm;
    Solver::add_output_parameters(prm);
    Discretization::add_output_parameters(prm);
    System::add_output_parameters(prm);
    Time::add_output_parameters(prm);
    PreciceAdapterConfiguration::add_output_parameters(prm);
    Block::add_output_parameters(prm);

    prm.parse_input(input_file); // strict, parameters.cc:187

    lambda = 2 * mu * nu / (1 - 2 * nu); // :189

    // :192-200 -- the kind of read data is derived from its name
    if (read_data_name.find("Stress") == 0)
      data_consistent = true;
    else if (read_data_name.find("Force") == 0)
      data_consistent = false;
    else
      throw std::runtime_error(
        "Unknown read data type. Please use 'Force' or 'Stress' in the read data naming.");
  }
} // namespace Parameters
