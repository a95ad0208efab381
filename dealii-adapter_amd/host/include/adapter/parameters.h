// Parameters -- the structs filled from parameters.prm; same members, keys, defaults and validation as the
// reference's include/adapter/parameters.{h,cc} (key table: SURVEY.md 5.1).  One additive subsection ("Block")
// describes the synthetic box meshes of the benchmark configurations; shipped files parse unchanged.
#pragma once
#include <string>

#include "parameter_handler.h"

namespace Parameters
{
  struct Time // parameters.h:17-27, parameters.cc:8-27
  {
    double      end_time        = 1;
    double      delta_t         = 0.1;
    int         output_interval = 1;
    std::string output_folder   = "";
    void        add_output_parameters(prm::Handler &prm);
  };

  struct System // parameters.h:32-42, parameters.cc:29-53
  {
    double nu            = 0.3;
    double mu            = 1538462;
    double lambda        = -1;
    double rho           = 1000;
    double body_force[3] = {0, 0, 0};
    void   add_output_parameters(prm::Handler &prm);
  };

  struct Solver // parameters.h:48-60, parameters.cc:55-102
  {
    std::string  model              = "linear";
    std::string  type_lin           = "Direct";
    double       tol_lin            = 1e-6;
    double       max_iterations_lin = 1;
    unsigned int max_iterations_NR  = 10;
    double       tol_f              = 1e-9;
    double       tol_u              = 1e-6;
    void         add_output_parameters(prm::Handler &prm);
  };

  struct Discretization // parameters.h:68-79, parameters.cc:104-128
  {
    unsigned int poly_degree = 3;
    double       theta       = 0.5;  // linear model (theta scheme)
    double       beta        = 0.25; // nonlinear model (Newmark)
    double       gamma       = 0.5;
    void         add_output_parameters(prm::Handler &prm);
  };

  struct PreciceAdapterConfiguration // parameters.h:87-100, parameters.cc:130-174
  {
    std::string scenario         = "FSI3";
    std::string config_file      = "precice-config.xml";
    std::string participant_name = "dealiisolver";
    std::string mesh_name        = "dealii-mesh";
    std::string read_data_name   = "Stress";
    std::string write_data_name  = "Displacement";
    double      flap_location    = 0.0;
    bool        data_consistent  = true;
    void        add_output_parameters(prm::Handler &prm);
  };

  // additive: geometry of "Scenario = Block" (clamped x-, coupling interface on the other sides)
  struct Block
  {
    int    repetitions[3] = {8, 8, 8};
    double lower[3]       = {0, 0, 0};
    double upper[3]       = {1, 1, 1};
    void   add_output_parameters(prm::Handler &prm);
  };

  struct AllParameters : public Solver,
                         public Discretization,
                         public System,
                         public Time,
                         public PreciceAdapterConfiguration,
                         public Block
  {
    AllParameters(const std::string &input_file);
  };
} // namespace Parameters
