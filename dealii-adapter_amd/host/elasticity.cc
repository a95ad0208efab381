// elasticity -- entry point of the solid solver; same command line, banner, output-folder handling, model
// dispatch and exit codes as the reference's elasticity.cc:7-129.
//   elasticity [parameters.prm]      (compile-time -DDIM=2|3, CMakeLists.txt:15-18)
#include <sys/stat.h>

#include <cerrno>
#include <cstdlib>
#include <iostream>
#include <string>
#include <thread>

#include <hip/hip_runtime_api.h>

#include <adapter/parameters.h>

#include "source/linear_elasticity/linear_elasticity.h"
#include "source/nonlinear_elasticity/nonlinear_elasticity.h"

#ifndef GIT_SHORTREV
#define GIT_SHORTREV ""
#endif
#ifndef GIT_BRANCH
#define GIT_BRANCH ""
#endif

int main(int argc, char **argv)
{
  try
    {
      const unsigned int n_threads = 1; // host side is single threaded; the parallelism is on the device
      // decomposed run (mi/device_vector.h): every rank executes the same program on global views; rank 0 speaks
      if (mi::host_rank() > 0)
        std::cout.setstate(std::ios_base::badbit);

      const std::string adapter_info =
        GIT_SHORTREV == std::string("") ? "unknown" : (GIT_SHORTREV + std::string(" on branch ") + GIT_BRANCH);
      std::string    device_info = "none";
      hipDeviceProp_t prop;
      if (hipGetDeviceProperties(&prop, 0) == hipSuccess)
        device_info = std::string(prop.name) + " (" + prop.gcnArchName + ")";

      std::cout << "-----------------------------------------------------------------------------" << std::endl
                << "--     . running with " << n_threads << " thread" << (n_threads == 1 ? "" : "s") << std::endl;
      std::cout << "--     . adapter revision " << adapter_info << std::endl;
      std::cout << "--     . device " << device_info << " (DIM=" << DIM << ")" << std::endl;
      if (mi::host_world_size() > 1)
        std::cout << "--     . " << mi::host_world_size() << " processes, one slab and one GPU each (RCCL; cut along the direction with most cell layers)" << std::endl;
      else if (std::getenv("MI_SLABS") && std::atoi(std::getenv("MI_SLABS")) > 1)
        std::cout << "--     . " << std::atoi(std::getenv("MI_SLABS")) << " slabs emulated on one GPU (cut along the direction with most cell layers)" << std::endl;
      std::cout << "-----------------------------------------------------------------------------" << std::endl
                << std::endl;

      const std::string parameter_file = argc > 1 ? argv[1] : "parameters.prm";

      // lenient partial parse for the output folder (elasticity.cc:51-55), then mkdir -p (:56-81)
      prm::Handler     prm;
      Parameters::Time time;
      time.add_output_parameters(prm);
      prm.parse_input(parameter_file, "", true);

      std::string pathname = time.output_folder;
      if (pathname.empty() || pathname[pathname.size() - 1] != '/')
        pathname += '/';
      size_t       pre = 0, pos;
      const mode_t mode = S_IRWXU | S_IRGRP | S_IXGRP | S_IROTH | S_IXOTH;
      while ((pos = pathname.find_first_of('/', pre)) != std::string::npos)
        {
          const std::string subdir = pathname.substr(0, pos++);
          pre                      = pos;
          if (subdir.size() == 0)
            continue;
          if (mkdir(subdir.c_str(), mode) && errno != EEXIST)
            throw std::runtime_error("Can't create: " + pathname);
        }

      // lenient partial parse for the model (:84-86)
      Parameters::Solver solver;
      solver.add_output_parameters(prm);
      prm.parse_input(parameter_file, "", true);

      if (solver.model == "neo-Hookean")
        {
          Nonlinear_Elasticity::Solid<DIM> solid(parameter_file);
          solid.run();
        }
      else if (solver.model == "linear")
        {
          Linear_Elasticity::ElastoDynamics<DIM> elastic_solver(parameter_file);
          elastic_solver.run();
        }
      else
        throw std::runtime_error("not implemented");
    }
  catch (std::exception &exc)
    {
      std::cerr << std::endl
                << std::endl
                << "----------------------------------------------------" << std::endl;
      std::cerr << "Exception on processing: " << std::endl
                << exc.what() << std::endl
                << "Aborting!" << std::endl
                << "----------------------------------------------------" << std::endl;
      return 1;
    }
  catch (...)
    {
      std::cerr << std::endl
                << std::endl
                << "----------------------------------------------------" << std::endl;
      std::cerr << "Unknown exception!" << std::endl
                << "Aborting!" << std::endl
                << "----------------------------------------------------" << std::endl;
      return 1;
    }
  return 0;
}
