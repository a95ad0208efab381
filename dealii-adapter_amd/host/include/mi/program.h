// mi/program.h -- what the `elasticity` executable does after its command line is known: banner, output folder,
// model dispatch, error block and exit code of the reference's elasticity.cc:27-129, as a function, so that the same
// program can also be run by the rank threads of the multi-rank test harness (tests/fake_rccl/elasticity_ranks.cc).
#pragma once
#include <sys/stat.h>

#include <cerrno>
#include <cstdlib>
#include <iostream>
#include <string>

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

namespace mi
{
  // `Output folder` of the parameter file, created with all its parents (elasticity.cc:56-81: mode 0755)
  inline void make_output_folder(const std::string &folder)
  {
    std::string path = folder;
    if (path.empty() || path.back() != '/')
      path += '/';
    for (size_t at = path.find('/'); at != std::string::npos; at = path.find('/', at + 1))
      {
        if (at == 0)
          continue; // the root of an absolute path
        const std::string dir = path.substr(0, at);
        if (mkdir(dir.c_str(), 0755) != 0 && errno != EEXIST)
          throw std::runtime_error("Can't create: " + path);
      }
  }

  inline void print_banner(std::ostream &os)
  {
    const unsigned int n_threads = 1; // host side is single threaded; the parallelism is on the device
    const std::string  rev = GIT_SHORTREV, adapter_info = rev.empty() ? "unknown" : rev + " on branch " + GIT_BRANCH;
    std::string        device_info = "none";
    hipDeviceProp_t    prop;
    if (hipGetDeviceProperties(&prop, 0) == hipSuccess)
      device_info = std::string(prop.name) + " (" + prop.gcnArchName + ")";
    const std::string rule(77, '-');
    os << rule << std::endl << "--     . running with " << n_threads << " thread" << (n_threads == 1 ? "" : "s") << std::endl;
    os << "--     . adapter revision " << adapter_info << std::endl;
    os << "--     . device " << device_info << " (DIM=" << DIM << ")" << std::endl;
    const char *slabs = std::getenv("MI_SLABS");
    if (host_world_size() > 1)
      os << "--     . " << host_world_size()
         << " processes, one slab and one GPU each (RCCL; cut along the direction with most cell layers); rank 0 couples" << std::endl;
    else if (slabs && std::atoi(slabs) > 1)
      os << "--     . " << std::atoi(slabs) << " slabs emulated on one GPU (cut along the direction with most cell layers)" << std::endl;
    os << rule << std::endl << std::endl;
  }

  // returns the process exit code: 0, or 1 after the "Exception on processing:" block (elasticity.cc:101-126)
  inline int program(const std::string &parameter_file)
  {
    try
      {
        print_banner(std::cout);

        // two lenient partial parses before the solver reads everything strictly: output folder, then model (:51-55, :84-86)
        prm::Handler     prm;
        Parameters::Time time;
        time.add_output_parameters(prm);
        prm.parse_input(parameter_file, "", true);
        make_output_folder(time.output_folder);

        Parameters::Solver solver;
        solver.add_output_parameters(prm);
        prm.parse_input(parameter_file, "", true);

        if (solver.model == "neo-Hookean")
          Nonlinear_Elasticity::Solid<DIM>(parameter_file).run();
        else if (solver.model == "linear")
          Linear_Elasticity::ElastoDynamics<DIM>(parameter_file).run();
        else
          throw std::runtime_error("not implemented");
        return 0;
      }
    catch (std::exception &exc)
      {
        const std::string rule(52, '-');
        std::cerr << "\n\n" << rule << std::endl;
        std::cerr << "Exception on processing: " << std::endl << exc.what() << std::endl << "Aborting!" << std::endl << rule << std::endl;
      }
    catch (...)
      {
        const std::string rule(52, '-');
        std::cerr << "\n\n" << rule << std::endl;
        std::cerr << "Unknown exception!" << std::endl << "Aborting!" << std::endl << rule << std::endl;
      }
    return 1;
  }
} // namespace mi
