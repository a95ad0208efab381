// elasticity -- entry point of the solid solver; same command line, banner, output-folder handling, model
// dispatch and exit codes as the reference's elasticity.cc:7-129 (the program itself: include/mi/program.h).
//   elasticity [parameters.prm]      (compile-time -DDIM=2|3, CMakeLists.txt:15-18)
#include <mi/program.h>

int main(int argc, char **argv)
{
  // decomposed run (mi/rank_identity.h): every rank executes the same program on global views; rank 0 speaks
  if (mi::host_rank() > 0)
    std::cout.setstate(std::ios_base::badbit);
  return mi::program(argc > 1 ? argv[1] : "parameters.prm");
}
