// elasticity_ranks -- TEST HARNESS: the `elasticity` program (host/include/mi/program.h) run by N rank THREADS of one
// process, each with its own slab of the mesh, through the library's RCCL branch against the RCCL test double
// (fake_rccl.cpp).  A single-GPU box cannot host two RCCL ranks, so this is how the multi-rank path of the executables --
// mi::Device with a rank identity, Adapter::RankZeroParticipant (rank 0 owns the participant, the others receive what it
// reads through mi_comm_broadcast), rank-0-only output -- is exercised end to end.
//   elasticity_ranks N [parameters.prm]        (compile-time -DDIM, optionally -DMI_WITH_PRECICE)
#include <atomic>
#include <streambuf>
#include <thread>
#include <vector>

#include <mi/program.h>

namespace
{
  thread_local bool t_quiet = false;
  // std::cout of a quiet thread goes nowhere (ranks > 0 of a process-per-rank run have their stdout closed instead)
  struct RankFilter : std::streambuf
  {
    std::streambuf *to;
    explicit RankFilter(std::streambuf *t)
      : to(t)
    {}
    int_type overflow(int_type c) override { return t_quiet || c == traits_type::eof() ? traits_type::not_eof(c) : to->sputc(char(c)); }
    std::streamsize xsputn(const char *s, std::streamsize n) override { return t_quiet ? n : to->sputn(s, n); }
    int sync() override { return to->pubsync(); }
  };
} // namespace

int main(int argc, char **argv)
{
  const int         world = argc > 1 ? std::atoi(argv[1]) : 0;
  const std::string prm   = argc > 2 ? argv[2] : "parameters.prm";
  if (world < 1)
    {
      std::cerr << "usage: elasticity_ranks N [parameters.prm]" << std::endl;
      return 2;
    }
  unsigned char uid[128] = {0};
  if (mi_comm_unique_id(uid) != MI_OK)
    {
      std::cerr << "mi_comm_unique_id: " << mi_last_error(nullptr) << std::endl;
      return 2;
    }
  RankFilter filter(std::cout.rdbuf());
  std::cout.rdbuf(&filter);
  std::vector<int>         rc(size_t(world), 1);
  std::vector<std::thread> ranks;
  for (int r = 0; r < world; ++r)
    ranks.emplace_back([&, r] {
      mi::thread_identity() = mi::RankIdentity{r, world, uid, 0};
      t_quiet               = r > 0;
      rc[size_t(r)]         = mi::program(prm);
    });
  for (std::thread &t : ranks)
    t.join();
  std::cout.rdbuf(filter.to);
  int worst = 0;
  for (int v : rc)
    worst = worst ? worst : v;
  return worst;
}
