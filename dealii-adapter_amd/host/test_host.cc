// CPU-only unit tests of the host-side boundary code (no device calls): parameter file reader, Adapter::Time,
// replay participant, Adapter call sequence with a mock vector type.  Run by tests/test_host_cpu.py.
#include <cassert>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <functional>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <filesystem>
#include <unistd.h>
#include <fstream>
#include <iostream>
#include <stdexcept>
#include <string>
#include <vector>

#include <adapter/adapter.h>
#include <adapter/parameters.h>
#include <adapter/time_handler.h>
#include <mi/vtk_lagrange.h>

static int g_fail = 0;
#define CHECK(cond)                                                          \
  do                                                                         \
    {                                                                        \
      if (!(cond))                                                           \
        {                                                                    \
          std::printf("CHECK FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); \
          ++g_fail;                                                          \
        }                                                                    \
    }                                                                        \
  while (0)

static void write_file(const std::string &name, const std::string &txt)
{
  std::ofstream f(name);
  f << txt;
}

template <typename F>
static bool throws(F f, const std::string &needle = "")
{
  try
    {
      f();
    }
  catch (const std::exception &e)
    {
      return needle.empty() || std::string(e.what()).find(needle) != std::string::npos;
    }
  return false;
}

// mock of the device vector: a host array, "interface" = first n*dim entries
struct MockVector
{
  std::vector<double> v = std::vector<double>(12, 0.0);
  void gather_interface(std::vector<double> &out, int n) const
  {
    for (size_t i = 0; i < out.size(); ++i)
      out[i] = v[i];
    (void)n;
  }
  void scatter_interface(const std::vector<double> &in, int n)
  {
    for (size_t i = 0; i < in.size(); ++i)
      v[i] = in[i];
    (void)n;
  }
};
struct MockDofs
{
  int  n_interface_nodes() const { return 3; }
  void interface_nodes(int *ids, double *xyz) const
  {
    for (int i = 0; i < 3; ++i)
      {
        ids[i]         = 10 + i;
        xyz[2 * i]     = 0.1 * i;
        xyz[2 * i + 1] = 1.0;
      }
  }
};

static const char *PRM = R"(# test file
subsection Time
  set End time              = 10
  set Time step size        = 0.005   # trailing comment
  set Output interval       = 10
   set Output folder   = dealii-output
end
subsection Discretization
  set Polynomial degree   = 3
end
subsection System properties
  set Poisson's ratio = 0.4
  set Shear modulus   = 0.5e6
  set rho	      = 1000
  set body forces     = 0.0,-9.81,0.0
end
subsection Solver
  set Model                     = linear
  set Solver type               = Direct
  set Max iteration multiplier  = 1
  set Residual                  = 1e-6
  set Max iterations Newton-Raphson = 10
  set Tolerance displacement        = 1.0e-6
  set Tolerance force               = 1.0e-9
end
subsection precice configuration
  set Scenario            = FSI3
  set precice config-file = precice-config.xml
  set Participant name    = Solid
  set Mesh name           = Solid-Mesh
  set Read data name      = Stress
  set Write data name     = Displacement
end
)";

// VTK's node order of Lagrange cells (higher-order output, nonlinear_elasticity.cc:1222-1225): a bijection for every
// order, the linear cell's order at order 1, and the 27 positions of the quadratic hexahedron by name
static void test_vtk_lagrange_order()
{
  for (int dim = 2; dim <= 3; ++dim)
    for (int order = 1; order <= 4; ++order)
      {
        const int         n1 = order + 1, npc = dim == 2 ? n1 * n1 : n1 * n1 * n1;
        std::vector<int> seen((size_t)npc, 0);
        for (int a = 0; a < npc; ++a)
          {
            const int v = mi::vtk_detail::lagrange_index(dim, order, a % n1, (a / n1) % n1, dim == 3 ? a / (n1 * n1) : 0);
            CHECK(v >= 0 && v < npc);
            if (v >= 0 && v < npc)
              ++seen[size_t(v)];
          }
        for (int v = 0; v < npc; ++v)
          CHECK(seen[size_t(v)] == 1);
      }
  auto L3 = [](int i, int j, int k) { return mi::vtk_detail::lagrange_index(3, 2, i, j, k); };
  // corners: the linear hexahedron's order
  CHECK(L3(0, 0, 0) == 0 && L3(2, 0, 0) == 1 && L3(2, 2, 0) == 2 && L3(0, 2, 0) == 3);
  CHECK(L3(0, 0, 2) == 4 && L3(2, 0, 2) == 5 && L3(2, 2, 2) == 6 && L3(0, 2, 2) == 7);
  // edge midpoints: bottom face x y x y, top face, the four vertical edges (x0y0, x1y0, x0y1, x1y1)
  CHECK(L3(1, 0, 0) == 8 && L3(2, 1, 0) == 9 && L3(1, 2, 0) == 10 && L3(0, 1, 0) == 11);
  CHECK(L3(1, 0, 2) == 12 && L3(2, 1, 2) == 13 && L3(1, 2, 2) == 14 && L3(0, 1, 2) == 15);
  CHECK(L3(0, 0, 1) == 16 && L3(2, 0, 1) == 17 && L3(0, 2, 1) == 18 && L3(2, 2, 1) == 19);
  // face centres x- x+ y- y+ z- z+, body centre
  CHECK(L3(0, 1, 1) == 20 && L3(2, 1, 1) == 21 && L3(1, 0, 1) == 22 && L3(1, 2, 1) == 23 && L3(1, 1, 0) == 24 && L3(1, 1, 2) == 25);
  CHECK(L3(1, 1, 1) == 26);
  auto L2 = [](int i, int j) { return mi::vtk_detail::lagrange_index(2, 3, i, j, 0); };
  CHECK(L2(0, 0) == 0 && L2(3, 0) == 1 && L2(3, 3) == 2 && L2(0, 3) == 3);
  CHECK(L2(1, 0) == 4 && L2(2, 0) == 5 && L2(3, 1) == 6 && L2(3, 2) == 7 && L2(1, 3) == 8 && L2(2, 3) == 9 && L2(0, 1) == 10 && L2(0, 2) == 11);
  CHECK(L2(1, 1) == 12 && L2(2, 1) == 13 && L2(1, 2) == 14 && L2(2, 2) == 15);
}

int main()
{
  test_vtk_lagrange_order();
  // every file the tests write goes into a scratch directory of their own (under $TMPDIR), removed at the end
  std::string scratch = std::string(std::getenv("TMPDIR") ? std::getenv("TMPDIR") : "/tmp") + "/mi_test_host_XXXXXX";
  if (!mkdtemp(scratch.data()) || chdir(scratch.c_str()) != 0)
    {
      std::perror("test_host: scratch directory");
      return 2;
    }
  // ---------------- Adapter::Time (time_handler.h:21-84)
  {
    Adapter::Time t(10.0, 0.005);
    for (int i = 0; i < 7; ++i)
      t.increment();
    CHECK(t.get_timestep() == 7 && std::abs(t.current() - 0.035) < 1e-15);
    t.set_absolute_time(0.015);
    CHECK(t.get_timestep() == 3 && t.current() == 0.015);
    t.set_absolute_time(0.005 * 2.9999999999); // rounds at 1e-10, then truncates
    CHECK(t.get_timestep() == 2);
    t.set_absolute_time(0.005 * 2.99999999999);
    CHECK(t.get_timestep() == 3);
    CHECK(t.end() == 10.0 && t.get_delta_t() == 0.005);
  }
  // ---------------- parameter files (parameters.cc:8-205)
  {
    write_file("t_ok.prm", PRM);
    Parameters::AllParameters p("t_ok.prm");
    CHECK(p.end_time == 10 && p.delta_t == 0.005 && p.output_interval == 10 && p.output_folder == "dealii-output");
    CHECK(p.poly_degree == 3 && p.theta == 0.5 && p.beta == 0.25 && p.gamma == 0.5); // defaults kept
    CHECK(p.nu == 0.4 && p.mu == 0.5e6 && p.rho == 1000 && p.body_force[1] == -9.81);
    CHECK(std::abs(p.lambda - 2 * 0.5e6 * 0.4 / (1 - 0.8)) < 1e-6); // :189
    CHECK(p.model == "linear" && p.type_lin == "Direct" && p.max_iterations_NR == 10);
    CHECK(p.scenario == "FSI3" && p.participant_name == "Solid" && p.mesh_name == "Solid-Mesh");
    CHECK(p.data_consistent);
    // "Force..." read data -> conservative (:194-195); anything else is an error (:196-200)
    std::string s = PRM;
    s.replace(s.find("= Stress"), 8, "= Force-Data");
    write_file("t_force.prm", s);
    CHECK(!Parameters::AllParameters("t_force.prm").data_consistent);
    s = PRM;
    s.replace(s.find("= Stress"), 8, "= Pressure");
    write_file("t_bad.prm", s);
    CHECK(throws([] { Parameters::AllParameters("t_bad.prm"); }, "Unknown read data type"));
    // strict parse: unknown key / subsection, pattern violations
    s = PRM;
    s.replace(s.find("set rho"), 7, "set rhx");
    write_file("t_key.prm", s);
    CHECK(throws([] { Parameters::AllParameters("t_key.prm"); }, "No entry with name"));
    s = std::string(PRM) + "subsection Linear solver\n  set x = 1\nend\n";
    write_file("t_sec.prm", s);
    CHECK(throws([] { Parameters::AllParameters("t_sec.prm"); }, "no such subsection"));
    s = PRM;
    s.replace(s.find("= 0.4"), 5, "= 0.6");
    write_file("t_nu.prm", s);
    CHECK(throws([] { Parameters::AllParameters("t_nu.prm"); }, "does not match its pattern"));
    s = PRM;
    s.replace(s.find("= linear"), 8, "= plastic");
    write_file("t_model.prm", s);
    CHECK(throws([] { Parameters::AllParameters("t_model.prm"); }, "pattern"));
    CHECK(throws([] { Parameters::AllParameters("does-not-exist.prm"); }, "Cannot open"));
    // lenient partial parse (elasticity.cc:51-55, :84-86): foreign sections and keys are skipped
    prm::Handler     h;
    Parameters::Time time;
    time.add_output_parameters(h);
    h.parse_input("t_ok.prm", "", true);
    CHECK(time.output_folder == "dealii-output" && time.delta_t == 0.005);
    // additive Block scenario
    s = PRM;
    s.replace(s.find("= FSI3"), 6, "= Block");
    s += "subsection Block\n  set Repetitions = 4, 5, 6\n  set Upper corner = 2,1,1\nend\n";
    write_file("t_block.prm", s);
    Parameters::AllParameters pb("t_block.prm");
    CHECK(pb.scenario == "Block" && pb.repetitions[0] == 4 && pb.repetitions[2] == 6 && pb.upper[0] == 2.0);
  }
  // ---------------- replay participant + Adapter call order (adapter.h:229-489)
  {
    write_file("t_explicit.xml", R"(<precice-configuration dimensions="2">
  <!-- replay: read-data = ramp 4 0 -40 0 -->
  <!-- replay: write-log = t_explicit.log -->
  <coupling-scheme:serial-explicit>
    <max-time value="0.03" /> <time-window-size value="0.01" />
  </coupling-scheme:serial-explicit>
</precice-configuration>)");
    struct P
    {
      std::string participant_name = "Solid", config_file = "t_explicit.xml", mesh_name = "Solid-Mesh",
                  read_data_name = "Stress", write_data_name = "Displacement";
    } par;
    Adapter::Adapter<2, MockVector, P> ad(par, 7);
    CHECK(ad.deal_boundary_interface_id == 7);
    MockVector   u, stress;
    Adapter::Time time(1.0, 0.01);
    ad.initialize(MockDofs(), u);
    std::vector<MockVector *> state = {&u};
    int                       steps = 0;
    while (ad.precice.isCouplingOngoing())
      {
        ad.save_current_state_if_required(state, time);
        time.increment();
        CHECK(std::abs(ad.precice.getMaxTimeStepSize() - 0.01) < 1e-15);
        ad.read_data(0.01, stress);
        // ramp over 4 windows, read at the END of the window: y traction = -40 * (k+1)/4
        CHECK(std::abs(stress.v[1] - (-40.0 * std::min(1.0, (steps + 1) / 4.0))) < 1e-12 && stress.v[0] == 0.0);
        CHECK(stress.v[3] == stress.v[1] && stress.v[5] == stress.v[1]);
        u.v[0] = 1.0 + steps;
        ad.advance(u, 0.01);
        ad.reload_old_state_if_required(state, time);
        CHECK(ad.precice.isTimeWindowComplete());
        ++steps;
      }
    ad.precice.finalize();
    CHECK(steps == 3 && time.get_timestep() == 3);
    std::ifstream log("t_explicit.log");
    std::string   line;
    int           rows = 0;
    while (std::getline(log, line))
      if (!line.empty() && line[0] != '#')
        ++rows;
    CHECK(rows == 3);
    // wrong dimension is rejected at initialize (:235-242)
    Adapter::Adapter<3, MockVector, P> ad3(par, 7);
    CHECK(throws([&] { ad3.initialize(MockDofs(), u); }, "dimension"));
  }
  {
    write_file("t_implicit.xml", R"(<precice-configuration dimensions="2">
  <!-- replay: read-data = constant 0 -8 0 -->
  <coupling-scheme:serial-implicit>
    <max-time-windows value="2" /> <time-window-size value="0.01" /> <max-iterations value="3" />
  </coupling-scheme:serial-implicit>
</precice-configuration>)");
    struct P
    {
      std::string participant_name = "Solid", config_file = "t_implicit.xml", mesh_name = "m", read_data_name = "Stress",
                  write_data_name = "Displacement";
    } par;
    Adapter::Adapter<2, MockVector, P> ad(par, 7);
    MockVector   u, stress;
    Adapter::Time time(1.0, 0.01);
    ad.initialize(MockDofs(), u);
    std::vector<MockVector *> state = {&u};
    int                       calls = 0, completed = 0;
    std::vector<double>       seen;
    while (ad.precice.isCouplingOngoing())
      {
        ad.save_current_state_if_required(state, time);
        time.increment();
        ad.read_data(0.01, stress);
        seen.push_back(stress.v[1]);
        u.v[0] += 1.0; // "solve": the state advances every call ...
        ad.advance(u, 0.01);
        ad.reload_old_state_if_required(state, time); // ... and is rewound unless the window converged
        if (ad.precice.isTimeWindowComplete())
          ++completed;
        ++calls;
        CHECK(calls < 50);
      }
    CHECK(calls == 6 && completed == 2);
    CHECK(time.get_timestep() == 2 && std::abs(time.current() - 0.02) < 1e-15); // set_absolute_time rewinds
    CHECK(std::abs(u.v[0] - 2.0) < 1e-15); // one net advance per window: checkpoints restored twice per window
    CHECK(std::abs(seen[0] + 4.0) < 1e-12 && std::abs(seen[1] + 6.0) < 1e-12 && std::abs(seen[2] + 8.0) < 1e-12);
  }
  // vector count mismatch between save and reload is an error (:478-480)
  {
    struct P
    {
      std::string participant_name = "Solid", config_file = "t_implicit.xml", mesh_name = "m", read_data_name = "Stress",
                  write_data_name = "Displacement";
    } par;
    Adapter::Adapter<2, MockVector, P> ad(par, 7);
    MockVector   u, w;
    Adapter::Time time(1.0, 0.01);
    ad.initialize(MockDofs(), u);
    std::vector<MockVector *> one = {&u}, two = {&u, &w};
    ad.save_current_state_if_required(one, time);
    time.increment();
    ad.advance(u, 0.01);
    CHECK(throws([&] { ad.reload_old_state_if_required(two, time); }, "not the same as previously saved"));
  }
  // several ranks, ONE participant (round 4; adapter.h:152-154, 213-225): two rank THREADS drive an Adapter each through an
  // implicit-coupling run; rank 0 owns the participant, rank 1 receives read data and every coupling decision through the
  // broadcast its DoF source provides (here: a buffer in shared memory and a two-thread barrier standing in for
  // mi_comm_broadcast).  Both ranks must see the same call results, and only one participant may write the log.
  {
    write_file("t_ranks.xml", R"(<precice-configuration dimensions="2">
  <!-- replay: read-data = constant 0 -8 0 -->
  <!-- replay: write-log = t_ranks.log -->
  <coupling-scheme:serial-implicit>
    <max-time-windows value="2" /> <time-window-size value="0.01" /> <max-iterations value="3" />
  </coupling-scheme:serial-implicit>
</precice-configuration>)");
    struct P
    {
      std::string participant_name = "Solid", config_file = "t_ranks.xml", mesh_name = "m", read_data_name = "Stress",
                  write_data_name = "Displacement";
    };
    struct Shared
    {
      std::mutex              m;
      std::condition_variable cv;
      int                     arrived = 0, generation = 0;
      std::vector<double>     buf;
      void barrier()
      {
        std::unique_lock<std::mutex> lk(m);
        const int                    g = generation;
        if (++arrived == 2)
          {
            arrived = 0;
            ++generation;
            cv.notify_all();
          }
        else
          cv.wait(lk, [&] { return generation != g; });
      }
    } sh;
    struct RankDofs : MockDofs
    {
      Shared *sh;
      int     rank;
      std::function<void(double *, int)> broadcaster() const
      {
        Shared   *s = sh;
        const int r = rank;
        return [s, r](double *v, int n) {
          if (r == 0)
            s->buf.assign(v, v + n);
          s->barrier(); // rank 0 has published
          if (r != 0)
            std::copy(s->buf.begin(), s->buf.begin() + n, v);
          s->barrier(); // everyone has read: the buffer may be reused
        };
      }
    };
    std::vector<std::vector<double>> seen(2);
    std::vector<int>                 calls(2, 0), completed(2, 0), steps(2, 0);
    std::vector<std::string>         errors(2);
    auto rank_main = [&](int r) {
      try
        {
          mi::thread_identity() = mi::RankIdentity{r, 2, nullptr, 0};
          P                                  par;
          Adapter::Adapter<2, MockVector, P> ad(par, 7);
          MockVector                         u, stress;
          Adapter::Time                      time(1.0, 0.01);
          RankDofs                           dofs;
          dofs.sh   = &sh;
          dofs.rank = r;
          ad.initialize(dofs, u);
          std::vector<MockVector *> state = {&u};
          while (ad.precice.isCouplingOngoing())
            {
              ad.save_current_state_if_required(state, time);
              time.increment();
              if (std::abs(ad.precice.getMaxTimeStepSize() - 0.01) > 1e-15)
                errors[size_t(r)] = "window size";
              ad.read_data(0.01, stress);
              seen[size_t(r)].push_back(stress.v[1]);
              u.v[0] += 1.0;
              ad.advance(u, 0.01);
              ad.reload_old_state_if_required(state, time);
              if (ad.precice.isTimeWindowComplete())
                ++completed[size_t(r)];
              if (++calls[size_t(r)] > 50)
                break;
            }
          ad.precice.finalize();
          steps[size_t(r)] = int(time.get_timestep());
          mi::thread_identity() = mi::RankIdentity{};
        }
      catch (std::exception &e)
        {
          errors[size_t(r)] = e.what();
        }
    };
    std::thread t1(rank_main, 1);
    rank_main(0);
    t1.join();
    CHECK(errors[0].empty() && errors[1].empty());
    CHECK(calls[0] == 6 && calls[1] == 6 && completed[0] == 2 && completed[1] == 2 && steps[0] == 2 && steps[1] == 2);
    CHECK(seen[0] == seen[1] && seen[0].size() == 6 && std::abs(seen[1][0] + 4.0) < 1e-12 && std::abs(seen[1][2] + 8.0) < 1e-12);
    std::ifstream log("t_ranks.log");
    std::string   line;
    int           rows = 0;
    while (std::getline(log, line))
      rows += (!line.empty() && line[0] != '#');
    CHECK(rows == 2); // one participant, one row per completed window
    // an error of the coupling library on rank 0 (here: a per-vertex trace that misses a vertex, thrown by
    // setMeshVertices) ends BOTH ranks at the next collective instead of leaving rank 1 blocked in it
    {
      write_file("t_ranks_bad.txt", "0.01 0 1 10\n0.01 1 2 20\n");
      write_file("t_ranks_bad.xml", R"(<precice-configuration dimensions="2">
  <!-- replay: read-data = vertex-trace t_ranks_bad.txt -->
  <coupling-scheme:serial-explicit><max-time-windows value="1" /><time-window-size value="0.01" /></coupling-scheme:serial-explicit>
</precice-configuration>)");
      struct PB
      {
        std::string participant_name = "Solid", config_file = "t_ranks_bad.xml", mesh_name = "m", read_data_name = "Stress",
                    write_data_name = "Displacement";
      };
      std::vector<std::string> what(2);
      auto failing_rank = [&](int r) {
        try
          {
            mi::thread_identity() = mi::RankIdentity{r, 2, nullptr, 0};
            PB                                  par;
            Adapter::Adapter<2, MockVector, PB> ad(par, 7);
            MockVector                          u;
            RankDofs                            dofs;
            dofs.sh   = &sh;
            dofs.rank = r;
            ad.initialize(dofs, u);
            what[size_t(r)] = "no exception";
          }
        catch (std::exception &e)
          {
            what[size_t(r)] = e.what();
          }
        mi::thread_identity() = mi::RankIdentity{};
      };
      std::thread t2(failing_rank, 1);
      failing_rank(0);
      t2.join();
      CHECK(what[0].find("every vertex exactly once") != std::string::npos);
      CHECK(what[1].find("rank 0") != std::string::npos && what[1].find("rank 1 stops") != std::string::npos);
    }
    // several ranks but no broadcast bound: refused, not silently wrong
    mi::thread_identity() = mi::RankIdentity{1, 2, nullptr, 0};
    {
      P                                  par;
      Adapter::Adapter<2, MockVector, P> ad(par, 7);
      MockVector                         u;
      CHECK(throws([&] { ad.initialize(MockDofs(), u); }, "no broadcast bound"));
    }
    mi::thread_identity() = mi::RankIdentity{};
  }
  // per-vertex trace replay: rows "t vertex fx fy", linear interpolation in time per vertex, constant outside the
  // recorded range; the coupling-mesh vertices are written out for whoever records such a trace
  {
    write_file("t_vtrace.xml", R"(<precice-configuration dimensions="2">
  <!-- replay: read-data = vertex-trace t_vtrace.txt -->
  <!-- replay: write-vertices = t_vertices.txt -->
  <coupling-scheme:serial-explicit><max-time-windows value="4" /><time-window-size value="0.01" /></coupling-scheme:serial-explicit>
</precice-configuration>)");
    write_file("t_vtrace.txt", "# t vertex fx fy\n0.01 0 1 10\n0.01 1 2 20\n0.01 2 3 30\n0.03 0 3 30\n0.03 1 4 40\n0.03 2 5 50\n");
    precice::Participant pp("Solid", "t_vtrace.xml", 0, 1);
    std::vector<double>  pos = {0.0, 0.0, 0.5, 0.0, 1.0, 0.25}, val(6);
    std::vector<int>     ids(3);
    pp.setMeshVertices("m", pos, ids);
    pp.initialize();
    pp.readData("m", "Stress", ids, 0.01, val); // t = 0.01: first frame
    CHECK(val[0] == 1 && val[1] == 10 && val[4] == 3 && val[5] == 30);
    pp.advance(0.01);
    pp.readData("m", "Stress", ids, 0.01, val); // t = 0.02: half way
    CHECK(std::abs(val[0] - 2.0) < 1e-13 && std::abs(val[3] - 30.0) < 1e-12 && std::abs(val[5] - 40.0) < 1e-12);
    pp.advance(0.01);
    pp.advance(0.01);
    pp.readData("m", "Stress", ids, 0.01, val); // t = 0.04: beyond the last frame
    CHECK(val[0] == 3 && val[5] == 50);
    std::ifstream vf("t_vertices.txt");
    std::string   line;
    int           rows = 0;
    while (std::getline(vf, line))
      rows += (!line.empty() && line[0] != '#');
    CHECK(rows == 3);
    write_file("t_vtrace_bad.txt", "0.01 0 1 10\n0.01 1 2 20\n");
    write_file("t_vtrace_bad.xml", R"(<precice-configuration dimensions="2">
  <!-- replay: read-data = vertex-trace t_vtrace_bad.txt -->
  <coupling-scheme:serial-explicit><max-time-windows value="1" /><time-window-size value="0.01" /></coupling-scheme:serial-explicit>
</precice-configuration>)");
    precice::Participant bad("Solid", "t_vtrace_bad.xml", 0, 1);
    CHECK(throws([&] { bad.setMeshVertices("m", pos, ids); }, "every vertex exactly once"));
  }
  std::printf(g_fail ? "HOST TESTS FAILED (%d)\n" : "HOST TESTS OK\n", g_fail);
  if (chdir("/") == 0)
    std::filesystem::remove_all(scratch);
  return g_fail ? 1 : 0;
}
