// TimerOutput -- wall-clock section timer with a summary table at destruction, standing in for
// dealii::TimerOutput(std::cout, summary, wall_times) (nonlinear_elasticity.cc:79).  Section names are the
// reference's ("Setup system", "Assemble linear system", "Linear solver", "Advance adapter", "Output results").
#pragma once
#include <chrono>
#include <iomanip>
#include <iostream>
#include <map>
#include <string>
#include <vector>

namespace mi
{
  class TimerOutput
  {
  public:
    explicit TimerOutput(std::ostream &out)
      : out_(out)
      , start_(clock::now())
    {}
    ~TimerOutput() { print_summary(); }

    void enter_subsection(const std::string &name)
    {
      if (!sections_.count(name))
        order_.push_back(name);
      open_[name] = clock::now();
    }
    void leave_subsection(const std::string &name = "")
    {
      const std::string key = name.empty() ? last_open() : name;
      auto              it  = open_.find(key);
      if (it == open_.end())
        return;
      auto &s = sections_[key];
      s.seconds += std::chrono::duration<double>(clock::now() - it->second).count();
      s.calls += 1;
      open_.erase(it);
    }
    void print_summary() const
    {
      const double total = std::chrono::duration<double>(clock::now() - start_).count();
      out_ << "\n\n+---------------------------------------------+------------+------------+\n"
           << "| Total wallclock time elapsed since start    |" << std::setw(10) << std::setprecision(3)
           << std::scientific << total << "s |            |\n"
           << "|                                             |            |            |\n"
           << "| Section                         | no. calls |  wall time | % of total |\n"
           << "+---------------------------------+-----------+------------+------------+\n";
      for (const auto &name : order_)
        {
          const auto it = sections_.find(name);
          if (it == sections_.end())
            continue;
          out_ << "| " << std::left << std::setw(32) << name << "|" << std::right << std::setw(10) << it->second.calls
               << " |" << std::setw(10) << std::setprecision(3) << std::scientific << it->second.seconds << "s |"
               << std::setw(10) << std::fixed << std::setprecision(1) << 100.0 * it->second.seconds / total << "% |\n";
        }
      out_ << "+---------------------------------+-----------+------------+------------+\n" << std::endl;
    }

  private:
    using clock = std::chrono::steady_clock;
    struct Section
    {
      double seconds = 0;
      long   calls   = 0;
    };
    std::string last_open() const { return open_.empty() ? std::string() : open_.rbegin()->first; }
    std::ostream                            &out_;
    clock::time_point                        start_;
    std::map<std::string, Section>           sections_;
    std::map<std::string, clock::time_point> open_;
    std::vector<std::string>                 order_;
  };
} // namespace mi
