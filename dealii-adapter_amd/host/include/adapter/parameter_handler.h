// prm::Handler -- a small reader for deal.II-style parameter files ("subsection X ... end",
// "set Key = value", '#' comments), standing in for dealii::ParameterHandler, which the reference uses
// (include/adapter/parameters.cc:8-187) but which is not available here.
//
// Semantics kept: declared entries are bound to variables with a validating pattern; parse_input() is strict
// (unknown subsection or key -> exception, as in AllParameters' parse at parameters.cc:187) unless
// skip_undefined is set (the two lenient partial parses of elasticity.cc:54,86).
#pragma once
#include <cstdlib>
#include <fstream>
#include <functional>
#include <map>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace prm
{
  inline std::string trim(const std::string &s)
  {
    const auto b = s.find_first_not_of(" \t\r\n");
    if (b == std::string::npos)
      return "";
    const auto e = s.find_last_not_of(" \t\r\n");
    return s.substr(b, e - b + 1);
  }

  // collapse runs of blanks so that "Time  step   size" matches "Time step size"
  inline std::string squeeze(const std::string &s)
  {
    std::string out;
    bool        blank = false;
    for (char ch : trim(s))
      {
        if (ch == ' ' || ch == '\t')
          {
            blank = true;
            continue;
          }
        if (blank && !out.empty())
          out += ' ';
        blank = false;
        out += ch;
      }
    return out;
  }

  struct Pattern
  {
    std::function<bool(const std::string &)> ok;
    std::string                              description;
  };

  inline bool parse_double(const std::string &s, double &v)
  {
    char *end = nullptr;
    v         = std::strtod(s.c_str(), &end);
    return end != s.c_str() && trim(end).empty();
  }

  inline Pattern Double(double lo = -1e300, double hi = 1e300)
  {
    return {[lo, hi](const std::string &s) {
              double v;
              return parse_double(s, v) && v >= lo && v <= hi;
            },
            "a floating point number in [" + std::to_string(lo) + ", " + std::to_string(hi) + "]"};
  }
  inline Pattern Integer(long lo = -2147483647L, long hi = 2147483647L)
  {
    return {[lo, hi](const std::string &s) {
              char *end = nullptr;
              long  v   = std::strtol(s.c_str(), &end, 10);
              return end != s.c_str() && trim(end).empty() && v >= lo && v <= hi;
            },
            "an integer in [" + std::to_string(lo) + ", " + std::to_string(hi) + "]"};
  }
  inline Pattern Selection(const std::string &alternatives)
  {
    return {[alternatives](const std::string &s) {
              std::stringstream ss(alternatives);
              std::string       item;
              while (std::getline(ss, item, '|'))
                if (trim(item) == s)
                  return true;
              return false;
            },
            "one of " + alternatives};
  }
  inline Pattern Anything()
  {
    return {[](const std::string &) { return true; }, "any text"};
  }
  inline Pattern ListOfDoubles(unsigned min_n, unsigned max_n)
  {
    return {[min_n, max_n](const std::string &s) {
              std::stringstream ss(s);
              std::string       item;
              unsigned          n = 0;
              while (std::getline(ss, item, ','))
                {
                  double v;
                  if (!parse_double(trim(item), v))
                    return false;
                  ++n;
                }
              return n >= min_n && n <= max_n;
            },
            "a comma separated list of numbers"};
  }

  class Handler
  {
  public:
    void enter_subsection(const std::string &name) { path_.push_back(squeeze(name)); }
    void leave_subsection()
    {
      if (path_.empty())
        throw std::runtime_error("leave_subsection without matching enter_subsection");
      path_.pop_back();
    }

    // bind `key` of the current subsection to a setter; the current value of the variable is the default
    void add(const std::string &key, const Pattern &pattern, std::function<void(const std::string &)> setter)
    {
      entries_[full(path_, squeeze(key))] = Entry{pattern, std::move(setter)};
      for (size_t i = 1; i <= path_.size(); ++i)
        sections_[join(std::vector<std::string>(path_.begin(), path_.begin() + i))] = true;
    }
    void add_parameter(const std::string &key, double &v, const std::string & /*doc*/ = "",
                       const Pattern &p = Double())
    {
      add(key, p, [&v](const std::string &s) { parse_double(s, v); });
    }
    void add_parameter(const std::string &key, int &v, const std::string & = "", const Pattern &p = Integer())
    {
      add(key, p, [&v](const std::string &s) { v = int(std::strtol(s.c_str(), nullptr, 10)); });
    }
    void add_parameter(const std::string &key, unsigned int &v, const std::string & = "",
                       const Pattern &p = Integer(0))
    {
      add(key, p, [&v](const std::string &s) { v = unsigned(std::strtoul(s.c_str(), nullptr, 10)); });
    }
    void add_parameter(const std::string &key, std::string &v, const std::string & = "",
                       const Pattern &p = Anything())
    {
      add(key, p, [&v](const std::string &s) { v = s; });
    }
    void add_parameter(const std::string &key, double (&v)[3], const std::string & = "",
                       const Pattern &p = ListOfDoubles(3, 3))
    {
      add(key, p, [&v](const std::string &s) {
        std::stringstream ss(s);
        std::string       item;
        for (int i = 0; i < 3 && std::getline(ss, item, ','); ++i)
          parse_double(trim(item), v[i]);
      });
    }
    void add_parameter(const std::string &key, int (&v)[3], const std::string & = "",
                       const Pattern &p = ListOfDoubles(1, 3))
    {
      add(key, p, [&v](const std::string &s) {
        std::stringstream ss(s);
        std::string       item;
        for (int i = 0; i < 3 && std::getline(ss, item, ','); ++i)
          v[i] = int(std::strtol(trim(item).c_str(), nullptr, 10));
      });
    }

    void parse_input(const std::string &filename, const std::string & /*last_line*/ = "",
                     const bool skip_undefined = false)
    {
      std::ifstream in(filename);
      if (!in)
        throw std::runtime_error("Cannot open parameter file <" + filename + ">");
      std::vector<std::string> where;
      std::vector<bool>        known; // is the subsection at this depth declared?
      std::string              raw, line;
      int                      lineno = 0;
      while (std::getline(in, raw))
        {
          ++lineno;
          // line continuation with a trailing backslash
          line += raw;
          if (!line.empty() && line.back() == '\\')
            {
              line.pop_back();
              continue;
            }
          const auto hash = line.find('#');
          if (hash != std::string::npos)
            line.erase(hash);
          const std::string text = trim(line);
          line.clear();
          if (text.empty())
            continue;
          auto fail = [&](const std::string &msg) {
            throw std::runtime_error("Line <" + std::to_string(lineno) + "> of file <" + filename + ">: " + msg);
          };
          if (text.rfind("subsection", 0) == 0 && (text.size() == 10 || text[10] == ' ' || text[10] == '\t'))
            {
              where.push_back(squeeze(text.substr(10)));
              const bool k = sections_.count(join(where)) > 0;
              if (!k && !skip_undefined)
                fail("There is no such subsection to be entered: " + join(where));
              known.push_back(k);
            }
          else if (squeeze(text) == "end" || squeeze(text) == "END")
            {
              if (where.empty())
                fail("There is no subsection to leave here.");
              where.pop_back();
              known.pop_back();
            }
          else if (text.rfind("set", 0) == 0 && text.size() > 3 && (text[3] == ' ' || text[3] == '\t'))
            {
              const auto eq = text.find('=');
              if (eq == std::string::npos)
                fail("Invalid format of 'set' statement (no '=').");
              const std::string key   = squeeze(text.substr(3, eq - 3));
              const std::string value = trim(text.substr(eq + 1));
              const auto        it    = entries_.find(full(where, key));
              if (it == entries_.end())
                {
                  if (skip_undefined)
                    continue;
                  fail("No entry with name <" + key + "> was declared in the current subsection.");
                }
              if (!it->second.pattern.ok(value))
                fail("The entry <" + key + "> with value <" + value + "> does not match its pattern: " +
                     it->second.pattern.description);
              it->second.set(value);
            }
          else
            fail("The line <" + text + "> could not be parsed.");
        }
      if (!where.empty())
        throw std::runtime_error("Unbalanced 'subsection'/'end' in file <" + filename + ">");
    }

  private:
    struct Entry
    {
      Pattern                                  pattern;
      std::function<void(const std::string &)> set;
    };
    static std::string join(const std::vector<std::string> &p)
    {
      std::string s;
      for (const auto &x : p)
        s += "/" + x;
      return s;
    }
    static std::string full(const std::vector<std::string> &p, const std::string &key) { return join(p) + "//" + key; }
    std::vector<std::string>    path_;
    std::map<std::string, Entry> entries_;
    std::map<std::string, bool>  sections_;
  };
} // namespace prm
