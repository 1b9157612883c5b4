// TEST INFRASTRUCTURE (oracle/_ref build only): boost::filesystem::path / exists as LeggedInterface.cpp:55-88 uses them.
#pragma once
#include <fstream>
#include <ostream>
#include <string>
namespace boost { namespace filesystem {
class path {
 public:
  path(const std::string& s) : s_(s) {}
  const std::string& string() const { return s_; }
 private:
  std::string s_;
};
inline std::ostream& operator<<(std::ostream& os, const path& p) { return os << '"' << p.string() << '"'; }
inline bool exists(const path& p) { return std::ifstream(p.string()).good(); }
} }
