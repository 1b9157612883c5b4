// TEST INFRASTRUCTURE (oracle/_ref build only): toDelimitedString as gait/ModeSequenceTemplate.cpp:51-52 prints with it.
#pragma once
#include <sstream>
#include <string>
#include <vector>
namespace ocs2 {
template <class T>
std::string toDelimitedString(const std::vector<T>& v, const std::string& delim = ", ") {
  std::ostringstream os;
  for (size_t i = 0; i < v.size(); ++i) os << (i ? delim : "") << v[i];
  return os.str();
}
}  // namespace ocs2
