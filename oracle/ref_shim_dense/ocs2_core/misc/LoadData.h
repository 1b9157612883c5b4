// TEST INFRASTRUCTURE (oracle/_ref build only).  OCS2 loadData over the INFO reader [OCS2-knowledge: loadPtreeValue leaves the
// value untouched (and says so when verbose) if the key is missing; loadEigenMatrix zero-fills, reads `(i,j) v`, applies `scaling`].
#pragma once
#include <iostream>
#include <string>
#include <boost/property_tree/ptree.hpp>
#include <ocs2_core/Types.h>
namespace ocs2 {
namespace loadData {
template <class T>
void loadPtreeValue(const boost::property_tree::ptree& pt, T& value, const std::string& name, bool verbose) {
  if (pt.root.has(name)) value = T(pt.root.number(name));
  if (verbose) std::cerr << " #### '" << name << "': " << value << (pt.root.has(name) ? "\n" : " (default)\n");
}
inline void loadPtreeValue(const boost::property_tree::ptree& pt, bool& value, const std::string& name, bool) {
  if (pt.root.has(name)) value = pt.root.boolean(name);
}
inline void loadPtreeValue(const boost::property_tree::ptree& pt, std::string& value, const std::string& name, bool) {
  if (pt.root.has(name)) value = pt.root.str(name);
}
template <class T>
void loadCppDataType(const std::string& file, const std::string& name, T& value) {
  boost::property_tree::ptree pt;
  boost::property_tree::read_info(file, pt);
  value = T(pt.root.number(name));
}
inline void loadCppDataType(const std::string& file, const std::string& name, bool& value) {
  boost::property_tree::ptree pt;
  boost::property_tree::read_info(file, pt);
  value = pt.root.boolean(name);
}
template <class M>
void loadEigenMatrix(const std::string& file, const std::string& name, M& m) {
  const hunter_hip::InfoNode root = hunter_hip::read_info_file(file);
  const std::vector<double> v = root.matrix(name, m.rows(), m.cols());
  for (int i = 0; i < m.rows(); ++i)
    for (int j = 0; j < m.cols(); ++j) m(i, j) = v[size_t(i) * size_t(m.cols()) + size_t(j)];
}
}  // namespace loadData
}  // namespace ocs2
