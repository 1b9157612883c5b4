// TEST INFRASTRUCTURE (oracle/_ref build only). Stand-in for OCS2's <ocs2_core/misc/LoadData.h>: the loaders are declared
// so that the reference sources compile; the golden-vector entry points set every configuration value directly and never
// call them (they would need boost::property_tree).
#pragma once
#include <stdexcept>
#include <string>
#include <ocs2_core/Types.h>
namespace ocs2 {
namespace loadData {
template <class PT, class T>
void loadPtreeValue(const PT&, T&, const std::string&, bool) { throw std::runtime_error("ref_shim: loadPtreeValue is not available"); }
template <class T>
void loadCppDataType(const std::string&, const std::string&, T&) { throw std::runtime_error("ref_shim: loadCppDataType is not available"); }
template <class M>
void loadEigenMatrix(const std::string&, const std::string&, M&) { throw std::runtime_error("ref_shim: loadEigenMatrix is not available"); }
}  // namespace loadData
}  // namespace ocs2
