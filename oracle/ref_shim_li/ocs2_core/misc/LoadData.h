// TEST INFRASTRUCTURE (oracle/_ref build only).  The dense stand-in plus loadStdVector [OCS2-knowledge: entries `[i] value` of the
// block `name`, in index order; the vector is left untouched when the block is missing].
#pragma once
#include "../../../ref_shim_dense/ocs2_core/misc/LoadData.h"
#include <sstream>
#include <vector>
namespace ocs2 {
namespace loadData {
template <class T> T li_convert(const std::string& s) { T v{}; std::istringstream is(s); is >> v; return v; }
template <> inline std::string li_convert<std::string>(const std::string& s) { return s; }
template <class T>
void loadStdVector(const std::string& file, const std::string& name, std::vector<T>& out, bool = true) {
  const hunter_hip::InfoNode root = hunter_hip::read_info_file(file);
  const hunter_hip::InfoNode* n = root.find(name);
  if (!n) return;
  std::vector<T> v;
  for (size_t i = 0;; ++i) {
    const hunter_hip::InfoNode* c = n->child("[" + std::to_string(i) + "]");
    if (!c) break;
    v.push_back(li_convert<T>(c->value));
  }
  if (!v.empty()) out = v;
}
}  // namespace loadData
}  // namespace ocs2
