// TEST INFRASTRUCTURE (oracle/_ref build only).  [OCS2-knowledge: entries `[i] "first, second"` of the block `name`.]
#pragma once
#include <ocs2_core/misc/LoadData.h>
#include <utility>
namespace ocs2 {
namespace loadData {
template <class T1, class T2>
void loadStdVectorOfPair(const std::string& file, const std::string& name, std::vector<std::pair<T1, T2>>& out, bool = true) {
  const hunter_hip::InfoNode root = hunter_hip::read_info_file(file);
  const hunter_hip::InfoNode* n = root.find(name);
  if (!n) return;
  for (size_t i = 0;; ++i) {
    const hunter_hip::InfoNode* c = n->child("[" + std::to_string(i) + "]");
    if (!c) break;
    const size_t comma = c->value.find(',');
    if (comma == std::string::npos) continue;
    auto trim = [](std::string s) { const size_t a = s.find_first_not_of(" \t"), b = s.find_last_not_of(" \t"); return a == std::string::npos ? std::string() : s.substr(a, b - a + 1); };
    out.emplace_back(li_convert<T1>(trim(c->value.substr(0, comma))), li_convert<T2>(trim(c->value.substr(comma + 1))));
  }
}
}  // namespace loadData
}  // namespace ocs2
