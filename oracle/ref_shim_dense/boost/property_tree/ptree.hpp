// TEST INFRASTRUCTURE (oracle/_ref build only).  boost::property_tree::ptree / read_info over the product's INFO reader
// (include/hunter_info.hpp), so that the reference's own loadTasksSetting / loadSettings read the reference's own task.info.
#pragma once
#include <string>
#include "../../../../include/hunter_info.hpp"
namespace boost { namespace property_tree {
struct ptree { hunter_hip::InfoNode root; };
inline void read_info(const std::string& file, ptree& pt) { pt.root = hunter_hip::read_info_file(file); }
} }
