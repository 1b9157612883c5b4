// TEST INFRASTRUCTURE (oracle/_ref build only). Empty stand-in: the settings loaders that use it are never called.
#pragma once
#include <string>
namespace boost { namespace property_tree { struct ptree {}; inline void read_info(const std::string&, ptree&) {} } }
