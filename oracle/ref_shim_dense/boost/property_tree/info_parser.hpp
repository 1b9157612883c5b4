#pragma once
#include <boost/property_tree/ptree.hpp>
