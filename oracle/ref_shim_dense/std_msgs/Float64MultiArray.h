#pragma once
#include <memory>
#include <vector>
namespace std_msgs { struct Float64MultiArray { std::vector<double> data; typedef std::shared_ptr<const Float64MultiArray> ConstPtr; }; }
