#pragma once
#include <memory>
namespace std_msgs { struct Float32 { float data = 0; typedef std::shared_ptr<const Float32> ConstPtr; }; }
