#pragma once
#include <memory>
namespace geometry_msgs {
struct Vector3 { double x = 0, y = 0, z = 0; };
struct Twist { Vector3 linear, angular; typedef std::shared_ptr<const Twist> ConstPtr; };
}
