// TEST INFRASTRUCTURE (oracle/_ref build only).  angles::shortest_angular_distance ([ROS-knowledge] angles/angles.h: published formulas).
#pragma once
#include <cmath>
namespace angles {
inline double normalize_angle_positive(double a) { return std::fmod(std::fmod(a, 2.0 * M_PI) + 2.0 * M_PI, 2.0 * M_PI); }
inline double normalize_angle(double a) { double r = normalize_angle_positive(a); if (r > M_PI) r -= 2.0 * M_PI; return r; }
inline double shortest_angular_distance(double from, double to) { return normalize_angle(to - from); }
}  // namespace angles
