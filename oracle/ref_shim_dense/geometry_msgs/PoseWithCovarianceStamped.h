// TEST INFRASTRUCTURE (oracle/_ref build only).  Message structs of geometry_msgs the estimator fills (plain data).
#pragma once
#include <array>
#include <memory>
#include <string>
#include <ros/ros.h>
namespace std_msgs { struct Header { ros::Time stamp; std::string frame_id; unsigned seq = 0; }; }
namespace geometry_msgs {
struct Point { double x = 0, y = 0, z = 0; };
struct Vector3 { double x = 0, y = 0, z = 0; };
struct Quaternion { double x = 0, y = 0, z = 0, w = 1; };
struct Pose { Point position; Quaternion orientation; };
struct Twist { Vector3 linear, angular; typedef std::shared_ptr<const Twist> ConstPtr; };
struct PoseWithCovariance { Pose pose; std::array<double, 36> covariance{}; };
struct TwistWithCovariance { Twist twist; std::array<double, 36> covariance{}; };
struct PoseWithCovarianceStamped { std_msgs::Header header; PoseWithCovariance pose; };
struct Transform { Vector3 translation; Quaternion rotation; };
struct TransformStamped { std_msgs::Header header; std::string child_frame_id; Transform transform; };
}  // namespace geometry_msgs
