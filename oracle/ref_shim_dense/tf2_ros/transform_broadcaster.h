#pragma once
#include <tf2_ros/transform_listener.h>
namespace tf2_ros { struct TransformBroadcaster { template <class M> void sendTransform(const M&) {} }; }
