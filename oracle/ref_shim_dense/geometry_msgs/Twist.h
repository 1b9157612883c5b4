#pragma once
#include <memory>
#include <geometry_msgs/PoseWithCovarianceStamped.h>
namespace geometry_msgs { struct TwistMsgPtrHolder {}; }
namespace geometry_msgs { struct TwistStampedDummy {}; }
// geometry_msgs::Twist as a topic message (ConstPtr); the struct itself lives in PoseWithCovarianceStamped.h of this stand-in set
namespace geometry_msgs { typedef std::shared_ptr<const Twist> TwistConstPtr; }
