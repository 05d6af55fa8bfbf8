// ros_compat.hpp — plain-struct mirrors of the few ROS 2 / Nav2 message and costmap types that appear in the
// reference's Optimizer::optimize signature (optimizer.hpp:167-170). ROS 2, Nav2, tf2, people_msgs and
// obstacle_distance_msgs are absent from this image; on a ROS machine define SMPC_HOST_WITH_ROS and the real headers
// are included instead (field names below are the real ones, so optimizer.cpp compiles against either).
#pragma once

#ifdef SMPC_HOST_WITH_ROS
#include <geometry_msgs/msg/pose_stamped.hpp>
#include <geometry_msgs/msg/twist_stamped.hpp>
#include <nav2_costmap_2d/costmap_2d.hpp>
#include <nav_msgs/msg/path.hpp>
#include <obstacle_distance_msgs/msg/obstacle_distance.hpp>
#include <people_msgs/msg/people.hpp>
#else
#include <cstdint>
#include <string>
#include <vector>

namespace builtin_interfaces { namespace msg { struct Time { int32_t sec = 0; uint32_t nanosec = 0; }; } }
namespace std_msgs { namespace msg { struct Header { builtin_interfaces::msg::Time stamp; std::string frame_id; }; } }
namespace geometry_msgs { namespace msg {
struct Point { double x = 0, y = 0, z = 0; };
struct Quaternion { double x = 0, y = 0, z = 0, w = 1; };
struct Vector3 { double x = 0, y = 0, z = 0; };
struct Pose { Point position; Quaternion orientation; };
struct PoseStamped { std_msgs::msg::Header header; Pose pose; };
struct Twist { Vector3 linear, angular; };
struct TwistStamped { std_msgs::msg::Header header; Twist twist; };
} }
namespace nav_msgs { namespace msg {
struct Path { std_msgs::msg::Header header; std::vector<geometry_msgs::msg::PoseStamped> poses; };
struct MapMetaData { float resolution = 0; uint32_t width = 0, height = 0; geometry_msgs::msg::Pose origin; };
} }
namespace people_msgs { namespace msg {
struct Person { std::string name; geometry_msgs::msg::Point position, velocity; double reliability = 0; };
struct People { std_msgs::msg::Header header; std::vector<Person> people; };
} }
namespace obstacle_distance_msgs { namespace msg {
struct ObstacleDistance { std_msgs::msg::Header header; nav_msgs::msg::MapMetaData info; std::vector<float> distances; std::vector<uint32_t> indexes; };
} }
namespace nav2_costmap_2d {
// Only the accessors the hot path reads (src/optimizer.cpp:167-168, src/critics/obstacle_cost_function.cpp:25-26).
class Costmap2D {
public:
  Costmap2D(unsigned size_x, unsigned size_y, double resolution, double origin_x, double origin_y)
  : size_x_(size_x), size_y_(size_y), resolution_(resolution), origin_x_(origin_x), origin_y_(origin_y), data_(size_x * size_y, 0) {}
  unsigned char * getCharMap() const { return const_cast<unsigned char *>(data_.data()); }
  unsigned int getSizeInCellsX() const { return size_x_; }
  unsigned int getSizeInCellsY() const { return size_y_; }
  double getOriginX() const { return origin_x_; }
  double getOriginY() const { return origin_y_; }
  double getResolution() const { return resolution_; }
private:
  unsigned size_x_, size_y_;
  double resolution_, origin_x_, origin_y_;
  std::vector<unsigned char> data_;
};
}  // namespace nav2_costmap_2d
#endif
