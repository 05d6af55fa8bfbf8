// path_handler.hpp — host-side mirror of mpc::PathHandler (reference include/nav2_social_mpc_controller/tools/
// path_handler.hpp:36-92, src/path_handler.cpp:39-137) with the reference's method names: setPlan,
// transformGlobalPlan(pose, max_robot_pose_search_dist), getTransformedGoal(goal_dist, transformed_plan, robot_pose).
// The reference's constructor takes a tf2 buffer and the Costmap2DROS (frame lookups, costmap extent); ROS is absent
// here, so this class takes the costmap itself and assumes the robot pose, the global plan and the costmap share one
// frame (the batch entry point smpc_transform_global_plan_batch takes a rigid transform per robot instead). It is a
// CPU statement of the same algorithm as csrc/smpc_path_window.hpp, used by tests as an independent check and by a
// single-robot integration as the drop-in class. nav2_core::PlannerException becomes std::runtime_error with the
// reference's messages.
#pragma once
#include <stdexcept>

#include "ros_compat.hpp"

namespace mpc
{

class PathHandler
{
public:
  explicit PathHandler(const nav2_costmap_2d::Costmap2D * costmap) : costmap_(costmap) {}

  // src/path_handler.cpp:39-108
  nav_msgs::msg::Path transformGlobalPlan(const geometry_msgs::msg::PoseStamped & pose, double max_robot_pose_search_dist);
  // src/path_handler.cpp:110-113
  void setPlan(const nav_msgs::msg::Path & path) { global_plan_ = path; }
  // src/path_handler.cpp:115-137 (a PointStamped in the reference; its point is what is used)
  geometry_msgs::msg::Point getTransformedGoal(const double & goal_dist, const nav_msgs::msg::Path & transformed_plan,
                                               const geometry_msgs::msg::PoseStamped & robot_pose);
  const nav_msgs::msg::Path & getPlan() const { return global_plan_; }

protected:
  const nav2_costmap_2d::Costmap2D * costmap_;
  nav_msgs::msg::Path global_plan_;
};

}  // namespace mpc
