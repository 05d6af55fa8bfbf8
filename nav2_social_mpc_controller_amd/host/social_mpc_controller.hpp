// social_mpc_controller.hpp — the part of the reference's plugin shell (SocialMPCController,
// social_mpc_controller.hpp:70-112, src/social_mpc_controller.cpp:162-257) that is arithmetic and control flow rather
// than ROS plumbing: trajectorize -> field-of-view filter -> optimize -> fallbacks -> first command. SURVEY §8 row f4
// proper (pluginlib export, lifecycle node, subscribers, TF, RViz markers, PathHandler's transform / prune of the global
// plan) needs ROS 2 / Nav2, which this image does not have; here the inputs those parts deliver are handed in directly.
// Method names follow nav2_core::Controller; the real shell — class SocialMPCController : public nav2_core::Controller with
// the pluginlib export — is host/ros/social_mpc_controller_plugin.{hpp,cpp} (needs ROS 2 / Nav2: -DSMPC_HOST_WITH_ROS)
// and is laid over this class.
#pragma once
#include <memory>
#include <vector>

#include "optimizer.hpp"
#include "path_trajectorizer.hpp"

namespace nav2_social_mpc_controller
{

struct ControllerParams
{
  double fov_angle = M_PI / 4;  // <plugin>.fov_angle (src/social_mpc_controller.cpp:60)
  TrajectorizerParams trajectorizer;
  OptimizerParams optimizer;
};

class SocialMPCControllerCore
{
public:
  void configure(const ControllerParams & params);
  void cleanup() { optimizer_.reset(); trajectorizer_.reset(); }
  void activate() {}
  void deactivate() {}

  // the plan already transformed into the frame of robot_pose and pruned (PathHandler::transformGlobalPlan, :172-174)
  void setPlan(const nav_msgs::msg::Path & transformed_plan) { plan_ = transformed_plan; }
  void setSpeedLimit(const double &, const bool &) {}  // without effect in the reference too (:264-275)
  // what PeopleInterface / ObstacleDistInterface / Costmap2DROS deliver in the reference
  void setPeople(const people_msgs::msg::People & people) { people_ = people; }
  void setObstacleDistance(const obstacle_distance_msgs::msg::ObstacleDistance & od) { od_ = od; }
  void setCostmap(const nav2_costmap_2d::Costmap2D * costmap) { costmap_ = costmap; }

  geometry_msgs::msg::TwistStamped computeVelocityCommands(
    const geometry_msgs::msg::PoseStamped & robot_pose, const geometry_msgs::msg::Twist & speed, void * goal_checker = nullptr);

  // the field-of-view filter of computeVelocityCommands (:196-214), public for tests
  people_msgs::msg::People filterPeople(const people_msgs::msg::People & people, const geometry_msgs::msg::PoseStamped & robot_pose) const;

  // what the reference publishes for RViz (local_path_pub_, people_traj_pub_): kept for inspection
  const nav_msgs::msg::Path & lastLocalPath() const { return last_path_; }
  const AgentsTrajectories & lastProjectedPeople() const { return last_people_; }
  bool lastOptimized() const { return last_optimized_; }

protected:
  double fov_angle_ = M_PI / 4;
  std::unique_ptr<PathTrajectorizer> trajectorizer_;
  std::unique_ptr<Optimizer> optimizer_;
  nav_msgs::msg::Path plan_, last_path_;
  people_msgs::msg::People people_;
  obstacle_distance_msgs::msg::ObstacleDistance od_;
  const nav2_costmap_2d::Costmap2D * costmap_ = nullptr;
  AgentsTrajectories last_people_;
  bool last_optimized_ = false;
};

#ifndef SMPC_HOST_WITH_ROS
// without ROS the core answers to the reference's class name (tests, demos); with ROS that name belongs to the plugin
using SocialMPCController = SocialMPCControllerCore;
#endif

}  // namespace nav2_social_mpc_controller
