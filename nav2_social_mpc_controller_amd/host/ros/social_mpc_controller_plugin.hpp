// social_mpc_controller_plugin.hpp — the Nav2 controller plugin over the MI355X solver: SURVEY §8 row f4 / boundary
// "outer, verbatim". NEEDS ROS 2 + Nav2 (rclcpp_lifecycle, nav2_core, nav2_costmap_2d, tf2_ros, pluginlib, people_msgs,
// obstacle_distance_msgs): absent from the image this repository is developed in, so this file has never been compiled
// there — host/ros/CMakeLists.txt builds it on a ROS machine. No stand-in headers exist for it.
//
// Interface kept from the reference (include/nav2_social_mpc_controller/social_mpc_controller.hpp:50-112):
//   class nav2_social_mpc_controller::SocialMPCController : public nav2_core::Controller
//     configure(LifecycleNode::WeakPtr, name, tf buffer, Costmap2DROS) / cleanup / activate / deactivate
//     computeVelocityCommands(PoseStamped, Twist, GoalChecker *) -> TwistStamped
//     setPlan(Path) / setSpeedLimit(double, bool)
//   exported with PLUGINLIB_EXPORT_CLASS under the same type string (src/social_mpc_controller.cpp:325,
//   nav2_social_mpc_controller.xml:1-9), so `plugin: "nav2_social_mpc_controller::SocialMPCController"` in a
//   controller_server configuration selects this library unchanged; the ROS parameters are the reference's
//   (<name>.fov_angle, <name>.transform_tolerance, <name>.trajectorizer.*, <name>.optimizer.*, <name>.optimizer.weights.*)
//   plus <name>.optimizer.device (HIP device index, default 0).
// What does the work: SocialMPCControllerCore (host/social_mpc_controller.hpp: trajectorize -> field-of-view filter ->
// Optimizer::optimize -> fallbacks), whose Optimizer::optimize is one smpc_solve_batch call on the GPU; mpc::PathHandler
// (host/path_handler.hpp) for the plan window. This file adds only what needs ROS: parameters, TF, the two
// subscriptions, the two publishers.
#pragma once
#ifndef SMPC_HOST_WITH_ROS
#error "social_mpc_controller_plugin.hpp needs ROS 2 / Nav2: build it through host/ros/CMakeLists.txt (-DSMPC_HOST_WITH_ROS)"
#endif

#include <memory>
#include <mutex>
#include <string>

#include "nav2_core/controller.hpp"
#include "nav2_costmap_2d/costmap_2d_ros.hpp"
#include "nav_msgs/msg/path.hpp"
#include "obstacle_distance_msgs/msg/obstacle_distance.hpp"
#include "people_msgs/msg/people.hpp"
#include "rclcpp/rclcpp.hpp"
#include "rclcpp_lifecycle/lifecycle_node.hpp"
#include "rclcpp_lifecycle/lifecycle_publisher.hpp"
#include "tf2_ros/buffer.h"
#include "visualization_msgs/msg/marker_array.hpp"

#include "../path_handler.hpp"
#include "../social_mpc_controller.hpp"

namespace nav2_social_mpc_controller
{

class SocialMPCController : public nav2_core::Controller
{
public:
  SocialMPCController() = default;
  ~SocialMPCController() override = default;

  void configure(
    const rclcpp_lifecycle::LifecycleNode::WeakPtr & parent, std::string name, std::shared_ptr<tf2_ros::Buffer> tf,
    std::shared_ptr<nav2_costmap_2d::Costmap2DROS> costmap_ros) override;
  void cleanup() override;
  void activate() override;
  void deactivate() override;
  geometry_msgs::msg::TwistStamped computeVelocityCommands(
    const geometry_msgs::msg::PoseStamped & pose, const geometry_msgs::msg::Twist & velocity,
    nav2_core::GoalChecker * goal_checker) override;
  void setPlan(const nav_msgs::msg::Path & path) override;
  void setSpeedLimit(const double & speed_limit, const bool & percentage) override;

protected:
  // the stored global plan moved into the costmap's global frame (the frame the window, the people and the solve live in)
  nav_msgs::msg::Path planInCostmapFrame(const rclcpp::Time & stamp) const;
  void publishPeople(const AgentsTrajectories & people, const std_msgs::msg::Header & header);

  std::string plugin_name_;
  rclcpp::Logger logger_{rclcpp::get_logger("SocialMPCController")};
  rclcpp::Clock::SharedPtr clock_;
  std::shared_ptr<tf2_ros::Buffer> tf_;
  std::shared_ptr<nav2_costmap_2d::Costmap2DROS> costmap_ros_;
  nav2_costmap_2d::Costmap2D * costmap_ = nullptr;
  tf2::Duration transform_tolerance_{};
  double desired_linear_vel_ = 0.5;  // read like the reference does, and like there without effect on the command

  SocialMPCControllerCore core_;
  std::unique_ptr<mpc::PathHandler> path_handler_;
  nav_msgs::msg::Path global_plan_;  // as received by setPlan, in its own frame
  bool plan_changed_ = false;

  // the latest people / obstacle-distance messages (PeopleInterface, ObstacleDistInterface of the reference)
  std::mutex inputs_mutex_;
  people_msgs::msg::People people_;
  obstacle_distance_msgs::msg::ObstacleDistance obstacle_distance_;
  rclcpp::Subscription<people_msgs::msg::People>::SharedPtr people_sub_;
  rclcpp::Subscription<obstacle_distance_msgs::msg::ObstacleDistance>::SharedPtr obstacle_sub_;

  std::shared_ptr<rclcpp_lifecycle::LifecyclePublisher<nav_msgs::msg::Path>> local_path_pub_;
  std::shared_ptr<rclcpp_lifecycle::LifecyclePublisher<visualization_msgs::msg::MarkerArray>> people_traj_pub_;
};

}  // namespace nav2_social_mpc_controller
