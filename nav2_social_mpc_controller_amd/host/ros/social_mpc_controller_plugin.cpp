// social_mpc_controller_plugin.cpp — see social_mpc_controller_plugin.hpp. NEEDS ROS 2 + Nav2; never compiled in the
// image this repository is developed in (host/ros/CMakeLists.txt builds it where Nav2 is installed).
//
// Behaviour follows the reference's shell (src/social_mpc_controller.cpp): parameters of configure() :48-86 (plus the
// trajectorizer's, src/path_trajectorizer.cpp:43-88, and the optimiser's, src/optimizer.cpp:16-85), the control tick of
// computeVelocityCommands() :162-257 — plan window, trajectorize (or the 0.1 m/s fallback), field-of-view filter,
// optimize (or the trajectorizer's commands), publish, first command — setPlan :259-262, and a setSpeedLimit that has
// no effect, as there (:264-283).
#include "social_mpc_controller_plugin.hpp"

#include <algorithm>
#include <cmath>
#include <stdexcept>
#include <utility>
#include <vector>

#include "nav2_core/exceptions.hpp"
#include "nav2_util/node_utils.hpp"
#include "pluginlib/class_list_macros.hpp"
#include "tf2/utils.h"
#include "tf2_geometry_msgs/tf2_geometry_msgs.hpp"

using nav2_util::declare_parameter_if_not_declared;

namespace nav2_social_mpc_controller
{
namespace
{
template <typename T>
T param(const rclcpp_lifecycle::LifecycleNode::SharedPtr & node, const std::string & key, const T & fallback)
{
  declare_parameter_if_not_declared(node, key, rclcpp::ParameterValue(fallback));
  T value = fallback;
  node->get_parameter(key, value);
  return value;
}
}  // namespace

void SocialMPCController::configure(
  const rclcpp_lifecycle::LifecycleNode::WeakPtr & parent, std::string name, std::shared_ptr<tf2_ros::Buffer> tf,
  std::shared_ptr<nav2_costmap_2d::Costmap2DROS> costmap_ros)
{
  auto node = parent.lock();
  if (!node) throw std::runtime_error("SocialMPCController::configure: the lifecycle node is gone");
  plugin_name_ = std::move(name);
  logger_ = node->get_logger();
  clock_ = node->get_clock();
  tf_ = std::move(tf);
  costmap_ros_ = std::move(costmap_ros);
  costmap_ = costmap_ros_->getCostmap();

  ControllerParams p;
  // <name>.*  (src/social_mpc_controller.cpp:58-65)
  desired_linear_vel_ = param(node, plugin_name_ + ".desired_linear_vel", 0.5);
  p.fov_angle = param(node, plugin_name_ + ".fov_angle", M_PI / 4);
  transform_tolerance_ = tf2::durationFromSec(param(node, plugin_name_ + ".transform_tolerance", 0.1));
  // <name>.trajectorizer.*  (src/path_trajectorizer.cpp:52-84; its own transform_tolerance / base_frame are read there for
  // TF lookups the core does not make: the window handed to it is already in the costmap frame)
  const std::string t = plugin_name_ + ".trajectorizer.";
  p.trajectorizer.omnidirectional = param(node, t + "omnidirectional", false);
  p.trajectorizer.desired_linear_vel = param(node, t + "desired_linear_vel", 0.4);
  p.trajectorizer.lookahead_dist = param(node, t + "lookahead_dist", 0.4);
  p.trajectorizer.max_angular_vel = param(node, t + "max_angular_vel", 1.0);
  p.trajectorizer.time_step = param(node, t + "time_step", 0.05);
  p.trajectorizer.max_time = param(node, t + "max_time", 3.0);
  // <name>.optimizer.* and <name>.optimizer.weights.*  (src/optimizer.cpp:26-84)
  const std::string o = plugin_name_ + ".optimizer.";
  OptimizerParams & q = p.optimizer;
  q.linear_solver_type = param(node, o + "linear_solver_type", std::string("SPARSE_NORMAL_CHOLESKY"));
  q.validate();  // "Invalid parameter: linear_solver_type", as :31-45
  q.param_tol = param(node, o + "param_tol", 1e-15);
  q.fn_tol = param(node, o + "fn_tol", 1e-7);
  q.gradient_tol = param(node, o + "gradient_tol", 1e-10);
  q.max_iterations = param(node, o + "max_iterations", 100);
  q.debug = param(node, o + "debug_optimizer", false);
  q.control_horizon_ = param(node, o + "control_horizon", 5);
  q.parameter_block_length_ = param(node, o + "parameter_block_length", 5);
  q.current_path_w = static_cast<float>(param(node, o + "current_path_weight", 1.0));
  q.current_cmds_w = static_cast<float>(param(node, o + "current_cmds_weight", 1.0));
  q.max_time = static_cast<float>(p.trajectorizer.max_time);  // :84 reads <name>.trajectorizer.max_time
  q.distance_w_ = param(node, o + "weights.distance_weight", 3.0);
  q.socialwork_w_ = param(node, o + "weights.social_weight", 1.0);
  q.velocity_w_ = param(node, o + "weights.velocity_weight", 0.5);
  q.angle_w_ = param(node, o + "weights.angle_weight", 0.0);
  q.agent_angle_w_ = param(node, o + "weights.agent_angle_weight", 0.5);
  q.proxemics_w_ = param(node, o + "weights.proxemics_weight", 90.0);
  q.velocity_feasibility_w_ = param(node, o + "weights.velocity_feasibility_weight", 0.5);
  q.obstacle_w_ = param(node, o + "weights.obstacle_weight", 0.0);
  q.goal_align_w_ = param(node, o + "weights.goal_align_weight", 0.0);
  q.device = param(node, o + "device", 0);  // HIP device index (no reference counterpart)
  core_.configure(p);  // creates the solver handle: throws if no MI355X is present (there is no CPU solve path)
  core_.setCostmap(costmap_);
  path_handler_ = std::make_unique<mpc::PathHandler>(costmap_);

  // the latest people and obstacle-distance messages (src/people_interface.cpp:9-21, src/obstacle_distance_interface.cpp:15-43)
  people_sub_ = node->create_subscription<people_msgs::msg::People>(
    "people", rclcpp::SensorDataQoS(), [this](const people_msgs::msg::People::SharedPtr msg) {
      std::lock_guard<std::mutex> lock(inputs_mutex_);
      people_ = *msg;
    });
  obstacle_sub_ = node->create_subscription<obstacle_distance_msgs::msg::ObstacleDistance>(
    "obstacle_distance", rclcpp::SystemDefaultsQoS(),
    [this](const obstacle_distance_msgs::msg::ObstacleDistance::SharedPtr msg) {
      // the reference re-expresses a grid that arrives in another frame before using it; a grid whose frame is not the
      // costmap's global frame is not taken here (its origin would be wrong for the nearest-obstacle lookup)
      if (msg->header.frame_id != costmap_ros_->getGlobalFrameID()) {
        RCLCPP_ERROR_THROTTLE(
          logger_, *clock_, 2000, "obstacle_distance arrives in frame '%s', the costmap is in '%s': ignored",
          msg->header.frame_id.c_str(), costmap_ros_->getGlobalFrameID().c_str());
        return;
      }
      std::lock_guard<std::mutex> lock(inputs_mutex_);
      obstacle_distance_ = *msg;
    });
  local_path_pub_ = node->create_publisher<nav_msgs::msg::Path>("local_plan", 1);
  people_traj_pub_ = node->create_publisher<visualization_msgs::msg::MarkerArray>("people_projected_trajectory", 1);
}

void SocialMPCController::cleanup()
{
  RCLCPP_INFO(logger_, "Cleaning up controller: %s of type nav2_social_mpc_controller::SocialMPCController", plugin_name_.c_str());
  local_path_pub_.reset();
  people_traj_pub_.reset();
  people_sub_.reset();
  obstacle_sub_.reset();
  core_.cleanup();
}

void SocialMPCController::activate()
{
  RCLCPP_INFO(logger_, "Activating controller: %s of type nav2_social_mpc_controller::SocialMPCController", plugin_name_.c_str());
  core_.activate();
  local_path_pub_->on_activate();
  people_traj_pub_->on_activate();
}

void SocialMPCController::deactivate()
{
  RCLCPP_INFO(logger_, "Deactivating controller: %s of type nav2_social_mpc_controller::SocialMPCController", plugin_name_.c_str());
  core_.deactivate();
  local_path_pub_->on_deactivate();
  people_traj_pub_->on_deactivate();
}

void SocialMPCController::setPlan(const nav_msgs::msg::Path & path)
{
  global_plan_ = path;
  plan_changed_ = true;  // handed to the path handler (whose copy carries the pruning state) at the next tick
}

void SocialMPCController::setSpeedLimit(const double & /*speed_limit*/, const bool & /*percentage*/)
{
  // the reference computes a throw-away value here and changes nothing (src/social_mpc_controller.cpp:264-283)
}

nav_msgs::msg::Path SocialMPCController::planInCostmapFrame(const rclcpp::Time & stamp) const
{
  const std::string frame = costmap_ros_->getGlobalFrameID();
  nav_msgs::msg::Path out;
  out.header.frame_id = frame;
  out.header.stamp = stamp;
  if (global_plan_.header.frame_id == frame) { out.poses = global_plan_.poses; return out; }
  out.poses.reserve(global_plan_.poses.size());
  for (const auto & in : global_plan_.poses) {
    geometry_msgs::msg::PoseStamped stamped = in, moved;
    stamped.header.frame_id = global_plan_.header.frame_id;
    stamped.header.stamp = stamp;
    try {
      tf_->transform(stamped, moved, frame, transform_tolerance_);
    } catch (const tf2::TransformException & ex) {
      throw nav2_core::PlannerException(std::string("Unable to transform plan pose into the costmap frame: ") + ex.what());
    }
    moved.header.frame_id = frame;
    out.poses.push_back(moved);
  }
  return out;
}

geometry_msgs::msg::TwistStamped SocialMPCController::computeVelocityCommands(
  const geometry_msgs::msg::PoseStamped & pose, const geometry_msgs::msg::Twist & velocity, nav2_core::GoalChecker * goal_checker)
{
  if (goal_checker == nullptr) RCLCPP_WARN(logger_, "Goal checker is null");
  // robot pose and plan in the costmap's global frame: a rigid transform keeps every distance transformGlobalPlan looks
  // at, so pruning and window are those of the reference, which searches in the plan's frame and moves the window over
  geometry_msgs::msg::PoseStamped robot_pose = pose;
  const std::string frame = costmap_ros_->getGlobalFrameID();
  if (pose.header.frame_id != frame) {
    try {
      tf_->transform(pose, robot_pose, frame, transform_tolerance_);
    } catch (const tf2::TransformException & ex) {
      throw nav2_core::PlannerException(std::string("Unable to transform robot pose into global plan's frame: ") + ex.what());
    }
    robot_pose.header.frame_id = frame;
  }
  nav_msgs::msg::Path transformed_plan;
  try {
    // the pruning state lives in the handler's copy of the plan: a plan is handed over once, when setPlan brought it
    if (plan_changed_) {
      path_handler_->setPlan(planInCostmapFrame(pose.header.stamp));
      plan_changed_ = false;
    }
    transformed_plan = path_handler_->transformGlobalPlan(robot_pose, 4.0);  // 4.0: src/social_mpc_controller.cpp:172
    (void)path_handler_->getTransformedGoal(2.5, transformed_plan, robot_pose);  // computed and unused there too (:173)
  } catch (const std::runtime_error & e) {
    throw nav2_core::PlannerException(e.what());  // "Received plan with zero length" / "Resulting plan has 0 poses in it."
  }
  {
    std::lock_guard<std::mutex> lock(inputs_mutex_);
    core_.setPeople(people_);
    core_.setObstacleDistance(obstacle_distance_);
  }
  core_.setCostmap(costmap_ros_->getCostmap());
  core_.setPlan(transformed_plan);
  geometry_msgs::msg::TwistStamped cmd_vel;
  try {
    cmd_vel = core_.computeVelocityCommands(robot_pose, velocity, goal_checker);
  } catch (const std::runtime_error & e) {
    throw nav2_core::PlannerException(e.what());  // computeObstacle's out-of-bounds cases (src/optimizer.cpp:676-712)
  }
  if (!core_.lastOptimized()) RCLCPP_WARN(logger_, "Optimization failed or no valid trajectory: fallback command");
  if (!core_.lastProjectedPeople().empty()) publishPeople(core_.lastProjectedPeople(), transformed_plan.header);
  local_path_pub_->publish(core_.lastLocalPath());
  RCLCPP_DEBUG(logger_, "cmd_vel: %f, %f", cmd_vel.twist.linear.x, cmd_vel.twist.angular.z);
  return cmd_vel;
}

void SocialMPCController::publishPeople(const AgentsTrajectories & people, const std_msgs::msg::Header & header)
{
  // one LINE_STRIP per valid person over the projected steps (src/social_mpc_controller.cpp:121-160)
  visualization_msgs::msg::MarkerArray ma;
  const size_t n = people.front().size();
  std::vector<int> marker_of(n, -1);
  for (size_t a = 0; a < n; ++a) {
    if (people.front()[a][3] == -1.0) continue;
    visualization_msgs::msg::Marker m;
    m.header = header;
    m.type = visualization_msgs::msg::Marker::LINE_STRIP;
    m.action = visualization_msgs::msg::Marker::ADD;
    m.id = static_cast<int>(a);
    m.scale.x = 0.05;
    m.color.a = 1.0; m.color.r = 1.0; m.color.g = 0.0; m.color.b = 1.0;
    marker_of[a] = static_cast<int>(ma.markers.size());
    ma.markers.push_back(m);
  }
  for (const auto & step : people) {
    for (size_t a = 0; a < step.size() && a < n; ++a) {
      if (marker_of[a] < 0 || step[a][3] == -1.0) continue;
      geometry_msgs::msg::Point pt;
      pt.x = step[a][0]; pt.y = step[a][1]; pt.z = 0.1;
      ma.markers[marker_of[a]].points.push_back(pt);
    }
  }
  people_traj_pub_->publish(ma);
}

}  // namespace nav2_social_mpc_controller

// Register this controller as a nav2_core plugin, under the reference's type string
PLUGINLIB_EXPORT_CLASS(nav2_social_mpc_controller::SocialMPCController, nav2_core::Controller)
