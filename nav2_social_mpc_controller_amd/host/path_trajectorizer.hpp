// path_trajectorizer.hpp — host-side mirror of the reference's PathTrajectorizer (path_trajectorizer.hpp:49-94): the
// pure-pursuit simulation that turns the pruned global plan into the optimiser's initial trajectory and commands.
// Same method names and argument meaning; the ROS lifecycle node of configure() is replaced by a plain parameter
// struct because ROS 2 is absent from this image (with SMPC_HOST_WITH_ROS a maintainer fills it from the node's
// parameters exactly as src/path_trajectorizer.cpp:53-84 does). This class is the CPU statement of row f3; the batch
// version for B plans is smpc_trajectorize_path_batch (include/smpc.h).
#pragma once
#include <cmath>
#include <vector>

#include "ros_compat.hpp"

namespace nav2_social_mpc_controller
{

struct TrajectorizerParams  // declare_parameter defaults of src/path_trajectorizer.cpp:53-60
{
  bool omnidirectional = false;
  double desired_linear_vel = 0.4;
  double lookahead_dist = 0.4;
  double max_angular_vel = 1.0;
  double time_step = 0.05;
  double max_time = 3.0;
};

class PathTrajectorizer
{
public:
  PathTrajectorizer() = default;
  void configure(const TrajectorizerParams & params);
  void cleanup() {}
  void activate() {}
  void deactivate() {}

  // In-out path: the plan on entry, the trajectorized path (robot pose first) on return; cmds are appended.
  bool trajectorize(
    nav_msgs::msg::Path & path, const geometry_msgs::msg::PoseStamped & path_robot_pose,
    std::vector<geometry_msgs::msg::TwistStamped> & cmds);

  float inline getTimeStep() { return time_step_; }

private:
  // pure-pursuit parameters (configure); the unicycle / holonomic step of path_trajectorizer.hpp:106-135 lives in the .cpp
  double desired_linear_vel_ = 0.4;
  double lookahead_dist_ = 0.4;
  double max_angular_vel_ = 1.0;
  bool omnidirectional_ = false;
  double time_step_ = 0.05;
  double max_steps_ = 60.0;
};

}  // namespace nav2_social_mpc_controller
