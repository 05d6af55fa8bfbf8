// optimizer.hpp — host-side mirror of the reference's Optimizer (include/nav2_social_mpc_controller/optimizer.hpp:
// OptimizerParams :59-101, Optimizer::initialize :152, Optimizer::optimize :167-170) over the C ABI of include/smpc.h.
// Same class name, method names, argument order and error behaviour, so SocialMPCController::computeVelocityCommands
// (src/social_mpc_controller.cpp:240) calls it unchanged. The Ceres problem build + solve + unpack
// (src/optimizer.cpp:241-446) is ONE smpc_solve_batch call with B = 1; there is no CPU solve path.
#pragma once

#include <array>
#include <map>
#include <string>
#include <vector>

#include "../../include/smpc.h"
#include "ros_compat.hpp"

// tools/type_definitions.hpp:6-9 (Eigen::Matrix<double,6,1> there; a plain array here — no Eigen in this image)
typedef std::array<double, 6> AgentStatus;             // x, y, yaw, timestamp, lv, av
typedef std::vector<AgentStatus> AgentsStates;         // different agents at the same time
typedef std::vector<AgentStatus> AgentTrajectory;      // one agent over time
typedef std::vector<AgentsStates> AgentsTrajectories;  // all agents over time

namespace nav2_social_mpc_controller
{

struct OptimizerParams
{
  // OptimizerParams::get (src/optimizer.cpp:16-85) reads these from ROS parameters; defaults are its defaults.
  const std::map<std::string, int> solver_types = {
    {"DENSE_SCHUR", SMPC_DENSE_SCHUR}, {"SPARSE_SCHUR", SMPC_SPARSE_SCHUR},
    {"DENSE_NORMAL_CHOLESKY", SMPC_DENSE_NORMAL_CHOLESKY}, {"DENSE_QR", SMPC_DENSE_QR},
    {"SPARSE_NORMAL_CHOLESKY", SMPC_SPARSE_NORMAL_CHOLESKY}};
  std::string linear_solver_type = "SPARSE_NORMAL_CHOLESKY";
  double param_tol = 1e-15, fn_tol = 1e-7, gradient_tol = 1e-10;
  double socialwork_w_ = 1.0, distance_w_ = 3.0, velocity_w_ = 0.5, angle_w_ = 0.0, agent_angle_w_ = 0.5;
  double velocity_feasibility_w_ = 0.5, goal_align_w_ = 0.0, obstacle_w_ = 0.0, proxemics_w_ = 90.0;
  float current_path_w = 1.0f, current_cmds_w = 1.0f, max_time = 3.0f;
  int discretization_ = 1, control_horizon_ = 5, parameter_block_length_ = 5;
  bool debug = false;
  int max_iterations = 100;
  int device = 0;  // HIP device the solver binds to (no reference counterpart)
  // throws std::runtime_error("Invalid parameter: linear_solver_type") like src/optimizer.cpp:31-45
  void validate() const;
};

// trajectory_memory.hpp:32-49: process-wide warm-start memory
class TrajectoryMemory
{
public:
  static TrajectoryMemory & getInstance() { static TrajectoryMemory m; return m; }
  nav_msgs::msg::Path previous_path;
  std::vector<geometry_msgs::msg::TwistStamped> previous_cmds;
  void clear() { previous_path.poses.clear(); previous_cmds.clear(); }
private:
  TrajectoryMemory() {}
};

class Optimizer
{
public:
  Optimizer();
  ~Optimizer();

  void initialize(const OptimizerParams params);

  bool optimize(
    nav_msgs::msg::Path & path, AgentsTrajectories & people_proj, const nav2_costmap_2d::Costmap2D * costmap,
    const obstacle_distance_msgs::msg::ObstacleDistance & obstacles,
    std::vector<geometry_msgs::msg::TwistStamped> & cmds, const people_msgs::msg::People & people,
    const geometry_msgs::msg::Twist & speed, const float time_step);

  // diagnostics of the last solve (no reference counterpart)
  int last_status() const { return last_status_; }
  int last_iterations() const { return last_iterations_; }
  double last_final_cost() const { return last_final_cost_; }

  // reference private helpers, public here so that tests can exercise them (src/optimizer.cpp:454-728)
  AgentsStates people_to_status(const people_msgs::msg::People & people);
  AgentTrajectory format_to_optimize(
    nav_msgs::msg::Path & path, const nav_msgs::msg::Path & previous_path,
    const std::vector<geometry_msgs::msg::TwistStamped> & cmds,
    const std::vector<geometry_msgs::msg::TwistStamped> & previous_cmds, const geometry_msgs::msg::Twist & speed,
    const float current_path_w, const float current_cmds_w, const float maxtime, const float timestep);
  AgentsTrajectories project_people(
    const AgentsStates & init_people, const AgentTrajectory & robot_path,
    const obstacle_distance_msgs::msg::ObstacleDistance & od, const float & maxtime, const float & timestep);
  std::array<double, 2> computeObstacle(
    const std::array<double, 2> & apos, const obstacle_distance_msgs::msg::ObstacleDistance & od);

private:
  smpc_params prm_;
  smpc_handle * handle_ = nullptr;
  int device_ = 0;
  float max_time = 3.0f, current_path_w = 1.0f, current_cmds_w = 1.0f;
  int last_status_ = SMPC_FAILURE, last_iterations_ = 0;
  double last_final_cost_ = 0.0;
};

// tf2 helpers restated (tf2::Quaternion::setRPY(0,0,yaw) / tf2::getYaw) — used where the reference uses them
geometry_msgs::msg::Quaternion quaternion_from_yaw(double yaw);
double yaw_from_quaternion(const geometry_msgs::msg::Quaternion & q);

}  // namespace nav2_social_mpc_controller
