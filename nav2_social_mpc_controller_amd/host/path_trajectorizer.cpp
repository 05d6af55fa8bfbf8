// path_trajectorizer.cpp — CPU statement of PathTrajectorizer::trajectorize (reference src/path_trajectorizer.cpp:120-288).
#include "path_trajectorizer.hpp"

#include <cstdio>

namespace nav2_social_mpc_controller
{
namespace
{
// tf2::getYaw / setRPY(0, 0, yaw) for pure-yaw orientations
double yaw_of(const geometry_msgs::msg::Quaternion & q)
{
  return std::atan2(2.0 * (q.w * q.z + q.x * q.y), q.w * q.w + q.x * q.x - q.y * q.y - q.z * q.z);
}
geometry_msgs::msg::Quaternion from_yaw(double yaw)
{
  geometry_msgs::msg::Quaternion q;
  q.x = 0.0; q.y = 0.0; q.z = std::sin(yaw * 0.5); q.w = std::cos(yaw * 0.5);
  return q;
}
// One explicit-Euler step of the motion model (path_trajectorizer.hpp:106-135): body-frame velocity (vx, vy) rotated by
// theta — the lateral axis written as an angle of theta + pi/2 like the reference — and the heading advanced by wz.
struct Pose2 { double x, y, theta; };
Pose2 advance(const Pose2 & p, double vx, double vy, double wz, double dt)
{
  const double c = std::cos(p.theta), s = std::sin(p.theta);
  const double cl = std::cos(M_PI_2 + p.theta), sl = std::sin(M_PI_2 + p.theta);
  return Pose2{p.x + (vx * c + vy * cl) * dt, p.y + (vx * s + vy * sl) * dt, p.theta + wz * dt};
}
// angles::normalize_angle (ros/angles, ROS 2 form)
double normalize_angle(double a)
{
  const double r = std::fmod(a + M_PI, 2.0 * M_PI);
  return r <= 0.0 ? r + M_PI : r - M_PI;
}
}  // namespace

void PathTrajectorizer::configure(const TrajectorizerParams & p)
{
  omnidirectional_ = p.omnidirectional;
  desired_linear_vel_ = p.desired_linear_vel;
  lookahead_dist_ = p.lookahead_dist;
  max_angular_vel_ = p.max_angular_vel;
  time_step_ = p.time_step;
  max_steps_ = (int)std::round(p.max_time / p.time_step);  // :84
}

bool PathTrajectorizer::trajectorize(
  nav_msgs::msg::Path & path, const geometry_msgs::msg::PoseStamped & path_robot_pose,
  std::vector<geometry_msgs::msg::TwistStamped> & cmds)
{
  if (path.poses.size() < 2) return false;  // :123-127
  nav_msgs::msg::Path out;
  out.header.frame_id = path.header.frame_id;
  geometry_msgs::msg::PoseStamped robot_pose = path_robot_pose;
  out.poses.push_back(robot_pose);
  double rx = robot_pose.pose.position.x, ry = robot_pose.pose.position.y;
  double rtheta = yaw_of(robot_pose.pose.orientation);
  const auto & goal = path.poses.back().pose.position;
  double goal_dist = 1000.0;
  int steps = 0;
  while (goal_dist > 0.2 && steps < max_steps_) {
    // look-ahead point: the last plan pose inside the look-ahead circle, else the closest one (:156-178)
    int wp_index = -1;
    double min_dist = 100.0;
    for (int i = (int)path.poses.size() - 1; i >= 0; i--) {
      const double wx = path.poses[i].pose.position.x, wy = path.poses[i].pose.position.y;
      const double wp_dist = std::sqrt((rx - wx) * (rx - wx) + (ry - wy) * (ry - wy));
      if (wp_dist <= lookahead_dist_) { wp_index = i; break; }
      if (wp_dist < min_dist) { min_dist = wp_dist; wp_index = i; }
    }
    if (wp_index < 0) break;  // the reference indexes poses[-1] here (undefined); the batch ABI reports SMPC_TRAJ_NO_WAYPOINT
    const double wpx = path.poses[wp_index].pose.position.x, wpy = path.poses[wp_index].pose.position.y;
    const double dx = (wpx - rx) * std::cos(rtheta) + (wpy - ry) * std::sin(rtheta);
    const double dy = -(wpx - rx) * std::sin(rtheta) + (wpy - ry) * std::cos(rtheta);
    const double dtheta = normalize_angle(std::atan2(dy, dx));
    double vx = 0.0, vy = 0.0, wz = 0.0;
    if (omnidirectional_) {
      vx = desired_linear_vel_ * std::cos(dtheta);
      vy = desired_linear_vel_ * std::sin(dtheta);
    } else {
      const double point_dist2 = dx * dx + dy * dy;
      double curvature = 0.0;
      if (point_dist2 > 0.001) curvature = 2.0 * dy / point_dist2;
      vx = desired_linear_vel_;
      if (std::fabs(dtheta) > M_PI / 2.0) {  // rotate in place
        vx = 0.0;
        wz = max_angular_vel_ * (dtheta > 0 ? 1.0 : -1.0);
      } else {
        wz = vx * curvature;
      }
    }
    const Pose2 next = advance(Pose2{rx, ry, rtheta}, vx, vy, wz, time_step_);
    rx = next.x; ry = next.y; rtheta = next.theta;
    robot_pose.pose.position.x = rx;
    robot_pose.pose.position.y = ry;
    robot_pose.pose.orientation = from_yaw(rtheta);
    out.poses.push_back(robot_pose);
    geometry_msgs::msg::TwistStamped vel;
    vel.twist.linear.x = vx; vel.twist.linear.y = vy; vel.twist.angular.z = wz;
    cmds.push_back(vel);
    goal_dist = std::sqrt((rx - goal.x) * (rx - goal.x) + (ry - goal.y) * (ry - goal.y));
    steps++;
  }
  path.poses = out.poses;
  return true;
}

}  // namespace nav2_social_mpc_controller

// Test hook: one plan through the class above with plain arrays. plan [n][2], robot_pose [3] (x, y, yaw);
// out_path [max_steps+1][3] (yaw read back with getYaw), out_cmds [max_steps][3] (vx, vy, wz). Returns poses written,
// 0 when trajectorize() returned false.
extern "C" int smpc_host_trajectorize(const double * plan, int n, const double * robot_pose, int omnidirectional,
                                      double desired_linear_vel, double lookahead_dist, double max_angular_vel,
                                      double time_step, double max_time, double * out_path, double * out_cmds)
{
  using namespace nav2_social_mpc_controller;
  TrajectorizerParams tp;
  tp.omnidirectional = omnidirectional != 0; tp.desired_linear_vel = desired_linear_vel; tp.lookahead_dist = lookahead_dist;
  tp.max_angular_vel = max_angular_vel; tp.time_step = time_step; tp.max_time = max_time;
  PathTrajectorizer t;
  t.configure(tp);
  nav_msgs::msg::Path path;
  path.poses.resize(n);
  for (int i = 0; i < n; ++i) { path.poses[i].pose.position.x = plan[2 * i]; path.poses[i].pose.position.y = plan[2 * i + 1]; }
  geometry_msgs::msg::PoseStamped rp;
  rp.pose.position.x = robot_pose[0]; rp.pose.position.y = robot_pose[1];
  rp.pose.orientation.z = std::sin(robot_pose[2] * 0.5); rp.pose.orientation.w = std::cos(robot_pose[2] * 0.5);
  std::vector<geometry_msgs::msg::TwistStamped> cmds;
  if (!t.trajectorize(path, rp, cmds)) return 0;
  for (size_t k = 0; k < path.poses.size(); ++k) {
    const auto & q = path.poses[k].pose.orientation;
    out_path[3 * k] = path.poses[k].pose.position.x;
    out_path[3 * k + 1] = path.poses[k].pose.position.y;
    out_path[3 * k + 2] = k == 0 ? robot_pose[2] : std::atan2(2.0 * (q.w * q.z), q.w * q.w - q.z * q.z);
  }
  for (size_t k = 0; k < cmds.size(); ++k) {
    out_cmds[3 * k] = cmds[k].twist.linear.x; out_cmds[3 * k + 1] = cmds[k].twist.linear.y; out_cmds[3 * k + 2] = cmds[k].twist.angular.z;
  }
  return (int)path.poses.size();
}
