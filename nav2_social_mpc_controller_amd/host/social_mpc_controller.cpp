// social_mpc_controller.cpp — see social_mpc_controller.hpp. Restates src/social_mpc_controller.cpp:162-257.
#include "social_mpc_controller.hpp"

#include <cmath>
#include <stdexcept>

namespace nav2_social_mpc_controller
{
namespace
{
// Costmap2D::worldToMap (nav2_costmap_2d, third party): false left of / below the origin or beyond the last cell
bool world_to_map(const nav2_costmap_2d::Costmap2D & c, double wx, double wy, unsigned & mx, unsigned & my)
{
  if (wx < c.getOriginX() || wy < c.getOriginY()) return false;
  mx = static_cast<unsigned>((wx - c.getOriginX()) / c.getResolution());
  my = static_cast<unsigned>((wy - c.getOriginY()) / c.getResolution());
  return mx < c.getSizeInCellsX() && my < c.getSizeInCellsY();
}
// angles::shortest_angular_distance (ros/angles, ROS 2 form) = normalize_angle(to - from)
double shortest_angular_distance(double from, double to)
{
  const double r = std::fmod((to - from) + M_PI, 2.0 * M_PI);
  return r <= 0.0 ? r + M_PI : r - M_PI;
}
}  // namespace

void SocialMPCControllerCore::configure(const ControllerParams & params)
{
  fov_angle_ = params.fov_angle;
  trajectorizer_ = std::make_unique<PathTrajectorizer>();
  trajectorizer_->configure(params.trajectorizer);
  optimizer_ = std::make_unique<Optimizer>();
  optimizer_->initialize(params.optimizer);
}

people_msgs::msg::People SocialMPCControllerCore::filterPeople(
  const people_msgs::msg::People & people_unf, const geometry_msgs::msg::PoseStamped & robot_pose) const
{
  people_msgs::msg::People people;
  if (!costmap_) throw std::runtime_error("SocialMPCController: no costmap set");
  for (const auto & p : people_unf.people) {  // only people in the FOV of the robot (:196-214)
    unsigned mx, my;
    if (!world_to_map(*costmap_, p.position.x, p.position.y, mx, my)) continue;
    const float angle_to_person = std::atan2(p.position.y - robot_pose.pose.position.y, p.position.x - robot_pose.pose.position.x);
    const float robot_yaw = yaw_from_quaternion(robot_pose.pose.orientation);
    const float relative_angle = shortest_angular_distance(robot_yaw, angle_to_person);
    if (std::fabs(relative_angle) < fov_angle_) people.people.push_back(p);
  }
  people.header.frame_id = people_unf.header.frame_id;
  return people;
}

geometry_msgs::msg::TwistStamped SocialMPCControllerCore::computeVelocityCommands(
  const geometry_msgs::msg::PoseStamped & robot_pose, const geometry_msgs::msg::Twist & speed, void * /*goal_checker*/)
{
  if (!trajectorizer_ || !optimizer_) throw std::runtime_error("SocialMPCController::configure was not called");
  nav_msgs::msg::Path traj_path = plan_;
  std::vector<geometry_msgs::msg::TwistStamped> cmds;
  if (!trajectorizer_->trajectorize(traj_path, robot_pose, cmds)) {  // fallback of :180-189
    geometry_msgs::msg::TwistStamped cmd_vel;
    cmd_vel.header = robot_pose.header;
    cmd_vel.twist.linear.x = 0.1;
    cmd_vel.twist.linear.y = 0.0;
    cmd_vel.twist.angular.z = 0.0;
    last_optimized_ = false;
    return cmd_vel;
  }
  const std::vector<geometry_msgs::msg::TwistStamped> init_cmds = cmds;
  const people_msgs::msg::People people = filterPeople(people_, robot_pose);
  // (:216-231 transforms the people into the plan's frame on copies, i.e. not at all; frames are the caller's here)
  const float ts = trajectorizer_->getTimeStep();
  AgentsTrajectories projected_people;
  const bool optimized = optimizer_->optimize(traj_path, projected_people, costmap_, od_, cmds, people, speed, ts);
  if (!optimized) cmds = init_cmds;  // "Optimization failed, using initial commands" (:241-245)
  last_optimized_ = optimized;
  last_people_ = projected_people;
  last_path_ = traj_path;
  geometry_msgs::msg::TwistStamped cmd_vel;  // :250-256
  cmd_vel.header = cmds[0].header;
  cmd_vel.twist.linear.x = cmds[0].twist.linear.x;
  cmd_vel.twist.linear.y = 0;
  cmd_vel.twist.angular.z = cmds[0].twist.angular.z;
  return cmd_vel;
}

}  // namespace nav2_social_mpc_controller

// Test hook: the field-of-view filter for one scene with plain arrays. people [n][2] positions; keep [n] receives 0 / 1.
extern "C" int smpc_host_fov_filter(const double * people_xy, int n, const double * robot_pose, double fov_angle,
                                    double origin_x, double origin_y, int size_x, int size_y, double resolution, int * keep)
{
  using namespace nav2_social_mpc_controller;
  class Probe : public SocialMPCControllerCore { public: void fov(double a) { fov_angle_ = a; } } c;
  c.fov(fov_angle);
  nav2_costmap_2d::Costmap2D cm(size_x, size_y, resolution, origin_x, origin_y);
  c.setCostmap(&cm);
  people_msgs::msg::People in;
  for (int i = 0; i < n; ++i) {
    people_msgs::msg::Person p;
    p.position.x = people_xy[2 * i]; p.position.y = people_xy[2 * i + 1];
    p.name = std::to_string(i);
    in.people.push_back(p);
    keep[i] = 0;
  }
  geometry_msgs::msg::PoseStamped rp;
  rp.pose.position.x = robot_pose[0]; rp.pose.position.y = robot_pose[1];
  rp.pose.orientation = quaternion_from_yaw(robot_pose[2]);
  const people_msgs::msg::People out = c.filterPeople(in, rp);
  for (const auto & p : out.people) keep[std::stoi(p.name)] = 1;
  return (int)out.people.size();
}
