// controller_demo.cpp — drives the host mirror of SocialMPCController (host/social_mpc_controller.hpp) for a few control
// ticks: global plan -> PathTrajectorizer::trajectorize -> field-of-view filter -> Optimizer::optimize (people projection
// on the host, the solve on the GPU through the C ABI) -> first command, like
// SocialMPCController::computeVelocityCommands (src/social_mpc_controller.cpp:162-257).
// usage: controller_demo [n_ticks] [dump_prefix]   (dump_prefix: every solve's C-ABI inputs/outputs, see optimizer.cpp)
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "social_mpc_controller.hpp"

using namespace nav2_social_mpc_controller;

int main(int argc, char ** argv)
{
  const int n_ticks = argc > 1 ? std::atoi(argv[1]) : 3;
  if (argc > 2) setenv("SMPC_HOST_DUMP", argv[2], 1);
  ControllerParams cp;
  OptimizerParams & p = cp.optimizer;  // README.md:66-98 parameter set
  p.linear_solver_type = "DENSE_SCHUR"; p.param_tol = 1e-9; p.fn_tol = 1e-5; p.gradient_tol = 1e-8; p.max_iterations = 40;
  p.control_horizon_ = 18; p.parameter_block_length_ = 6; p.current_path_w = 1.0f; p.current_cmds_w = 0.5f;
  p.distance_w_ = 20; p.socialwork_w_ = 120; p.velocity_w_ = 10; p.angle_w_ = 250; p.agent_angle_w_ = 40;
  p.velocity_feasibility_w_ = 5; p.goal_align_w_ = 10; p.obstacle_w_ = 0.15; p.proxemics_w_ = 100; p.max_time = 1.5f;
  cp.trajectorizer.desired_linear_vel = 0.6; cp.trajectorizer.max_time = 1.5; cp.trajectorizer.time_step = 0.05;
  SocialMPCController ctrl;
  try {
    ctrl.configure(cp);
  } catch (const std::exception & e) {
    std::fprintf(stderr, "configure failed: %s\n", e.what());
    return 2;
  }
  nav2_costmap_2d::Costmap2D costmap(160, 160, 0.05, -4.0, -4.0);
  for (int r = 0; r < 160; ++r)
    for (int c = 0; c < 160; ++c) {
      const double wx = -4.0 + (c + 0.5) * 0.05, wy = -4.0 + (r + 0.5) * 0.05;
      const double d = std::hypot(wx - 1.4, wy - 0.7) - 0.25;
      costmap.getCharMap()[r * 160 + c] = d <= 0 ? 254 : (d <= 0.7 ? (unsigned char)std::floor(252.0 * std::exp(-3.0 * d)) : 0);
    }
  ctrl.setCostmap(&costmap);
  obstacle_distance_msgs::msg::ObstacleDistance od;
  od.info.width = 200; od.info.height = 200; od.info.resolution = 0.05f;
  od.info.origin.position.x = -5.0; od.info.origin.position.y = -5.0;
  od.distances.assign(200 * 200, 1.0f);
  const unsigned ocx = (unsigned)((1.4 + 5.0) / 0.05), ocy = (unsigned)((0.7 + 5.0) / 0.05);
  od.indexes.assign(200 * 200, ocx + ocy * 200);
  ctrl.setObstacleDistance(od);
  // global plan: a gentle left arc of 8 m, 0.05 m spacing, already in the robot's frame
  nav_msgs::msg::Path plan; plan.header.frame_id = "odom";
  {
    double px = 0.0, py = 0.0, pyaw = 0.05;
    for (int i = 0; i < 160; ++i) {
      geometry_msgs::msg::PoseStamped ps; ps.pose.position.x = px; ps.pose.position.y = py; ps.pose.orientation = quaternion_from_yaw(pyaw);
      plan.poses.push_back(ps);
      px += 0.05 * std::cos(pyaw); py += 0.05 * std::sin(pyaw); pyaw += 0.15 * 0.05;
    }
  }
  ctrl.setPlan(plan);
  people_msgs::msg::People people; people.header.frame_id = "odom";
  { people_msgs::msg::Person a; a.name = "ahead"; a.position.x = 1.6; a.position.y = -0.3; a.velocity.x = -0.5; a.velocity.y = 0.1; people.people.push_back(a); }
  { people_msgs::msg::Person b; b.name = "behind"; b.position.x = -1.0; b.position.y = 0.2; b.velocity.x = 0.4; people.people.push_back(b); }   // outside the FOV
  { people_msgs::msg::Person c; c.name = "far"; c.position.x = 30.0; c.position.y = 0.0; people.people.push_back(c); }                          // off the costmap
  geometry_msgs::msg::Twist speed; speed.linear.x = 0.3; speed.angular.z = 0.05;
  double x = 0.0, y = 0.0, yaw = 0.05;
  const double dt = 0.05;
  for (int tick = 0; tick < n_ticks; ++tick) {
    geometry_msgs::msg::PoseStamped rp; rp.header.frame_id = "odom";
    rp.pose.position.x = x; rp.pose.position.y = y; rp.pose.orientation = quaternion_from_yaw(yaw);
    ctrl.setPeople(people);
    const people_msgs::msg::People seen = ctrl.filterPeople(people, rp);
    const geometry_msgs::msg::TwistStamped cmd = ctrl.computeVelocityCommands(rp, speed);
    std::printf("tick %d optimized=%d people_in_fov=%zu cmd=(%.12f, %.12f) local_path=%zu projected=%zu\n", tick, (int)ctrl.lastOptimized(),
                seen.people.size(), cmd.twist.linear.x, cmd.twist.angular.z, ctrl.lastLocalPath().poses.size(), ctrl.lastProjectedPeople().size());
    if (!ctrl.lastOptimized()) return 1;
    speed.linear.x = cmd.twist.linear.x; speed.angular.z = cmd.twist.angular.z;
    x += speed.linear.x * std::cos(yaw) * dt; y += speed.linear.x * std::sin(yaw) * dt; yaw += speed.angular.z * dt;
    for (auto & q : people.people) { q.position.x += q.velocity.x * dt; q.position.y += q.velocity.y * dt; }
  }
  return 0;
}
