// host_demo.cpp — drives the reference-shaped Optimizer (host/optimizer.hpp) the way
// SocialMPCController::computeVelocityCommands does (src/social_mpc_controller.cpp:235-256) for a few control
// ticks on a synthetic corridor scene, and (optionally) dumps every solve's C-ABI inputs/outputs so that the test
// suite can replay them through the CPU oracle.   usage: host_demo [n_ticks] [dump_prefix]
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>

#include "optimizer.hpp"

using namespace nav2_social_mpc_controller;

int main(int argc, char ** argv)
{
  const int n_ticks = argc > 1 ? std::atoi(argv[1]) : 3;
  if (argc > 2) setenv("SMPC_HOST_DUMP", argv[2], 1);
  OptimizerParams p;  // README.md:66-98 parameter set
  p.linear_solver_type = "DENSE_SCHUR"; p.param_tol = 1e-9; p.fn_tol = 1e-5; p.gradient_tol = 1e-8; p.max_iterations = 40;
  p.control_horizon_ = 18; p.parameter_block_length_ = 6; p.current_path_w = 1.0f; p.current_cmds_w = 0.5f;
  p.distance_w_ = 20; p.socialwork_w_ = 120; p.velocity_w_ = 10; p.angle_w_ = 250; p.agent_angle_w_ = 40;
  p.velocity_feasibility_w_ = 5; p.goal_align_w_ = 10; p.obstacle_w_ = 0.15; p.proxemics_w_ = 100; p.max_time = 1.5f;
  Optimizer opt;
  try {
    opt.initialize(p);
  } catch (const std::exception & e) {
    std::fprintf(stderr, "initialize failed: %s\n", e.what());
    return 2;
  }
  const float dt = 0.05f;
  // local costmap 80 x 80 @ 0.05 m around the origin with one inflated obstacle ahead-left of the robot
  nav2_costmap_2d::Costmap2D costmap(80, 80, 0.05, -2.0, -2.0);
  for (int r = 0; r < 80; ++r)
    for (int c = 0; c < 80; ++c) {
      const double wx = -2.0 + (c + 0.5) * 0.05, wy = -2.0 + (r + 0.5) * 0.05;
      const double d = std::hypot(wx - 1.2, wy - 0.6) - 0.25;
      costmap.getCharMap()[r * 80 + c] = d <= 0 ? 254 : (d <= 0.7 ? (unsigned char)std::floor(252.0 * std::exp(-3.0 * d)) : 0);
    }
  // obstacle-distance grid 120 x 120 @ 0.05 m: nearest obstacle index = that obstacle's cell for every cell
  obstacle_distance_msgs::msg::ObstacleDistance od;
  od.info.width = 120; od.info.height = 120; od.info.resolution = 0.05f;
  od.info.origin.position.x = -3.0; od.info.origin.position.y = -3.0;
  od.distances.assign(120 * 120, 1.0f);
  const unsigned ocx = (unsigned)((1.2 + 3.0) / 0.05), ocy = (unsigned)((0.6 + 3.0) / 0.05);
  od.indexes.assign(120 * 120, ocx + ocy * 120);
  people_msgs::msg::People people;
  { people_msgs::msg::Person a; a.position.x = 1.5; a.position.y = -0.4; a.velocity.x = -0.5; a.velocity.y = 0.1; people.people.push_back(a); }
  { people_msgs::msg::Person b; b.position.x = 0.8; b.position.y = 0.9; b.velocity.x = 0.0; b.velocity.y = 0.0; people.people.push_back(b); }
  geometry_msgs::msg::Twist speed; speed.linear.x = 0.3; speed.angular.z = 0.1;
  double x = 0.0, y = 0.0, yaw = 0.1;
  for (int tick = 0; tick < n_ticks; ++tick) {
    // trajectorized plan: 40 poses of a gentle left arc at 0.6 m/s (what PathTrajectorizer hands over)
    nav_msgs::msg::Path path; path.header.frame_id = "odom";
    std::vector<geometry_msgs::msg::TwistStamped> cmds;
    double px = x, py = y, pyaw = yaw;
    for (int i = 0; i < 40; ++i) {
      geometry_msgs::msg::PoseStamped ps; ps.pose.position.x = px; ps.pose.position.y = py; ps.pose.orientation = quaternion_from_yaw(pyaw);
      path.poses.push_back(ps);
      geometry_msgs::msg::TwistStamped c; c.twist.linear.x = 0.6; c.twist.angular.z = 0.25; cmds.push_back(c);
      px += 0.6 * std::cos(pyaw) * dt; py += 0.6 * std::sin(pyaw) * dt; pyaw += 0.25 * dt;
    }
    AgentsTrajectories people_proj;
    const bool ok = opt.optimize(path, people_proj, &costmap, od, cmds, people, speed, dt);
    std::printf("tick %d ok=%d status=%d iterations=%d cost=%.9e cmd0=(%.12f, %.12f) poses=%zu cmds=%zu proj=%zux%zu\n", tick, (int)ok,
                opt.last_status(), opt.last_iterations(), opt.last_final_cost(), ok ? cmds[0].twist.linear.x : 0.0,
                ok ? cmds[0].twist.angular.z : 0.0, path.poses.size(), cmds.size(), people_proj.size(), people_proj.empty() ? 0 : people_proj[0].size());
    if (!ok) return 1;
    // apply the first command (what the controller returns, src/social_mpc_controller.cpp:250-256) and move people
    speed.linear.x = cmds[0].twist.linear.x; speed.angular.z = cmds[0].twist.angular.z;
    x += speed.linear.x * std::cos(yaw) * dt; y += speed.linear.x * std::sin(yaw) * dt; yaw += speed.angular.z * dt;
    for (auto & q : people.people) { q.position.x += q.velocity.x * dt; q.position.y += q.velocity.y * dt; }
  }
  return 0;
}
