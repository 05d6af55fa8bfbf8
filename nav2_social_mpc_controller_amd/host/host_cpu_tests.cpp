// host_cpu_tests.cpp — CPU-only checks of the host rows restated in optimizer.cpp (no GPU, no solve).
#include <cassert>
#include <cmath>
#include <cstdio>
#include <stdexcept>

#include "optimizer.hpp"

using namespace nav2_social_mpc_controller;

static obstacle_distance_msgs::msg::ObstacleDistance make_od(unsigned w, unsigned h)
{
  obstacle_distance_msgs::msg::ObstacleDistance od;
  od.info.width = w; od.info.height = h; od.info.resolution = 0.1f;
  od.info.origin.position.x = -5.0; od.info.origin.position.y = -5.0;
  od.distances.assign(w * h, 1.0f);
  od.indexes.assign(w * h, 3 + 4 * w);  // every cell points at obstacle cell (3,4)
  return od;
}

int main()
{
  Optimizer opt;  // helpers do not need initialize()
  // people_to_status: pads with t = -1 up to 3, truncates above 3 (src/optimizer.cpp:467-479)
  people_msgs::msg::People ppl;
  { people_msgs::msg::Person p; p.position.x = 1; p.position.y = 2; p.velocity.x = 0.3; p.velocity.y = 0.4; ppl.people.push_back(p); }
  AgentsStates st = opt.people_to_status(ppl);
  assert(st.size() == 3 && st[0][3] == 0.0 && std::fabs(st[0][4] - 0.5) < 1e-15 && st[1][3] == -1.0 && st[2][3] == -1.0);
  for (int i = 0; i < 4; ++i) ppl.people.push_back(ppl.people[0]);
  assert(opt.people_to_status(ppl).size() == 3);
  // yaw round trip
  for (double yaw = -3.0; yaw <= 3.0; yaw += 0.37) assert(std::fabs(yaw_from_quaternion(quaternion_from_yaw(yaw)) - yaw) < 1e-14);
  // format_to_optimize: 40 poses, max_time 1.5 / dt 0.05 -> cut to 29; first call (previous == current) is the identity blend
  nav_msgs::msg::Path path;
  std::vector<geometry_msgs::msg::TwistStamped> cmds;
  for (int i = 0; i < 40; ++i) {
    geometry_msgs::msg::PoseStamped ps; ps.pose.position.x = 0.03 * i; ps.pose.position.y = 0.01 * i; ps.pose.orientation = quaternion_from_yaw(0.02 * i);
    path.poses.push_back(ps);
    geometry_msgs::msg::TwistStamped c; c.twist.linear.x = 0.6; c.twist.angular.z = 0.4; cmds.push_back(c);
  }
  geometry_msgs::msg::Twist speed; speed.linear.x = 0.2; speed.angular.z = -0.1;
  nav_msgs::msg::Path prev = path;
  AgentTrajectory rs = opt.format_to_optimize(path, prev, cmds, cmds, speed, 1.0f, 0.5f, 1.5f, 0.05f);
  assert(rs.size() == 29 && path.poses.size() == 29);
  assert(rs[0][4] == 0.2 && rs[0][5] == -0.1 && std::fabs(rs[1][4] - 0.6) < 1e-15 && std::fabs(rs[5][0] - 0.15) < 1e-12);
  assert(std::fabs(rs[3][3] - 3 * 0.05f) < 1e-12);
  // project_people: T+1 entries, valid agents first, phantom padding, robot removed again
  AgentsStates init = opt.people_to_status(people_msgs::msg::People{{}, {ppl.people[0]}});
  auto od = make_od(120, 120);
  AgentsTrajectories proj = opt.project_people(init, rs, od, 1.5f, 0.05f);
  assert(proj.size() == rs.size());
  for (auto & step : proj) { assert(step.size() == 3 && step[1][3] == -1.0 && step[2][3] == -1.0); }
  assert(proj[1][0][3] > 0.0 && std::isfinite(proj[28][0][0]) && proj[28][0][4] <= 0.5 + 1e-12);  // desiredVelocity cap 0.5
  // a 100 x 100 distance grid is declared "NOT valid": the person is dropped, every later step holds only phantoms
  auto od100 = make_od(100, 100);
  AgentsTrajectories dropped = opt.project_people(init, rs, od100, 1.5f, 0.05f);
  assert(dropped[0][0][3] == 0.0 && dropped[1][0][3] == -1.0 && dropped[5][2][3] == -1.0);
  // computeObstacle: returns agent - obstacle and throws on an empty grid / out-of-bounds cell
  auto diff = opt.computeObstacle({0.0, 0.0}, od);
  assert(std::fabs(diff[0] - (0.0 - (3 * 0.1f + -5.0))) < 1e-6 && std::fabs(diff[1] - (0.0 - (4 * 0.1f + -5.0))) < 1e-6);
  bool threw = false;
  try { obstacle_distance_msgs::msg::ObstacleDistance e; opt.computeObstacle({0, 0}, e); } catch (const std::runtime_error &) { threw = true; }
  assert(threw);
  threw = false;
  try { opt.computeObstacle({100.0, 0.0}, od); } catch (const std::runtime_error &) { threw = true; }
  assert(threw);
  // parameter validation mirrors src/optimizer.cpp:31-45
  threw = false;
  try { OptimizerParams bad; bad.linear_solver_type = "NOPE"; bad.validate(); } catch (const std::runtime_error & e) { threw = std::string(e.what()) == "Invalid parameter: linear_solver_type"; }
  assert(threw);
  std::puts("host_cpu_tests: all checks passed");
  return 0;
}
