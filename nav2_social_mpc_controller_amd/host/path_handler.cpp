// path_handler.cpp — see path_handler.hpp. nav2_util::geometry_utils (ROS 2 Humble, unpinned by the reference) is
// restated where it is used: euclidean_distance(a, b) = std::hypot(dx, dy), first_after_integrated_distance, min_by.
#include "path_handler.hpp"

#include <algorithm>
#include <cmath>

namespace mpc
{
namespace
{
double euclidean_distance(const geometry_msgs::msg::PoseStamped & a, const geometry_msgs::msg::PoseStamped & b)
{
  return std::hypot(a.pose.position.x - b.pose.position.x, a.pose.position.y - b.pose.position.y);
}
}  // namespace

nav_msgs::msg::Path PathHandler::transformGlobalPlan(const geometry_msgs::msg::PoseStamped & pose, double max_robot_pose_search_dist)
{
  if (global_plan_.poses.empty()) throw std::runtime_error("Received plan with zero length");
  const geometry_msgs::msg::PoseStamped & robot_pose = pose;  // same frame as the plan (see the header)
  auto & poses = global_plan_.poses;
  // first pose after max_robot_pose_search_dist of integrated path length (:56-59)
  auto upper = poses.end();
  {
    double dist = 0.0;
    for (auto it = poses.begin(); it != poses.end() - 1; ++it) {
      dist += euclidean_distance(*it, *(it + 1));
      if (dist > max_robot_pose_search_dist) { upper = it + 1; break; }
    }
  }
  // closest pose before it, first minimum (:61-66)
  auto begin_it = poses.begin();
  {
    double lowest = euclidean_distance(robot_pose, *begin_it);
    for (auto it = poses.begin() + 1; it != upper; ++it) {
      const double d = euclidean_distance(robot_pose, *it);
      if (d < lowest) { lowest = d; begin_it = it; }
    }
  }
  // poses outside the local costmap are dropped (:68-75)
  const double sx = costmap_->getSizeInCellsX() * costmap_->getResolution(), sy = costmap_->getSizeInCellsY() * costmap_->getResolution();
  const double dist_threshold = std::max(sx, sy) / 2.0;
  auto end_it = std::find_if(begin_it, poses.end(), [&](const geometry_msgs::msg::PoseStamped & ps) {
    return euclidean_distance(ps, robot_pose) > dist_threshold;
  });
  nav_msgs::msg::Path transformed_plan;
  transformed_plan.header = global_plan_.header;
  transformed_plan.header.stamp = robot_pose.header.stamp;
  for (auto it = begin_it; it != end_it; ++it) {  // the transform into the costmap frame is the identity here (:77-96)
    geometry_msgs::msg::PoseStamped ps = *it;
    ps.pose.position.z = 0.0;
    transformed_plan.poses.push_back(ps);
  }
  poses.erase(poses.begin(), begin_it);  // pruning (:98)
  if (transformed_plan.poses.empty()) throw std::runtime_error("Resulting plan has 0 poses in it.");
  return transformed_plan;
}

geometry_msgs::msg::Point PathHandler::getTransformedGoal(const double & goal_dist, const nav_msgs::msg::Path & transformed_plan,
                                                          const geometry_msgs::msg::PoseStamped & robot_pose)
{
  auto it = std::find_if(transformed_plan.poses.begin(), transformed_plan.poses.end(), [&](const geometry_msgs::msg::PoseStamped & ps) {
    return euclidean_distance(ps, robot_pose) >= goal_dist;
  });
  if (it == transformed_plan.poses.end()) it = std::prev(transformed_plan.poses.end());
  return it->pose.position;
}

}  // namespace mpc

// Test hook: one robot through the class above with plain arrays. plan [n][2] with its first `start` poses already
// pruned, robot (x, y), costmap of size_x x size_y cells of `resolution`. window [n][2]. Returns the window length, -1
// for "Received plan with zero length", -2 for "Resulting plan has 0 poses in it."; *new_start = poses pruned so far.
extern "C" int smpc_host_transform_global_plan(const double * plan, int n, int start, const double * robot_xy, double max_search_dist,
                                               unsigned size_x, unsigned size_y, double resolution, double * window, int * new_start)
{
  nav2_costmap_2d::Costmap2D cm(size_x, size_y, resolution, 0.0, 0.0);
  mpc::PathHandler ph(&cm);
  nav_msgs::msg::Path path;
  for (int i = start; i < n; ++i) {
    geometry_msgs::msg::PoseStamped ps;
    ps.pose.position.x = plan[2 * i]; ps.pose.position.y = plan[2 * i + 1];
    path.poses.push_back(ps);
  }
  ph.setPlan(path);
  geometry_msgs::msg::PoseStamped rp;
  rp.pose.position.x = robot_xy[0]; rp.pose.position.y = robot_xy[1];
  *new_start = start;
  try {
    const nav_msgs::msg::Path w = ph.transformGlobalPlan(rp, max_search_dist);
    *new_start = n - (int)ph.getPlan().poses.size();
    for (size_t k = 0; k < w.poses.size(); ++k) { window[2 * k] = w.poses[k].pose.position.x; window[2 * k + 1] = w.poses[k].pose.position.y; }
    return (int)w.poses.size();
  } catch (const std::runtime_error & e) {
    *new_start = n - (int)ph.getPlan().poses.size();
    return std::string(e.what()).find("zero length") != std::string::npos ? -1 : -2;
  }
}
