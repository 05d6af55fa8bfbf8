// optimizer.cpp — reference Optimizer interface over the HIP C ABI (see optimizer.hpp).
//
// Host-side rows restated here from the reference, because Optimizer::optimize needs them before the hot path starts
// (SURVEY.md §8f "next" rows f1, f2 — plain C++ on the host for now, once per control tick, not per LM iteration):
//   people_to_status   src/optimizer.cpp:454-482
//   format_to_optimize src/optimizer.cpp:484-551   (warm-start blend with TrajectoryMemory)
//   project_people     src/optimizer.cpp:554-671   (Social Force Model rollout of the people, sfm.hpp)
//   computeObstacle    src/optimizer.cpp:673-728
// The hot path itself (src/optimizer.cpp:241-446: problem build, ceres::Solve, unpack) is smpc_solve_batch.
#include "optimizer.hpp"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <list>
#include <stdexcept>

namespace nav2_social_mpc_controller
{

// ---- tf2 restated: Quaternion::setRPY(0, 0, yaw) and tf2::getYaw --------------------------------------------------
geometry_msgs::msg::Quaternion quaternion_from_yaw(double yaw)
{
  geometry_msgs::msg::Quaternion q;
  const double half = yaw * 0.5;
  q.x = 0.0; q.y = 0.0; q.z = std::sin(half); q.w = std::cos(half);
  return q;
}

double yaw_from_quaternion(const geometry_msgs::msg::Quaternion & q)
{
  const double sqx = q.x * q.x, sqy = q.y * q.y, sqz = q.z * q.z, sqw = q.w * q.w;
  const double sarg = -2 * (q.x * q.z - q.w * q.y) / (sqx + sqy + sqz + sqw);
  if (sarg <= -0.99999) return -2 * std::atan2(q.y, q.x);
  if (sarg >= 0.99999) return 2 * std::atan2(q.y, q.x);
  return std::atan2(2 * (q.x * q.y + q.w * q.z), sqw + sqx - sqy - sqz);
}

void OptimizerParams::validate() const
{
  if (solver_types.find(linear_solver_type) == solver_types.end()) {
    throw std::runtime_error("Invalid parameter: linear_solver_type");  // src/optimizer.cpp:44
  }
}

Optimizer::Optimizer() { smpc_params_default(&prm_); }

Optimizer::~Optimizer()
{
  if (handle_) smpc_destroy(handle_);
}

void Optimizer::initialize(const OptimizerParams params)  // src/optimizer.cpp:98-132
{
  params.validate();
  smpc_params_default(&prm_);
  prm_.obstacle_w = params.obstacle_w_;
  prm_.goal_align_w = params.goal_align_w_;
  prm_.velocity_feasibility_w = params.velocity_feasibility_w_;
  prm_.socialwork_w = params.socialwork_w_;
  prm_.distance_w = params.distance_w_;
  prm_.velocity_w = params.velocity_w_;
  prm_.angle_w = params.angle_w_;
  prm_.agent_angle_w = params.agent_angle_w_;
  prm_.proxemics_w = params.proxemics_w_;
  prm_.control_horizon = params.control_horizon_;
  prm_.parameter_block_length = params.parameter_block_length_;
  prm_.linear_solver_type = params.solver_types.at(params.linear_solver_type);
  prm_.max_iterations = params.max_iterations;
  prm_.fn_tol = params.fn_tol;
  prm_.gradient_tol = params.gradient_tol;
  prm_.param_tol = params.param_tol;
  max_time = params.max_time;
  current_path_w = params.current_path_w;
  current_cmds_w = params.current_cmds_w;
  device_ = params.device;
  // options_.max_solver_time_in_seconds (:131) is a wall-clock cap of 1.5-2 s; a batched device solve takes
  // microseconds to milliseconds, the cap is not modelled.
  if (handle_) { smpc_destroy(handle_); handle_ = nullptr; }
  handle_ = smpc_create(&prm_, device_);
  if (!handle_) throw std::runtime_error(std::string("smpc_create failed: ") + smpc_last_error());
}

// ---- src/optimizer.cpp:454-482 -------------------------------------------------------------------------------------
AgentsStates Optimizer::people_to_status(const people_msgs::msg::People & people)
{
  AgentsStates people_status;
  for (const auto & p : people.people) {
    const double yaw = std::atan2(p.velocity.y, p.velocity.x);
    const double lv = std::sqrt(p.velocity.x * p.velocity.x + p.velocity.y * p.velocity.y);
    people_status.push_back(AgentStatus{(double)p.position.x, (double)p.position.y, yaw, 0.0, lv, (double)p.velocity.z});
  }
  while ((int)people_status.size() < 3) people_status.push_back(AgentStatus{0.0, 0.0, 0.0, -1.0, 0.0, 0.0});
  while ((int)people_status.size() > 3) people_status.pop_back();
  return people_status;
}

// ---- src/optimizer.cpp:484-551 -------------------------------------------------------------------------------------
AgentTrajectory Optimizer::format_to_optimize(
  nav_msgs::msg::Path & path, const nav_msgs::msg::Path & previous_path,
  const std::vector<geometry_msgs::msg::TwistStamped> & cmds,
  const std::vector<geometry_msgs::msg::TwistStamped> & previous_cmds, const geometry_msgs::msg::Twist & speed,
  const float current_path_w, const float current_cmds_w, const float maxtime, const float timestep)
{
  const int maxsize = (int)std::round(maxtime / timestep);
  if ((int)path.poses.size() > maxsize) {
    std::vector<geometry_msgs::msg::PoseStamped> p(path.poses.begin(), path.poses.begin() + (maxsize - 1));
    path.poses = p;
  }
  AgentTrajectory robot_status;
  for (unsigned int i = 0; i < path.poses.size(); i++) {
    if (!previous_path.poses.empty() && i < previous_path.poses.size()) {
      geometry_msgs::msg::Pose smoothed;
      smoothed.position.x = current_path_w * path.poses[i].pose.position.x + (1.0 - current_path_w) * previous_path.poses[i].pose.position.x;
      smoothed.position.y = current_path_w * path.poses[i].pose.position.y + (1.0 - current_path_w) * previous_path.poses[i].pose.position.y;
      const double yaw_current = yaw_from_quaternion(path.poses[i].pose.orientation);
      const double yaw_prev = yaw_from_quaternion(previous_path.poses[i].pose.orientation);
      const double smoothed_yaw = current_path_w * yaw_current + (1.0 - current_path_w) * yaw_prev;
      smoothed.orientation = quaternion_from_yaw(smoothed_yaw);
      path.poses[i].pose = smoothed;
    }
    AgentStatus r;
    r[0] = path.poses[i].pose.position.x;
    r[1] = path.poses[i].pose.position.y;
    r[2] = yaw_from_quaternion(path.poses[i].pose.orientation);
    r[3] = i * timestep;
    if (i == 0) {
      r[4] = speed.linear.x;
      r[5] = speed.angular.z;
    } else {
      r[4] = current_cmds_w * cmds[i - 1].twist.linear.x + (1.0 - current_cmds_w) * previous_cmds[i - 1].twist.linear.x;
      r[5] = current_cmds_w * cmds[i - 1].twist.angular.z + (1.0 - current_cmds_w) * previous_cmds[i - 1].twist.angular.z;
    }
    robot_status.push_back(r);
  }
  return robot_status;
}

// ---- Social Force Model of the people projection (sfm.hpp), restated without Eigen ---------------------------------
namespace
{
struct V2 { double x = 0, y = 0; };
inline V2 operator+(V2 a, V2 b) { return {a.x + b.x, a.y + b.y}; }
inline V2 operator-(V2 a, V2 b) { return {a.x - b.x, a.y - b.y}; }
inline V2 operator*(double s, V2 a) { return {s * a.x, s * a.y}; }
inline V2 operator*(V2 a, double s) { return {s * a.x, s * a.y}; }
inline V2 operator/(V2 a, double s) { return {a.x / s, a.y / s}; }
inline double norm(V2 a) { return std::sqrt(a.x * a.x + a.y * a.y); }
inline V2 normalized(V2 a) { const double z = a.x * a.x + a.y * a.y; return z > 0 ? a / std::sqrt(z) : a; }
inline double wrap(double a) { while (a <= -M_PI) a += 2 * M_PI; while (a > M_PI) a -= 2 * M_PI; return a; }

struct SfmGoal { V2 center; double radius; };
struct SfmAgent {  // sfm.hpp:84-142 with the defaults project_people overrides
  int id = 0;
  V2 position, velocity;
  double yaw = 0, desiredVelocity = 0.6, radius = 0.35, linearVelocity = 0, angularVelocity = 0;
  std::list<SfmGoal> goals;
  std::vector<V2> obstacles1;
  V2 globalForce;
};
// sfm.hpp:38-58 Parameters defaults
const double kFactorDesired = 2.0, kFactorObstacle = 20.0, kSigmaObstacle = 0.2, kFactorSocial = 2.1, kLambda = 2.0,
             kGamma = 0.35, kN = 2.0, kNPrime = 3.0, kRelaxationTime = 0.5;

// SocialForceModel::computeForces(std::vector<Agent>&) (sfm.hpp:462-485); groupId is -1 for every agent here, so the
// group force is identically zero (sfm.hpp:339-343).
void sfm_compute_forces(std::vector<SfmAgent> & agents)
{
  for (unsigned idx = 0; idx < agents.size(); idx++) {
    SfmAgent & me = agents[idx];
    // computeDesiredForce (sfm.hpp:188-205)
    V2 desired;
    if (!me.goals.empty() && norm(me.goals.front().center - me.position) > me.goals.front().radius) {
      const V2 dir = normalized(me.goals.front().center - me.position);
      desired = kFactorDesired * (dir * me.desiredVelocity - me.velocity) / kRelaxationTime;
    } else {
      desired = (-1.0 * me.velocity) / kRelaxationTime;
    }
    // computeObstacleForce (sfm.hpp:207-235): obstacles1 entries are treated as obstacle POSITIONS
    V2 obstacle;
    if (!me.obstacles1.empty()) {
      for (const V2 & o : me.obstacles1) {
        const V2 minDiff = me.position - o;
        const double distance = norm(minDiff) - me.radius;
        obstacle = obstacle + kFactorObstacle * std::exp(-distance / kSigmaObstacle) * normalized(minDiff);
      }
      obstacle = obstacle / (double)me.obstacles1.size();
    }
    // computeSocialForce(index, agents) (sfm.hpp:237-281)
    V2 social;
    for (unsigned i = 0; i < agents.size(); i++) {
      if (i == idx) continue;
      const V2 diff = agents[i].position - me.position;
      const V2 diffDirection = normalized(diff);
      const V2 velDiff = me.velocity - agents[i].velocity;
      const V2 iv = kLambda * velDiff + diffDirection;
      const double il = norm(iv);
      const V2 idir = iv / il;
      const double a1 = wrap(std::atan2(idir.y, idir.x));
      const double a2 = wrap(std::atan2(diffDirection.y, diffDirection.x));
      const double theta = wrap(a2 - a1);
      const double B = kGamma * il;
      const double fv = -std::exp(-norm(diff) / B - (kNPrime * B * theta) * (kNPrime * B * theta));
      double thetaSign = -1.0;
      if (theta == 0) thetaSign = 0; else if (theta > 0) thetaSign = 1;  // sfm.hpp:265-270: 0 at theta == 0
      const double fa = -thetaSign * std::exp(-norm(diff) / B - (kN * B * theta) * (kN * B * theta));
      const V2 left{-idir.y, idir.x};
      social = social + kFactorSocial * (fv * idir + fa * left);
    }
    me.globalForce = desired + social + obstacle;
  }
}

// SocialForceModel::updatePosition(std::vector<Agent>&, dt) (sfm.hpp:525-551)
void sfm_update_position(std::vector<SfmAgent> & agents, double dt)
{
  for (SfmAgent & a : agents) {
    a.velocity = a.velocity + a.globalForce * dt;
    if (norm(a.velocity) > a.desiredVelocity) a.velocity = normalized(a.velocity) * a.desiredVelocity;
    const double initYaw = a.yaw;
    a.yaw = wrap(std::atan2(a.velocity.y, a.velocity.x));
    a.angularVelocity = wrap(a.yaw - initYaw) / dt;
    a.position = a.position + a.velocity * dt;
    a.linearVelocity = norm(a.velocity);
    if (!a.goals.empty() && norm(a.goals.front().center - a.position) <= a.goals.front().radius) a.goals.pop_front();
  }
}
}  // namespace

// ---- src/optimizer.cpp:673-728 -------------------------------------------------------------------------------------
std::array<double, 2> Optimizer::computeObstacle(
  const std::array<double, 2> & apos, const obstacle_distance_msgs::msg::ObstacleDistance & od)
{
  if (od.distances.empty() || od.indexes.empty()) throw std::runtime_error("ObstacleDistance grid is empty");
  if (od.info.width <= 0 || od.info.height <= 0) throw std::runtime_error("ObstacleDistance grid has invalid size");
  if (od.info.resolution <= 0.0) throw std::runtime_error("ObstacleDistance grid has invalid resolution");
  unsigned int xcell = (unsigned int)std::floor((apos[0] - od.info.origin.position.x) / od.info.resolution);
  unsigned int ycell = (unsigned int)std::floor((apos[1] - od.info.origin.position.y) / od.info.resolution);
  if (xcell >= (unsigned int)od.info.width || ycell >= (unsigned int)od.info.height)
    throw std::runtime_error("ObstacleDistance grid cell out of bounds");
  const unsigned int index = xcell + ycell * od.info.width;
  const unsigned int ob_idx = od.indexes[index];
  if (ob_idx >= od.info.width * od.info.height) throw std::runtime_error("ObstacleDistance grid index out of bounds");
  ycell = std::floor(ob_idx / od.info.width);
  xcell = ob_idx % od.info.width;
  const float x = xcell * od.info.resolution + od.info.origin.position.x;
  const float y = ycell * od.info.resolution + od.info.origin.position.y;
  // the reference returns the DIFFERENCE agent - obstacle (:724-727) although its caller stores it as a position
  return {apos[0] - (double)x, apos[1] - (double)y};
}

// ---- src/optimizer.cpp:554-671 -------------------------------------------------------------------------------------
AgentsTrajectories Optimizer::project_people(
  const AgentsStates & init_people, const AgentTrajectory & robot_path,
  const obstacle_distance_msgs::msg::ObstacleDistance & od, const float & maxtime, const float & timestep)
{
  const float naive_goal_time = maxtime;
  AgentsTrajectories people_traj;
  people_traj.push_back(init_people);
  std::vector<SfmAgent> agents;
  for (unsigned int i = 0; i < init_people.size(); i++) {
    if (init_people[i][3] == -1) continue;
    SfmAgent a;
    a.id = i + 1;
    a.position = {init_people[i][0], init_people[i][1]};
    a.yaw = init_people[i][2];
    a.linearVelocity = init_people[i][4];
    a.angularVelocity = init_people[i][5];
    a.velocity = {a.linearVelocity * std::cos(a.yaw), a.linearVelocity * std::sin(a.yaw)};
    a.desiredVelocity = 0.5;
    a.radius = 0.5;
    a.goals.push_back(SfmGoal{a.position + naive_goal_time * a.velocity, 0.25});
    if (od.info.width == 100 && od.info.height == 100) continue;  // "NOT valid" grid: the person is dropped (:598-603)
    const auto ob = computeObstacle({a.position.x, a.position.y}, od);
    a.obstacles1.push_back({ob[0], ob[1]});
    agents.push_back(a);
  }
  for (unsigned int i = 0; i + 1 < robot_path.size(); i++) {
    SfmAgent robot;
    robot.desiredVelocity = 0.6;
    robot.radius = 0.5;
    robot.id = 0;
    robot.position = {robot_path[i][0], robot_path[i][1]};
    robot.yaw = robot_path[i][2];
    robot.linearVelocity = robot_path[i][4];
    robot.angularVelocity = robot_path[i][5];
    robot.velocity = {robot.linearVelocity * std::cos(robot.yaw), robot.linearVelocity * std::sin(robot.yaw)};
    robot.goals.push_back(SfmGoal{{robot_path.back()[0], robot_path.back()[1]}, 0.25});
    agents.push_back(robot);
    sfm_compute_forces(agents);
    sfm_update_position(agents, timestep);
    agents.pop_back();
    for (SfmAgent & a : agents) {
      a.obstacles1.clear();
      const auto ob = computeObstacle({a.position.x, a.position.y}, od);
      a.obstacles1.push_back({ob[0], ob[1]});
    }
    AgentsStates humans;
    for (const SfmAgent & p : agents) {
      humans.push_back(AgentStatus{p.position.x, p.position.y, p.yaw, (double)((i + 1) * timestep), p.linearVelocity, p.angularVelocity});
    }
    while (humans.size() < init_people.size()) humans.push_back(AgentStatus{0.0, 0.0, 0.0, -1.0, 0.0, 0.0});
    people_traj.push_back(humans);
  }
  return people_traj;
}

// ---- src/optimizer.cpp:148-452 -------------------------------------------------------------------------------------
bool Optimizer::optimize(
  nav_msgs::msg::Path & path, AgentsTrajectories & people_proj, const nav2_costmap_2d::Costmap2D * costmap,
  const obstacle_distance_msgs::msg::ObstacleDistance & obstacles,
  std::vector<geometry_msgs::msg::TwistStamped> & cmds, const people_msgs::msg::People & people,
  const geometry_msgs::msg::Twist & speed, const float time_step)
{
  if (!handle_) throw std::runtime_error("Optimizer::initialize was not called");
  AgentsStates init_people = people_to_status(people);
  if (path.poses.size() < 2) return false;  // :158-162

  auto & memory = TrajectoryMemory::getInstance();
  if (memory.previous_path.poses.size() == 0) {  // :177-183
    memory.previous_path = path;
    memory.previous_cmds = cmds;
  }
  const nav_msgs::msg::Path previous_path = memory.previous_path;
  const std::vector<geometry_msgs::msg::TwistStamped> previous_cmds = memory.previous_cmds;

  AgentsStates optim_status = format_to_optimize(path, previous_path, cmds, previous_cmds, speed, current_path_w, current_cmds_w, max_time, time_step);
  people_proj = project_people(init_people, optim_status, obstacles, max_time, time_step);

  // ---- the hot path: one scene through the C ABI (replaces :197-446) ----
  const int T = (int)optim_status.size() - 1;  // optim_velocities.size() after pop_back (:237)
  if (T < 1) return false;
  const int N = (int)init_people.size();
  int nb = 0, P = 0;
  if (smpc_dims(&prm_, T, 1, nullptr, nullptr, &nb, &P, nullptr, nullptr) != SMPC_OK) return false;
  // evolving_poses[0]: orientation = setRPY(0,0,yaw) (:224-226); the functors read it back with tf2::getYaw
  const double pose0[3] = {optim_status[0][0], optim_status[0][1], yaw_from_quaternion(quaternion_from_yaw(optim_status[0][2]))};
  std::vector<double> init_params(P), path_pts(2 * (T + 1)), ppl((size_t)(T + 1) * 6 * N);
  for (int b = 0; b < nb; ++b) {  // parameter blocks alias optim_velocities[0..nb-1] (:254-261)
    init_params[2 * b] = optim_status[b][4];
    init_params[2 * b + 1] = optim_status[b][5];
  }
  for (int k = 0; k <= T; ++k) { path_pts[2 * k] = optim_status[k][0]; path_pts[2 * k + 1] = optim_status[k][1]; }
  const double goal_yaw = optim_status.back()[2];  // optim_headings.back().params[1] (:298)
  for (int k = 0; k <= T; ++k)
    for (int f = 0; f < 6; ++f)
      for (int a = 0; a < N; ++a) ppl[((size_t)k * 6 + f) * N + a] = people_proj[k][a][f];
  const uint8_t has_people = people.people.size() != 0 ? 1 : 0;  // :263
  const double origin[2] = {costmap->getOriginX(), costmap->getOriginY()};

  smpc_scene_batch sb{};
  sb.B = 1; sb.T = T; sb.N = N; sb.on_device = 0; sb.dt = (double)time_step;
  sb.pose0 = pose0; sb.init_params = init_params.data(); sb.path_pts = path_pts.data(); sb.goal_yaw = &goal_yaw;
  sb.people = ppl.data(); sb.has_people = &has_people;
  sb.costmap = costmap->getCharMap(); sb.costmap_shared = 1;
  sb.size_x = (int)costmap->getSizeInCellsX(); sb.size_y = (int)costmap->getSizeInCellsY();
  sb.costmap_origin = origin; sb.resolution = costmap->getResolution();

  std::vector<double> out_cmds(2 * (T + 1)), out_path(3 * (T + 1));
  int32_t status = SMPC_FAILURE, iterations = 0;
  double final_cost = 0.0;
  smpc_result_batch rb{};
  rb.cmds = out_cmds.data(); rb.path = out_path.data(); rb.status = &status; rb.iterations = &iterations; rb.final_cost = &final_cost;
  if (smpc_solve_batch(handle_, &sb, &rb) != SMPC_OK) throw std::runtime_error(std::string("smpc_solve_batch failed: ") + smpc_last_error());
  last_status_ = status; last_iterations_ = iterations; last_final_cost_ = final_cost;
  if (const char * prefix = std::getenv("SMPC_HOST_DUMP")) {
    // test hook: raw dump of this solve's C-ABI inputs and outputs (replayed through the CPU oracle by tests/)
    static int tick = 0;
    const std::string fn = std::string(prefix) + "_" + std::to_string(tick++) + ".bin";
    if (FILE * f = std::fopen(fn.c_str(), "wb")) {
      const int32_t hdr[8] = {T, N, P, sb.size_x, sb.size_y, has_people, status, iterations};
      const double scal[3] = {sb.dt, sb.resolution, goal_yaw};
      std::fwrite(hdr, sizeof(hdr), 1, f);
      std::fwrite(scal, sizeof(scal), 1, f);
      std::fwrite(pose0, sizeof(pose0), 1, f);
      std::fwrite(origin, sizeof(origin), 1, f);
      std::fwrite(init_params.data(), sizeof(double), init_params.size(), f);
      std::fwrite(path_pts.data(), sizeof(double), path_pts.size(), f);
      std::fwrite(ppl.data(), sizeof(double), ppl.size(), f);
      std::fwrite(out_cmds.data(), sizeof(double), out_cmds.size(), f);
      std::fwrite(out_path.data(), sizeof(double), out_path.size(), f);
      std::fwrite(sb.costmap, 1, (size_t)sb.size_x * sb.size_y, f);
      std::fclose(f);
    }
  }
  if (status == SMPC_FAILURE) return false;  // !summary.IsSolutionUsable() (:384-388)

  cmds.resize(T + 1);  // :412-419
  for (int i = 0; i <= T; ++i) {
    cmds[i].header = path.header;
    cmds[i].twist.linear.x = out_cmds[2 * i];
    cmds[i].twist.linear.y = 0.0;
    cmds[i].twist.angular.z = out_cmds[2 * i + 1];
  }
  path.poses.clear();  // :420-446
  for (int i = 0; i <= T; ++i) {
    geometry_msgs::msg::PoseStamped p;
    p.header = path.header;
    p.pose.position.x = out_path[3 * i];
    p.pose.position.y = out_path[3 * i + 1];
    p.pose.orientation = quaternion_from_yaw(out_path[3 * i + 2]);
    path.poses.push_back(p);
  }
  memory.previous_path = path;  // :448-449
  memory.previous_cmds = cmds;
  return true;
}

}  // namespace nav2_social_mpc_controller

// C-callable view of Optimizer::project_people for the test-suite (ctypes): arrays in, arrays out.
// init_people [N][6], robot_path [T+1][6], od_indexes [h][w] -> out [T+1][N][6]. Returns 0, or -1 if the reference
// would throw (message in `err`, up to errlen bytes).
extern "C" int smpc_host_project_people(const double * init_people, int N, const double * robot_path, int T,
                                        const uint32_t * od_indexes, int od_width, int od_height, float od_resolution,
                                        double od_origin_x, double od_origin_y, float max_time, float time_step,
                                        double * out, char * err, int errlen)
{
  using namespace nav2_social_mpc_controller;
  AgentsStates init(N);
  for (int a = 0; a < N; ++a) for (int f = 0; f < 6; ++f) init[a][f] = init_people[a * 6 + f];
  AgentTrajectory path(T + 1);
  for (int k = 0; k <= T; ++k) for (int f = 0; f < 6; ++f) path[k][f] = robot_path[k * 6 + f];
  obstacle_distance_msgs::msg::ObstacleDistance od;
  od.info.width = od_width; od.info.height = od_height; od.info.resolution = od_resolution;
  od.info.origin.position.x = od_origin_x; od.info.origin.position.y = od_origin_y;
  if (od_indexes) {
    od.indexes.assign(od_indexes, od_indexes + (size_t)od_width * od_height);
    od.distances.assign((size_t)od_width * od_height, 0.0f);
  }
  try {
    Optimizer opt;
    AgentsTrajectories traj = opt.project_people(init, path, od, max_time, time_step);
    for (size_t k = 0; k < traj.size(); ++k)
      for (int a = 0; a < N; ++a) for (int f = 0; f < 6; ++f) out[(k * N + a) * 6 + f] = traj[k][a][f];
  } catch (const std::exception & e) {
    if (err && errlen > 0) std::snprintf(err, errlen, "%s", e.what());
    return -1;
  }
  return 0;
}

// Test hook: Optimizer::format_to_optimize for one scene through plain arrays (tests/test_format.py compares it with
// the numpy restatement and with the device kernel behind smpc_format_to_optimize_batch).
// path [n][3] (x, y, yaw), cmds [n][2], prev_path [nprev][3], prev_cmds [nprev][2] (nprev = 0: empty memory, which
// optimize() first fills with the current path / cmds), speed [2]; out [n][6]; returns the number of states written.
extern "C" int smpc_host_format_to_optimize(const double * path, const double * cmds, int n, const double * prev_path,
                                            const double * prev_cmds, int nprev, const double * speed,
                                            float current_path_w, float current_cmds_w, float max_time, float time_step,
                                            double * out)
{
  using namespace nav2_social_mpc_controller;
  nav_msgs::msg::Path p, pp;
  std::vector<geometry_msgs::msg::TwistStamped> c(n), pc(nprev);
  p.poses.resize(n);
  for (int i = 0; i < n; ++i) {
    p.poses[i].pose.position.x = path[3 * i]; p.poses[i].pose.position.y = path[3 * i + 1];
    p.poses[i].pose.orientation = quaternion_from_yaw(path[3 * i + 2]);
    c[i].twist.linear.x = cmds[2 * i]; c[i].twist.angular.z = cmds[2 * i + 1];
  }
  pp.poses.resize(nprev);
  for (int i = 0; i < nprev; ++i) {
    pp.poses[i].pose.position.x = prev_path[3 * i]; pp.poses[i].pose.position.y = prev_path[3 * i + 1];
    pp.poses[i].pose.orientation = quaternion_from_yaw(prev_path[3 * i + 2]);
    pc[i].twist.linear.x = prev_cmds[2 * i]; pc[i].twist.angular.z = prev_cmds[2 * i + 1];
  }
  if (nprev == 0) { pp = p; pc = c; }  // src/optimizer.cpp:177-183
  geometry_msgs::msg::Twist sp;
  sp.linear.x = speed[0]; sp.angular.z = speed[1];
  Optimizer opt;
  const AgentTrajectory st = opt.format_to_optimize(p, pp, c, pc, sp, current_path_w, current_cmds_w, max_time, time_step);
  for (size_t i = 0; i < st.size(); ++i) for (int f = 0; f < 6; ++f) out[i * 6 + f] = st[i][f];
  return (int)st.size();
}
