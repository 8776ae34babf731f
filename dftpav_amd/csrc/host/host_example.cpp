// host_example.cpp — what the call site of TrajPlanner::RunMINCOParking (traj_manager.cpp:608-610)
// looks like against the drop-in; doubles as the compile/link check of the C++ host mirror and as a
// tiny end-to-end run when a GPU is present (exit 0 either way; prints what happened).
#include <cstdio>

#include "poly_traj_optimizer.hpp"
#include "traj_planner_steps.hpp"

using namespace plan_manage;

int main() {
  PolyTrajOptimizer opt;
  dftpav_params p;
  dftpav_default_params(&p);
  p.traj_resolution = 4;
  p.des_traj_resolution = 6;
  opt.setParam(p);
  // one forward segment of 3 pieces along +x at 2 m/s, generous corridor
  const int N = 3;
  std::vector<Mat> ini(1, Mat(2, 3)), fin(1, Mat(2, 3)), inner(1, Mat(2, N - 1));
  ini[0](0, 0) = 0.0; ini[0](0, 1) = 2.0;
  fin[0](0, 0) = 6.0; fin[0](0, 1) = 2.0;
  inner[0](0, 0) = 2.0; inner[0](0, 1) = 4.0;
  std::vector<double> Ts{3.0};
  size_t npts = (size_t)(N - 2) * (p.traj_resolution + 1) + 2 * (p.des_traj_resolution + 1);
  std::vector<std::vector<Mat>> polys(1);
  for (size_t k = 0; k < npts; k++) {
    Mat h(4, 4); // columns (n_x, n_y, p_x, p_y), outward normals
    h(0, 0) = 0; h(1, 0) = 1; h(2, 0) = 0; h(3, 0) = 10;
    h(0, 1) = 1; h(1, 1) = 0; h(2, 1) = 30; h(3, 1) = 0;
    h(0, 2) = 0; h(1, 2) = -1; h(2, 2) = 0; h(3, 2) = -10;
    h(0, 3) = -1; h(1, 3) = 0; h(2, 3) = -20; h(3, 3) = 0;
    polys[0].push_back(h);
  }
  bool ok = opt.OptimizeTrajectory(ini, fin, inner, Ts, polys, {1}, 0.0, 0.0);
  if (!ok && opt.last_error() == DFTPAV_E_NO_DEVICE) {
    std::printf("host mirror built; no HIP device here (DFTPAV_E_NO_DEVICE), nothing computed\n");
    return 0;
  }
  std::printf("OptimizeTrajectory -> %d, status %d, cost %.6f, %d iterations, %zu segments, dt %.4f\n", (int)ok,
              opt.last_status(), opt.last_cost(), opt.last_iterations(), opt.getMinJerkOptPtr()->size(),
              ok ? (*opt.getMinJerkOptPtr())[0].getDt() : 0.0);
  // ---- the read-out of the plan the way the server publishes it, and the plan as bytes
  bool ok3 = ok;
  if (ok) {
    TrajPlannerSteps out(opt.handle());
    std::vector<std::vector<State>> st;
    ok3 = out.GetStates(opt.solved_batch(), 1, 0.0, 0.01, 400, st);
    const MinJerkOptView &mj = (*opt.getMinJerkOptPtr())[0];
    const double total = mj.getDt() * N;
    ok3 = ok3 && !st[0].empty() && st[0].size() <= 400 && st[0].front().x == 0.0 && st[0].front().velocity > 1.9;
    std::vector<double> c((size_t)12 * N), dtp(1);
    ok3 = ok3 && dftpav_batch_coeffs(opt.solved_batch(), c.data(), dtp.data()) == DFTPAV_OK;
    const int pn[1] = {N}, sg[1] = {1};
    dftpav_layout lay{1, pn, sg, 4};
    std::vector<unsigned char> blob;
    ok3 = ok3 && out.SerializeTraj(lay, c.data(), dtp.data(), 1, 1, 0.0, blob) && out.setSurroundTrajsFromWire({blob});
    int S = 0, np = 0;
    ok3 = ok3 && dftpav_get_surround(opt.handle(), &S, &np, nullptr, nullptr, nullptr, nullptr, nullptr) == DFTPAV_OK && S == 1 && np == N;
    ok3 = ok3 && out.setSurroundTrajsFromWire({});
    std::printf("GetStates -> %zu states over %.3f s, last x = %.3f, v = %.3f; plan serialised to %zu bytes and installed as an obstacle -> %d\n",
                st.empty() ? (size_t)0 : st[0].size(), total, (st.empty() || st[0].empty()) ? 0.0 : st[0].back().x,
                (st.empty() || st[0].empty()) ? 0.0 : st[0].back().velocity, blob.size(), (int)ok3);
  }
  // a size error is reported the reference's way: false, no throw (traj_optimizer.cpp:44-48)
  polys[0].pop_back();
  bool bad = opt.OptimizeTrajectory(ini, fin, inner, Ts, polys, {1}, 0.0, 0.0);
  std::printf("short corridor -> %d (error %d)\n", (int)bad, opt.last_error());
  // ---- the steps around the solve, with the reference's member names
  dftpav_handle *h = nullptr;
  if (dftpav_create(&p, 0, &h) != DFTPAV_OK) return 1;
  TrajPlannerSteps steps(h);
  std::vector<unsigned char> cells(200 * 200, 127);
  for (int iy = 0; iy < 200; iy++) cells[150 + 200 * iy] = 80; // a wall at x = 15
  bool ok2 = steps.setObstacleMap(cells.data(), 200, 200, 0.3, -30.0, -30.0);
  std::vector<std::array<double, 3>> path;
  for (int i = 0; i <= 60; i++) path.push_back({0.15 * i, 0.0, 0.0}); // 9 m straight ahead
  std::vector<FlatTrajData> flat;
  dftpav_frontend_params fp{5.0, 8.0, 2.0, 4.0, 0.2, 2.85, 1.0, p.traj_resolution, p.des_traj_resolution};
  ok2 = ok2 && steps.getKinoNode(path, {0, 0, 0, 0.5}, {9.0, 0, 0, 0.2}, {0, 0}, fp, flat);
  ok2 = ok2 && flat.size() == 1 && steps.getRectangleConst(flat[0].states);
  double front = 0.0;
  if (ok2) front = steps.hPolys()[0](2, 1); // x of the front edge of the first rectangle: stops short of the wall
  // the analytic shot to a goal behind the wall at x = 15 collides, the one to a goal in front of it is free
  std::vector<std::array<double, 3>> shot;
  double shot_len = 0.0;
  ok2 = ok2 && steps.computeShotTraj({0.0, 0.0, 0.0}, {8.0, 3.0, 0.6}, shot, shot_len, 0.5, 0.2);
  const bool free_shot = steps.is_shot_sucess({0.0, 0.0, 0.0}, {8.0, 3.0, 0.6}, 0.5), blocked_shot = steps.is_shot_sucess({0.0, 0.0, 0.0}, {25.0, 0.0, 0.0}, 0.5);
  ok2 = ok2 && free_shot && !blocked_shot && shot.size() > 10 && shot_len > 8.0 && shot_len < 12.0;
  std::printf("computeShotTraj -> %zu poses over %.3f m; is_shot_sucess: free %d, through the wall %d\n", shot.size(), shot_len,
              (int)free_shot, (int)blocked_shot);
  std::vector<std::vector<PredictedState>> sur(1);
  for (int k = 0; k <= 10; k++) sur[0].push_back({-5.0 + 1.0 * k, 4.0, 0.0, 1.0, 0.0, 0.0, 1.0 * k});
  ok2 = ok2 && steps.ConverSurroundTrajFromPoints(sur);
  std::printf("getKinoNode -> %zu segment(s), %d pieces of %.3f s, %zu states; first rectangle's front edge at x = %.2f; "
              "surround fit -> %d (error %d)\n",
              flat.size(), ok2 ? flat[0].piece_nums : 0, ok2 ? flat[0].piece_duration : 0.0, ok2 ? flat[0].states.size() : (size_t)0,
              front, (int)ok2, steps.last_error());
  dftpav_destroy(h);
  return (ok && !bad && ok2 && ok3 && front < 15.0 && front > 3.0) ? 0 : 1;
}
