// host_example.cpp — what the call site of TrajPlanner::RunMINCOParking (traj_manager.cpp:608-610)
// looks like against the drop-in; doubles as the compile/link check of the C++ host mirror and as a
// tiny end-to-end run when a GPU is present (exit 0 either way; prints what happened).
#include <cstdio>

#include "poly_traj_optimizer.hpp"

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
  // a size error is reported the reference's way: false, no throw (traj_optimizer.cpp:44-48)
  polys[0].pop_back();
  bool bad = opt.OptimizeTrajectory(ini, fin, inner, Ts, polys, {1}, 0.0, 0.0);
  std::printf("short corridor -> %d (error %d)\n", (int)bad, opt.last_error());
  return (ok && !bad) ? 0 : 1;
}
