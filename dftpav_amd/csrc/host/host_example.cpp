// host_example.cpp — what the call site of TrajPlanner::RunMINCOParking (traj_manager.cpp:608-610)
// looks like against the drop-in; doubles as the compile/link check of the C++ host mirror and as a
// tiny end-to-end run when a GPU is present (exit 0 either way; prints what happened).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <thread>

#include <hip/hip_runtime_api.h>

#include "poly_traj_optimizer.hpp"
#include "traj_planner_steps.hpp"

using namespace plan_manage;

// `host_example --ranks N`: the multi-GPU hand-off of SURVEY 8(e) from a C++ host, through the C-ABI only -- N ranks (one
// thread and one dftpav_handle per GPU here; one process per GPU works the same way, the 128-byte id then travels over the
// host's own channel), a batch of restarts sharded contiguously, every rank solves its shard, ONE RCCL all-gather of the
// 16-byte records, every rank ends with every result.  Needs N GPUs (RCCL ranks cannot share a device).
// The gathered buffer of dftpav_batch_allgather_results -> the records of the global batch in order: rank r's shard sits at the
// start of block r (dftpav_comm_layout), the pad behind a short shard is zero.  Used after the real collective below and by
// `--placement` (no GPU), which stands in for the collective with a plain copy.
static bool unpack_gathered(const std::vector<unsigned char> &all, int nranks, int B, std::vector<unsigned char> &out) {
  out.assign((size_t)B * 16, 0);
  int expect_first = 0;
  for (int r = 0; r < nranks; r++) {
    int first = 0, count = 0, block = 0;
    if (dftpav_comm_layout(B, nranks, r, &first, &count, &block) != DFTPAV_OK) return false;
    if (first != expect_first || count < 0 || count > block || all.size() != (size_t)nranks * block * 16) return false; // contiguous, in rank order
    std::memcpy(&out[(size_t)first * 16], &all[(size_t)r * block * 16], (size_t)count * 16);
    for (size_t q = ((size_t)r * block + count) * 16; q < (size_t)(r + 1) * block * 16; q++)
      if (all[q] != 0) return false; // the pad of a short shard
    expect_first = first + count;
  }
  return expect_first == B;
}

// `host_example --placement N B`: the sharding and the placement of the records for N ranks and B trajectories WITHOUT a device:
// every rank's send buffer is built as dftpav_batch_allgather_results builds it (its `count` records, zero up to `block`),
// the collective is a plain concatenation, and the unpacked result must be records 0 .. B-1 in order.  Run by the CPU tests
// for uneven shards (B % N != 0, B < N ...), so that the first real N > 1 run cannot fail on bookkeeping.
static int run_placement(int nranks, int B) {
  int block0 = 0;
  if (dftpav_comm_layout(B, nranks, 0, nullptr, nullptr, &block0) != DFTPAV_OK) {
    std::printf("--placement %d %d: invalid\n", nranks, B);
    return 1;
  }
  std::vector<unsigned char> all((size_t)nranks * block0 * 16, 0xff); // what the collective overwrites entirely
  for (int r = 0; r < nranks; r++) {
    int first = 0, count = 0, block = 0;
    dftpav_comm_layout(B, nranks, r, &first, &count, &block);
    if (block != block0) return 1;
    std::vector<unsigned char> send((size_t)block * 16, 0);
    for (int i = 0; i < count; i++) { // record of global trajectory g: cost = g + 0.5, status = -g, iterations = 7 g
      const int g = first + i;
      const double c = g + 0.5;
      const int st = -g, it = 7 * g;
      std::memcpy(&send[(size_t)i * 16], &c, 8);
      std::memcpy(&send[(size_t)i * 16 + 8], &st, 4);
      std::memcpy(&send[(size_t)i * 16 + 12], &it, 4);
    }
    std::memcpy(&all[(size_t)r * block * 16], send.data(), send.size()); // the all-gather, stubbed
  }
  std::vector<unsigned char> out;
  bool ok = unpack_gathered(all, nranks, B, out);
  for (int g = 0; ok && g < B; g++) {
    double c;
    int st, it;
    std::memcpy(&c, &out[(size_t)g * 16], 8);
    std::memcpy(&st, &out[(size_t)g * 16 + 8], 4);
    std::memcpy(&it, &out[(size_t)g * 16 + 12], 4);
    ok = c == g + 0.5 && st == -g && it == 7 * g;
  }
  std::printf("--placement %d ranks, %d trajectories, blocks of %d: %s\n", nranks, B, block0, ok ? "every record in its place" : "FAILED");
  return ok ? 0 : 1;
}

static int run_ranks(int nranks) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < nranks) {
    std::printf("--ranks %d: %d HIP device(s) here, nothing run\n", nranks, ndev);
    return 0;
  }
  const int N = 3, B = 8 * nranks + 3; // uneven shards on purpose
  dftpav_params p;
  dftpav_default_params(&p);
  p.traj_resolution = 4;
  p.des_traj_resolution = 6;
  const int pn[1] = {N}, sg[1] = {1};
  dftpav_layout lay{1, pn, sg, 4};
  const int npts = dftpav_num_points(&p, &lay);
  unsigned char id[DFTPAV_UNIQUE_ID_BYTES];
  if (dftpav_comm_unique_id(id) != DFTPAV_OK) {
    std::printf("--ranks: RCCL is not loadable here, nothing run\n");
    return 0;
  }
  std::vector<int> ok(nranks, 0);
  std::vector<double> wait_ms(nranks, 0.0), solve_ms(nranks, 0.0);
  std::vector<std::vector<unsigned char>> gathered(nranks);
  std::vector<std::thread> th;
  for (int r = 0; r < nranks; r++)
    th.emplace_back([&, r] {
      int first = 0, count = 0, block = 0;
      dftpav_comm_layout(B, nranks, r, &first, &count, &block);
      dftpav_handle *h = nullptr;
      dftpav_batch *b = nullptr;
      if (dftpav_create(&p, r, &h) != DFTPAV_OK) return;
      bool good = dftpav_comm_create(h, nranks, r, id) == DFTPAV_OK && dftpav_batch_create(h, &lay, count, &b) == DFTPAV_OK;
      // trajectory g of the global batch: the straight run of main() with its middle waypoints moved by g centimetres
      std::vector<double> ini((size_t)count * 6, 0.0), fin((size_t)count * 6, 0.0), inner((size_t)count * 2 * (N - 1)), Ts(count, 3.0),
          cor((size_t)count * npts * 16);
      for (int i = 0; i < count; i++) {
        const int g = first + i;
        ini[6 * i + 2] = 2.0; fin[6 * i + 0] = 6.0; fin[6 * i + 2] = 2.0;
        inner[4 * i + 0] = 2.0; inner[4 * i + 1] = 0.01 * g; inner[4 * i + 2] = 4.0; inner[4 * i + 3] = -0.01 * g;
        for (int k = 0; k < npts; k++) {
          const double hp[16] = {0, 1, 0, 10, 1, 0, 30, 0, 0, -1, 0, -10, -1, 0, -20, 0};
          std::memcpy(&cor[((size_t)i * npts + k) * 16], hp, sizeof(hp));
        }
      }
      dftpav_batch_data d{};
      d.ini_states = ini.data(); d.fin_states = fin.data(); d.inner_pts = inner.data(); d.init_Ts = Ts.data(); d.corridor = cor.data();
      void *all = nullptr;
      good = good && dftpav_batch_upload(b, &d) == DFTPAV_OK && dftpav_batch_solve_async(b) == DFTPAV_OK;
      good = good && hipSetDevice(r) == hipSuccess && hipMalloc(&all, (size_t)nranks * block * 16) == hipSuccess;
      // how long a rank waits for the COLLECTIVE after its own solve is done: the all-gather completes at the pace of the slowest
      // rank, so this is the coupling between the ranks (the first N > 1 run shows it directly)
      const auto t0 = std::chrono::steady_clock::now();
      good = good && dftpav_batch_sync(b) == DFTPAV_OK; // this rank's solve is complete
      const auto t1 = std::chrono::steady_clock::now();
      good = good && dftpav_batch_allgather_results(b, B, all) == DFTPAV_OK && dftpav_batch_sync(b) == DFTPAV_OK;
      const auto t2 = std::chrono::steady_clock::now();
      solve_ms[r] = std::chrono::duration<double, std::milli>(t1 - t0).count();
      wait_ms[r] = std::chrono::duration<double, std::milli>(t2 - t1).count();
      gathered[r].resize((size_t)nranks * block * 16);
      good = good && hipMemcpy(gathered[r].data(), all, gathered[r].size(), hipMemcpyDeviceToHost) == hipSuccess;
      std::vector<double> cost(count);
      good = good && dftpav_batch_results(b, nullptr, cost.data(), nullptr, nullptr, nullptr, nullptr, nullptr, nullptr) == DFTPAV_OK;
      for (int i = 0; good && i < count; i++) { // this rank's own block of the gathered buffer holds its own results
        double c;
        std::memcpy(&c, &gathered[r][((size_t)r * block + i) * 16], 8);
        good = c == cost[i];
      }
      if (!good) std::printf("rank %d: %s\n", r, dftpav_last_error(h));
      ok[r] = good;
      if (all) (void)hipFree(all);
      dftpav_batch_destroy(b);
      dftpav_destroy(h);
    });
  for (auto &t : th) t.join();
  bool all_ok = true;
  for (int r = 0; r < nranks; r++) all_ok = all_ok && ok[r] && gathered[r] == gathered[0]; // every rank sees every result
  std::vector<unsigned char> in_order;
  all_ok = all_ok && unpack_gathered(gathered[0], nranks, B, in_order);
  for (int r = 0; r < nranks; r++) std::printf("rank %d: solve %.2f ms, then %.2f ms in the collective\n", r, solve_ms[r], wait_ms[r]);
  std::printf("--ranks %d: %d trajectories sharded, solved, one all-gather of 16-byte records -> %s\n", nranks, B, all_ok ? "identical on every rank" : "FAILED");
  return all_ok ? 0 : 1;
}

int main(int argc, char **argv) {
  if (argc >= 3 && std::strcmp(argv[1], "--ranks") == 0) return run_ranks(std::atoi(argv[2]));
  if (argc >= 4 && std::strcmp(argv[1], "--placement") == 0) return run_placement(std::atoi(argv[2]), std::atoi(argv[3]));
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
  // the same call in the reference's own floating-point order (the bits of the CPU planner)
  opt.setReferenceOrder(true);
  const bool ok_ref = opt.OptimizeTrajectory(ini, fin, inner, Ts, polys, {1}, 0.0, 0.0);
  std::printf("  in reference order -> %d (order %d), status %d, cost %.6f, %d iterations\n", (int)ok_ref, opt.last_order(), opt.last_status(),
              opt.last_cost(), opt.last_iterations());
  ok = ok && ok_ref && opt.last_order() == DFTPAV_ORDER_REFERENCE;
  opt.setReferenceOrder(false);
  ok = ok && opt.OptimizeTrajectory(ini, fin, inner, Ts, polys, {1}, 0.0, 0.0);
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
