// e4_plan.h — host side: the lane plan of the penalty integral (solver.hip, block_eval E4) for one layout and one
// workgroup size.  See e4_group_size in device_types.h for the rule; this file turns it into the tables the kernel walks.
#pragma once
#include <utility>
#include <vector>

#include "device_types.h"

namespace dftpav {

struct E4Plan {
  int T = 0, nw = 0, rounds = 0, groups = 0, left = 0, lcap = 16;
  std::vector<int> gtab;  // [groups]        piece | (first j of the group) << 16
  std::vector<int> ltab;  // [left]          piece | j << 16 of a leftover point
  std::vector<int> wave;  // [rounds][nw][3] kind (32 / 16 group size, 0 leftovers, -1 idle), base (first group id / first leftover
                          //                 index), count (groups / leftover points the wave holds)
  std::vector<int> round; // [rounds][2]     leftovers evaluated in the round, index of the first of them
  std::vector<int> piece; // [Ntot][4]       first group, groups, first leftover, leftovers
};

inline E4Plan build_e4_plan(const DevLayout &L, int T) {
  E4Plan pl;
  pl.T = T;
  pl.nw = T / kWave;
  struct Grp {
    int piece, j0, G;
  };
  std::vector<Grp> grp;
  std::vector<std::pair<int, int>> left; // (piece, j) in point order
  pl.piece.assign((size_t)4 * L.Ntot, 0);
  for (int sg = 0; sg < L.M; sg++)
    for (int lp = 0; lp < L.piece_nums[sg]; lp++) {
      const int p = L.seg_piece0[sg] + lp;
      const int K = (lp == 0 || lp == L.piece_nums[sg] - 1) ? L.Kd : L.K;
      const int P = K + 1, G = e4_group_size(P), nf = G ? P / G : 0;
      pl.piece[4 * p + 0] = (int)grp.size();
      pl.piece[4 * p + 1] = nf;
      for (int g = 0; g < nf; g++) grp.push_back({p, g * G, G});
      pl.piece[4 * p + 2] = (int)left.size();
      pl.piece[4 * p + 3] = P - nf * G;
      for (int j = nf * G; j < P; j++) left.push_back({p, j});
    }
  pl.groups = (int)grp.size();
  pl.left = (int)left.size();
  // wave tasks: first every group (a wave holds 64 / G consecutive groups of one size), then the leftovers, 64 per wave;
  // so any round that evaluates leftovers comes after (or holds) the last group, and its chain pass sees every group sum
  struct Task {
    int kind, base;
    int lane[kWave];
  };
  std::vector<Task> tasks;
  for (size_t i = 0; i < grp.size();) {
    Task t;
    t.kind = grp[i].G;
    t.base = (int)i;
    for (int l = 0; l < kWave; l++) t.lane[l] = -1;
    const int per = kWave / t.kind;
    for (int q = 0; q < per && i < grp.size() && grp[i].G == t.kind; q++, i++)
      for (int l = 0; l < t.kind; l++) t.lane[q * t.kind + l] = grp[i].piece | ((grp[i].j0 + l) << 16);
    tasks.push_back(t);
  }
  for (size_t i = 0; i < left.size(); i += kWave) {
    Task t;
    t.kind = 0;
    t.base = (int)i;
    for (int l = 0; l < kWave; l++) t.lane[l] = i + l < left.size() ? (left[i + l].first | (left[i + l].second << 16)) : -1;
    tasks.push_back(t);
  }
  pl.rounds = (int)((tasks.size() + pl.nw - 1) / pl.nw);
  if (pl.rounds < 1) pl.rounds = 1;
  pl.gtab.resize(grp.size() ? grp.size() : 1, 0);
  for (size_t i = 0; i < grp.size(); i++) pl.gtab[i] = grp[i].piece | (grp[i].j0 << 16);
  pl.ltab.resize(left.size() ? left.size() : 1, 0);
  for (size_t i = 0; i < left.size(); i++) pl.ltab[i] = left[i].first | (left[i].second << 16);
  pl.wave.assign((size_t)pl.rounds * pl.nw * 3, -1);
  pl.round.assign((size_t)pl.rounds * 2, 0);
  int lmax = 0;
  for (int r = 0; r < pl.rounds; r++) {
    int cnt = 0, first = -1;
    for (int w = 0; w < pl.nw; w++) {
      const size_t ti = (size_t)r * pl.nw + w;
      if (ti >= tasks.size()) continue;
      const Task &t = tasks[ti];
      int held = 0;
      for (int l = 0; l < kWave; l += (t.kind ? t.kind : 1)) held += t.lane[l] >= 0;
      pl.wave[ti * 3 + 0] = t.kind;
      pl.wave[ti * 3 + 1] = t.base;
      pl.wave[ti * 3 + 2] = held;
      if (t.kind == 0) {
        if (first < 0) first = t.base;
        cnt += held;
      }
    }
    pl.round[2 * r + 0] = cnt;
    pl.round[2 * r + 1] = first < 0 ? 0 : first;
    if (cnt > lmax) lmax = cnt;
  }
  pl.lcap = ((lmax + 15) / 16) * 16;
  if (pl.lcap < 16) pl.lcap = 16;
  return pl;
}

} // namespace dftpav
