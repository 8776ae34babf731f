"""Seeded synthetic inputs for the solve path.

What the optimiser receives in the reference comes from
TrajPlanner::RunMINCOParking (traj_manager.cpp:509-641): per gear segment a
resampled front-end path (waypoints at piece ends, traj_manager.cpp:551-568), one
vehicle-aligned rectangle per constraint point (getRectangleConst,
traj_manager.cpp:1213-1469), flat boundary states (kino_astar.cpp:834-857) and
optionally moving-obstacle trajectories fitted with a uniform MINCO
(traj_manager.cpp:743-789) from the scripted cars of
ai_agent_planner/cfg/dynamicObs.yaml.  None of that is on the hot path; this
module produces inputs of the same shape from a seed so that CPU oracle and
GPU see identical arrays.

Config numbering follows BASELINE.json `configs` (1-based).
"""
from dataclasses import dataclass, field

import numpy as np

from .pods import BatchData, LayoutSpec, SurroundSet, dptr

# raw vehicle, common/basics/semantics.h:66-76
VEH_W, VEH_L, VEH_DCR, VEH_WB = 1.90, 4.88, 1.015, 2.85
HALF_MARGIN = 0.15          # minco_config.pb.txt:74
MAP_RESL = 0.3              # minco_config.pb.txt:57
NON_SIGUAV = 0.2            # kino_astar.h:207
LIMIT_BOUND = 10.0          # traj_manager.cpp:1219


@dataclass
class Scenario:
    """One homogeneous batch == B calls of OptimizeTrajectory with a shared layout."""
    name: str
    layout: LayoutSpec
    K: int
    Kd: int
    B: int
    ini_states: np.ndarray   # [B][M][6]
    fin_states: np.ndarray   # [B][M][6]
    inner_pts: np.ndarray    # [B][n_inner]
    init_Ts: np.ndarray      # [B][M]
    corridor: np.ndarray     # [B][Npts][H][4]
    t_now: float = 0.0
    help_eps: float = 0.0
    surround: SurroundSet = None
    meta: dict = field(default_factory=dict)

    @property
    def n_points(self):
        return self.layout.n_points(self.K, self.Kd)

    def batch_data(self):
        d = BatchData()
        d.ini_states = dptr(self.ini_states)
        d.fin_states = dptr(self.fin_states)
        d.inner_pts = dptr(self.inner_pts)
        d.init_Ts = dptr(self.init_Ts)
        d.corridor = dptr(self.corridor)
        d.t_now = float(self.t_now)
        d.help_eps = float(self.help_eps)
        return d

    def apply_resolution(self, params):
        params.traj_resolution = self.K
        params.des_traj_resolution = self.Kd
        return params

    def subset(self, idx):
        idx = np.asarray(idx)
        return Scenario(self.name, self.layout, self.K, self.Kd, len(idx),
                        np.ascontiguousarray(self.ini_states[idx]), np.ascontiguousarray(self.fin_states[idx]),
                        np.ascontiguousarray(self.inner_pts[idx]), np.ascontiguousarray(self.init_Ts[idx]),
                        np.ascontiguousarray(self.corridor[idx]), self.t_now, self.help_eps, self.surround,
                        dict(self.meta))

    def with_restarts(self, handle, K, b=0, sigma=0.3, lo=0.8, hi=1.25, seed=20240):
        """Element b of this batch as slot 0 of a batch of K, slots 1 .. K-1 its seeded restarts from the device's sampler
        (dftpav_sample_restarts: waypoints moved by N(0, sigma^2), durations scaled by U[lo, hi]) -- what the drop-in builds with
        DFTPAV_DROPIN_RESTARTS=K (csrc/host/dropin/traj_optimizer_hip.cpp); boundary states and corridor are shared."""
        inner, durs = handle.sample_restarts(self.inner_pts[b:b + 1], self.init_Ts[b:b + 1], K, sigma=sigma, lo=lo, hi=hi, seed=seed)
        rep = lambda a: np.ascontiguousarray(np.repeat(a[b:b + 1], K, axis=0))
        return Scenario(self.name + "+restarts", self.layout, self.K, self.Kd, K, rep(self.ini_states), rep(self.fin_states),
                        np.ascontiguousarray(inner), np.ascontiguousarray(durs), rep(self.corridor), self.t_now, self.help_eps, self.surround,
                        dict(self.meta))


# --------------------------------------------------------------------------
# nominal ("front-end") path: a kinematic car driven by a seeded control script
# --------------------------------------------------------------------------
def _drive(rng, pose, singul, duration, v_start, v_end, v_cruise, kappa_max=0.5, dt=0.005):
    """Integrate x' = s v cos(yaw), y' = s v sin(yaw), yaw' = s v kappa with a
    trapezoid speed magnitude and piecewise-linear curvature.  Returns a
    function t -> (x, y, yaw, v, a, kappa) by linear interpolation of the dense
    integration (the role of KinoAstar::evaluatePos, kino_astar.cpp:468-521)."""
    n = int(round(duration / dt))
    t = np.arange(n + 1) * dt
    acc = 1.5
    t_up = max((v_cruise - v_start) / acc, 0.0)
    t_dn = max((v_cruise - v_end) / acc, 0.0)
    if t_up + t_dn > duration:  # triangle
        sc = duration / (t_up + t_dn)
        t_up *= sc
        t_dn *= sc
        v_cruise = v_start + acc * t_up
    v = np.where(t < t_up, v_start + acc * t,
                 np.where(t > duration - t_dn, v_end + acc * (duration - t), v_cruise))
    a = np.where(t < t_up, acc, np.where(t > duration - t_dn, -acc, 0.0))
    n_knots = max(int(np.ceil(duration / 2.5)) + 1, 2)
    knots_t = np.linspace(0.0, duration, n_knots)
    knots_k = rng.uniform(-kappa_max, kappa_max, n_knots)
    kappa = np.interp(t, knots_t, knots_k)
    yaw = pose[2] + np.concatenate([[0.0], np.cumsum(singul * 0.5 * (v[1:] * kappa[1:] + v[:-1] * kappa[:-1]) * dt)])
    vx = singul * v * np.cos(yaw)
    vy = singul * v * np.sin(yaw)
    x = pose[0] + np.concatenate([[0.0], np.cumsum(0.5 * (vx[1:] + vx[:-1]) * dt)])
    y = pose[1] + np.concatenate([[0.0], np.cumsum(0.5 * (vy[1:] + vy[:-1]) * dt)])

    def ev(tq):
        tq = np.clip(np.asarray(tq, dtype=np.float64), 0.0, duration)
        return (np.interp(tq, t, x), np.interp(tq, t, y), np.interp(tq, t, yaw), np.interp(tq, t, v),
                np.interp(tq, t, a), np.interp(tq, t, kappa))

    return ev


def _flat_state(x, y, yaw, vel, acc, kappa, singul):
    """KinoAstar::getFlatState, kino_astar.cpp:834-857 (col-major 2x3)."""
    vel = singul * NON_SIGUAV if abs(vel) <= NON_SIGUAV else singul * vel
    c, s = np.cos(yaw), np.sin(yaw)
    v2 = (c * vel, s * vel)
    lon, lat = acc, kappa * vel * vel
    a2 = (c * lon - s * lat, s * lon + c * lat)
    return np.array([x, y, v2[0], v2[1], a2[0], a2[1]])


def _constraint_times(N, K, Kd, piece_dur):
    """Sampling times of traj_manager.cpp:551-568 (both piece-boundary samples kept)."""
    ts = []
    res_time = 0.0
    for i in range(N):
        res = Kd if (i == 0 or i == N - 1) else K
        for k in range(res + 1):
            ts.append(res_time + 1.0 * k / res * piece_dur)
        res_time += piece_dur
    return np.array(ts)


# --------------------------------------------------------------------------
# static map + rectangle corridor (analytic restatement of getRectangleConst)
# --------------------------------------------------------------------------
def _body_frame(px, py, yaw, ox, oy):
    dx = ox[None, :] - px[:, None]
    dy = oy[None, :] - py[:, None]
    c, s = np.cos(yaw)[:, None], np.sin(yaw)[:, None]
    return c * dx + s * dy, -s * dx + c * dy


def _rect_disc_dist(bx, by, x0, x1, y0, y1):
    """distance from points (bx,by) [P,O] to the axis-aligned boxes [x0,x1]x[y0,y1] ([P,1] each)."""
    ddx = np.maximum(np.maximum(x0 - bx, bx - x1), 0.0)
    ddy = np.maximum(np.maximum(y0 - by, by - y1), 0.0)
    return np.hypot(ddx, ddy)


def sample_obstacles(rng, n_obs, paths_xyyaw, arena=60.0, centre=(0.0, 0.0), clearance=0.3):
    """Discs, radius U[0.5,1.5], rejection-sampled to stay `clearance` away from
    the inflated footprint swept along every nominal path."""
    px, py, yaw = paths_xyyaw
    W = VEH_W + 2 * HALF_MARGIN
    L = VEH_L + 2 * HALF_MARGIN
    out = []
    tries = 0
    while len(out) < n_obs and tries < 200 * n_obs:
        tries += 1
        r = rng.uniform(0.5, 1.5)
        ox = centre[0] + rng.uniform(-arena / 2, arena / 2)
        oy = centre[1] + rng.uniform(-arena / 2, arena / 2)
        bx, by = _body_frame(px, py, yaw, np.array([ox]), np.array([oy]))
        d = _rect_disc_dist(bx, by, VEH_DCR - L / 2, VEH_DCR + L / 2, -W / 2, W / 2)
        if d.min() >= r + clearance:
            out.append((ox, oy, r))
    return np.array(out).reshape(-1, 3)


def occupancy_grid(obstacles, arena=80.0, centre=(0.0, 0.0), resolution=MAP_RESL):
    """The obstacle map getRectangleConst queries (GridMapND<uint8_t, 2>, semantics.h:351-358): a cell is
    OCCUPIED (80) when its centre lies inside a disc, FREE (127) otherwise.  Returns (grid [size_y][size_x], origin):
    cell (ix, iy) is centred at origin + (ix, iy) * resolution (semantics.cc:214-221)."""
    n = int(round(arena / resolution)) + 1
    origin = (centre[0] - arena / 2.0, centre[1] - arena / 2.0)
    xs = origin[0] + np.arange(n) * resolution
    ys = origin[1] + np.arange(n) * resolution
    X, Y = np.meshgrid(xs, ys)
    grid = np.full((n, n), 127, dtype=np.uint8)
    for ox, oy, r in np.asarray(obstacles).reshape(-1, 3):
        grid[(X - ox) ** 2 + (Y - oy) ** 2 <= r * r] = 80
    return grid, origin


def fill_polygons(grid, origin, resolution, poly_xy, poly_off, value=80):
    """Polygons into an occupancy grid the way DataRenderer::GetObstacleMap does it (data_renderer.cc:206-231): the
    vertices are snapped to cell coordinates (GridMapND::GetCoordUsingGlobalPosition: round((p - origin) / resolution),
    semantics.h:498-506), the integer polygon is filled even-odd over the cell centres, and the cells its edges pass
    through are set as well (cv::fillPoly draws the outline too).  OpenCV is not in this image; the rule is its documented
    one, a cell on the outline of a polygon may differ from cv::fillPoly's choice by one cell."""
    ny, nx = grid.shape
    for k in range(len(poly_off) - 1):
        pts = np.asarray(poly_xy[poly_off[k]:poly_off[k + 1]], dtype=np.float64)
        cx = np.rint((pts[:, 0] - origin[0]) / resolution).astype(np.int64)
        cy = np.rint((pts[:, 1] - origin[1]) / resolution).astype(np.int64)
        if cx.max() < 0 or cy.max() < 0 or cx.min() >= nx or cy.min() >= ny:
            continue
        ex0, ey0 = cx, cy
        ex1, ey1 = np.roll(cx, -1), np.roll(cy, -1)
        y_lo, y_hi = max(int(cy.min()), 0), min(int(cy.max()), ny - 1)
        for y in range(y_lo, y_hi + 1):                       # even-odd fill at the cell centres of row y
            yc = y + 0.0
            crossing = ((ey0 <= yc) & (ey1 > yc)) | ((ey1 <= yc) & (ey0 > yc))
            if not crossing.any():
                continue
            t = (yc - ey0[crossing]) / (ey1[crossing] - ey0[crossing])
            xs = np.sort(ex0[crossing] + t * (ex1[crossing] - ex0[crossing]))
            for a, b in zip(xs[0::2], xs[1::2]):
                i0, i1 = max(int(np.ceil(a)), 0), min(int(np.floor(b)), nx - 1)
                if i1 >= i0:
                    grid[y, i0:i1 + 1] = value
        for a0, b0, a1, b1 in zip(ex0, ey0, ex1, ey1):        # the outline
            n = int(max(abs(a1 - a0), abs(b1 - b0))) + 1
            xi = np.rint(np.linspace(a0, a1, n)).astype(np.int64)
            yi = np.rint(np.linspace(b0, b1, n)).astype(np.int64)
            ok = (xi >= 0) & (xi < nx) & (yi >= 0) & (yi < ny)
            grid[yi[ok], xi[ok]] = value
    return grid


def default_sim_map(fixture=None):
    """The reference's default simulation arena (playgrounds/ring_exp_v1.0: 36 obstacle polygons, ego start pose, the
    1500 x 1500 x 0.2 m obstacle map of agent 0) from tests/golden/default_map.npz (data only, written by
    tests/golden/make_default_map.py), rasterised as DataRenderer::GetObstacleMap lays it out around the ego vehicle:
    origin = round(ego - extent / 2) (data_renderer.cc:157-165).  Returns (grid [1500][1500] uint8, origin, resolution,
    ego_init (x, y, yaw))."""
    import os
    if fixture is None:
        fixture = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "default_map.npz")
    z = np.load(fixture)
    w, h, res = int(z["map_meta"][0]), int(z["map_meta"][1]), float(z["map_meta"][2])
    ego = z["ego_init"]
    origin = (float(np.round(ego[0] - h * res / 2.0)), float(np.round(ego[1] - w * res / 2.0)))
    grid = np.full((h, w), 127, dtype=np.uint8)
    fill_polygons(grid, origin, res, z["poly_xy"], z["poly_off"])
    return grid, origin, res, ego.copy()


def rectangle_corridor(px, py, yaw, obstacles, step=MAP_RESL, limit=LIMIT_BOUND):
    """One 4-plane rectangle per state (x, y, yaw): the growth rule of
    TrajPlanner::getRectangleConst (traj_manager.cpp:1296-1441) — sides
    +dy,+dx,-dy,-dx grown in turn by `step` until the new strip touches an
    obstacle or `limit` is reached — with disc obstacles tested analytically
    instead of through the occupancy grid (discs are inflated by step/2, the
    sampling pitch of CheckIfCollisionUsingLine, map_adapter.cpp:117-129).
    Returns [P][4][4] columns (n_x,n_y,p_x,p_y) in the order of
    traj_manager.cpp:1442-1465."""
    P = len(px)
    e = np.zeros((P, 4))
    alive = np.ones((P, 4), dtype=bool)
    if len(obstacles):
        bx, by = _body_frame(px, py, yaw, obstacles[:, 0], obstacles[:, 1])
        rad = obstacles[None, :, 2] + 0.5 * step
    hl, hw = VEH_L / 2.0, VEH_W / 2.0
    while alive.any():
        for i in range(4):
            a = alive[:, i]
            if not a.any():
                continue
            x0 = (VEH_DCR - hl - e[:, 3])[:, None]
            x1 = (VEH_DCR + hl + e[:, 1])[:, None]
            y0 = (-hw - e[:, 2])[:, None]
            y1 = (hw + e[:, 0])[:, None]
            if i == 0:
                sx0, sx1, sy0, sy1 = x0, x1, y1, y1 + step
            elif i == 1:
                sx0, sx1, sy0, sy1 = x1, x1 + step, y0, y1
            elif i == 2:
                sx0, sx1, sy0, sy1 = x0, x1, y0 - step, y0
            else:
                sx0, sx1, sy0, sy1 = x0 - step, x0, y0, y1
            if len(obstacles):
                occ = (_rect_disc_dist(bx, by, sx0, sx1, sy0, sy1) <= rad).any(axis=1)
            else:
                occ = np.zeros(P, dtype=bool)
            grow = a & ~occ
            e[grow, i] += step
            alive[a & occ, i] = False
            alive[grow & (e[:, i] >= limit), i] = False
    c, s = np.cos(yaw), np.sin(yaw)

    def pt(bxv, byv):
        return px + c * bxv - s * byv, py + s * bxv + c * byv

    H = np.zeros((P, 4, 4))
    p1 = pt(hl + VEH_DCR + e[:, 1], hw + e[:, 0])
    p2 = pt(hl + VEH_DCR + e[:, 1], -hw - e[:, 2])
    p3 = pt(-hl + VEH_DCR - e[:, 3], -hw - e[:, 2])
    p4 = pt(-hl + VEH_DCR - e[:, 3], hw + e[:, 0])
    H[:, 0] = np.stack([-s, c, p1[0], p1[1]], axis=1)
    H[:, 1] = np.stack([c, s, p2[0], p2[1]], axis=1)
    H[:, 2] = np.stack([s, -c, p3[0], p3[1]], axis=1)
    H[:, 3] = np.stack([-c, -s, p4[0], p4[1]], axis=1)
    return H


# --------------------------------------------------------------------------
# uniform-time quintic MINCO fit (numpy, dense) for the moving obstacles
# --------------------------------------------------------------------------
def minco_matrix(N):
    """The constant 6N x 6N matrix of MinJerkOpt::reset (poly_traj_utils.hpp:895-947), dense."""
    A = np.zeros((6 * N, 6 * N))
    A[0, 0] = 1.0
    A[1, 1] = 1.0
    A[2, 2] = 2.0
    for i in range(N - 1):
        r = 6 * i
        A[r + 3, r + 3:r + 6] = [6.0, 24.0, 60.0]
        A[r + 3, r + 9] = -6.0
        A[r + 4, r + 4:r + 6] = [24.0, 120.0]
        A[r + 4, r + 10] = -24.0
        A[r + 5, r:r + 6] = 1.0
        A[r + 6, r:r + 6] = 1.0
        A[r + 6, r + 6] = -1.0
        A[r + 7, r + 1:r + 6] = [1.0, 2.0, 3.0, 4.0, 5.0]
        A[r + 7, r + 7] = -1.0
        A[r + 8, r + 2:r + 6] = [2.0, 6.0, 12.0, 20.0]
        A[r + 8, r + 8] = -2.0
    A[6 * N - 3, 6 * N - 6:] = 1.0
    A[6 * N - 2, 6 * N - 5:] = [1.0, 2.0, 3.0, 4.0, 5.0]
    A[6 * N - 1, 6 * N - 4:] = [2.0, 6.0, 12.0, 20.0]
    return A


def minco_fit(inner, dT, head, tail):
    """MinJerkOpt::generate (poly_traj_utils.hpp:953-986) with a dense solve.
    inner [N-1][2], head/tail col-major 2x3.  Returns c [N][6][2] (row k multiplies s^k)."""
    N = inner.shape[0] + 1
    A = minco_matrix(N)
    rhs = np.zeros((6 * N, 2))
    head = np.asarray(head).reshape(3, 2)
    tail = np.asarray(tail).reshape(3, 2)
    rhs[0] = head[0]
    rhs[1] = head[1] * dT
    rhs[2] = head[2] * dT * dT
    for i in range(N - 1):
        rhs[6 * i + 5] = inner[i]
    rhs[6 * N - 3] = tail[0]
    rhs[6 * N - 2] = tail[1] * dT
    rhs[6 * N - 1] = tail[2] * dT * dT
    b = np.linalg.solve(A, rhs).reshape(N, 6, 2)
    tinv = (1.0 / dT) ** np.arange(6)
    return b * tinv[None, :, None]


# dynamicObs.yaml:4-32 (centre x, centre y, desired_vel, radius, inityaw)
DYNAMIC_OBS_YAML = [
    (-33.5759, 21.998, 4.5, 12.0, 4.57),
    (-37.804, 7.38686, 4.5, 12.0, 0.0),
    (-37.804, 1.121, 4.5, 12.0, -1.0),
    (-37.804, -5.0256, 4.5, 12.0, 4.7),
]


def predicted_states(pre_time=30.0, deltatime=1.0, cars=DYNAMIC_OBS_YAML, start_time=0.0):
    """The predicted state sequences ConverSurroundTrajFromPoints receives (traj_manager.cpp:743): scripted circle
    cars (parking_moving_obstacles.cc:41-57).  [S][n][7] = x, y, angle, velocity, acceleration, curvature, time_stamp."""
    out = []
    for (cx, cy, vel, rad, yaw0) in cars:
        omg = vel / rad
        ts = np.arange(0.0, pre_time + 1e-9, deltatime)
        ang = yaw0 + ts * omg
        st = np.zeros((len(ts), 7))
        st[:, 0] = rad * np.cos(ang) + cx
        st[:, 1] = rad * np.sin(ang) + cy
        st[:, 2] = ang + np.pi / 2
        st[:, 3] = vel
        st[:, 5] = 1.0 / rad
        st[:, 6] = start_time + ts
        out.append(st)
    return np.array(out)


def moving_obstacles(pre_time=30.0, deltatime=1.0, cars=DYNAMIC_OBS_YAML, start_time=0.0):
    """Scripted circle cars (parking_moving_obstacles.cc:41-57) predicted
    `pre_time` s at `deltatime` steps and fitted as TrajPlanner::
    ConverSurroundTrajFromPoints does (traj_manager.cpp:743-789)."""
    offs = [0]
    durs, coeffs, tot, st = [], [], [], []
    for (cx, cy, vel, rad, yaw0) in cars:
        omg = vel / rad
        ts = np.arange(0.0, pre_time + 1e-9, deltatime)
        ang = yaw0 + ts * omg
        pos = np.stack([rad * np.cos(ang) + cx, rad * np.sin(ang) + cy], axis=1)
        yaw = ang + np.pi / 2
        nP = len(ts) - 1
        dT = (ts[-1] - ts[0]) / nP

        def flat(k):  # state_to_flat_output, traj_manager.cpp:139-158
            c, s = np.cos(yaw[k]), np.sin(yaw[k])
            lat = (1.0 / rad) * vel ** 2
            return np.array([pos[k, 0], pos[k, 1], c * vel, s * vel, -s * lat, c * lat])

        c = minco_fit(pos[1:-1], dT, flat(0), flat(-1))  # [nP][6][2], row k = s^k
        # Piece::coeffMat is 2x6 with column 0 = t^5 (poly_traj_utils.hpp:993): col-major [x5,y5,...,x0,y0]
        cm = c[:, ::-1, :].reshape(nP, 12)
        coeffs.append(cm)
        durs.append(np.full(nP, dT))
        offs.append(offs[-1] + nP)
        tot.append(dT * nP)
        st.append(start_time)
    return SurroundSet(np.array(offs), np.concatenate(durs), np.concatenate(coeffs), np.array(tot), np.array(st))


def searched_paths(n_hyp, seed=0, gears=(1, -1), seg_duration=8.0, spacing=0.15, max_path=1024):
    """Stand-ins for KinoAstar's SampleTraj (kino_astar.cpp:566-610): dense pose lists (x, y, yaw in (-pi, pi]) at
    `spacing` metres along a kinematic-car path with the given gear sequence.  Returns paths [n_hyp][max_path][3],
    path_len, start_states [n_hyp][4], end_states [n_hyp][4], start_ctrl [n_hyp][2]."""
    paths = np.zeros((n_hyp, max_path, 3))
    plen = np.zeros(n_hyp, dtype=np.int32)
    ss, es, sc_ = np.zeros((n_hyp, 4)), np.zeros((n_hyp, 4)), np.zeros((n_hyp, 2))
    for h in range(n_hyp):
        rng = _rng(seed * 7919 + h)
        pose = np.array([rng.uniform(-5, 5), rng.uniform(-5, 5), rng.uniform(-np.pi, np.pi)])
        pts = [pose.copy()]
        v0 = rng.uniform(0.0, 1.5)
        for gi, sg in enumerate(gears):
            vs = v0 if gi == 0 else NON_SIGUAV
            ev = _drive(rng, pose, sg, seg_duration, vs, NON_SIGUAV, 3.0 if sg > 0 else 1.5)
            tt = np.linspace(0.0, seg_duration, 4001)
            x, y, yaw, _, _, _ = ev(tt)
            arc = np.concatenate([[0.0], np.cumsum(np.hypot(np.diff(x), np.diff(y)))])
            sa = np.arange(spacing, arc[-1], spacing)
            xs, ys, yw = np.interp(sa, arc, x), np.interp(sa, arc, y), np.interp(sa, arc, yaw)
            for k in range(len(sa)):
                pts.append(np.array([xs[k], ys[k], yw[k]]))
            pose = np.array([x[-1], y[-1], yaw[-1]])
            pts.append(pose.copy())
        P = np.array(pts)
        P[:, 2] = np.arctan2(np.sin(P[:, 2]), np.cos(P[:, 2]))  # normalize_angle
        n = min(len(P), max_path)
        paths[h, :n] = P[:n]
        plen[h] = n
        ss[h] = [P[0, 0], P[0, 1], P[0, 2], gears[0] * v0]
        es[h] = [P[n - 1, 0], P[n - 1, 1], P[n - 1, 2], gears[-1] * NON_SIGUAV]
        sc_[h] = [rng.uniform(-0.3, 0.3), rng.uniform(-1.0, 1.0)]
    return paths, plen, ss, es, sc_


# --------------------------------------------------------------------------
# scenario assembly
# --------------------------------------------------------------------------
def _rng(seed):
    return np.random.Generator(np.random.PCG64(seed))


def make_scenario(piece_nums, singuls, K, Kd, B, seed, n_hyp=None, n_obs=50, with_moving=False,
                  start_centre=(0.0, 0.0), name="custom", restart_sigma=0.3, dur_scale=(0.8, 1.25)):
    """B trajectories = n_hyp nominal paths ("hypotheses") through one static
    map x B/n_hyp seeded restarts each.  Restart r>0 of a hypothesis perturbs
    the inner waypoints by N(0, restart_sigma^2) and each segment duration by
    U[dur_scale]; restart 0 is the unperturbed front-end guess."""
    piece_nums = list(piece_nums)
    singuls = list(singuls)
    M = len(piece_nums)
    layout = LayoutSpec(piece_nums, singuls, H=4)
    if n_hyp is None:
        n_hyp = min(max(1, B // 16), 128)
    n_hyp = min(n_hyp, B)
    rng = _rng(seed)
    hyps = []
    all_x, all_y, all_yaw = [], [], []
    for h in range(n_hyp):
        hr = _rng(seed * 7919 + 1000 + h)
        pose = np.array([start_centre[0] + hr.uniform(-5, 5), start_centre[1] + hr.uniform(-5, 5),
                         hr.uniform(-np.pi, np.pi)])
        segs = []
        for i in range(M):
            N = piece_nums[i]
            sg = singuls[i]
            dur = N * 1.0  # traj_piece_duration 1.0, minco_config.pb.txt:76
            v_c = hr.uniform(2.0, 4.0) if sg > 0 else hr.uniform(1.0, 1.6)
            v0 = NON_SIGUAV if (i > 0 or hr.uniform() < 0.5) else hr.uniform(0.5, 2.0)
            v1 = NON_SIGUAV
            ev = _drive(hr, pose, sg, dur, v0, v1, v_c)
            tc = _constraint_times(N, K, Kd, dur / N)
            x, y, yaw, v, a, kap = ev(tc)
            xe, ye, yawe, ve, ae, ke = ev(np.array([0.0, dur]))
            ini = _flat_state(xe[0], ye[0], yawe[0], ve[0], ae[0], ke[0], sg)
            fin = _flat_state(xe[1], ye[1], yawe[1], ve[1], ae[1], ke[1], sg)
            tw = (np.arange(1, N)) * (dur / N)
            wx, wy, *_ = ev(tw)
            segs.append(dict(x=x, y=y, yaw=yaw, ini=ini, fin=fin, wp=np.stack([wx, wy], axis=1), dur=dur))
            dense = ev(np.arange(0.0, dur + 1e-9, 0.1))
            all_x.append(dense[0])
            all_y.append(dense[1])
            all_yaw.append(dense[2])
            pose = np.array([xe[1], ye[1], yawe[1]])
        hyps.append(segs)
    paths = (np.concatenate(all_x), np.concatenate(all_y), np.concatenate(all_yaw))
    obstacles = sample_obstacles(rng, n_obs, paths, arena=60.0, centre=start_centre) if n_obs > 0 else np.zeros((0, 3))
    n_inner = layout.n_inner
    npts = layout.n_points(K, Kd)
    ini_states = np.zeros((B, M, 6))
    fin_states = np.zeros((B, M, 6))
    inner_pts = np.zeros((B, n_inner))
    init_Ts = np.zeros((B, M))
    corridor = np.zeros((B, npts, 4, 4))
    hyp_of = np.zeros(B, dtype=np.int32)
    cors = []
    states = np.zeros((n_hyp, npts, 3))  # constraint-point poses, the statelist of getRectangleConst
    for h in range(n_hyp):
        px = np.concatenate([s["x"] for s in hyps[h]])
        py = np.concatenate([s["y"] for s in hyps[h]])
        yw = np.concatenate([s["yaw"] for s in hyps[h]])
        states[h] = np.stack([px, py, yw], axis=1)
        cors.append(rectangle_corridor(px, py, yw, obstacles))
    for b in range(B):
        h = b % n_hyp
        r = b // n_hyp
        hyp_of[b] = h
        br = _rng(seed * 104729 + 1000 * 3 + b)
        segs = hyps[h]
        off = 0
        for i in range(M):
            wp = segs[i]["wp"].copy()
            dur = segs[i]["dur"]
            if r > 0:
                wp = wp + br.normal(0.0, restart_sigma, wp.shape)
                dur = dur * br.uniform(dur_scale[0], dur_scale[1])
            inner_pts[b, off:off + wp.size] = wp.reshape(-1)
            off += wp.size
            init_Ts[b, i] = dur
            ini_states[b, i] = segs[i]["ini"]
            fin_states[b, i] = segs[i]["fin"]
        corridor[b] = cors[h]
    sur = moving_obstacles() if with_moving else None
    return Scenario(name, layout, K, Kd, B, ini_states, fin_states, inner_pts, init_Ts, corridor, 0.0, 0.0, sur,
                    meta=dict(seed=seed, n_hyp=n_hyp, obstacles=obstacles, hyp_of=hyp_of, states=states))


def baseline_config(config, B=None, seed=20240, n_hyp=None):
    """The five BASELINE.json configs (1-based)."""
    base = seed + 1000 * config
    if config == 1:   # single forward goal, 8 pieces, reference resolutions 16/32
        return make_scenario([8], [1], 16, 32, B or 1, base, n_hyp=n_hyp, name="cfg1_fwd8")
    if config == 2:   # one gear shift, 8+8 pieces, 32 pts/piece, 50 obstacles
        return make_scenario([8, 8], [1, -1], 32, 32, B or 1, base, n_hyp=n_hyp, name="cfg2_gear8+8")
    if config == 3:   # batch 256 restarts, 16 pieces
        return make_scenario([16], [1], 32, 32, B or 256, base, n_hyp=n_hyp, name="cfg3_batch256")
    if config == 4:   # batch 4096 over 8 GPUs (512 per rank)
        return make_scenario([16], [1], 32, 32, B or 4096, base, n_hyp=n_hyp, name="cfg4_batch4096")
    if config == 5:   # moving obstacles, 32 pieces, 64 pts/piece
        return make_scenario([32], [1], 64, 64, B or 1024, base, n_hyp=n_hyp, n_obs=30, with_moving=True,
                             start_centre=(-38.0, 5.0), name="cfg5_moving32")
    raise ValueError(config)
