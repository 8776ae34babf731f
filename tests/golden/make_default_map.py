"""Writes tests/golden/default_map.npz from the reference's default simulation arena (run in the build container,
where /root/reference exists; the GPU box only sees the fixture):

  src/Sim/core/playgrounds/ring_exp_v1.0/obstacles_norm.json   the 37 static obstacle polygons (GeoJSON MultiPolygon rings)
  src/Sim/core/playgrounds/ring_exp_v1.0/vehicle_set.json      the ego vehicle's initial state (vehicle id 0)
  src/Sim/core/playgrounds/ring_exp_v1.0/agent_config.json     the obstacle map of agent 0: 1500 x 1500 cells of 0.2 m

Only data is taken (vertex coordinates, one pose, three numbers).  The map itself is rasterised from the polygons by
dftpav_amd.scenarios.default_sim_map, the way DataRenderer::GetObstacleMap does it (semantic_map_manager/src/
data_renderer.cc:146-233: origin = round(ego - extent / 2), polygons filled as OCCUPIED).

    python tests/golden/make_default_map.py
"""
import json, os
import numpy as np

REF = "/root/reference/src/Sim/core/playgrounds/ring_exp_v1.0"
HERE = os.path.dirname(os.path.abspath(__file__))

obs = json.load(open(os.path.join(REF, "obstacles_norm.json")))
xy, off, ids = [], [0], []
for f in obs["features"]:
    if not f["properties"].get("is_valid", 1):
        continue
    for poly in f["geometry"]["coordinates"]:          # MultiPolygon: polygons -> rings; the first ring is the outline
        ring = np.asarray(poly[0], dtype=np.float64)
        xy.append(ring)
        off.append(off[-1] + len(ring))
        ids.append(int(f["properties"]["id"]))
veh = json.load(open(os.path.join(REF, "vehicle_set.json")))["vehicles"]["info"]
ego = next(v for v in veh if v["id"] == 0)["init_state"]
agent = json.load(open(os.path.join(REF, "agent_config.json")))["agent_config"]["info"]
meta = next(a for a in agent if a["id"] == 0)["obstacle_map_meta_info"]
out = os.path.join(HERE, "default_map.npz")
np.savez_compressed(out, poly_xy=np.concatenate(xy), poly_off=np.asarray(off, np.int32), poly_id=np.asarray(ids, np.int32),
                    ego_init=np.array([ego["x"], ego["y"], ego["angle"]]),
                    map_meta=np.array([meta["width"], meta["height"], meta["resolution"]], dtype=np.float64))
print("wrote", out, ":", len(ids), "polygons,", off[-1], "vertices; ego", ego["x"], ego["y"], ego["angle"], "; map", meta)
