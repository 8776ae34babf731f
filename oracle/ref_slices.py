#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (oracle/_ref): cuts the reference's OWN code of the SURVEY.md §8(f) steps -- the functions either side of
the solve path -- out of /root/reference by file and line range, verbatim, into oracle/_ref/slices/*.inc, so that
oracle/ref_next_driver.cpp can compile them against interface stand-ins (the files they live in need ROS, OMPL, PCL, OpenCV and the
simulator's lane / vehicle library, none of which exists here, so they cannot be compiled whole as traj_optimizer.cpp is).

Nothing is edited: a slice is the byte range of whole lines first..last of its file.  Every slice carries the text its first and
last line must begin with; a reference that moved fails the build instead of silently cutting something else.  The generated files
are build products (oracle/_ref/ is git-ignored): no reference source enters the repository.

    python3 oracle/ref_slices.py [reference-root] [out-dir]      (called by oracle/Makefile.ref)
"""
import hashlib
import os
import sys

SRC = "src"
TP = "src/Plan/traj_planner"
CM = "src/Sim/core/common"
SMM = "src/Sim/core/semantic_map_manager"

# (name, file, first line, last line, how the first line must begin, how the last line must begin)
SLICES = [
    # ---- the occupancy grid and the vehicle outline (simulator library) ------------------------------------------------------
    ("gridmap_class", CM + "/inc/common/basics/semantics.h", 350, 604, "template <typename T, int N_DIM>", "};"),
    ("gridmap_members", CM + "/src/common/basics/semantics.cc", 130, 322, "template <typename T, int N_DIM>", "}"),
    ("obb_struct", CM + "/inc/common/basics/shapes.h", 84, 108, "struct OrientedBoundingBox2D {", "};"),
    ("obb_ctors", CM + "/src/common/basics/shapes.cc", 29, 36, "OrientedBoundingBox2D::OrientedBoundingBox2D() {}", ": x(x_), y(y_), angle(angle_), width(width_), length(length_) {}"),
    ("dense_vertices_decl", CM + "/inc/common/basics/shapes.h", 200, 201, "static ErrorType GetDenseVerticesOfOrientedBoundingBox(", "const OrientedBoundingBox2D& obb, vec_E<Vecf<2>>* vertices,double res = 0.1);"),
    ("dense_vertices", CM + "/src/common/basics/shapes.cc", 110, 149, "ErrorType ShapeUtils::GetDenseVerticesOfOrientedBoundingBox(", "}"),
    ("normalize_angle", CM + "/src/common/math/calculations.cc", 18, 23, "decimal_t normalize_angle(const decimal_t& theta) {", "}"),
    ("smm_pos_and_yaw", SMM + "/src/semantic_map_manager.cc", 639, 662, "ErrorType SemanticMapManager::CheckCollisionUsingPosAndYaw(", "}"),
    ("smm_global_position", SMM + "/src/semantic_map_manager.cc", 710, 715, "ErrorType SemanticMapManager::CheckCollisionUsingGlobalPosition(", "}"),
    # ---- the planner's map adapter --------------------------------------------------------------------------------------------
    ("adapter_pos_and_yaw", TP + "/src/map_adapter.cpp", 110, 115, "ErrorType TrajPlannerAdapter::CheckIfCollisionUsingPosAndYaw(", "}"),
    ("adapter_line", TP + "/src/map_adapter.cpp", 117, 129, "ErrorType TrajPlannerAdapter::CheckIfCollisionUsingLine(const Eigen::Vector2d p1,", "}"),
    ("adapter_obstacle_map", TP + "/src/map_adapter.cpp", 93, 97, "ErrorType TrajPlannerAdapter::GetObstacleMap(GridMap2D* grid_map) {", "}"),
    # ---- (f)-1 corridor rectangles --------------------------------------------------------------------------------------------
    ("rectangle", TP + "/src/traj_manager.cpp", 1213, 1469, "ErrorType TrajPlanner::getRectangleConst(std::vector<Eigen::Vector3d> statelist){", "}"),
    # ---- (f)-4 moving-obstacle fit --------------------------------------------------------------------------------------------
    ("state_to_flat_output", TP + "/src/traj_manager.cpp", 139, 158, "Eigen::MatrixXd TrajPlanner::state_to_flat_output(const State& state) {", "}"),
    ("fit_surround", TP + "/src/traj_manager.cpp", 743, 789, "ErrorType TrajPlanner::ConverSurroundTrajFromPoints(", "}"),
    # ---- (f)-3 front-end resampling: RunMINCOParking's loop over the gear segments, up to the corridor call (statement range
    #      inside the function: the driver supplies the declarations of :514-529 and closes the loop) ---------------------------
    ("resample_locals", TP + "/src/traj_manager.cpp", 514, 516, "Eigen::MatrixXd flat_finalState(2, 3),  flat_headState(2,3);", "Eigen::MatrixXd ego_innerPs;"),
    ("resample_containers", TP + "/src/traj_manager.cpp", 521, 529, "double basetime = 0.0;", "duration_container.resize(kino_trajs_.size());"),
    ("resample_loop", TP + "/src/traj_manager.cpp", 531, 568, "for(unsigned int i = 0; i < kino_trajs_.size(); i++){", "}"),
    ("resample_loop_tail", TP + "/src/traj_manager.cpp", 573, 577, "display_hPolys_.insert(display_hPolys_.end(),hPolys_.begin(),hPolys_.end());", "basetime += initTotalduration;"),
    # ---- (f)-3 KinoAstar: members, evaluatePos, getKinoNode from SampleTraj on, the trapezoid profile, flat states -----------
    ("kino_members_states", TP + "/include/path_searching/kino_astar.h", 140, 141, "Eigen::Vector4d start_state_, end_state_;", "Eigen::Vector2d start_ctrl;"),
    ("kino_members_limits", TP + "/include/path_searching/kino_astar.h", 150, 153, "double max_forward_vel = 4.0;", "double max_backward_acc = 1.0;"),
    ("kino_members_shot", TP + "/include/path_searching/kino_astar.h", 177, 181, "std::vector<double>  shot_timeList;", "std::vector<Eigen::Vector3d> SampleTraj;"),
    ("kino_members_flat", TP + "/include/path_searching/kino_astar.h", 191, 192, "void getFlatState(Eigen::Vector4d state, Eigen::Vector2d control_input,", "Eigen::MatrixXd &flat_state, int singul);"),
    ("kino_members_profile", TP + "/include/path_searching/kino_astar.h", 201, 202, "double evaluateLength(double curt,double locallength,double localtime, double max_vel, double max_acc, double startV = 0.0, double endV = 0.0);", "double evaluateDuration(double length, double max_vel, double max_acc, double startV = 0.0, double endV = 0.0);"),
    ("kino_members_vehicle", TP + "/include/path_searching/kino_astar.h", 206, 207, "common::VehicleParam vp_;", "double non_siguav=0.2;"),
    ("kino_members_total", TP + "/include/path_searching/kino_astar.h", 253, 254, "double totalTrajTime;", "double checkl = 0.2;"),
    ("kino_evaluate_pos", TP + "/src/kino_astar.cpp", 468, 521, "Eigen::Vector3d KinoAstar::evaluatePos(double t){", "}"),
    ("kino_node_locals_a", TP + "/src/kino_astar.cpp", 559, 559, "flat_trajs.clear();", "flat_trajs.clear();"),
    ("kino_node_locals_b", TP + "/src/kino_astar.cpp", 561, 562, "double startvel = fabs(start_state_[3]);", "double endvel = fabs(end_state_[3]);"),
    ("kino_node_locals_c", TP + "/src/kino_astar.cpp", 564, 565, "std::vector<Eigen::Vector3d> traj_pts;  // 3, N", "std::vector<double> thetas;"),
    ("kino_node_body", TP + "/src/kino_astar.cpp", 613, 743, "/*divide the whole shot traj into different segments*/", "}"),
    ("kino_profile", TP + "/src/kino_astar.cpp", 744, 795, "double KinoAstar::evaluateDuration(double length, double max_vel, double max_acc, double startV, double endV){", "}"),
    ("kino_flat_state", TP + "/src/kino_astar.cpp", 834, 857, "void KinoAstar::getFlatState(Eigen::Vector4d state, Eigen::Vector2d control_input,", "}"),
    # ---- (f)-2 the server: playback of the result and the collision re-check --------------------------------------------------
    ("server_filter", TP + "/src/traj_server_ros.cpp", 335, 356, "ErrorType TrajPlannerServer::FilterSingularityState(", "}"),
    ("server_playback", TP + "/src/traj_server_ros.cpp", 248, 259, "if (executing_traj_->at(exe_traj_index_).end_time <= t){", "if (ctrl_state_hist_.size() > 100) ctrl_state_hist_.erase(ctrl_state_hist_.begin());"),
    ("server_recheck", TP + "/src/traj_server_ros.cpp", 385, 397, "for(int i = 0; i < executing_traj_->size(); i++){", "}"),
]


def main():
    root = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    out = sys.argv[2] if len(sys.argv) > 2 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref", "slices")
    os.makedirs(out, exist_ok=True)
    files = {}
    manifest = []
    for name, rel, first, last, want_first, want_last in SLICES:
        path = os.path.join(root, rel)
        if path not in files:
            with open(path, "rb") as f:
                files[path] = f.read().decode("utf-8", errors="surrogateescape").split("\n")
        lines = files[path]
        got_first, got_last = lines[first - 1].strip(), lines[last - 1].strip()
        if not got_first.startswith(want_first.strip()) or not got_last.startswith(want_last.strip()):
            sys.exit("ref_slices: %s:%d-%d is not where slice '%s' expects it:\n  first: %r\n  last:  %r" % (rel, first, last, name, got_first, got_last))
        body = "\n".join(lines[first - 1:last]) + "\n"
        # a #line directive so that diagnostics and debuggers point at the reference's file
        text = '#line %d "%s"\n%s' % (first, path, body)
        with open(os.path.join(out, name + ".inc"), "w", encoding="utf-8", errors="surrogateescape") as f:
            f.write(text)
        manifest.append("%s  %s:%d-%d  %s" % (hashlib.sha256(body.encode("utf-8", errors="surrogateescape")).hexdigest(), rel, first, last, name))
    with open(os.path.join(out, "SLICES.sha256"), "w") as f:
        f.write("\n".join(manifest) + "\n")
    print("ref_slices: %d slices of %d reference files -> %s" % (len(SLICES), len(files), out))


if __name__ == "__main__":
    main()
