// Stand-in for ROS message headers: plain structs with the fields traj_optimizer.cpp:1785-1914 fills before publishing
// to a publisher that swallows them.  TEST INFRASTRUCTURE for oracle/_ref.
#pragma once
#include <ros/ros.h>
#include <string>
#include <vector>
namespace std_msgs {
struct Header { std::string frame_id; ros::Time stamp; };
struct ColorRGBA { float r = 0, g = 0, b = 0, a = 0; };
}  // namespace std_msgs
namespace geometry_msgs {
struct Point { double x = 0, y = 0, z = 0; };
struct Vector3 { double x = 0, y = 0, z = 0; };
struct Quaternion { double x = 0, y = 0, z = 0, w = 0; };
struct Pose { Point position; Quaternion orientation; };
}  // namespace geometry_msgs
namespace visualization_msgs {
struct Marker {
  enum { ARROW = 0, CUBE = 1, SPHERE = 2, CYLINDER = 3, LINE_STRIP = 4, LINE_LIST = 5, CUBE_LIST = 6, SPHERE_LIST = 7, POINTS = 8 };
  enum { ADD = 0, MODIFY = 0, DELETE = 2, DELETEALL = 3 };
  std_msgs::Header header;
  std::string ns;
  int id = 0, type = 0, action = 0;
  geometry_msgs::Pose pose;
  geometry_msgs::Vector3 scale;
  std_msgs::ColorRGBA color;
  std::vector<geometry_msgs::Point> points;
  std::vector<std_msgs::ColorRGBA> colors;
};
}  // namespace visualization_msgs
