/* Stand-in for source/cell_map_keyframe.hpp when compiling point_cloud_registration.hpp for oracle/_ref
 * (g++ -I <this dir> -I- ...).  The registrar includes that header (PCR:26) but uses nothing from it except what it
 * pulls in through common_tools.h (tools_random.hpp; tools_json.hpp is left out) (Common_tools::Random_generator_float, PCR:104); the real file needs OpenCV,
 * pcl::octree and boost::format, none of which is on the hot path.  TEST INFRASTRUCTURE ONLY. */
#pragma once
#include "tools_logger.hpp"
#include "tools_random.hpp"
#include "tools_timer.hpp"
#include <set>
