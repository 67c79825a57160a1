/* stand-in: cell_map_keyframe.hpp includes boost/format.hpp and never uses it.  TEST INFRASTRUCTURE ONLY. */
#pragma once
