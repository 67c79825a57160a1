"""Builds the link-and-run form of tests/test_adapter_verbatim.py: the VERBATIM excerpts of the reference's mapping node
(laser_mapping.hpp:1266-1297 init_pointcloud_registration, :1405-1445 / :1494-1512 process_new_scan) wrapped in a harness
with a main(), compiled twice from the same text --

    tests/cpp/_verbatim/verbatim_adapter    Point_cloud_registration = loam_livox_hip::Point_cloud_registration (links the C-ABI library)
    tests/cpp/_verbatim/verbatim_reference  Point_cloud_registration = the reference's own class (point_cloud_registration.hpp
                                            against oracle/ref_stubs, the recipe of oracle/Makefile's `ref` target)

The excerpts are pulled out of /root/reference by line range at build time; the translation unit lives in a temporary
directory and only the binaries stay (git-ignored, like oracle/_ref; they travel to the GPU box with the tree).  Test
infrastructure only."""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "cpp", "_verbatim")
EXE_A, EXE_B = os.path.join(OUT, "verbatim_adapter"), os.path.join(OUT, "verbatim_reference")

HEAD = r'''
#include <cstdio>
#include <cstdlib>
#include <map>
#include <memory>
#include <mutex>
#include <vector>
#include <iostream>
#include <Eigen/Eigen>
#include <pcl/point_cloud.h>
#include <pcl/point_types.h>
#include <pcl/kdtree/kdtree_flann.h>
#ifdef LL_USE_ADAPTER
#include "tools/common.h"
#include "tools/tools_logger.hpp"
#include "tools/tools_timer.hpp"
#include "loam_livox_adapter.hpp"
using Point_cloud_registration = loam_livox_hip::Point_cloud_registration;   // INTEGRATION.md section 3
typedef loam_livox_hip::Point_cloud_registration::Opt_summary Summary_t;
typedef loam_livox_hip::VoxelGrid<pcl::PointCloud<PointType>> Voxel_t;
#else
#include "point_cloud_registration.hpp"  // the reference's own class
typedef ceres::Solver::Summary Summary_t;
struct Voxel_t {  // the down-sampled clouds of the excerpt feed the history (outside the excerpt); the pose does not depend on them
    void setLeafSize( float, float, float ) {}
    template <class P> void setInputCloud( const P & ) {}
    template <class C> void filter( C & ) {}
};
#endif
using namespace std;
int g_if_undistore = 0;  // laser_mapping.hpp:80
'''

HARNESS = r'''
class Laser_mapping_harness
{
  public:
    Common_tools::File_logger m_logger_common, m_logger_pcd, m_logger_timer;
    Common_tools::Timer       m_timer;
    int    if_motion_deblur = 0, m_current_frame_index = 100, m_mapping_init_accumulate_frames = 50;
    float  m_last_time_stamp = 0, m_para_max_angular_rate = 20, m_para_max_speed = 0.3f, m_max_final_cost = 1000;
    int    m_para_icp_max_iterations = 20, m_para_cere_max_iterations = 20, m_para_optimization_maximum_residual_block = 100000;
    float  m_minimum_pt_time_stamp = 0, m_maximum_pt_time_stamp = 1;
    double m_minimum_icp_R_diff = 0.01, m_minimum_icp_T_diff = 0.01;
    double m_para_buffer_RT[ 7 ] = { 0, 0, 0, 1, 0, 0, 0 };
    Eigen::Map<Eigen::Quaterniond> m_q_w_curr = Eigen::Map<Eigen::Quaterniond>( m_para_buffer_RT );
    Eigen::Map<Eigen::Vector3d>    m_t_w_curr = Eigen::Map<Eigen::Vector3d>( m_para_buffer_RT + 4 );
    Eigen::Quaterniond m_last_his_add_q;
    Eigen::Vector3d    m_last_his_add_t;
    std::mutex m_mutex_mapping;
    double m_lastest_pc_reg_time = 0;
    Summary_t m_final_opt_summary;
    Voxel_t down_sample_filter_corner, down_sample_filter_surface;
    ADD_SCREEN_PRINTF_OUT_METHOD;
    float refine_blur( float in_blur, const float &min_blur, const float &max_blur ) { return ( in_blur - min_blur ) / ( max_blur - min_blur ); }

// ---- verbatim: laser_mapping.hpp:1266-1297
@INIT@
// ---- end of excerpt

    int process_new_scan_excerpt( pcl::PointCloud<PointType>::Ptr laser_cloud_corner_from_map, pcl::PointCloud<PointType>::Ptr laser_cloud_surf_from_map,
                                  pcl::KdTreeFLANN<PointType> &kdtree_corner_from_map, pcl::KdTreeFLANN<PointType> &kdtree_surf_from_map,
                                  pcl::PointCloud<PointType>::Ptr laserCloudCornerStack, pcl::PointCloud<PointType>::Ptr laserCloudSurfStack,
                                  pcl::PointCloud<PointType> &current_laser_cloud_full )
    {
        Point_cloud_registration pc_reg;        // laser_mapping.hpp:1348
        init_pointcloud_registration( pc_reg ); // :1349
        int    reg_res = 0;
        int    laser_corner_pt_num = laserCloudCornerStack->points.size(), laser_surface_pt_num = laserCloudSurfStack->points.size();
        double point_cloud_current_timestamp = 0;
// ---- verbatim: laser_mapping.hpp:1405-1445
@REG@
// ---- end of excerpt
        m_mutex_mapping.unlock();
        ( void ) r_diff;
        ( void ) t_diff;
        m_final_opt_summary = pc_reg.m_final_opt_summary;
// ---- verbatim: laser_mapping.hpp:1494-1512
@POSE@
// ---- end of excerpt
        return reg_res;
    }
};

typedef pcl::PointCloud<PointType> Cloud;
static std::shared_ptr<Cloud> load_cloud( const char *path )
{
    std::shared_ptr<Cloud> c( new Cloud() );
    FILE *f = fopen( path, "rb" );
    if ( !f ) { fprintf( stderr, "cannot open %s\n", path ); exit( 2 ); }
    float v[ 4 ];
    while ( fread( v, sizeof( float ), 4, f ) == 4 )
    {
        PointType p;
        p.x = v[ 0 ]; p.y = v[ 1 ]; p.z = v[ 2 ]; p.intensity = v[ 3 ];
        c->points.push_back( p );
    }
    fclose( f );
    return c;
}

// argv: map_corner.bin map_surf.bin scan_corner.bin scan_surf.bin pose7.bin out.txt
// Three passes through the excerpt, each from the same initial pose and with FRESH cloud objects, the way the node allocates
// and deep-copies its match buffer for every scan (laser_mapping.hpp:1391-1392, 1396-1401): (1) the clouds as given, (2) new
// objects with equal contents, (3) new objects, one surface map point moved.  Per pass: return value, pose, and (adapter
// build) the number of structures the shared device map has published per kind.
int main( int argc, char **argv )
{
    if ( argc < 7 ) return 2;
    double pose[ 7 ];
    FILE *f = fopen( argv[ 5 ], "rb" );
    if ( !f || fread( pose, sizeof( double ), 7, f ) != 7 ) return 3;
    fclose( f );
    FILE *out = fopen( argv[ 6 ], "w" );
    Laser_mapping_harness node;
    node.m_if_verbose_screen_printf = 1;  // screen_out silent
    for ( int pass = 0; pass < 3; pass++ )
    {
        std::shared_ptr<Cloud> map_corner = load_cloud( argv[ 1 ] ), map_surf = load_cloud( argv[ 2 ] );
        std::shared_ptr<Cloud> scan_corner = load_cloud( argv[ 3 ] ), scan_surf = load_cloud( argv[ 4 ] );
        if ( pass == 2 ) map_surf->points[ map_surf->points.size() / 3 + 7 ].z += 0.25f;
        pcl::KdTreeFLANN<PointType> kd_corner, kd_surf;
#ifndef LL_USE_ADAPTER
        kd_corner.setInputCloud( map_corner );  // laser_mapping.hpp:1396-1401 (the adapter ignores the trees)
        kd_surf.setInputCloud( map_surf );
#endif
        for ( int i = 0; i < 7; i++ ) node.m_para_buffer_RT[ i ] = pose[ i ];
        node.m_lastest_pc_reg_time = -1;
        Cloud full = *scan_surf;
        const int res = node.process_new_scan_excerpt( map_corner, map_surf, kd_corner, kd_surf, scan_corner, scan_surf, full );
        long long gen_c = -1, gen_s = -1;
#ifdef LL_USE_ADAPTER
        gen_c = ( long long ) ll_map_generation( loam_livox_hip::Handle_pool::instance().shared_map( 0 ), LL_MAP_CORNER );
        gen_s = ( long long ) ll_map_generation( loam_livox_hip::Handle_pool::instance().shared_map( 0 ), LL_MAP_SURF );
#endif
        fprintf( out, "%d %lld %lld", res, gen_c, gen_s );
        for ( int i = 0; i < 7; i++ ) fprintf( out, " %.17g", node.m_para_buffer_RT[ i ] );
        fprintf( out, " %.9g %.9g %.9g\n", ( double ) full.points[ 0 ].x, ( double ) full.points[ 0 ].y, ( double ) full.points[ 0 ].z );
    }
    fclose( out );
    return 0;
}
'''


def _lines(path, a, b):
    with open(os.path.join(REF, path)) as f:
        src = f.read().split("\n")
    return "\n".join(src[a - 1:b])


def have_reference():
    return os.path.exists(os.path.join(REF, "source", "laser_mapping.hpp"))


def build(force=False):
    """-> (adapter exe, reference exe), built where /root/reference exists; elsewhere whatever travelled with the tree (or None)"""
    if not have_reference():
        return (EXE_A if os.path.exists(EXE_A) else None, EXE_B if os.path.exists(EXE_B) else None)
    from loam_livox_amd import build as libbuild
    lib = libbuild.build()
    deps = [os.path.abspath(__file__), os.path.join(ROOT, "include", "loam_livox_adapter.hpp"), os.path.join(ROOT, "include", "loam_livox_hip.h"), lib]
    deps += [os.path.join(dp, f) for dp, _, fs in os.walk(os.path.join(ROOT, "oracle", "ref_stubs")) for f in fs]
    if not force and all(os.path.exists(e) and all(os.path.getmtime(d) <= os.path.getmtime(e) for d in deps) for e in (EXE_A, EXE_B)):
        return EXE_A, EXE_B
    os.makedirs(OUT, exist_ok=True)
    tu = HEAD + (HARNESS.replace("@INIT@", _lines("source/laser_mapping.hpp", 1266, 1297))
                        .replace("@REG@", _lines("source/laser_mapping.hpp", 1405, 1445))
                        .replace("@POSE@", _lines("source/laser_mapping.hpp", 1494, 1512)))
    stubs = os.path.join(ROOT, "oracle", "ref_stubs")
    with tempfile.TemporaryDirectory() as td:
        src = os.path.join(td, "verbatim_run.cpp")
        with open(src, "w") as f:
            f.write(tu)
        inc = ["-I", os.path.join(stubs, "override"), "-I-", "-I", stubs, "-I", os.path.join(REF, "source"), "-I", os.path.join(REF, "include"),
               "-I", os.path.join(REF, "include", "tools"), "-I", os.path.join(ROOT, "include")]
        common = ["g++", "-std=c++14", "-O2", "-ffp-contract=off", "-fno-fast-math", "-w"]
        subprocess.check_call(common + ["-DLL_USE_ADAPTER"] + inc + [src, "-o", EXE_A, lib, "-Wl,-rpath," + os.path.dirname(lib), "-Wl,-rpath,/opt/rocm/lib", "-lpthread"])
        subprocess.check_call(common + inc + [src, "-o", EXE_B, "-lpthread"])
    return EXE_A, EXE_B


if __name__ == "__main__":
    print(build(force=True))
