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
#include <pcl/filters/voxel_grid.h>
typedef ceres::Solver::Summary Summary_t;
#ifdef LL_USE_SEQ_REF
typedef pcl::VoxelGrid<PointType> Voxel_t;  // the stand-in of oracle/ref_stubs (PCL 1.9 semantics, oracle/ll_oracle_voxel.c's definition)
#else
struct Voxel_t {  // the down-sampled clouds of the excerpt feed the history (outside the excerpt); the pose does not depend on them
    void setLeafSize( float, float, float ) {}
    template <class P> void setInputCloud( const P & ) {}
    template <class C> void filter( C & ) {}
};
#endif
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




# ---------------------------------------------------------------------------------------------------------------------------------
# The mapping loop: process_new_scan from its first line to the pose read-back, the history add rule and the history branch of
# update_buff_for_matching, executed frame after frame -- again ONE translation unit, built with the adapter classes and with the
# reference's own Point_cloud_registration + a stand-in pcl::VoxelGrid.  Verbatim line ranges of source/laser_mapping.hpp:
#   1243-1258  find_min_max_intensity, refine_blur            1266-1297  init_pointcloud_registration
#   1299-1314  if_matchbuff_and_pc_sync                        1330-1350, 1352-1404  process_new_scan up to the registration
#   1405-1445  registration, "Add new frame"                   1446-1463, 1468-1490  history add rule and trimming
#   1496-1512  pose read-back                                  467-470, 518-531, 533-565  update_buff_for_matching, history branch
# Left out (declared by the harness instead): the by-value copies of the two filters and the k-d tree locals (:1325-1328, :465-468 --
# they name pcl::VoxelGrid, which INTEGRATION.md replaces by loam_livox_hip::VoxelGrid), ros::Time::now() (:1351), the two
# screen_out lines about the cell maps (:1465-1466) and the cell-map appends / service thread (:1491-1493, :1496ff).
EXE_SA, EXE_SB = os.path.join(OUT, "sequence_adapter"), os.path.join(OUT, "sequence_reference")

SEQ_HARNESS = r'''
#include <chrono>
#include <list>
#include <thread>
#define PCD_SAVE_RAW 1  // laser_mapping.hpp:76
double history_add_t_step = 0.00;      // :83
double history_add_angle_step = 0.00;  // :84
struct Pcl_tools_stub { template <class C> void save_to_pcd_files( const char *, C &, int ) {} };
typedef pcl::KdTreeFLANN<PointType> Kd_t;

class Laser_mapping_harness
{
  public:
    // members of Laser_mapping the excerpts touch, with the reference's types and defaults (laser_mapping.hpp:98-280)
    Common_tools::File_logger m_logger_common, m_logger_pcd, m_logger_timer, m_logger_matching_buff;
    Common_tools::Timer       m_timer;
    int    if_motion_deblur = 0, m_current_frame_index = 0, m_mapping_init_accumulate_frames = 50;
    double m_last_time_stamp = 0, m_minimum_pt_time_stamp = 0, m_maximum_pt_time_stamp = 1.0, m_time_odom = 0;
    int    m_if_input_downsample_mode = 1, m_maximum_history_size = 100, m_if_save_to_pcd_files = 0;
    float  m_para_max_angular_rate = 200.0 / 50.0, m_para_max_speed = 100.0 / 50.0, m_max_final_cost = 100.0;
    int    m_para_icp_max_iterations = 20, m_para_cere_max_iterations = 100, m_para_optimization_maximum_residual_block = 1e5;
    double m_minimum_icp_R_diff = 0.01, m_minimum_icp_T_diff = 0.01;
    float  m_line_resolution = 0.4, m_plane_resolution = 0.8;
    double m_lastest_pc_reg_time = -3e8, m_lastest_pc_matching_refresh_time = -1, m_lastest_pc_income_time = -3e8;
    double m_maximum_pointcloud_delay_time = 0.1;
    std::list<pcl::PointCloud<PointType>> m_laser_cloud_corner_history, m_laser_cloud_surface_history, m_laser_cloud_full_history;
    std::list<double>  m_his_reg_error;
    Eigen::Quaterniond m_last_his_add_q;
    Eigen::Vector3d    m_last_his_add_t;
    int m_if_mapping_updated_corner = true, m_if_mapping_updated_surface = true;
    pcl::PointCloud<PointType>::Ptr m_laser_cloud_corner_from_map_last, m_laser_cloud_surf_from_map_last;
    pcl::PointCloud<PointType>::Ptr m_laser_cloud_full_res, m_laser_cloud_corner_last, m_laser_cloud_surf_last;
    Kd_t m_kdtree_corner_from_map_last, m_kdtree_surf_from_map_last;
    double m_para_buffer_RT[ 7 ] = { 0, 0, 0, 1, 0, 0, 0 };
    double m_para_buffer_RT_last[ 7 ] = { 0, 0, 0, 1, 0, 0, 0 };
    Eigen::Map<Eigen::Quaterniond> m_q_w_curr = Eigen::Map<Eigen::Quaterniond>( m_para_buffer_RT );
    Eigen::Map<Eigen::Vector3d>    m_t_w_curr = Eigen::Map<Eigen::Vector3d>( m_para_buffer_RT + 4 );
    Eigen::Map<Eigen::Quaterniond> m_q_w_last = Eigen::Map<Eigen::Quaterniond>( m_para_buffer_RT_last );
    Eigen::Map<Eigen::Vector3d>    m_t_w_last = Eigen::Map<Eigen::Vector3d>( m_para_buffer_RT_last + 4 );
    std::mutex m_mutex_mapping, m_mutex_querypointcloud, m_mutex_buff_for_matching_corner, m_mutex_buff_for_matching_surface,
        m_mutex_dump_full_history;
    Summary_t      m_final_opt_summary;
    Voxel_t        m_down_sample_filter_corner, m_down_sample_filter_surface;
    Pcl_tools_stub m_pcl_tools_raw;
    int            m_matching_mode = 0;
    ADD_SCREEN_PRINTF_OUT_METHOD;

    Laser_mapping_harness()
    {
        m_laser_cloud_corner_from_map_last.reset( new pcl::PointCloud<PointType>() );
        m_laser_cloud_surf_from_map_last.reset( new pcl::PointCloud<PointType>() );
        m_laser_cloud_full_res.reset( new pcl::PointCloud<PointType>() );
        m_laser_cloud_corner_last.reset( new pcl::PointCloud<PointType>() );
        m_laser_cloud_surf_last.reset( new pcl::PointCloud<PointType>() );
        m_last_his_add_q.setIdentity();
        m_last_his_add_t.setZero();
    }

// ---- verbatim: laser_mapping.hpp:1243-1258
@HELPERS@
// ---- verbatim: laser_mapping.hpp:1266-1297
@INIT@
// ---- verbatim: laser_mapping.hpp:1299-1314
@SYNC@
// ---- end of excerpts

    void update_buff_for_matching_history_branch()
    {
        Voxel_t down_sample_filter_corner = m_down_sample_filter_corner;    // :465-466
        Voxel_t down_sample_filter_surface = m_down_sample_filter_surface;
// ---- verbatim: laser_mapping.hpp:467-470
@UPD_HEAD@
// ---- end of excerpt
// ---- verbatim: laser_mapping.hpp:518-531 (the body of the else branch: m_matching_mode == 0)
@UPD_HIST@
// ---- end of excerpt
// ---- verbatim: laser_mapping.hpp:533-565
@UPD_TAIL@
// ---- end of excerpt
    }

    int process_new_scan()
    {
        m_timer.tic( "Frame process" );          // :1318-1323
        m_timer.tic( "Query points for match" );
        pcl::PointCloud<PointType> current_laser_cloud_full, current_laser_cloud_corner_last, current_laser_cloud_surf_last;
        Voxel_t down_sample_filter_corner = m_down_sample_filter_corner;    // :1325-1328
        Voxel_t down_sample_filter_surface = m_down_sample_filter_surface;
        Kd_t    kdtree_corner_from_map;
        Kd_t    kdtree_surf_from_map;
// ---- verbatim: laser_mapping.hpp:1330-1350
@PNS_A@
// ---- verbatim: laser_mapping.hpp:1352-1404
@PNS_B@
// ---- verbatim: laser_mapping.hpp:1405-1445
@REG@
// ---- verbatim: laser_mapping.hpp:1446-1463
@HIS_A@
// ---- verbatim: laser_mapping.hpp:1468-1490
@HIS_B@
// ---- end of excerpts
        m_mutex_mapping.unlock();  // :1495
        m_final_opt_summary = pc_reg.m_final_opt_summary;
// ---- verbatim: laser_mapping.hpp:1494-1512 is the registration's pose read-back; here :1496-1512
@POSE@
// ---- end of excerpt
        return 1;
    }
};

typedef pcl::PointCloud<PointType> Cloud;
static std::shared_ptr<Cloud> load_cloud( FILE *f, int n )
{
    std::shared_ptr<Cloud> c( new Cloud() );
    for ( int i = 0; i < n; i++ )
    {
        float v[ 4 ];
        if ( fread( v, sizeof( float ), 4, f ) != 4 ) { fprintf( stderr, "short read\n" ); exit( 3 ); }
        PointType p;
        p.x = v[ 0 ]; p.y = v[ 1 ]; p.z = v[ 2 ]; p.intensity = v[ 3 ];
        c->points.push_back( p );
    }
    return c;
}

// argv: frames.bin out.txt line_res plane_res init_accumulate_frames maximum_history_size icp_max_iterations max_final_cost
// frames.bin: int32 n_frames, then per frame int32 n_corner, n_surf, n_full and the three xyzi clouds (what /pc2_corners,
// /pc2_surface, /pc2_full carry into the mapping node).  Per frame: process_new_scan, then the match-buffer refresh, like the
// node's service thread would (synchronously: its timing is not reproducible).  Output per frame: return value, pose, sizes of
// the history and of the match buffer, checksum-like sums of the match-buffer clouds.
int main( int argc, char **argv )
{
    if ( argc < 9 ) return 2;
    FILE *f = fopen( argv[ 1 ], "rb" );
    if ( !f ) return 3;
    int n_frames = 0;
    if ( fread( &n_frames, sizeof( int ), 1, f ) != 1 ) return 3;
    FILE *out = fopen( argv[ 2 ], "w" );
    Laser_mapping_harness node;
    node.m_if_verbose_screen_printf = 1;
    node.m_line_resolution = ( float ) atof( argv[ 3 ] );
    node.m_plane_resolution = ( float ) atof( argv[ 4 ] );
    node.m_mapping_init_accumulate_frames = atoi( argv[ 5 ] );
    node.m_maximum_history_size = atoi( argv[ 6 ] );
    node.m_para_icp_max_iterations = atoi( argv[ 7 ] );
    node.m_max_final_cost = ( float ) atof( argv[ 8 ] );
    node.m_para_max_angular_rate = 20.0f;   // the launch files' max_allow_incre_R / T
    node.m_para_max_speed = 0.3f;
    node.m_para_cere_max_iterations = 20;
    node.m_down_sample_filter_corner.setLeafSize( node.m_line_resolution, node.m_line_resolution, node.m_line_resolution );       // :742-743
    node.m_down_sample_filter_surface.setLeafSize( node.m_plane_resolution, node.m_plane_resolution, node.m_plane_resolution );
    for ( int k = 0; k < n_frames; k++ )
    {
        int n[ 3 ];
        if ( fread( n, sizeof( int ), 3, f ) != 3 ) return 3;
        node.m_laser_cloud_corner_last = load_cloud( f, n[ 0 ] );
        node.m_laser_cloud_surf_last = load_cloud( f, n[ 1 ] );
        node.m_laser_cloud_full_res = load_cloud( f, n[ 2 ] );
        const int res = node.process_new_scan();
        node.update_buff_for_matching_history_branch();
        double sc = 0, ss = 0;
        for ( auto &p : node.m_laser_cloud_corner_from_map_last->points ) sc += ( double ) p.x + 2.0 * p.y + 3.0 * p.z;
        for ( auto &p : node.m_laser_cloud_surf_from_map_last->points ) ss += ( double ) p.x + 2.0 * p.y + 3.0 * p.z;
        fprintf( out, "%d %d %d %d %d", res, ( int ) node.m_laser_cloud_corner_history.size(), ( int ) node.m_laser_cloud_surface_history.size(),
                 ( int ) node.m_laser_cloud_corner_from_map_last->points.size(), ( int ) node.m_laser_cloud_surf_from_map_last->points.size() );
        for ( int i = 0; i < 7; i++ ) fprintf( out, " %.17g", node.m_para_buffer_RT[ i ] );
        fprintf( out, " %.17g %.17g\n", sc, ss );
    }
    fclose( out );
    return 0;
}
'''


def build_sequence(force=False):
    """-> (adapter exe, reference exe) of the mapping-loop harness; like build()"""
    if not have_reference():
        return (EXE_SA if os.path.exists(EXE_SA) else None, EXE_SB if os.path.exists(EXE_SB) else None)
    from loam_livox_amd import build as libbuild
    lib = libbuild.build()
    deps = [os.path.abspath(__file__), os.path.join(ROOT, "include", "loam_livox_adapter.hpp"), os.path.join(ROOT, "include", "loam_livox_hip.h"), lib]
    deps += [os.path.join(dp, f) for dp, _, fs in os.walk(os.path.join(ROOT, "oracle", "ref_stubs")) for f in fs]
    if not force and all(os.path.exists(e) and all(os.path.getmtime(d) <= os.path.getmtime(e) for d in deps) for e in (EXE_SA, EXE_SB)):
        return EXE_SA, EXE_SB
    os.makedirs(OUT, exist_ok=True)
    L = lambda a, b: _lines("source/laser_mapping.hpp", a, b)
    tu = HEAD + (SEQ_HARNESS.replace("@HELPERS@", L(1243, 1258)).replace("@INIT@", L(1266, 1297)).replace("@SYNC@", L(1299, 1314))
                 .replace("@UPD_HEAD@", L(467, 470)).replace("@UPD_HIST@", L(518, 531)).replace("@UPD_TAIL@", L(533, 565))
                 .replace("@PNS_A@", L(1330, 1350)).replace("@PNS_B@", L(1352, 1404)).replace("@REG@", L(1405, 1445))
                 .replace("@HIS_A@", L(1446, 1463)).replace("@HIS_B@", L(1468, 1490)).replace("@POSE@", L(1496, 1512)))
    stubs = os.path.join(ROOT, "oracle", "ref_stubs")
    with tempfile.TemporaryDirectory() as td:
        src = os.path.join(td, "verbatim_sequence.cpp")
        with open(src, "w") as f:
            f.write(tu)
        inc = ["-I", os.path.join(stubs, "override"), "-I-", "-I", stubs, "-I", os.path.join(REF, "source"), "-I", os.path.join(REF, "include"),
               "-I", os.path.join(REF, "include", "tools"), "-I", os.path.join(ROOT, "include")]
        common = ["g++", "-std=c++14", "-O2", "-ffp-contract=off", "-fno-fast-math", "-w"]
        subprocess.check_call(common + ["-DLL_USE_ADAPTER"] + inc + [src, "-o", EXE_SA, lib, "-Wl,-rpath," + os.path.dirname(lib), "-Wl,-rpath,/opt/rocm/lib", "-lpthread"])
        subprocess.check_call(common + ["-DLL_USE_SEQ_REF"] + inc + [src, "-o", EXE_SB, "-lpthread"])
    return EXE_SA, EXE_SB


if __name__ == "__main__":
    print(build(force=True))
    print(build_sequence(force=True))


# ---------------------------------------------------------------------------------------------------------------------------------
# The feature node: Laser_feature::laserCloudHandler from the start-up delay to the end of its Livox branch
# (source/laser_feature_extractor.hpp:256-392, one verbatim range), message after message -- built with loam_livox_hip::Livox_laser +
# loam_livox_hip::VoxelGrid and with the reference's own Livox_laser + the stand-in pcl::VoxelGrid.  The ROS surface the lines touch
# (PointCloud2 with header.stamp, pcl::fromROSMsg / toROSMsg, Publisher::publish, ros::Time::now) is a few structs in the harness.
EXE_FA, EXE_FB = os.path.join(OUT, "feature_adapter"), os.path.join(OUT, "feature_reference")

FEAT_HEAD = r'''
#include <cstdio>
#include <cstdlib>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>
#include <iostream>
#include <Eigen/Eigen>
#include <pcl/point_cloud.h>
#include <pcl/point_types.h>
#include "tools/common.h"
#include "tools/tools_logger.hpp"
#include "tools/tools_timer.hpp"
#ifdef LL_USE_ADAPTER
#include "loam_livox_adapter.hpp"
using Livox_laser = loam_livox_hip::Livox_laser;                             // INTEGRATION.md section 2
typedef loam_livox_hip::VoxelGrid<pcl::PointCloud<PointType>> Voxel_t;
#else
#include "livox_feature_extractor.hpp"   // the reference's own class
#include <pcl/filters/voxel_grid.h>
typedef pcl::VoxelGrid<PointType> Voxel_t;
#endif
using namespace std;
namespace ros
{
struct Time
{
    double      t = 0;
    static Time now() { return Time(); }
    double      toSec() const { return t; }
};
} // namespace ros
namespace sensor_msgs
{
struct PointCloud2
{
    struct Header
    {
        ros::Time   stamp;
        std::string frame_id;
    } header;
    pcl::PointCloud<PointType> cloud;
};
typedef std::shared_ptr<const PointCloud2> PointCloud2ConstPtr;
} // namespace sensor_msgs
namespace pcl
{
template <class C> void fromROSMsg( const sensor_msgs::PointCloud2 &m, C &c ) { c = m.cloud; }
template <class C> void toROSMsg( const C &c, sensor_msgs::PointCloud2 &m ) { m.cloud = c; }
} // namespace pcl
struct Capture_pub
{
    std::vector<pcl::PointCloud<PointType>> *sink = nullptr;
    void publish( const sensor_msgs::PointCloud2 &m ) { sink->push_back( m.cloud ); }
};
'''

FEAT_HARNESS = r'''
class Laser_feature_harness
{
  public:
    Livox_laser m_livox;                        // laser_feature_extractor.hpp:92
    int         m_laser_scan_number = 64;       // :90
    int         m_para_system_init_count = 0, m_para_system_delay = 20;
    bool        m_para_systemInited = false;
    int         m_if_pub_debug_feature = 1, m_piecewise_number = 3, m_if_motion_deblur = 0, m_lidar_type = 1, m_odom_mode = 0;
    int         m_maximum_input_lidar_pointcloud = 3;
    float       m_plane_resolution = 0.8f, m_line_resolution = 0.8f;
    double      m_minimum_range = 0.1;
    std::vector<std::vector<pcl::PointCloud<pcl::PointXYZI>>> m_map_pointcloud_corner_vec_vec, m_map_pointcloud_surface_vec_vec,
        m_map_pointcloud_full_vec_vec;          // :113-115
    Voxel_t                   m_voxel_filter_for_surface, m_voxel_filter_for_corner;
    sensor_msgs::PointCloud2  temp_out_msg;
    Capture_pub               m_pub_pc_livox_corners, m_pub_pc_livox_surface, m_pub_pc_livox_full;
    Common_tools::File_logger m_file_logger;
    std::vector<pcl::PointCloud<PointType>> out_full, out_surface, out_corners;

    void init()
    {
        m_map_pointcloud_full_vec_vec.resize( m_maximum_input_lidar_pointcloud );      // :163-172
        m_map_pointcloud_surface_vec_vec.resize( m_maximum_input_lidar_pointcloud );
        m_map_pointcloud_corner_vec_vec.resize( m_maximum_input_lidar_pointcloud );
        for ( int i = 0; i < m_maximum_input_lidar_pointcloud; i++ )
        {
            m_map_pointcloud_full_vec_vec[ i ].resize( m_piecewise_number );
            m_map_pointcloud_surface_vec_vec[ i ].resize( m_piecewise_number );
            m_map_pointcloud_corner_vec_vec[ i ].resize( m_piecewise_number );
        }
        m_voxel_filter_for_surface.setLeafSize( m_plane_resolution / 2, m_plane_resolution / 2, m_plane_resolution / 2 );   // :192-193
        m_voxel_filter_for_corner.setLeafSize( m_line_resolution, m_line_resolution, m_line_resolution );
        m_pub_pc_livox_full.sink = &out_full;
        m_pub_pc_livox_surface.sink = &out_surface;
        m_pub_pc_livox_corners.sink = &out_corners;
    }

    void laserCloudHandler_excerpt( const sensor_msgs::PointCloud2ConstPtr &laserCloudMsg, int current_lidar_index )
    {
// ---- verbatim: laser_feature_extractor.hpp:256-392
@HANDLER@
// ---- end of excerpt
    }
};

// argv: msgs.bin out.bin piecewise_number odom_mode maximum_input_lidar_pointcloud para_system_delay plane_res line_res if_motion_deblur
// msgs.bin: int32 n_msgs, then per message int32 n_points, int32 lidar, float64 stamp, the xyzi cloud.
// out.bin: per message int32 n_published, then per publication three clouds (full, surface, corners) as int32 n + n x xyzi.
int main( int argc, char **argv )
{
    if ( argc < 10 ) return 2;
    FILE *f = fopen( argv[ 1 ], "rb" );
    FILE *out = fopen( argv[ 2 ], "wb" );
    if ( !f || !out ) return 3;
    Laser_feature_harness node;
    node.m_piecewise_number = atoi( argv[ 3 ] );
    node.m_odom_mode = atoi( argv[ 4 ] );
    node.m_maximum_input_lidar_pointcloud = atoi( argv[ 5 ] );
    node.m_para_system_delay = atoi( argv[ 6 ] );
    node.m_plane_resolution = ( float ) atof( argv[ 7 ] );
    node.m_line_resolution = ( float ) atof( argv[ 8 ] );
    node.m_if_motion_deblur = atoi( argv[ 9 ] );
    node.m_livox.thr_corner_curvature = 0.05;   // :152-154, 854, 859
    node.m_livox.thr_surface_curvature = 0.01;
    node.m_livox.minimum_view_angle = 10;
    node.m_livox.m_livox_min_allow_dis = 0.1f;
    node.m_livox.m_livox_min_sigma = 7e-4f;
#ifdef LL_USE_ADAPTER
    node.m_livox.piecewise_number = node.m_if_motion_deblur ? 1 : node.m_piecewise_number;
    node.m_livox.max_points = 30000;
#else
    node.m_livox.m_if_verbose_screen_printf = 1;
    node.m_livox.m_last_maximum_time_stamp = 0;  // LFE:152 leaves it uninitialised; defined as 0 (DESIGN, oracle)
#endif
    node.init();
    int n_msgs = 0;
    if ( fread( &n_msgs, sizeof( int ), 1, f ) != 1 ) return 3;
    for ( int k = 0; k < n_msgs; k++ )
    {
        int    hdr[ 2 ];
        double stamp;
        if ( fread( hdr, sizeof( int ), 2, f ) != 2 || fread( &stamp, sizeof( double ), 1, f ) != 1 ) return 3;
        std::shared_ptr<sensor_msgs::PointCloud2> msg( new sensor_msgs::PointCloud2() );
        msg->header.stamp.t = stamp;
        msg->cloud.points.resize( hdr[ 0 ] );
        for ( int i = 0; i < hdr[ 0 ]; i++ )
        {
            float v[ 4 ];
            if ( fread( v, sizeof( float ), 4, f ) != 4 ) return 3;
            msg->cloud.points[ i ].x = v[ 0 ]; msg->cloud.points[ i ].y = v[ 1 ]; msg->cloud.points[ i ].z = v[ 2 ]; msg->cloud.points[ i ].intensity = v[ 3 ];
        }
        node.out_full.clear(); node.out_surface.clear(); node.out_corners.clear();
        node.laserCloudHandler_excerpt( msg, hdr[ 1 ] );
        const int n_pub = ( int ) node.out_full.size();
        fwrite( &n_pub, sizeof( int ), 1, out );
        for ( int j = 0; j < n_pub; j++ )
            for ( auto *cl : { &node.out_full[ j ], &node.out_surface[ j ], &node.out_corners[ j ] } )
            {
                const int n = ( int ) cl->points.size();
                fwrite( &n, sizeof( int ), 1, out );
                for ( auto &p : cl->points )
                {
                    const float v[ 4 ] = { p.x, p.y, p.z, p.intensity };
                    fwrite( v, sizeof( float ), 4, out );
                }
            }
    }
    fclose( out );
    return 0;
}
'''


def build_feature(force=False):
    """-> (adapter exe, reference exe) of the feature-node harness; like build()"""
    if not have_reference():
        return (EXE_FA if os.path.exists(EXE_FA) else None, EXE_FB if os.path.exists(EXE_FB) else None)
    from loam_livox_amd import build as libbuild
    lib = libbuild.build()
    deps = [os.path.abspath(__file__), os.path.join(ROOT, "include", "loam_livox_adapter.hpp"), os.path.join(ROOT, "include", "loam_livox_hip.h"), lib]
    deps += [os.path.join(dp, f) for dp, _, fs in os.walk(os.path.join(ROOT, "oracle", "ref_stubs")) for f in fs]
    if not force and all(os.path.exists(e) and all(os.path.getmtime(d) <= os.path.getmtime(e) for d in deps) for e in (EXE_FA, EXE_FB)):
        return EXE_FA, EXE_FB
    os.makedirs(OUT, exist_ok=True)
    tu = FEAT_HEAD + FEAT_HARNESS.replace("@HANDLER@", _lines("source/laser_feature_extractor.hpp", 256, 392))
    stubs = os.path.join(ROOT, "oracle", "ref_stubs")
    with tempfile.TemporaryDirectory() as td:
        src = os.path.join(td, "verbatim_feature.cpp")
        with open(src, "w") as f:
            f.write(tu)
        inc = ["-I", os.path.join(stubs, "override"), "-I-", "-I", stubs, "-I", os.path.join(REF, "source"), "-I", os.path.join(REF, "include"),
               "-I", os.path.join(REF, "include", "tools"), "-I", os.path.join(ROOT, "include")]
        common = ["g++", "-std=c++14", "-O2", "-ffp-contract=off", "-fno-fast-math", "-w"]
        subprocess.check_call(common + ["-DLL_USE_ADAPTER"] + inc + [src, "-o", EXE_FA, lib, "-Wl,-rpath," + os.path.dirname(lib), "-Wl,-rpath,/opt/rocm/lib", "-lpthread"])
        subprocess.check_call(common + inc + [src, "-o", EXE_FB, "-lpthread"])
    return EXE_FA, EXE_FB


# ---- key-frame assembly of the mapping loop (laser_mapping.hpp:1524-1562 + Maps_keyframe::add_cells, cell_map_keyframe.hpp:1243-1261) ----
# The two excerpts run on scripted "touched cell" sets against minimal stand-ins of the classes around them (no PCL / Eigen needed for
# list bookkeeping); tests/test_keyframes.py holds loam_livox_amd.keyframes.Keyframe_assembly to the state the reference's text reaches.
EXE_KF = os.path.join(OUT, "verbatim_keyframes")
KF_HARNESS = r'''
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <list>
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <vector>
using namespace std;
#define screen_out std::cout
#define FUNC_T
struct PointType {};
struct Cloud_stub {
    int n_added = 0;
    std::shared_ptr<Cloud_stub> makeShared() const { return std::make_shared<Cloud_stub>( *this ); }
    Cloud_stub &operator+=( const Cloud_stub & ) { n_added++; return *this; }
    void clear() {}
};
namespace pcl { struct PointXYZ { float x, y, z; PointXYZ( float a, float b, float c ) : x( a ), y( b ), z( c ) {} }; }
namespace PCL_TOOLS { template <class T, class P> std::vector<int> pcl_pts_to_eigen_pts( std::shared_ptr<Cloud_stub> ) { return std::vector<int>(); } }
struct Center_stub { int id; float operator()( int ) const { return ( float ) id; } bool operator<( const Center_stub &o ) const { return id < o.id; } };
struct Mapping_cell { Center_stub m_center; Center_stub get_center() const { return m_center; } };
struct Quat_stub { double v[ 4 ]; };
struct Vec3_stub { double v[ 3 ]; };
template <typename T> struct Points_cloud_map
{
    typedef std::shared_ptr<Mapping_cell> Mapping_cell_ptr;
    std::vector<std::vector<int>> script;  // the cells each scan touches
    size_t frame = 0;
    std::map<int, Mapping_cell_ptr> cells;
    Mapping_cell_ptr cell( int id )
    {
        auto it = cells.find( id );
        if ( it != cells.end() ) return it->second;
        Mapping_cell_ptr c = std::make_shared<Mapping_cell>();
        c->m_center.id = id;
        cells[ id ] = c;
        return c;
    }
    template <class V> void append_cloud( const V &, std::set<Mapping_cell_ptr> *cell_vec = nullptr )
    {
        if ( cell_vec )
        {
            cell_vec->clear();
            for ( int id : script[ frame ] ) cell_vec->insert( cell( id ) );
        }
        frame++;
    }
};
template <typename T> struct Maps_keyframe
{
    typedef std::shared_ptr<Mapping_cell> Mapping_cell_ptr;
    std::set<Mapping_cell_ptr>              m_set_cell;
    std::map<Center_stub, Mapping_cell_ptr> m_map_pt_cell;
    std::shared_ptr<std::vector<pcl::PointXYZ>> m_pcl_cells_center = std::make_shared<std::vector<pcl::PointXYZ>>();
    unsigned int m_accumulate_frames = 0;
    int          m_ending_frame_idx = 0;
    Quat_stub    m_pose_q;
    Vec3_stub    m_pose_t;
    Cloud_stub   m_accumulated_point_cloud;
@ADD_CELLS@
};
struct Laser_mapping_kf_harness
{
    int m_loop_closure_if_enable = 1, m_para_scans_of_each_keyframe = 300, m_para_scans_between_two_keyframe = 100, m_current_frame_index = 0;
    size_t m_loop_closure_maximum_keyframe_in_wating_list = 3;
    Quat_stub m_q_w_curr;
    Vec3_stub m_t_w_curr;
    std::mutex m_mutex_keyframe, m_mutex_dump_full_history;
    Points_cloud_map<float> m_pt_cell_map_full;
    std::list<std::shared_ptr<Maps_keyframe<float>>> m_keyframe_of_updating_list, m_keyframe_need_precession_list;
    std::list<Cloud_stub> m_laser_cloud_full_history;
    Laser_mapping_kf_harness() { m_keyframe_of_updating_list.push_back( std::make_shared<Maps_keyframe<float>>() ); }  // laser_mapping.hpp:626
    void one_scan()
    {
        Cloud_stub current_laser_cloud_full;
        m_laser_cloud_full_history.push_back( current_laser_cloud_full );
@ASSEMBLY@
    }
};
int main( int argc, char **argv )
{
    if ( argc < 6 ) return 2;
    Laser_mapping_kf_harness node;
    node.m_para_scans_of_each_keyframe = atoi( argv[ 1 ] );
    node.m_para_scans_between_two_keyframe = atoi( argv[ 2 ] );
    node.m_loop_closure_maximum_keyframe_in_wating_list = ( size_t ) atoi( argv[ 3 ] );
    FILE *f = fopen( argv[ 4 ], "r" ), *out = fopen( argv[ 5 ], "w" );
    if ( !f || !out ) return 3;
    int n_frames = 0;
    if ( fscanf( f, "%d", &n_frames ) != 1 ) return 4;
    for ( int k = 0; k < n_frames; k++ )
    {
        int n = 0;
        if ( fscanf( f, "%d", &n ) != 1 ) return 4;
        std::vector<int> ids( n );
        for ( int i = 0; i < n; i++ ) if ( fscanf( f, "%d", &ids[ i ] ) != 1 ) return 4;
        node.m_pt_cell_map_full.script.push_back( ids );
    }
    std::cout.setstate( std::ios_base::failbit );  // (the excerpt's own chatter)
    for ( int k = 0; k < n_frames; k++ )
    {
        node.m_current_frame_index = k + 1;
        node.one_scan();
        fprintf( out, "%d U", k );
        for ( auto &kf : node.m_keyframe_of_updating_list ) fprintf( out, " %u:%d", kf->m_accumulate_frames, ( int ) kf->m_set_cell.size() );
        fprintf( out, " W" );
        for ( auto &kf : node.m_keyframe_need_precession_list ) fprintf( out, " %u:%d:%d", kf->m_accumulate_frames, ( int ) kf->m_set_cell.size(), kf->m_ending_frame_idx );
        fprintf( out, "\n" );
    }
    fclose( out );
    return 0;
}
'''


def build_keyframes(force=False):
    """-> exe of the key-frame assembly harness (None where neither /root/reference nor a travelled binary exists)"""
    if not have_reference():
        return EXE_KF if os.path.exists(EXE_KF) else None
    deps = [os.path.abspath(__file__)]
    if not force and os.path.exists(EXE_KF) and all(os.path.getmtime(d) <= os.path.getmtime(EXE_KF) for d in deps):
        return EXE_KF
    os.makedirs(OUT, exist_ok=True)
    tu = (KF_HARNESS.replace("@ADD_CELLS@", _lines("source/cell_map_keyframe.hpp", 1243, 1261))
                    .replace("@ASSEMBLY@", _lines("source/laser_mapping.hpp", 1524, 1564)))
    with tempfile.TemporaryDirectory() as td:
        src = os.path.join(td, "verbatim_keyframes.cpp")
        with open(src, "w") as f:
            f.write(tu)
        subprocess.check_call(["g++", "-O1", "-std=c++14", "-w", "-o", EXE_KF, src])
    return EXE_KF


# ---- Scene_alignment::find_tranfrom_of_two_mappings, the WHOLE driver (source/scene_alignment.hpp:269-391), on the reference's own
# classes: Maps_keyframe / Points_cloud_map (cell_map_keyframe.hpp, compiled verbatim -- no override stub here) and
# Point_cloud_registration (point_cloud_registration.hpp, verbatim), against the stand-in third-party headers of oracle/ref_stubs
# (pcl::VoxelGrid, KdTreeFLANN, ceres::Solve, Eigen, the OpenCV calls).  scene_alignment.hpp itself is not included as a whole: its
# other includes (pcl/registration/ndt.h, icp.h, ceres_pose_graph_3d.hpp, g2o dump) are off the path.  The excerpted lines are the
# members the driver uses (:27-36), set_downsample_resolution (:214-222), the registrar set-up of init() (:233-243) and the driver
# (:269-391).  CPU only: pins oracle/orc_scene_alignment.py (tests/test_cellmap.py).
EXE_SCENE = os.path.join(OUT, "verbatim_scene_alignment")
SA_HARNESS = r'''
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <set>
#include <vector>
#include "cell_map_keyframe.hpp"
#include "point_cloud_registration.hpp"
#include <pcl/filters/voxel_grid.h>

template < typename PT_DATA_TYPE >
class Scene_alignment
{
  public:
    Common_tools::File_logger file_logger_commond, file_logger_timer;
    Common_tools::Timer       timer;
// ---- verbatim: scene_alignment.hpp:27-36
@SA_MEMBERS@
    ADD_SCREEN_PRINTF_OUT_METHOD;
    Scene_alignment()
    {
        m_if_verbose_screen_printf = 0;
// ---- verbatim: scene_alignment.hpp:233-243 (the registrar set-up of init(); the loggers are never opened: their streams
//      swallow what the registrar writes, as in oracle/ref_shim.cpp)
@SA_INIT@
        m_pc_reg.m_logger_pcd = &file_logger_pcd;
    }
    Common_tools::File_logger file_logger_pcd;
// ---- verbatim: scene_alignment.hpp:214-222
@SA_RES@
// ---- verbatim: scene_alignment.hpp:269-391
@SA_DRIVER@
};

typedef Points_cloud_map< float >                   Map_t;
typedef Maps_keyframe< float >                      Kf_t;
typedef Eigen::Matrix< float, 3, 1 >                Pt_t;

static std::vector< Pt_t > read_cloud( const char *path )
{
    std::vector< Pt_t > v;
    FILE *              f = fopen( path, "rb" );
    if ( !f ) exit( 3 );
    float p[ 3 ];
    while ( fread( p, sizeof( float ), 3, f ) == 3 ) v.push_back( Pt_t( p[ 0 ], p[ 1 ], p[ 2 ] ) );
    fclose( f );
    return v;
}

// usage: exe a.bin b.bin line_res plane_res max_icp accepted max_blocks out.txt   (clouds: float32 x y z)
int main( int argc, char **argv )
{
    if ( argc < 9 ) return 2;
    std::streambuf *old = std::cout.rdbuf( nullptr );  // the reference chats on std::cout
    Map_t map_a, map_b;
    Kf_t  kf_a, kf_b;
    Map_t *maps[ 2 ] = { &map_a, &map_b };
    Kf_t * kfs[ 2 ] = { &kf_a, &kf_b };
    for ( int i = 0; i < 2; i++ )
    {
        maps[ i ]->set_resolution( 1.0 );                                   // laser_mapping.hpp:616
        std::set< Map_t::Mapping_cell_ptr > cell_vec;
        maps[ i ]->append_cloud( read_cloud( argv[ 1 + i ] ), &cell_vec );  // every cell of the (first) cloud
        kfs[ i ]->add_cells( cell_vec );
        kfs[ i ]->update_features_of_each_cells( 1 );                       // as the detector does before it aligns (:927-934)
        kfs[ i ]->analyze( 1 );
    }
    Scene_alignment< float > sa;
    sa.set_downsample_resolution( atof( argv[ 3 ] ), atof( argv[ 4 ] ) );
    sa.m_maximum_icp_iteration = atoi( argv[ 5 ] );
    sa.m_accepted_threshold = atof( argv[ 6 ] );
    sa.m_para_scene_alignments_maximum_residual_block = atoi( argv[ 7 ] );
    const double thr = sa.find_tranfrom_of_two_mappings( &kf_a, &kf_b, 0 );
    std::cout.rdbuf( old );
    FILE *out = fopen( argv[ 8 ], "w" );
    fprintf( out, "%.17g %d %.17g\n", thr, sa.m_pc_reg.m_para_icp_max_iterations, ( double ) sa.m_pc_reg.m_inlier_threshold );
    fprintf( out, "%.17g %.17g %.17g %.17g %.17g %.17g %.17g\n", sa.m_pc_reg.m_q_w_curr.x(), sa.m_pc_reg.m_q_w_curr.y(), sa.m_pc_reg.m_q_w_curr.z(),
             sa.m_pc_reg.m_q_w_curr.w(), sa.m_pc_reg.m_t_w_curr( 0 ), sa.m_pc_reg.m_t_w_curr( 1 ), sa.m_pc_reg.m_t_w_curr( 2 ) );
    fclose( out );
    return 0;
}
'''


def build_scene_alignment(force=False):
    """-> exe of the scene-alignment harness (None where neither /root/reference nor a travelled binary exists)"""
    if not have_reference():
        return EXE_SCENE if os.path.exists(EXE_SCENE) else None
    stubs = os.path.join(ROOT, "oracle", "ref_stubs")
    deps = [os.path.abspath(__file__)] + [os.path.join(dp, f) for dp, _, fs in os.walk(stubs) for f in fs]
    if not force and os.path.exists(EXE_SCENE) and all(os.path.getmtime(d) <= os.path.getmtime(EXE_SCENE) for d in deps):
        return EXE_SCENE
    os.makedirs(OUT, exist_ok=True)
    tu = (SA_HARNESS.replace("@SA_MEMBERS@", _lines("source/scene_alignment.hpp", 27, 36))
                    .replace("@SA_INIT@", _lines("source/scene_alignment.hpp", 233, 243))
                    .replace("@SA_RES@", _lines("source/scene_alignment.hpp", 214, 222))
                    .replace("@SA_DRIVER@", _lines("source/scene_alignment.hpp", 269, 391)))
    with tempfile.TemporaryDirectory() as td:
        src = os.path.join(td, "verbatim_scene_alignment.cpp")
        with open(src, "w") as f:
            f.write(tu)
        subprocess.check_call(["g++", "-std=c++14", "-O2", "-ffp-contract=off", "-fno-fast-math", "-w", "-I", stubs, "-I", os.path.join(REF, "source"),
                               "-I", os.path.join(REF, "include"), "-I", os.path.join(REF, "include", "tools"), "-o", EXE_SCENE, src])
    return EXE_SCENE


# ---- service_loop_detection's loop over the earlier key frames (source/laser_mapping.hpp:988-1057 + 1110-1127): which pairs are
# compared, which are handed to the scene alignment, how `his` advances (continue / += 10 / += 5 inside a for), when a loop is
# declared.  Key frames, image similarity and the alignment are scripted stand-ins; the lines in between (:1058-1109: the pose
# graph, g2o dump, map refinement -- out of scope) are replaced by a record of the loop.  CPU only: pins
# loam_livox_amd/keyframes.py Keyframe_assembly.process_waiting (tests/test_keyframes.py).
EXE_LOOP = os.path.join(OUT, "verbatim_loop_detector")
LOOP_HARNESS = r'''
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <set>
#include <string>
#include <thread>
#include <vector>

using namespace std;  // include/tools/tools_logger.hpp:86 does this for every translation unit of the reference: the unqualified
                      // abs( float ) of laser_mapping.hpp:1004 is std::abs( float ), not ::abs( int )
static int                              g_K;
static std::vector< float >             g_sim[ 2 ], g_thr;  // [kind][a * K + b], thr[a * K + b]
static FILE *                           g_out;
struct Img { int kf, kind; };
struct Maps_keyframe_stub
{
    float           m_ratio_nonzero_plane, m_ratio_nonzero_line, m_roi_range;
    Img             m_feature_img_plane, m_feature_img_line, m_feature_img_plane_roi, m_feature_img_line_roi;
    std::set< int > m_set_cell;
    float max_similiarity_of_two_image( const Img &a, const Img &b )
    {
        if ( a.kind == 0 ) fprintf( g_out, "S %d %d\n", a.kf, b.kf );
        return g_sim[ a.kind ][ a.kf * g_K + b.kf ];
    }
};
struct Logger_stub { void printf( const char *, ... ) {} };
struct Summary_stub { std::string BriefReport() { return ""; } };
struct Reg_stub { double m_inlier_threshold = 0; Summary_stub m_final_opt_summary; };
struct Scene_alignment_stub
{
    Reg_stub m_pc_reg;
    int      m_para_scene_alignments_maximum_residual_block = 0;
    void set_downsample_resolution( float, float ) {}
    int find_tranfrom_of_two_mappings( std::shared_ptr< Maps_keyframe_stub > a, std::shared_ptr< Maps_keyframe_stub > b, int )
    {
        m_pc_reg.m_inlier_threshold = g_thr[ a->m_feature_img_plane.kf * g_K + b->m_feature_img_plane.kf ];
        fprintf( g_out, "A %d %d\n", a->m_feature_img_plane.kf, b->m_feature_img_plane.kf );
        return ( int ) m_pc_reg.m_inlier_threshold;
    }
};
#define screen_printf( ... ) do { } while ( 0 )

// usage: exe script.txt out.txt
int main( int argc, char **argv )
{
    if ( argc < 3 ) return 2;
    FILE *in = fopen( argv[ 1 ], "r" );
    g_out = fopen( argv[ 2 ], "w" );
    if ( !in || !g_out ) return 3;
    int   m_loop_closure_minimum_keyframe_differen;
    float avail_ratio_plane, avail_ratio_line, m_loop_closure_minimum_similarity_linear, m_loop_closure_minimum_similarity_planar;
    float m_loop_closure_map_alignment_inlier_threshold;
    if ( fscanf( in, "%d %d %f %f %f %f %f", &g_K, &m_loop_closure_minimum_keyframe_differen, &avail_ratio_plane, &avail_ratio_line,
                 &m_loop_closure_minimum_similarity_linear, &m_loop_closure_minimum_similarity_planar, &m_loop_closure_map_alignment_inlier_threshold ) != 7 )
        return 4;
    std::vector< std::shared_ptr< Maps_keyframe_stub > > all( g_K );
    for ( int k = 0; k < g_K; k++ )
    {
        all[ k ] = std::make_shared< Maps_keyframe_stub >();
        int n_cells;
        if ( fscanf( in, "%f %f %f %d", &all[ k ]->m_ratio_nonzero_plane, &all[ k ]->m_ratio_nonzero_line, &all[ k ]->m_roi_range, &n_cells ) != 4 ) return 4;
        for ( int c = 0; c < n_cells; c++ ) all[ k ]->m_set_cell.insert( c );
        all[ k ]->m_feature_img_plane = all[ k ]->m_feature_img_plane_roi = Img{ k, 0 };
        all[ k ]->m_feature_img_line = all[ k ]->m_feature_img_line_roi = Img{ k, 1 };
    }
    for ( int t = 0; t < 3; t++ )
    {
        std::vector< float > &v = t < 2 ? g_sim[ t ] : g_thr;
        v.resize( ( size_t ) g_K * g_K );
        for ( float &x : v )
            if ( fscanf( in, "%f", &x ) != 1 ) return 4;
    }
    std::vector< std::shared_ptr< Maps_keyframe_stub > > keyframe_vec;
    std::vector< std::string >                           m_filename_vec;
    Logger_stub          m_logger_loop_closure;
    Scene_alignment_stub m_scene_align;
    float                m_loop_closure_map_alignment_resolution = 0.2;
    int                  m_para_scene_alignments_maximum_residual_block = 5000, m_loop_closure_map_alignment_if_dump_matching_result = 0;
    int                  if_end = 0;
    for ( int k = 0; k < g_K && !if_end; k++ )  // `while ( 1 )` of the service thread: one pass per key frame that arrives
    {
        keyframe_vec.push_back( all[ k ] );
        m_filename_vec.push_back( std::to_string( k ) );
        std::shared_ptr< Maps_keyframe_stub > last_keyframe = keyframe_vec.back();
        float sim_plane_res_cv = 0, sim_plane_res = 0;
        float sim_line_res_cv = 0, sim_line_res = 0;
        float sim_plane_res_roi = 0, sim_line_res_roi = 0;
        fprintf( g_out, "K %d\n", k );
// ---- verbatim: laser_mapping.hpp:988-1057
@LOOP_HEAD@
                        fprintf( g_out, "L %d %d\n", ( int ) keyframe_vec.size() - 1, ( int ) his );  // (:1058-1109: pose graph, refinement)
// ---- verbatim: laser_mapping.hpp:1110-1127
@LOOP_TAIL@
    }
    fclose( g_out );
    return 0;
}
'''


def build_loop_detector(force=False):
    """-> exe of the loop-detector harness (None where neither /root/reference nor a travelled binary exists)"""
    if not have_reference():
        return EXE_LOOP if os.path.exists(EXE_LOOP) else None
    deps = [os.path.abspath(__file__)]
    if not force and os.path.exists(EXE_LOOP) and all(os.path.getmtime(d) <= os.path.getmtime(EXE_LOOP) for d in deps):
        return EXE_LOOP
    os.makedirs(OUT, exist_ok=True)
    tu = (LOOP_HARNESS.replace("@LOOP_HEAD@", _lines("source/laser_mapping.hpp", 988, 1057))
                      .replace("@LOOP_TAIL@", _lines("source/laser_mapping.hpp", 1110, 1127)))
    with tempfile.TemporaryDirectory() as td:
        src = os.path.join(td, "verbatim_loop_detector.cpp")
        with open(src, "w") as f:
            f.write(tu)
        subprocess.check_call(["g++", "-O1", "-std=c++14", "-w", "-o", EXE_LOOP, src, "-lpthread"])
    return EXE_LOOP


# ---- update_buff_for_matching, BOTH branches in one verbatim range (source/laser_mapping.hpp:465-537), with if_pt_in_fov (:309-324) and
# the cell-map appends of process_new_scan (:1492-1493), on the reference's own Points_cloud_map (cell_map_keyframe.hpp verbatim) and
# the stand-in pcl::VoxelGrid.  Pins the cell ("cube") branch of the match-buffer refresh: oracle/orc_mapping.py History.refresh_cells
# (tests/test_cellmap.py).  CPU only.
EXE_CELLREFRESH = os.path.join(OUT, "verbatim_cell_refresh")
CELLREFRESH_HARNESS = r'''
#include <cstdio>
#include <cstdlib>
#include <list>
#include <mutex>
#include <vector>
#include "cell_map_keyframe.hpp"
#include "tools_eigen_math.hpp"
#include <pcl/filters/voxel_grid.h>

typedef pcl::PointXYZI PointType;

struct Mapping_excerpt
{
    Points_cloud_map< float >               m_pt_cell_map_corners, m_pt_cell_map_planes;
    Eigen::Quaterniond                      m_q_w_curr = Eigen::Quaterniond( 1, 0, 0, 0 );
    Eigen::Vector3d                         m_t_w_curr = Eigen::Vector3d( 0, 0, 0 );
    float                                   m_maximum_in_fov_angle = 30, m_maximum_search_range_corner = 100, m_maximum_search_range_surface = 100;
    float                                   m_line_resolution = 0.1, m_plane_resolution = 0.4;
    int                                     m_matching_mode = 1, m_down_sample_replace = 1;
    pcl::VoxelGrid< PointType >             m_down_sample_filter_corner, m_down_sample_filter_surface;
    std::list< pcl::PointCloud< PointType > > m_laser_cloud_corner_history, m_laser_cloud_surface_history;
    std::mutex                              m_mutex_mapping;
    pcl::PointCloud< PointType >            out_corner, out_surf;

// ---- verbatim: laser_mapping.hpp:309-324
@FOV@

    void append( pcl::PointCloud< PointType >::Ptr pc_new_feature_corners, pcl::PointCloud< PointType >::Ptr pc_new_feature_surface )
    {
// ---- verbatim: laser_mapping.hpp:1492-1493
@APPEND@
    }

    void refresh()
    {
// ---- verbatim: laser_mapping.hpp:465-537
@REFRESH@
        out_corner = *laser_cloud_corner_from_map;
        out_surf = *laser_cloud_surf_from_map;
    }
};

static pcl::PointCloud< PointType >::Ptr read_cloud( FILE *f, int n )
{
    pcl::PointCloud< PointType >::Ptr c( new pcl::PointCloud< PointType >() );
    for ( int i = 0; i < n; i++ )
    {
        float v[ 4 ];
        if ( fread( v, sizeof( float ), 4, f ) != 4 ) exit( 3 );
        PointType p;
        p.x = v[ 0 ], p.y = v[ 1 ], p.z = v[ 2 ], p.intensity = v[ 3 ];
        c->points.push_back( p );
    }
    return c;
}
static void write_cloud( FILE *f, const pcl::PointCloud< PointType > &c )
{
    int n = ( int ) c.points.size();
    fwrite( &n, sizeof( int ), 1, f );
    for ( auto &p : c.points )
    {
        float v[ 4 ] = { p.x, p.y, p.z, p.intensity };
        fwrite( v, sizeof( float ), 4, f );
    }
}

// usage: exe in.bin out.bin cell_res revisit line_res plane_res range_c range_s fov replace
// in.bin: int n_frames; per frame: double pose[7] {qx qy qz qw tx ty tz}, int n_corner, int n_surf, corner xyzi, surf xyzi (map frame)
// out.bin: per frame the two match-buffer clouds after the refresh at that frame's pose
int main( int argc, char **argv )
{
    if ( argc < 11 ) return 2;
    std::streambuf *old = std::cout.rdbuf( nullptr );
    FILE *in = fopen( argv[ 1 ], "rb" ), *out = fopen( argv[ 2 ], "wb" );
    if ( !in || !out ) return 3;
    Mapping_excerpt m;
    m.m_pt_cell_map_corners.set_resolution( atof( argv[ 3 ] ) );   // laser_mapping.hpp:620-624
    m.m_pt_cell_map_planes.set_resolution( atof( argv[ 3 ] ) );
    m.m_pt_cell_map_corners.m_minimum_revisit_threshold = atoi( argv[ 4 ] );
    m.m_pt_cell_map_planes.m_minimum_revisit_threshold = atoi( argv[ 4 ] );
    m.m_line_resolution = atof( argv[ 5 ] );
    m.m_plane_resolution = atof( argv[ 6 ] );
    m.m_down_sample_filter_corner.setLeafSize( m.m_line_resolution, m.m_line_resolution, m.m_line_resolution );      // :742-743
    m.m_down_sample_filter_surface.setLeafSize( m.m_plane_resolution, m.m_plane_resolution, m.m_plane_resolution );
    m.m_maximum_search_range_corner = atof( argv[ 7 ] );
    m.m_maximum_search_range_surface = atof( argv[ 8 ] );
    m.m_maximum_in_fov_angle = atof( argv[ 9 ] );
    m.m_down_sample_replace = atoi( argv[ 10 ] );
    int n_frames = 0;
    if ( fread( &n_frames, sizeof( int ), 1, in ) != 1 ) return 4;
    for ( int k = 0; k < n_frames; k++ )
    {
        double pose[ 7 ];
        int    nc, ns;
        if ( fread( pose, sizeof( double ), 7, in ) != 7 || fread( &nc, sizeof( int ), 1, in ) != 1 || fread( &ns, sizeof( int ), 1, in ) != 1 ) return 4;
        auto c = read_cloud( in, nc ), s = read_cloud( in, ns );
        m.append( c, s );
        m.m_q_w_curr = Eigen::Quaterniond( pose[ 3 ], pose[ 0 ], pose[ 1 ], pose[ 2 ] );
        m.m_t_w_curr = Eigen::Vector3d( pose[ 4 ], pose[ 5 ], pose[ 6 ] );
        m.refresh();
        write_cloud( out, m.out_corner );
        write_cloud( out, m.out_surf );
    }
    std::cout.rdbuf( old );
    fclose( out );
    return 0;
}
'''


def build_cell_refresh(force=False):
    """-> exe of the cell-branch refresh harness (None where neither /root/reference nor a travelled binary exists)"""
    if not have_reference():
        return EXE_CELLREFRESH if os.path.exists(EXE_CELLREFRESH) else None
    stubs = os.path.join(ROOT, "oracle", "ref_stubs")
    deps = [os.path.abspath(__file__)] + [os.path.join(dp, f) for dp, _, fs in os.walk(stubs) for f in fs]
    if not force and os.path.exists(EXE_CELLREFRESH) and all(os.path.getmtime(d) <= os.path.getmtime(EXE_CELLREFRESH) for d in deps):
        return EXE_CELLREFRESH
    os.makedirs(OUT, exist_ok=True)
    tu = (CELLREFRESH_HARNESS.replace("@FOV@", _lines("source/laser_mapping.hpp", 309, 324))
                             .replace("@APPEND@", _lines("source/laser_mapping.hpp", 1492, 1493))
                             .replace("@REFRESH@", _lines("source/laser_mapping.hpp", 465, 537)))
    with tempfile.TemporaryDirectory() as td:
        src = os.path.join(td, "verbatim_cell_refresh.cpp")
        with open(src, "w") as f:
            f.write(tu)
        subprocess.check_call(["g++", "-std=c++14", "-O2", "-ffp-contract=off", "-fno-fast-math", "-w", "-I", stubs, "-I", os.path.join(REF, "source"),
                               "-I", os.path.join(REF, "include"), "-I", os.path.join(REF, "include", "tools"), "-o", EXE_CELLREFRESH, src])
    return EXE_CELLREFRESH


# ---------------------------------------------------------------------------------------------------------------------------------
# The mapping node's message surface (output side): the three cloud handlers with the queue of complete triples, the
# body of Laser_mapping::process's loop with the maximum_mapping_buffer drop rule, and what process_new_scan publishes after a
# registration -- the reference's own text compiled against stub ROS message types (ROS is absent from the image), driven by the io record
# tools/ll_node.cpp writes (--dump-io) and replays (--replay-io).  Verbatim line ranges of source/laser_mapping.hpp:
#   89-120     struct Data_pair                                 633-647    get_data_pair
#   749-780    the three handlers                               1701-1733, 1735  loop body of process(): drop rule, take, fromROSMsg
#   1570-1575  /velodyne_cloud_registered                       1613-1653  /aft_mapped_to_init, /aft_mapped_path, camera_init -> aft_mapped
# The harness logs every message a stub publisher / the stub broadcaster receives in the format of ll_node's recorder.
EXE_MAPIO = os.path.join(OUT, "verbatim_mapping_io")
MAPIO_HARNESS = r'''
#include <cinttypes>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <map>
#include <memory>
#include <mutex>
#include <queue>
#include <string>
#include <vector>
#include <unistd.h>
#include <Eigen/Eigen>
#include <pcl/point_cloud.h>
#include <pcl/point_types.h>
#include "tools/common.h"
#include "tools/tools_logger.hpp"
#include "tools/tools_timer.hpp"
using namespace std;

// ---- stub ROS message types: the fields the excerpts touch, named and nested as in ROS ------------------------------------------
namespace ros
{
struct Time
{
    double t = 0;
    Time &fromSec( double s ) { t = s; return *this; }
    double toSec() const { return t; }
};
} // namespace ros
namespace std_msgs
{
struct Header
{
    uint32_t    seq = 0;
    ros::Time   stamp;
    std::string frame_id;
};
} // namespace std_msgs
namespace sensor_msgs
{
struct PointField
{
    std::string name;
    uint32_t    offset = 0;
    uint8_t     datatype = 7;
    uint32_t    count = 1;
};
struct PointCloud2
{
    std_msgs::Header        header;
    uint32_t                height = 1, width = 0;
    std::vector<PointField> fields;
    bool                    is_bigendian = false;
    uint32_t                point_step = 0, row_step = 0;
    std::vector<uint8_t>    data;
    bool                    is_dense = true;
};
typedef std::shared_ptr<const PointCloud2> PointCloud2ConstPtr;
} // namespace sensor_msgs
namespace geometry_msgs
{
struct Point { double x = 0, y = 0, z = 0; };
struct Quaternion { double x = 0, y = 0, z = 0, w = 1; };
struct Pose { Point position; Quaternion orientation; };
struct PoseWithCovariance { Pose pose; double covariance[ 36 ] = { 0 }; };
struct PoseStamped { std_msgs::Header header; Pose pose; };
} // namespace geometry_msgs
namespace nav_msgs
{
struct Odometry
{
    std_msgs::Header                  header;
    std::string                       child_frame_id;
    geometry_msgs::PoseWithCovariance pose;
};
struct Path
{
    std_msgs::Header                        header;
    std::vector<geometry_msgs::PoseStamped> poses;
};
} // namespace nav_msgs
namespace pcl
{
// pcl::toROSMsg / fromROSMsg for PointXYZI: the point structs copied as they lie (32 bytes: x, y, z at 0 / 4 / 8, intensity at 16)
inline void toROSMsg( const PointCloud<PointXYZI> &c, sensor_msgs::PointCloud2 &m )
{
    static const char *names[ 4 ] = { "x", "y", "z", "intensity" };
    static const uint32_t offs[ 4 ] = { 0, 4, 8, 16 };
    m.fields.resize( 4 );
    for ( int i = 0; i < 4; i++ ) { m.fields[ i ].name = names[ i ]; m.fields[ i ].offset = offs[ i ]; m.fields[ i ].datatype = 7; m.fields[ i ].count = 1; }
    m.height = 1;
    m.width = ( uint32_t ) c.points.size();
    m.point_step = sizeof( PointXYZI );
    m.row_step = m.point_step * m.width;
    m.data.resize( ( size_t ) m.row_step );
    if ( m.width ) memcpy( m.data.data(), c.points.data(), m.data.size() );
}
inline void fromROSMsg( const sensor_msgs::PointCloud2 &m, PointCloud<PointXYZI> &c )
{
    int off[ 4 ] = { -1, -1, -1, -1 };
    for ( const auto &f : m.fields )
    {
        if ( f.name == "x" ) off[ 0 ] = f.offset;
        if ( f.name == "y" ) off[ 1 ] = f.offset;
        if ( f.name == "z" ) off[ 2 ] = f.offset;
        if ( f.name == "intensity" ) off[ 3 ] = f.offset;
    }
    const size_t n = ( size_t ) m.width * m.height;
    c.points.assign( n, PointXYZI() );
    for ( size_t i = 0; i < n; i++ )
    {
        const uint8_t *p = m.data.data() + i * m.point_step;
        memcpy( &c.points[ i ].x, p + off[ 0 ], 4 );
        memcpy( &c.points[ i ].y, p + off[ 1 ], 4 );
        memcpy( &c.points[ i ].z, p + off[ 2 ], 4 );
        if ( off[ 3 ] >= 0 ) memcpy( &c.points[ i ].intensity, p + off[ 3 ], 4 );
    }
}
} // namespace pcl

static FILE *g_log = nullptr;
static uint64_t cloud_hash( const pcl::PointCloud<PointType> &c ) // tools/ll_sequence.py: cloud_hash over (x, y, z, intensity)
{
    uint64_t h = 0, i = 0;
    for ( const auto &p : c.points )
    {
        uint32_t w[ 4 ];
        memcpy( &w[ 0 ], &p.x, 4 ); memcpy( &w[ 1 ], &p.y, 4 ); memcpy( &w[ 2 ], &p.z, 4 ); memcpy( &w[ 3 ], &p.intensity, 4 );
        for ( int k = 0; k < 4; k++, i++ ) h += ( uint64_t ) w[ k ] * ( uint64_t )( 2 * i + 1 );
    }
    return h;
}
namespace ros
{
struct Publisher
{
    std::string topic;
    void publish( const sensor_msgs::PointCloud2 &m ) const
    {
        pcl::PointCloud<PointType> c;
        pcl::fromROSMsg( m, c );
        fprintf( g_log, "CLOUD %s %.17g %s %u %u %u %d %016" PRIx64 "\n", topic.c_str(), m.header.stamp.toSec(), m.header.frame_id.c_str(), m.width, m.height,
                 m.point_step, ( int ) m.fields.size(), cloud_hash( c ) );
    }
    void publish( const nav_msgs::Odometry &o ) const
    {
        fprintf( g_log, "ODOM %s %.17g %s %s %.17g %.17g %.17g %.17g %.17g %.17g %.17g\n", topic.c_str(), o.header.stamp.toSec(), o.header.frame_id.c_str(),
                 o.child_frame_id.c_str(), o.pose.pose.position.x, o.pose.pose.position.y, o.pose.pose.position.z, o.pose.pose.orientation.x,
                 o.pose.pose.orientation.y, o.pose.pose.orientation.z, o.pose.pose.orientation.w );
    }
    void publish( const nav_msgs::Path &p ) const
    {
        const geometry_msgs::PoseStamped &l = p.poses.back();
        fprintf( g_log, "PATH %s %.17g %s %zu %.17g %s %.17g %.17g %.17g %.17g %.17g %.17g %.17g\n", topic.c_str(), p.header.stamp.toSec(), p.header.frame_id.c_str(),
                 p.poses.size(), l.header.stamp.toSec(), l.header.frame_id.c_str(), l.pose.position.x, l.pose.position.y, l.pose.position.z, l.pose.orientation.x,
                 l.pose.orientation.y, l.pose.orientation.z, l.pose.orientation.w );
    }
};
} // namespace ros
namespace tf
{
struct Vector3
{
    double v[ 3 ];
    Vector3( double x = 0, double y = 0, double z = 0 ) : v{ x, y, z } {}
};
struct Quaternion
{
    double x = 0, y = 0, z = 0, w = 1;
    void setW( double a ) { w = a; }
    void setX( double a ) { x = a; }
    void setY( double a ) { y = a; }
    void setZ( double a ) { z = a; }
};
struct Transform
{
    Vector3    origin;
    Quaternion rotation;
    void setOrigin( const Vector3 &o ) { origin = o; }
    void setRotation( const Quaternion &q ) { rotation = q; }
};
struct StampedTransform : Transform
{
    ros::Time   stamp_;
    std::string frame_id_, child_frame_id_;
    StampedTransform( const Transform &t, const ros::Time &s, const std::string &f, const std::string &c ) : Transform( t ), stamp_( s ), frame_id_( f ), child_frame_id_( c ) {}
};
struct TransformBroadcaster
{
    void sendTransform( const StampedTransform &t )
    {
        fprintf( g_log, "TF %.17g %s %s %.17g %.17g %.17g %.17g %.17g %.17g %.17g\n", t.stamp_.toSec(), t.frame_id_.c_str(), t.child_frame_id_.c_str(), t.origin.v[ 0 ],
                 t.origin.v[ 1 ], t.origin.v[ 2 ], t.rotation.x, t.rotation.y, t.rotation.z, t.rotation.w );
    }
};
} // namespace tf

// ---- verbatim: laser_mapping.hpp:89-120
@DATA_PAIR@
// ---- end of excerpt

#define ROS_WARN( ... ) fprintf( g_log, "DROP %.17g\n", m_queue_avail_data.front()->m_pc_corner->header.stamp.toSec() )

class Laser_mapping_harness
{
  public:
    int    m_current_frame_index = 0, m_max_buffer_size = 5;
    double m_time_pc_corner_past = 0;
    double m_para_buffer_RT[ 7 ] = { 0, 0, 0, 1, 0, 0, 0 };
    Eigen::Map<Eigen::Quaterniond> m_q_w_curr = Eigen::Map<Eigen::Quaterniond>( m_para_buffer_RT );
    Eigen::Map<Eigen::Vector3d>    m_t_w_curr = Eigen::Map<Eigen::Vector3d>( m_para_buffer_RT + 4 );
    std::mutex m_mutex_buf, m_mutex_querypointcloud, m_mutex_ros_pub;
    std::map<double, Data_pair *> m_map_data_pair;   // :215
    std::queue<Data_pair *>       m_queue_avail_data; // :216
    Common_tools::File_logger m_logger_common, m_logger_timer;
    Common_tools::Timer       m_timer;
    pcl::PointCloud<PointType>::Ptr m_laser_cloud_corner_last{ new pcl::PointCloud<PointType>() }, m_laser_cloud_surf_last{ new pcl::PointCloud<PointType>() },
        m_laser_cloud_full_res{ new pcl::PointCloud<PointType>() };
    ros::Publisher m_pub_laser_cloud_full_res{ "/velodyne_cloud_registered" }, m_pub_odom_aft_mapped{ "/aft_mapped_to_init" },
        m_pub_laser_aft_mapped_path{ "/aft_mapped_path" }; // :611, 612, 614
    nav_msgs::Path m_laser_after_mapped_path;

// ---- verbatim: laser_mapping.hpp:633-647
@GET_PAIR@
// ---- verbatim: laser_mapping.hpp:749-780
@HANDLERS@
// ---- end of excerpt

    void process_pass()
    {
        double first_time_stamp = 0;
        if ( m_queue_avail_data.empty() ) return; // (the reference waits, :1697-1700)
// ---- verbatim: laser_mapping.hpp:1701-1733
@PROCESS@
// ---- end of excerpt
        fprintf( g_log, "TAKE %.17g %.17g %.17g %zu %zu %zu %016" PRIx64 " %016" PRIx64 " %016" PRIx64 "\n", current_data_pair->m_pc_corner->header.stamp.toSec(),
                 current_data_pair->m_pc_plane->header.stamp.toSec(), current_data_pair->m_pc_full->header.stamp.toSec(), m_laser_cloud_corner_last->points.size(),
                 m_laser_cloud_surf_last->points.size(), m_laser_cloud_full_res->points.size(), cloud_hash( *m_laser_cloud_corner_last ),
                 cloud_hash( *m_laser_cloud_surf_last ), cloud_hash( *m_laser_cloud_full_res ) );
// ---- verbatim: laser_mapping.hpp:1735
@DELETE@
// ---- end of excerpt
    }

    void publish_excerpt( pcl::PointCloud<PointType> &current_laser_cloud_full, double time_odom )
    {
// ---- verbatim: laser_mapping.hpp:1570-1575 (takes m_mutex_ros_pub)
@PUB_CLOUD@
// ---- verbatim: laser_mapping.hpp:1613-1653
@PUB_ODOM@
// ---- end of excerpt
        m_mutex_ros_pub.unlock();
    }
};

static bool rd( FILE *f, void *p, size_t n ) { return n == 0 || fread( p, 1, n, f ) == n; }
static void read_cloud( FILE *f, pcl::PointCloud<PointType> &c )
{
    int32_t n = 0;
    if ( !rd( f, &n, 4 ) || n < 0 ) { fprintf( stderr, "truncated io record\n" ); exit( 1 ); }
    c.points.assign( ( size_t ) n, PointType() );
    for ( int32_t i = 0; i < n; i++ )
    {
        float v[ 4 ];
        if ( !rd( f, v, 16 ) ) { fprintf( stderr, "truncated io record\n" ); exit( 1 ); }
        c.points[ i ].x = v[ 0 ]; c.points[ i ].y = v[ 1 ]; c.points[ i ].z = v[ 2 ]; c.points[ i ].intensity = v[ 3 ];
    }
}

int main( int argc, char **argv )
{
    if ( argc < 3 ) { fprintf( stderr, "usage: verbatim_mapping_io io.bin log.txt [maximum_mapping_buffer]\n" ); return 2; }
    FILE *fi = fopen( argv[ 1 ], "rb" );
    g_log = fopen( argv[ 2 ], "w" );
    char magic[ 8 ];
    if ( !fi || !g_log || !rd( fi, magic, 8 ) || memcmp( magic, "LLIO0001", 8 ) != 0 ) { fprintf( stderr, "cannot open / not an LLIO0001 file\n" ); return 1; }
    Laser_mapping_harness node;
    if ( argc > 3 ) node.m_max_buffer_size = atoi( argv[ 3 ] );
    for ( int type = fgetc( fi ); type != EOF; type = fgetc( fi ) )
    {
        if ( type == 'C' || type == 'S' || type == 'F' )
        {
            double stamp = 0;
            rd( fi, &stamp, 8 );
            pcl::PointCloud<PointType> c;
            read_cloud( fi, c );
            std::shared_ptr<sensor_msgs::PointCloud2> m( new sensor_msgs::PointCloud2() );
            pcl::toROSMsg( c, *m );
            m->header.stamp.fromSec( stamp );
            m->header.frame_id = "camera_init";
            if ( type == 'C' ) node.laserCloudCornerLastHandler( m );
            if ( type == 'S' ) node.laserCloudSurfLastHandler( m );
            if ( type == 'F' ) node.laserCloudFullResHandler( m );
        }
        else if ( type == 'P' )
        {
            node.process_pass();
        }
        else if ( type == 'U' )
        {
            int32_t frame_index = 0;
            double  time_odom = 0;
            rd( fi, &frame_index, 4 );
            rd( fi, &time_odom, 8 );
            rd( fi, node.m_para_buffer_RT, 56 );
            pcl::PointCloud<PointType> c;
            read_cloud( fi, c );
            node.m_current_frame_index = frame_index;
            node.publish_excerpt( c, time_odom );
        }
        else { fprintf( stderr, "unknown event\n" ); return 1; }
    }
    fclose( g_log );
    return 0;
}
'''


def build_mapping_io(force=False):
    """-> the harness executable (built where /root/reference exists; elsewhere whatever travelled with the tree, or None)"""
    if not have_reference():
        return EXE_MAPIO if os.path.exists(EXE_MAPIO) else None
    deps = [os.path.abspath(__file__)] + [os.path.join(dp, f) for dp, _, fs in os.walk(os.path.join(ROOT, "oracle", "ref_stubs")) for f in fs]
    if not force and os.path.exists(EXE_MAPIO) and all(os.path.getmtime(d) <= os.path.getmtime(EXE_MAPIO) for d in deps):
        return EXE_MAPIO
    os.makedirs(OUT, exist_ok=True)
    lm = "source/laser_mapping.hpp"
    tu = (MAPIO_HARNESS.replace("@DATA_PAIR@", _lines(lm, 89, 120)).replace("@GET_PAIR@", _lines(lm, 633, 647)).replace("@HANDLERS@", _lines(lm, 749, 780))
          .replace("@PROCESS@", _lines(lm, 1701, 1733)).replace("@DELETE@", _lines(lm, 1735, 1735)).replace("@PUB_CLOUD@", _lines(lm, 1570, 1575))
          .replace("@PUB_ODOM@", _lines(lm, 1613, 1653)))
    stubs = os.path.join(ROOT, "oracle", "ref_stubs")
    with tempfile.TemporaryDirectory() as td:
        src = os.path.join(td, "verbatim_mapping_io.cpp")
        with open(src, "w") as f:
            f.write(tu)
        inc = ["-I", os.path.join(stubs, "override"), "-I-", "-I", stubs, "-I", os.path.join(REF, "source"), "-I", os.path.join(REF, "include"),
               "-I", os.path.join(REF, "include", "tools")]
        subprocess.check_call(["g++", "-std=c++14", "-O1", "-w"] + inc + [src, "-o", EXE_MAPIO, "-lpthread"])
    return EXE_MAPIO
