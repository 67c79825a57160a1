/*
 * ref_shim.cpp -- C entry points around the REFERENCE's own classes, compiled verbatim from /root/reference
 * (source/livox_feature_extractor.hpp, source/ceres_icp.hpp, source/point_cloud_registration.hpp,
 * include/tools/*) against the stand-in third-party headers in oracle/ref_stubs/.  Output: oracle/_ref/libll_ref.so
 * (git-ignored; built by `make -C oracle ref` only where /root/reference exists).
 *
 * TEST INFRASTRUCTURE ONLY: used by tests/ to pin oracle/ (the C restatement) to the reference's own text.  No reference
 * source is copied into this repository; this file only calls the reference's public members.
 */
#include <cstdint>
#include <cstring>

#include "livox_feature_extractor.hpp"   // /root/reference/source
#include "point_cloud_registration.hpp"  // /root/reference/source (includes ceres_icp.hpp)

typedef pcl::PointCloud<PointType> Cloud;

static Cloud cloud_from( const float *xyzi, int n )
{
    Cloud c;
    c.points.resize( n );
    for ( int i = 0; i < n; i++ )
    {
        c.points[ i ].x = xyzi[ 4 * i + 0 ];
        c.points[ i ].y = xyzi[ 4 * i + 1 ];
        c.points[ i ].z = xyzi[ 4 * i + 2 ];
        c.points[ i ].intensity = xyzi[ 4 * i + 3 ];
    }
    return c;
}
static void cloud_to( const Cloud &c, float *xyzi )
{
    for ( size_t i = 0; i < c.points.size(); i++ )
    {
        xyzi[ 4 * i + 0 ] = c.points[ i ].x;
        xyzi[ 4 * i + 1 ] = c.points[ i ].y;
        xyzi[ 4 * i + 2 ] = c.points[ i ].z;
        xyzi[ 4 * i + 3 ] = c.points[ i ].intensity;
    }
}

struct RefFe
{
    Livox_laser        laser;
    std::vector<Cloud> petals;
    RefFe()
    {
        laser.m_if_verbose_screen_printf = 1; // silence screen_out
        laser.m_last_maximum_time_stamp = 0;  // LFE:152 leaves it uninitialised; defined as 0 (DESIGN, oracle)
    }
};

extern "C" {

// ------------------------------------------------------------------------------------------------ Livox_laser
void *ref_fe_create() { return new RefFe(); }
void  ref_fe_destroy( void *h ) { delete ( RefFe * ) h; }

// public tunables, LFE:143-167 (max_fov enters through the constructor's m_max_edge_polar_pos, LFE:185)
void ref_fe_set_params( void *h, float corner_curvature, float surface_curvature, float minimum_view_angle, float min_allow_dis, float min_sigma,
                        float time_internal_pts )
{
    Livox_laser &l = ( ( RefFe * ) h )->laser;
    l.thr_corner_curvature = corner_curvature;
    l.thr_surface_curvature = surface_curvature;
    l.minimum_view_angle = minimum_view_angle;
    l.m_livox_min_allow_dis = min_allow_dis;
    l.m_livox_min_sigma = min_sigma;
    l.m_time_internal_pts = time_internal_pts;
}
float ref_fe_max_edge_polar_pos( void *h ) { return ( ( RefFe * ) h )->laser.m_max_edge_polar_pos; }
double ref_fe_current_time( void *h ) { return ( ( RefFe * ) h )->laser.m_current_time; }
double ref_fe_first_receive_time( void *h ) { return ( ( RefFe * ) h )->laser.m_first_receive_time; }
double ref_fe_last_maximum_time_stamp( void *h ) { return ( ( RefFe * ) h )->laser.m_last_maximum_time_stamp; }

// extract_laser_features (LFE:722-766): returns the number of petal clouds the reference returns
int ref_fe_extract( void *h, const float *xyzi, int n, double stamp )
{
    RefFe *f = ( RefFe * ) h;
    Cloud  in = cloud_from( xyzi, n );
    f->petals = f->laser.extract_laser_features( in, stamp );
    return ( int ) f->petals.size();
}
int ref_fe_num_points( void *h ) { return ( int ) ( ( RefFe * ) h )->laser.m_pts_info_vec.size(); }

// m_pts_info_vec (LFE:118-133), one array per field; any pointer may be null
void ref_fe_pts_info( void *h, int32_t *pt_type, int32_t *pt_label, int32_t *idx, float *raw_intensity, float *time_stamp, float *polar_angle,
                      int32_t *polar_direction, float *polar_dis_sq2, float *depth_sq2, float *curvature, float *view_angle, float *sigma,
                      float *img2d )
{
    const std::vector<Livox_laser::Pt_infos> &v = ( ( RefFe * ) h )->laser.m_pts_info_vec;
    for ( size_t i = 0; i < v.size(); i++ )
    {
        if ( pt_type ) pt_type[ i ] = v[ i ].pt_type;
        if ( pt_label ) pt_label[ i ] = v[ i ].pt_label;
        if ( idx ) idx[ i ] = v[ i ].idx;
        if ( raw_intensity ) raw_intensity[ i ] = v[ i ].raw_intensity;
        if ( time_stamp ) time_stamp[ i ] = v[ i ].time_stamp;
        if ( polar_angle ) polar_angle[ i ] = v[ i ].polar_angle;
        if ( polar_direction ) polar_direction[ i ] = v[ i ].polar_direction;
        if ( polar_dis_sq2 ) polar_dis_sq2[ i ] = v[ i ].polar_dis_sq2;
        if ( depth_sq2 ) depth_sq2[ i ] = v[ i ].depth_sq2;
        if ( curvature ) curvature[ i ] = v[ i ].curvature;
        if ( view_angle ) view_angle[ i ] = v[ i ].view_angle;
        if ( sigma ) sigma[ i ] = v[ i ].sigma;
        if ( img2d )
        {
            img2d[ 2 * i + 0 ] = v[ i ].pt_2d_img( 0 );
            img2d[ 2 * i + 1 ] = v[ i ].pt_2d_img( 1 );
        }
    }
}

// get_features (LFE:219-272).  Clouds are written as xyzi (capacity n points each); the *_idx arrays receive, for every
// returned point, Livox_laser::find_pt_info(pt)->idx -- the look-up the reference's caller uses (LFX:321-322).
void ref_fe_get_features( void *h, float minimum_blur, float maximum_blur, float *corners, int32_t *corner_idx, int32_t *n_corner, float *surface,
                          int32_t *surf_idx, int32_t *n_surf, float *full, int32_t *n_full )
{
    RefFe *f = ( RefFe * ) h;
    Cloud  pc, ps, pf;
    f->laser.get_features( pc, ps, pf, minimum_blur, maximum_blur );
    *n_corner = ( int ) pc.size();
    *n_surf = ( int ) ps.size();
    *n_full = ( int ) pf.size();
    if ( corners ) cloud_to( pc, corners );
    if ( surface ) cloud_to( ps, surface );
    if ( full ) cloud_to( pf, full );
    if ( corner_idx )
        for ( size_t i = 0; i < pc.size(); i++ )
            corner_idx[ i ] = f->laser.find_pt_info( pc.points[ i ] )->idx;
    if ( surf_idx )
        for ( size_t i = 0; i < ps.size(); i++ )
            surf_idx[ i ] = f->laser.find_pt_info( ps.points[ i ] )->idx;
}

// petal clouds returned by extract_laser_features (split_laser_scan, LFE:657-719)
int ref_fe_petal_size( void *h, int k ) { return ( int ) ( ( RefFe * ) h )->petals[ k ].size(); }
void ref_fe_petal( void *h, int k, float *xyzi, int32_t *idx )
{
    RefFe *f = ( RefFe * ) h;
    cloud_to( f->petals[ k ], xyzi );
    if ( idx )
        for ( size_t i = 0; i < f->petals[ k ].size(); i++ )
            idx[ i ] = f->laser.find_pt_info( f->petals[ k ].points[ i ] )->idx;
}

// ------------------------------------------------------------------------------------------------ ceres_icp.hpp functors
// kind: 0 = ceres_icp_point2line (ICP:238-301), 1 = ceres_icp_point2plane (ICP:306-380),
//       2 = ceres_icp_point2line_mb (ICP:81-148), 3 = ceres_icp_point2plane_mb (ICP:152-233).
// pa, pb, pc: the neighbour points the registrar passes (PCR:300-301, 416-418); pc unused for lines.
// q_last_wxyz / t_last as PCR:315-316 passes them; x = m_para_buffer_incremental {qx,qy,qz,qw,tx,ty,tz}.
// Outputs: residual[3], jac_q[3x4] and jac_t[3x3] row-major (AutoDiffCostFunction<.,3,4,3>::Evaluate).
int ref_icp_evaluate( int kind, const double f[ 3 ], const double pa[ 3 ], const double pb[ 3 ], const double pc[ 3 ], double s,
                      const double q_last_wxyz[ 4 ], const double t_last[ 3 ], const double x[ 7 ], double residual[ 3 ], double jac_q[ 12 ],
                      double jac_t[ 9 ] )
{
    typedef Eigen::Matrix<double, 3, 1> V3;
    V3                          F( f[ 0 ], f[ 1 ], f[ 2 ] ), A( pa[ 0 ], pa[ 1 ], pa[ 2 ] ), B( pb[ 0 ], pb[ 1 ], pb[ 2 ] );
    V3                          Cc = pc ? V3( pc[ 0 ], pc[ 1 ], pc[ 2 ] ) : V3( 0, 0, 0 );
    Eigen::Matrix<double, 4, 1> QL( q_last_wxyz[ 0 ], q_last_wxyz[ 1 ], q_last_wxyz[ 2 ], q_last_wxyz[ 3 ] );
    V3                          TL( t_last[ 0 ], t_last[ 1 ], t_last[ 2 ] );
    ceres::CostFunction *       cf = nullptr;
    switch ( kind )
    {
    case 0: cf = ceres_icp_point2line<double>::Create( F, A, B, QL, TL ); break;
    case 1: cf = ceres_icp_point2plane<double>::Create( F, A, B, Cc, QL, TL ); break;
    case 2: cf = ceres_icp_point2line_mb<double>::Create( F, A, B, s, QL, TL ); break;
    case 3: cf = ceres_icp_point2plane_mb<double>::Create( F, A, B, Cc, s, QL, TL ); break;
    default: return 0;
    }
    const double *pp[ 2 ] = { x, x + 4 };
    double *      jj[ 2 ] = { jac_q, jac_t };
    bool          ok = cf->Evaluate( pp, residual, ( jac_q || jac_t ) ? jj : nullptr );
    delete cf;
    return ok ? 1 : 0;
}

// ------------------------------------------------------------------------------------------------ Point_cloud_registration
struct RefReg
{
    Point_cloud_registration    reg;
    Common_tools::File_logger   log_common, log_pcd, log_timer;
    Common_tools::Timer         timer;
    Cloud::Ptr                  map_corner, map_surf;
    pcl::KdTreeFLANN<PointType> kd_corner, kd_surf;
    RefReg() : map_corner( new Cloud ), map_surf( new Cloud )
    {
        reg.m_logger_common = &log_common;
        reg.m_logger_pcd = &log_pcd;
        reg.m_logger_timer = &log_timer;
        reg.m_timer = &timer;
        reg.m_if_verbose_screen_printf = 1;
        reg.reset_incremtal_parameter(); // PCR:58,66-68 are uninitialised in the reference; defined as zero (oracle, DESIGN)
        reg.m_interpolatation_omega.setZero();
    }
};

void *ref_reg_create() { return new RefReg(); }
void  ref_reg_destroy( void *h ) { delete ( RefReg * ) h; }

// the fields Laser_mapping::init_pointcloud_registration sets (LM:1266-1297)
void ref_reg_set_params( void *h, int if_motion_deblur, int icp_max_iterations, int ceres_max_iterations, int ceres_prerun_times,
                         int current_frame_index, int init_accumulate_frames, float max_angular_rate, float max_speed, float max_final_cost,
                         float min_time_stamp, float max_time_stamp, double min_icp_R_diff, double min_icp_T_diff, double inliner_dis,
                         double inlier_ratio, int maximum_allow_residual_block, int icp_line, int icp_plane, int line_check, int plane_check )
{
    Point_cloud_registration &r = ( ( RefReg * ) h )->reg;
    r.m_if_motion_deblur = if_motion_deblur;
    r.m_para_icp_max_iterations = icp_max_iterations;
    r.m_para_cere_max_iterations = ceres_max_iterations;
    r.m_para_cere_prerun_times = ceres_prerun_times;
    r.m_current_frame_index = current_frame_index;
    r.m_mapping_init_accumulate_frames = init_accumulate_frames;
    r.m_para_max_angular_rate = max_angular_rate;
    r.m_para_max_speed = max_speed;
    r.m_max_final_cost = max_final_cost;
    r.m_minimum_pt_time_stamp = min_time_stamp;
    r.m_maximum_pt_time_stamp = max_time_stamp;
    r.m_minimum_icp_R_diff = min_icp_R_diff;
    r.m_minimum_icp_T_diff = min_icp_T_diff;
    r.m_inliner_dis = inliner_dis;
    r.m_inlier_ratio = inlier_ratio;
    r.m_maximum_allow_residual_block = maximum_allow_residual_block;
    r.ICP_LINE = icp_line;
    r.ICP_PLANE = icp_plane;
    r.IF_LINE_FEATURE_CHECK = line_check;
    r.IF_PLANE_FEATURE_CHECK = plane_check;
}

void ref_reg_set_maps( void *h, const float *corner_xyz, int64_t n_corner, const float *surf_xyz, int64_t n_surf, int stride )
{
    RefReg *R = ( RefReg * ) h;
    R->map_corner->points.resize( n_corner );
    for ( int64_t i = 0; i < n_corner; i++ )
    {
        R->map_corner->points[ i ].x = corner_xyz[ stride * i + 0 ];
        R->map_corner->points[ i ].y = corner_xyz[ stride * i + 1 ];
        R->map_corner->points[ i ].z = corner_xyz[ stride * i + 2 ];
    }
    R->map_surf->points.resize( n_surf );
    for ( int64_t i = 0; i < n_surf; i++ )
    {
        R->map_surf->points[ i ].x = surf_xyz[ stride * i + 0 ];
        R->map_surf->points[ i ].y = surf_xyz[ stride * i + 1 ];
        R->map_surf->points[ i ].z = surf_xyz[ stride * i + 2 ];
    }
    R->kd_corner.setInputCloud( R->map_corner );
    R->kd_surf.setInputCloud( R->map_surf );
}

// pose arrays {qx,qy,qz,qw,tx,ty,tz}.  Sets m_q/t_w_last and m_q/t_w_curr as LM:1290-1294 does, resets the increment
// to identity (a fresh Point_cloud_registration per scan, LM:1348), runs the 6-argument
// find_out_incremental_transfrom (PCR:163-583) and returns its return value.
int ref_reg_solve( void *h, const float *scan_corner_xyzi, int n_corner, const float *scan_surf_xyzi, int n_surf, const double pose_last[ 7 ],
                   double pose_curr[ 7 ], double pose_incre[ 7 ], double report[ 8 ] )
{
    RefReg *                  R = ( RefReg * ) h;
    Point_cloud_registration &r = R->reg;
    r.m_q_w_last = Eigen::Quaterniond( pose_last[ 3 ], pose_last[ 0 ], pose_last[ 1 ], pose_last[ 2 ] );
    r.m_t_w_last = Eigen::Vector3d( pose_last[ 4 ], pose_last[ 5 ], pose_last[ 6 ] );
    r.m_q_w_curr = Eigen::Quaterniond( pose_curr[ 3 ], pose_curr[ 0 ], pose_curr[ 1 ], pose_curr[ 2 ] );
    r.m_t_w_curr = Eigen::Vector3d( pose_curr[ 4 ], pose_curr[ 5 ], pose_curr[ 6 ] );
    for ( int i = 0; i < 7; i++ )
        r.m_para_buffer_incremental[ i ] = pose_incre[ i ];
    Cloud::Ptr sc( new Cloud( cloud_from( scan_corner_xyzi, n_corner ) ) ), ss( new Cloud( cloud_from( scan_surf_xyzi, n_surf ) ) );
    int        ret = r.find_out_incremental_transfrom( R->map_corner, R->map_surf, R->kd_corner, R->kd_surf, sc, ss );
    pose_curr[ 0 ] = r.m_q_w_curr.x();
    pose_curr[ 1 ] = r.m_q_w_curr.y();
    pose_curr[ 2 ] = r.m_q_w_curr.z();
    pose_curr[ 3 ] = r.m_q_w_curr.w();
    pose_curr[ 4 ] = r.m_t_w_curr.x();
    pose_curr[ 5 ] = r.m_t_w_curr.y();
    pose_curr[ 6 ] = r.m_t_w_curr.z();
    for ( int i = 0; i < 7; i++ )
        pose_incre[ i ] = r.m_para_buffer_incremental[ i ];
    if ( report )
    {
        report[ 0 ] = r.summary.final_cost;
        report[ 1 ] = r.summary.initial_cost;
        report[ 2 ] = r.m_inlier_threshold;
        report[ 3 ] = r.m_angular_diff;
        report[ 4 ] = r.m_t_diff;
        report[ 5 ] = r.summary.num_residual_blocks;
        report[ 6 ] = r.summary.ll_iterations;
        report[ 7 ] = 0;
    }
    return ret;
}

// pointcloudAssociateToMap (PCR:673-685) with the registrar's current pose
void ref_reg_cloud_transform( void *h, const double pose[ 7 ], const float *in_xyzi, float *out_xyzi, int n )
{
    Point_cloud_registration &r = ( ( RefReg * ) h )->reg;
    r.m_q_w_curr = Eigen::Quaterniond( pose[ 3 ], pose[ 0 ], pose[ 1 ], pose[ 2 ] );
    r.m_t_w_curr = Eigen::Vector3d( pose[ 4 ], pose[ 5 ], pose[ 6 ] );
    Cloud in = cloud_from( in_xyzi, n ), out;
    r.pointcloudAssociateToMap( in, out, 0 );
    cloud_to( out, out_xyzi );
}

// refine_blur (PCR:128-141)
float ref_reg_refine_blur( void *h, int deblur, float in_blur, float min_blur, float max_blur )
{
    Point_cloud_registration &r = ( ( RefReg * ) h )->reg;
    int                       keep = r.m_if_motion_deblur;
    r.m_if_motion_deblur = deblur;
    float v = r.refine_blur( in_blur, min_blur, max_blur );
    r.m_if_motion_deblur = keep;
    return v;
}

// compute_inlier_residual_threshold (PCR:153-161)
double ref_reg_inlier_threshold( void *h, const double *residuals, int n, double ratio )
{
    std::vector<double> v( residuals, residuals + n );
    return ( ( RefReg * ) h )->reg.compute_inlier_residual_threshold( v, ratio );
}

// the line search's three-sample fit of the Ceres stand-in (ll_stub_ceres_solver.h quintic_min): test tap for the adversarial fits
double ref_quintic_min( double f0, double g0, double x1, double f1, double g1, double x2, double f2, double g2, double lo, double hi )
{
    return ceres::ll_solver::quintic_min( f0, g0, x1, f1, g1, x2, f2, g2, lo, hi );
}

} // extern "C"
