// oracle/ref_cells_shim.cpp -- TEST INFRASTRUCTURE: the reference's OWN cell map and key-frame classes
// (source/cell_map_keyframe.hpp, compiled verbatim from /root/reference: Points_cloud_cell, Points_cloud_map, Maps_keyframe) behind a
// small C interface, so that oracle/orc_cellmap.py (CPU tier) and the cm_* kernels (-m gpu, through fixtures) can be held to them.
// Third-party headers are the stand-ins of oracle/ref_stubs/ (Eigen, PCL point types + an exact-scan octree, the OpenCV calls, boost::format);
// rapidjson is the copy the reference vendors.  Built by `make -C oracle ref` into oracle/_ref/libll_ref_cells.so (git-ignored).
#include <cstdio>
#include <cstring>
#include <iostream>
#include <memory>
#include <set>
#include <vector>

#include "cell_map_keyframe.hpp"

typedef Points_cloud_map<float>                         Map_t;
typedef Maps_keyframe<float>                            Kf_t;
typedef Points_cloud_map<float>::Mapping_cell_ptr       Cell_ptr;
typedef Eigen::Matrix<float, 3, 1>                      Pt_t;

namespace
{
struct Quiet  // the reference chats on std::cout
{
    std::streambuf *old;
    Quiet() : old( std::cout.rdbuf( nullptr ) ) {}
    ~Quiet() { std::cout.rdbuf( old ); }
};
std::vector<Pt_t> to_pts( const float *xyz, int n )
{
    std::vector<Pt_t> v( n );
    for ( int i = 0; i < n; i++ ) v[ i ] = Pt_t( xyz[ 3 * i ], xyz[ 3 * i + 1 ], xyz[ 3 * i + 2 ] );
    return v;
}
Cell_ptr cell_at( Map_t *m, const float *c )
{
    auto it = m->m_map_pt_cell.find( Pt_t( c[ 0 ], c[ 1 ], c[ 2 ] ) );
    return it == m->m_map_pt_cell.end() ? nullptr : it->second;
}
} // namespace

extern "C" {

void *refc_map_create( float resolution, int revisit_threshold )
{
    Quiet  q;
    Map_t *m = new Map_t();
    m->set_resolution( resolution );                         // laser_mapping.hpp:616
    m->m_minimum_revisit_threshold = revisit_threshold;      // :617
    return m;
}
void refc_map_destroy( void *m ) { delete ( Map_t * ) m; }

// append_cloud( pts, &cell_vec ) (cell_map_keyframe.hpp:619-672): returns cell_vec's size, its cells' centres in `centres` (capacity cap)
int refc_map_append( void *mp, const float *xyz, int n, int want_cells, float *centres, int cap )
{
    Quiet              q;
    Map_t *            m = ( Map_t * ) mp;
    std::set<Cell_ptr> cell_vec;
    m->append_cloud( to_pts( xyz, n ), want_cells ? &cell_vec : nullptr );
    int k = 0;
    for ( auto &c : cell_vec )
    {
        if ( k < cap )
            for ( int d = 0; d < 3; d++ ) centres[ 3 * k + d ] = c->m_center( d );
        k++;
    }
    return k;
}
int refc_map_n_cells( void *mp ) { return ( ( Map_t * ) mp )->get_cells_size(); }
int refc_map_frame_idx( void *mp ) { return ( ( Map_t * ) mp )->m_current_frame_idx; }
// every cell: centre, point count, m_last_update_frame_idx
int refc_map_cells( void *mp, float *centres, int *counts, int *last_update, int cap )
{
    Map_t *m = ( Map_t * ) mp;
    int    k = 0;
    for ( auto &kv : m->m_map_pt_cell )
    {
        if ( k < cap )
        {
            for ( int d = 0; d < 3; d++ ) centres[ 3 * k + d ] = kv.second->m_center( d );
            counts[ k ] = ( int ) kv.second->m_points_vec.size();
            last_update[ k ] = kv.second->m_last_update_frame_idx;
        }
        k++;
    }
    return k;
}
int refc_map_cell_points( void *mp, const float *centre, float *xyz, int cap )
{
    Cell_ptr c = cell_at( ( Map_t * ) mp, centre );
    if ( !c ) return -1;
    const int n = ( int ) c->m_points_vec.size();
    for ( int i = 0; i < n && i < cap; i++ )
        for ( int d = 0; d < 3; d++ ) xyz[ 3 * i + d ] = c->m_points_vec[ i ]( d );
    return n;
}
// find_cells_in_radius (:761-788): centres of the cells whose CENTRE lies within `radius` of pt (order of the stand-in octree: insertion)
int refc_map_cells_in_radius( void *mp, const float *pt, float radius, float *centres, int cap )
{
    Quiet                 q;
    Map_t *               m = ( Map_t * ) mp;
    std::vector<Cell_ptr> v = m->find_cells_in_radius( Pt_t( pt[ 0 ], pt[ 1 ], pt[ 2 ] ), radius );
    int                   k = 0;
    for ( auto &c : v )
    {
        if ( k < cap )
            for ( int d = 0; d < 3; d++ ) centres[ 3 * k + d ] = c->m_center( d );
        k++;
    }
    return k;
}
// Points_cloud_cell::determine_feature( if_recompute = 1 ) (:436-473) of the cell at `centre`
int refc_cell_feature( void *mp, const float *centre, int *type, float *vec, float *mean, float *cov, float *eval, float *evec )
{
    Quiet    q;
    Cell_ptr c = cell_at( ( Map_t * ) mp, centre );
    if ( !c ) return -1;
    *type = ( int ) c->determine_feature( 1 );
    c->get_mean();  // (determine_feature returns before the moments for cells of fewer than 5 points; the mean is asked for on its own)
    for ( int d = 0; d < 3; d++ )
    {
        vec[ d ] = c->m_feature_vector( d );
        mean[ d ] = c->m_mean( d );
        eval[ d ] = c->m_eigen_val( d );
    }
    for ( int r = 0; r < 3; r++ )
        for ( int cc = 0; cc < 3; cc++ )
        {
            cov[ 3 * r + cc ] = c->m_cov_mat( r, cc );
            evec[ 3 * r + cc ] = c->m_eigen_vec( r, cc );
        }
    return ( int ) c->m_points_vec.size();
}

// ---- Maps_keyframe ------------------------------------------------------------------------------------------------------------------
void *refc_kf_create()
{
    Quiet q;
    return new Kf_t();
}
void refc_kf_destroy( void *k ) { delete ( Kf_t * ) k; }
// add_cells (:1243-1261) with the cells of `map` at the given centres; returns m_accumulate_frames
int refc_kf_add_cells( void *kp, void *mp, const float *centres, int n )
{
    Quiet              q;
    std::set<Cell_ptr> cells;
    for ( int i = 0; i < n; i++ )
    {
        Cell_ptr c = cell_at( ( Map_t * ) mp, centres + 3 * i );
        if ( c ) cells.insert( c );
    }
    ( ( Kf_t * ) kp )->add_cells( cells );
    return ( int ) ( ( Kf_t * ) kp )->m_accumulate_frames;
}
int refc_kf_n_cells( void *kp ) { return ( int ) ( ( Kf_t * ) kp )->m_set_cell.size(); }
// update_features_of_each_cells + analyze (:1231-1241, 1486-1493): images [4][60][60] = line, plane, line_roi, plane_roi as img( phi, theta );
// ratio [2] = m_ratio_nonzero_line, _plane (of the LAST generate_feature_img call: the whole key frame, :1483); R [2][9] row-major =
// m_eigen_R, m_eigen_R_roi are not kept by the reference (locals of generate_feature_img): not returned; n_vec [4]; roi_range
int refc_kf_analyze( void *kp, float *images, float *ratio, int *n_vec, float *roi_range )
{
    Quiet q;
    Kf_t *k = ( Kf_t * ) kp;
    k->update_features_of_each_cells( 1 );
    k->analyze( 1 );
    const Eigen::Matrix<float, Eigen::Dynamic, Eigen::Dynamic> *img[ 4 ] = { &k->m_feature_img_line, &k->m_feature_img_plane, &k->m_feature_img_line_roi,
                                                                               &k->m_feature_img_plane_roi };
    for ( int w = 0; w < 4; w++ )
        for ( int i = 0; i < 60; i++ )
            for ( int j = 0; j < 60; j++ ) images[ ( w * 60 + i ) * 60 + j ] = ( *img[ w ] )( i, j );
    ratio[ 0 ] = k->m_ratio_nonzero_line;
    ratio[ 1 ] = k->m_ratio_nonzero_plane;
    n_vec[ 0 ] = ( int ) k->m_feature_vecs_line.size();
    n_vec[ 1 ] = ( int ) k->m_feature_vecs_plane.size();
    n_vec[ 2 ] = ( int ) k->m_feature_vecs_line_roi.size();
    n_vec[ 3 ] = ( int ) k->m_feature_vecs_plane_roi.size();
    *roi_range = k->m_roi_range;
    return 0;
}
// get_center (:1291-1301) and the two eigen frames of the last analyze (m_eigen_R, m_eigen_R_roi; R( i, j ) row-major)
void refc_kf_frames( void *kp, float *centre, float *eigen_R )
{
    Quiet                       q;
    Kf_t                       *k = ( Kf_t * ) kp;
    Eigen::Matrix<float, 3, 1> c = k->get_center();
    for ( int d = 0; d < 3; d++ ) centre[ d ] = c( d );
    for ( int i = 0; i < 3; i++ )
        for ( int j = 0; j < 3; j++ )
        {
            eigen_R[ 3 * i + j ] = k->m_eigen_R( i, j );
            eigen_R[ 9 + 3 * i + j ] = k->m_eigen_R_roi( i, j );
        }
}
// max_similiarity_of_two_image (:1156-1229) of two 60 x 60 images given as img( i, j ) row-major
float refc_max_similarity( const float *a, const float *b )
{
    Quiet                                                q;
    Eigen::Matrix<float, Eigen::Dynamic, Eigen::Dynamic> A, B;
    A.resize( 60, 60 );
    B.resize( 60, 60 );
    for ( int i = 0; i < 60; i++ )
        for ( int j = 0; j < 60; j++ )
        {
            A( i, j ) = a[ i * 60 + j ];
            B( i, j ) = b[ i * 60 + j ];
        }
    return Kf_t::max_similiarity_of_two_image( A, B );
}

} // extern "C"
