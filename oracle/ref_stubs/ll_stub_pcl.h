/*
 * ll_stub_pcl.h -- OUR minimal stand-in for the PCL types the reference's hot-path headers use
 * (pcl::PointXYZI, pcl::PointCloud, pcl::KdTreeFLANN::nearestKSearch / setInputCloud, PCD io names).
 * TEST INFRASTRUCTURE ONLY (oracle/_ref build); PCL is absent from this image and not vendored by the reference.
 *
 * KdTreeFLANN restates FLANN KDTreeSingleIndex + L2_Simple<float> as called at point_cloud_registration.hpp:249,351:
 * exact k-NN, squared distance accumulated in fp32 as ((dx*dx) + dy*dy) + dz*dz, results ascending; equal distances are
 * ordered by point index (FLANN's own tie order depends on its private tree layout).  Implemented as a median-split
 * k-d tree with exact pruning on the fp32 plane distance -- written for this stub, independent of oracle/ll_oracle_kdtree.c.
 */
#ifndef LL_STUB_PCL_H
#define LL_STUB_PCL_H
#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <memory>
#include <string>
#include <vector>

#define PCL_ERROR( ... ) fprintf( stderr, __VA_ARGS__ )

namespace pcl
{
struct PointXYZ
{
    float x = 0, y = 0, z = 0, data_w = 1.f;
    PointXYZ() {}
    PointXYZ( float xx, float yy, float zz ) : x( xx ), y( yy ), z( zz ) {}
};
struct PointXYZI
{
    float x = 0, y = 0, z = 0, data_w = 1.f; // PCL_ADD_POINT4D
    float intensity = 0;
    float pad_[ 3 ] = { 0, 0, 0 };
};
struct PointXYZRGBA
{
    float         x = 0, y = 0, z = 0, data_w = 1.f;
    unsigned char b = 0, g = 0, r = 0, a = 0;
    float         pad_[ 3 ] = { 0, 0, 0 };
};

template <typename PointT> class PointCloud
{
  public:
    typedef std::shared_ptr<PointCloud<PointT>>       Ptr;
    typedef std::shared_ptr<const PointCloud<PointT>> ConstPtr;
    std::vector<PointT> points;
    uint32_t            width = 0, height = 0;
    bool                is_dense = true;

    size_t        size() const { return points.size(); }
    bool          empty() const { return points.empty(); }
    void          resize( size_t n ) { points.resize( n ); }
    void          clear() { points.clear(); }
    void          reserve( size_t n ) { points.reserve( n ); }
    void          push_back( const PointT &p ) { points.push_back( p ); }
    PointT &      operator[]( size_t i ) { return points[ i ]; }
    const PointT &operator[]( size_t i ) const { return points[ i ]; }
    typename std::vector<PointT>::iterator       begin() { return points.begin(); }
    typename std::vector<PointT>::iterator       end() { return points.end(); }
    typename std::vector<PointT>::const_iterator begin() const { return points.begin(); }
    typename std::vector<PointT>::const_iterator end() const { return points.end(); }
    Ptr makeShared() const { return Ptr( new PointCloud<PointT>( *this ) ); }
    PointCloud &operator+=( const PointCloud &o )
    {
        points.insert( points.end(), o.points.begin(), o.points.end() );
        return *this;
    }
};

namespace io
{
template <typename C> int savePCDFileASCII( const std::string &, const C & ) { return 0; }
template <typename C> int savePCDFile( const std::string &, const C & ) { return 0; }
template <typename T> int loadPCDFile( const std::string &, PointCloud<T> & ) { return -1; }
} // namespace io

template <typename PointT> class KdTreeFLANN
{
    struct Node
    {
        int   lo, hi;     // point range in perm_ (leaf) / children
        int   left, right; // -1 for leaf
        int   dim;
        float split;
    };
    struct Tree
    {
        std::vector<float> xyz; // 3 per point, original order
        std::vector<int>   perm;
        std::vector<Node>  nodes;
    };
    std::shared_ptr<Tree> t_;

    static int build( Tree &t, int lo, int hi )
    {
        Node nd;
        nd.lo = lo;
        nd.hi = hi;
        nd.left = nd.right = -1;
        nd.dim = 0;
        nd.split = 0;
        const int id = ( int ) t.nodes.size();
        t.nodes.push_back( nd );
        if ( hi - lo > 12 )
        {
            float mn[ 3 ] = { INFINITY, INFINITY, INFINITY }, mx[ 3 ] = { -INFINITY, -INFINITY, -INFINITY };
            for ( int i = lo; i < hi; i++ )
                for ( int d = 0; d < 3; d++ )
                {
                    const float v = t.xyz[ 3 * ( size_t ) t.perm[ i ] + d ];
                    mn[ d ] = std::min( mn[ d ], v );
                    mx[ d ] = std::max( mx[ d ], v );
                }
            int dim = 0;
            for ( int d = 1; d < 3; d++ )
                if ( mx[ d ] - mn[ d ] > mx[ dim ] - mn[ dim ] )
                    dim = d;
            if ( mx[ dim ] > mn[ dim ] )
            {
                const int mid = ( lo + hi ) / 2;
                std::nth_element( t.perm.begin() + lo, t.perm.begin() + mid, t.perm.begin() + hi,
                                  [&]( int a, int b ) { return t.xyz[ 3 * ( size_t ) a + dim ] < t.xyz[ 3 * ( size_t ) b + dim ]; } );
                const float split = t.xyz[ 3 * ( size_t ) t.perm[ mid ] + dim ];
                const int   l = build( t, lo, mid );
                const int   r = build( t, mid, hi );
                t.nodes[ id ].left = l;
                t.nodes[ id ].right = r;
                t.nodes[ id ].dim = dim;
                t.nodes[ id ].split = split;
            }
        }
        return id;
    }

    struct Best
    {
        int   k, n;
        int   idx[ 16 ];
        float d2[ 16 ];
        bool  better( float d, int i, int slot ) const { return d < d2[ slot ] || ( d == d2[ slot ] && i < idx[ slot ] ); }
        void  push( float d, int i )
        {
            if ( !( d == d ) )
                return; // NaN distance never enters
            if ( n == k && !better( d, i, k - 1 ) )
                return;
            int pos = n < k ? n : k - 1;
            while ( pos > 0 && better( d, i, pos - 1 ) )
            {
                idx[ pos ] = idx[ pos - 1 ];
                d2[ pos ] = d2[ pos - 1 ];
                pos--;
            }
            idx[ pos ] = i;
            d2[ pos ] = d;
            if ( n < k )
                n++;
        }
    };

    void search( int id, const float q[ 3 ], Best &b ) const
    {
        const Node &nd = t_->nodes[ id ];
        if ( nd.left < 0 )
        {
            for ( int i = nd.lo; i < nd.hi; i++ )
            {
                const int    p = t_->perm[ i ];
                const float *v = &t_->xyz[ 3 * ( size_t ) p ];
                const float  dx = q[ 0 ] - v[ 0 ], dy = q[ 1 ] - v[ 1 ], dz = q[ 2 ] - v[ 2 ];
                float        d = dx * dx;
                d += dy * dy;
                d += dz * dz;
                b.push( d, p );
            }
            return;
        }
        const float diff = q[ nd.dim ] - nd.split;
        const int   near = diff < 0 ? nd.left : nd.right, far = diff < 0 ? nd.right : nd.left;
        search( near, q, b );
        // a point across the plane is at fp32 distance >= diff*diff (each squared term is non-negative and rounding is
        // monotone), so pruning on diff*diff > worst is exact; ties (==) must still be visited for the index order
        if ( b.n < b.k || !( diff * diff > b.d2[ b.k - 1 ] ) )
            search( far, q, b );
    }

  public:
    typedef std::shared_ptr<const PointCloud<PointT>> PointCloudConstPtr;
    KdTreeFLANN() {}
    void setInputCloud( const PointCloudConstPtr &cloud )
    {
        std::shared_ptr<Tree> t( new Tree );
        const size_t          m = cloud->points.size();
        t->xyz.resize( 3 * m );
        t->perm.resize( m );
        for ( size_t i = 0; i < m; i++ )
        {
            t->xyz[ 3 * i + 0 ] = cloud->points[ i ].x;
            t->xyz[ 3 * i + 1 ] = cloud->points[ i ].y;
            t->xyz[ 3 * i + 2 ] = cloud->points[ i ].z;
            t->perm[ i ] = ( int ) i;
        }
        if ( m )
            build( *t, 0, ( int ) m );
        t_ = t;
    }
    void setInputCloud( const std::shared_ptr<PointCloud<PointT>> &cloud ) { setInputCloud( PointCloudConstPtr( cloud ) ); }

    int nearestKSearch( const PointT &pt, int k, std::vector<int> &k_indices, std::vector<float> &k_sqr_distances ) const
    {
        k_indices.clear();
        k_sqr_distances.clear();
        if ( !t_ || t_->perm.empty() || k <= 0 || k > 16 )
            return 0;
        const float q[ 3 ] = { pt.x, pt.y, pt.z };
        if ( !std::isfinite( q[ 0 ] ) || !std::isfinite( q[ 1 ] ) || !std::isfinite( q[ 2 ] ) )
            return 0; // pcl::KdTreeFLANN asserts on a non-finite query; the stub returns "none found"
        Best b;
        b.k = k;
        b.n = 0;
        search( 0, q, b );
        k_indices.assign( b.idx, b.idx + b.n );
        k_sqr_distances.assign( b.d2, b.d2 + b.n );
        return b.n;
    }
};

// pcl::VoxelGrid<PointXYZI>::applyFilter as the reference calls it (laser_feature_extractor.hpp:372-381, laser_mapping.hpp:533-537,
// 1367-1373, 1434-1437), PCL 1.9 semantics with default settings -- the steps listed in oracle/ll_oracle_voxel.c, written for this
// stub: getMinMax3D over finite points, leaf index from float arithmetic, one centroid (x, y, z, intensity as float sums / count)
// per occupied leaf in ascending leaf order.  Defined like the oracle where PCL leaves it open: points of a leaf are added in input
// order, non-finite points are skipped, "leaf too small" copies the input.
template <typename PointT> class VoxelGrid
{
    float                                         leaf_[ 3 ] = { 0, 0, 0 };
    std::shared_ptr<const PointCloud<PointT>>     in_;

  public:
    void setLeafSize( float lx, float ly, float lz )
    {
        leaf_[ 0 ] = lx;
        leaf_[ 1 ] = ly;
        leaf_[ 2 ] = lz;
    }
    void setInputCloud( const std::shared_ptr<PointCloud<PointT>> &c ) { in_ = c; }
    void setInputCloud( const std::shared_ptr<const PointCloud<PointT>> &c ) { in_ = c; }
    void filter( PointCloud<PointT> &out )
    {
        const std::vector<PointT> src = in_ ? in_->points : std::vector<PointT>(); // (the node filters a cloud into itself)
        out.points.clear();
        float mn[ 3 ] = { 3.402823466e38f, 3.402823466e38f, 3.402823466e38f }, mx[ 3 ] = { -3.402823466e38f, -3.402823466e38f, -3.402823466e38f };
        size_t n_valid = 0;
        for ( const PointT &p : src )
        {
            if ( !std::isfinite( p.x ) || !std::isfinite( p.y ) || !std::isfinite( p.z ) )
                continue;
            const float v[ 3 ] = { p.x, p.y, p.z };
            for ( int c = 0; c < 3; c++ )
            {
                mn[ c ] = v[ c ] < mn[ c ] ? v[ c ] : mn[ c ];
                mx[ c ] = v[ c ] > mx[ c ] ? v[ c ] : mx[ c ];
            }
            n_valid++;
        }
        if ( n_valid == 0 )
            return;
        float   inv[ 3 ];
        int64_t d[ 3 ];
        bool    too_small = false;
        for ( int c = 0; c < 3; c++ )
        {
            inv[ c ] = 1.0f / leaf_[ c ];
            const float e = ( mx[ c ] - mn[ c ] ) * inv[ c ];
            if ( !( e < 9.0e18f ) )
            {
                too_small = true;
                d[ c ] = 0;
            }
            else
                d[ c ] = ( int64_t ) e + 1;
        }
        if ( too_small || d[ 0 ] > INT32_MAX || d[ 1 ] > INT32_MAX || d[ 0 ] * d[ 1 ] > INT32_MAX || d[ 2 ] > INT32_MAX || d[ 0 ] * d[ 1 ] * d[ 2 ] > INT32_MAX )
        {
            out.points = src; // "Leaf size is too small for the input dataset"
            return;
        }
        int32_t min_b[ 3 ], div_b[ 3 ], mul[ 3 ];
        for ( int c = 0; c < 3; c++ )
        {
            min_b[ c ] = ( int32_t ) std::floor( mn[ c ] * inv[ c ] );
            div_b[ c ] = ( int32_t ) std::floor( mx[ c ] * inv[ c ] ) - min_b[ c ] + 1;
        }
        mul[ 0 ] = 1;
        mul[ 1 ] = div_b[ 0 ];
        mul[ 2 ] = div_b[ 0 ] * div_b[ 1 ];
        std::vector<std::pair<uint32_t, int32_t>> keys;
        keys.reserve( n_valid );
        for ( size_t i = 0; i < src.size(); i++ )
        {
            const PointT &p = src[ i ];
            if ( !std::isfinite( p.x ) || !std::isfinite( p.y ) || !std::isfinite( p.z ) )
                continue;
            const int32_t i0 = ( int32_t )( std::floor( p.x * inv[ 0 ] ) - ( float ) min_b[ 0 ] );
            const int32_t i1 = ( int32_t )( std::floor( p.y * inv[ 1 ] ) - ( float ) min_b[ 1 ] );
            const int32_t i2 = ( int32_t )( std::floor( p.z * inv[ 2 ] ) - ( float ) min_b[ 2 ] );
            keys.push_back( std::make_pair( ( uint32_t )( i0 * mul[ 0 ] + i1 * mul[ 1 ] + i2 * mul[ 2 ] ), ( int32_t ) i ) );
        }
        std::sort( keys.begin(), keys.end() ); // (leaf, input index): input order inside a leaf
        size_t k = 0;
        while ( k < keys.size() )
        {
            size_t e = k;
            float  sx = 0.f, sy = 0.f, sz = 0.f, si = 0.f;
            while ( e < keys.size() && keys[ e ].first == keys[ k ].first )
            {
                const PointT &p = src[ keys[ e ].second ];
                sx = sx + p.x;
                sy = sy + p.y;
                sz = sz + p.z;
                si = si + p.intensity;
                e++;
            }
            const float cnt = ( float ) ( e - k );
            PointT      o;
            o.x = sx / cnt;
            o.y = sy / cnt;
            o.z = sz / cnt;
            o.intensity = si / cnt;
            out.points.push_back( o );
            k = e;
        }
    }
};
template <typename PointT> class StatisticalOutlierRemoval
{
};
} // namespace pcl
#endif
