/*
 * OUR stand-in for pcl::octree::OctreePointCloudSearch as source/cell_map_keyframe.hpp uses it (the octree of CELL CENTRES,
 * :612-613, 679, 709, 761-788, 1104-1123): setInputCloud / addPointsFromInputCloud / addPointToCloud / radiusSearch /
 * deleteTree.  TEST INFRASTRUCTURE ONLY (oracle/_ref build).  radiusSearch is an exact linear scan -- squared distance in float
 * like pcl::octree (pointSquaredDist), results in INPUT-CLOUD ORDER.  PCL returns them in its octree traversal order, which is
 * not reproducible here; callers of the stand-in compare SETS of cells (the repository defines its own order: ascending cell index).
 */
#pragma once
#include <memory>
#include <vector>
#include <pcl/point_cloud.h>
#include <pcl/point_types.h>
namespace pcl
{
namespace octree
{
template <typename PointT> class OctreePointCloudSearch
{
    typename pcl::PointCloud<PointT>::Ptr cloud_;
    std::vector<int>                      indexed_;  // points of cloud_ that are in the tree
    double                                resolution_;

  public:
    explicit OctreePointCloudSearch( double resolution = 1.0 ) : resolution_( resolution ) {}
    void   setResolution( double r ) { resolution_ = r; }
    double getResolution() const { return resolution_; }
    void   setInputCloud( const typename pcl::PointCloud<PointT>::Ptr &c ) { cloud_ = c; }
    typename pcl::PointCloud<PointT>::Ptr getInputCloud() const { return cloud_; }
    void addPointsFromInputCloud()
    {
        indexed_.clear();
        if ( cloud_ )
            for ( size_t i = 0; i < cloud_->points.size(); i++ ) indexed_.push_back( ( int ) i );
    }
    void addPointToCloud( const PointT &p, const typename pcl::PointCloud<PointT>::Ptr &c )
    {
        c->push_back( p );
        if ( !cloud_ ) cloud_ = c;
        indexed_.push_back( ( int ) c->points.size() - 1 );
    }
    void deleteTree() { indexed_.clear(); }
    int  radiusSearch( const PointT &p, double radius, std::vector<int> &idx, std::vector<float> &sqr, unsigned int max_nn = 0 ) const
    {
        idx.clear();
        sqr.clear();
        const double r2 = radius * radius;
        for ( int i : indexed_ )
        {
            const PointT &q = cloud_->points[ i ];
            const float   dx = q.x - p.x, dy = q.y - p.y, dz = q.z - p.z;
            const float   d2 = dx * dx + dy * dy + dz * dz;
            if ( d2 <= r2 )
            {
                idx.push_back( i );
                sqr.push_back( d2 );
            }
        }
        return ( int ) idx.size();
    }
};
} // namespace octree
} // namespace pcl
