/* ll_stub_misc.h -- empty stand-ins for the ROS / tf / OpenCV names that livox_feature_extractor.hpp mentions but the
 * hot path never executes (ros/ros.h, sensor_msgs/*, nav_msgs/*, tf/*, pcl_conversions/*; cv::Mat / cv::Scalar appear
 * only in the signature of the uninstantiated debug template Livox_laser::draw_dbg_img, LFE:308-320).
 * TEST INFRASTRUCTURE ONLY (oracle/_ref build). */
#ifndef LL_STUB_MISC_H
#define LL_STUB_MISC_H
namespace cv
{
struct Mat
{
    Mat clone() const { return *this; }
};
struct Scalar
{
    static Scalar all( double ) { return Scalar(); }
};
} // namespace cv
#endif
