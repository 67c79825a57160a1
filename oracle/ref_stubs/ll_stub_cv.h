/*
 * ll_stub_cv.h -- OUR minimal stand-in for the handful of OpenCV calls source/cell_map_keyframe.hpp makes
 * (Maps_keyframe::apply_guassian_blur :1360-1372, max_similiarity_of_two_image :1155-1229).  TEST INFRASTRUCTURE ONLY
 * (oracle/_ref build): OpenCV is absent from this image and is not vendored by the reference.
 *
 * Restated from OpenCV's documented behaviour (and therefore still "unpinned" third-party arithmetic):
 *   - cv::GaussianBlur( src, dst, Size( k, k ), sigma ) on CV_32F: separable convolution with getGaussianKernel( k, sigma, CV_32F )
 *     (exp( -( i - ( k - 1 ) / 2 )^2 / ( 2 sigma^2 ) ), normalised to sum 1), BORDER_REFLECT_101 at the image edges;
 *   - cv::matchTemplate( img, templ, result, TM_CCORR_NORMED / CV_TM_CCORR_NORMED ) for equal-sized or larger images:
 *     R( x, y ) = sum( T * I' ) / sqrt( sum( T^2 ) * sum( I'^2 ) ) over the template window;
 *   - cv::minMaxLoc; cv::eigen2cv / cv::cv2eigen for float matrices; cv::hconcat / vconcat / Mat::operator() ( Rect ).
 * Row-major float storage only.
 */
#ifndef LL_STUB_CV_H
#define LL_STUB_CV_H
#include <algorithm>
#include <cmath>
#include <vector>
#include <Eigen/Eigen>

#define CV_32F 5
#define CV_32FC1 5
#define CV_8UC1 0
#define CV_TM_CCORR_NORMED 3
#define CV_TM_CCOEFF_NORMED 5
#define CV_COMP_CORREL 0

namespace cv
{
enum { TM_SQDIFF = 0, TM_SQDIFF_NORMED = 1, TM_CCORR = 2, TM_CCORR_NORMED = 3, TM_CCOEFF = 4, TM_CCOEFF_NORMED = 5 };
enum { BORDER_DEFAULT = 4 };
struct Size
{
    int width, height;
    Size( int w = 0, int h = 0 ) : width( w ), height( h ) {}
};
struct Point
{
    int x, y;
    Point( int xx = 0, int yy = 0 ) : x( xx ), y( yy ) {}
};
struct Rect
{
    int x, y, width, height;
    Rect( int xx = 0, int yy = 0, int w = 0, int h = 0 ) : x( xx ), y( yy ), width( w ), height( h ) {}
};
class Mat
{
  public:
    int                rows = 0, cols = 0;
    std::vector<float> d;
    Mat() {}
    Mat( int r, int c, int /*type*/ ) : rows( r ), cols( c ), d( ( size_t ) r * c, 0.f ) {}
    void create( int r, int c, int /*type*/ )
    {
        rows = r;
        cols = c;
        d.assign( ( size_t ) r * c, 0.f );
    }
    float       &at( int r, int c ) { return d[ ( size_t ) r * cols + c ]; }
    const float &at( int r, int c ) const { return d[ ( size_t ) r * cols + c ]; }
    template <typename T> T       &at( int r, int c ) { return d[ ( size_t ) r * cols + c ]; }
    template <typename T> const T &at( int r, int c ) const { return d[ ( size_t ) r * cols + c ]; }
    Mat clone() const { return *this; }
    bool empty() const { return d.empty(); }
    Mat operator()( const Rect &r ) const
    {
        Mat o( r.height, r.width, CV_32F );
        for ( int i = 0; i < r.height; i++ )
            for ( int j = 0; j < r.width; j++ )
                o.at( i, j ) = at( r.y + i, r.x + j );
        return o;
    }
    void convertTo( Mat &o, int /*type*/, double alpha = 1.0, double beta = 0.0 ) const
    {
        Mat t = *this;
        for ( float &v : t.d ) v = ( float ) ( v * alpha + beta );
        o = t;
    }
};
typedef const Mat &InputArray;
typedef Mat       &OutputArray;

template <typename T, int R, int C> inline void eigen2cv( const Eigen::Matrix<T, R, C> &src, Mat &dst )
{
    dst.create( ( int ) src.rows(), ( int ) src.cols(), CV_32F );
    for ( int i = 0; i < ( int ) src.rows(); i++ )
        for ( int j = 0; j < ( int ) src.cols(); j++ )
            dst.at( i, j ) = ( float ) src( i, j );
}
template <typename T, int R, int C> inline void cv2eigen( const Mat &src, Eigen::Matrix<T, R, C> &dst )
{
    dst.resize( src.rows, src.cols );
    for ( int i = 0; i < src.rows; i++ )
        for ( int j = 0; j < src.cols; j++ )
            dst( i, j ) = ( T ) src.at( i, j );
}

inline int ll_reflect101( int p, int n )
{
    if ( n == 1 ) return 0;
    while ( p < 0 || p >= n )
    {
        if ( p < 0 ) p = -p;
        if ( p >= n ) p = 2 * ( n - 1 ) - p;
    }
    return p;
}
// getGaussianKernel( n, sigma, CV_32F ): float coefficients, normalised (the sum is accumulated in double, the scale applied in float)
inline std::vector<float> ll_gaussian_kernel( int n, double sigma )
{
    std::vector<float> k( n );
    if ( sigma <= 0 ) sigma = ( ( n - 1 ) * 0.5 - 1 ) * 0.3 + 0.8;
    const double scale2X = -0.5 / ( sigma * sigma );
    double       sum = 0;
    for ( int i = 0; i < n; i++ )
    {
        const double x = i - ( n - 1 ) * 0.5;
        k[ i ] = ( float ) std::exp( scale2X * x * x );
        sum += k[ i ];
    }
    sum = 1. / sum;
    for ( int i = 0; i < n; i++ ) k[ i ] = ( float ) ( k[ i ] * sum );
    return k;
}
inline void GaussianBlur( const Mat &src, Mat &dst, Size ksize, double sigmaX, double sigmaY = 0, int /*border*/ = BORDER_DEFAULT )
{
    if ( sigmaY <= 0 ) sigmaY = sigmaX;
    const std::vector<float> kx = ll_gaussian_kernel( ksize.width, sigmaX ), ky = ll_gaussian_kernel( ksize.height, sigmaY );
    Mat                      tmp( src.rows, src.cols, CV_32F ), out( src.rows, src.cols, CV_32F );
    const int                rx = ksize.width / 2, ry = ksize.height / 2;
    for ( int i = 0; i < src.rows; i++ )
        for ( int j = 0; j < src.cols; j++ )
        {
            float s = 0.f;
            for ( int k = 0; k < ksize.width; k++ ) s += kx[ k ] * src.at( i, ll_reflect101( j + k - rx, src.cols ) );
            tmp.at( i, j ) = s;
        }
    for ( int i = 0; i < src.rows; i++ )
        for ( int j = 0; j < src.cols; j++ )
        {
            float s = 0.f;
            for ( int k = 0; k < ksize.height; k++ ) s += ky[ k ] * tmp.at( ll_reflect101( i + k - ry, src.rows ), j );
            out.at( i, j ) = s;
        }
    dst = out;
}
// the normalised methods' division as OpenCV's matchTemplate documents / performs it: |num| < t -> num / t; within 12.5 % above t -> +-1
// (rounding); anything else, including a zero denominator (an all-zero window or template) -> 0
inline double normed( double num, double t )
{
    if ( std::fabs( num ) < t ) return num / t;
    if ( std::fabs( num ) < t * 1.125 ) return num > 0 ? 1.0 : -1.0;
    return 0.0;
}
inline void matchTemplate( const Mat &img, const Mat &templ, Mat &result, int method )
{
    const int rr = img.rows - templ.rows + 1, rc = img.cols - templ.cols + 1;
    result.create( rr, rc, CV_32F );
    double tt = 0, tsum = 0;
    for ( float v : templ.d ) tt += ( double ) v * v, tsum += v;
    const double n = ( double ) templ.rows * templ.cols;
    for ( int y = 0; y < rr; y++ )
        for ( int x = 0; x < rc; x++ )
        {
            double ti = 0, ii = 0, isum = 0;
            for ( int i = 0; i < templ.rows; i++ )
                for ( int j = 0; j < templ.cols; j++ )
                {
                    const double a = templ.at( i, j ), b = img.at( y + i, x + j );
                    ti += a * b;
                    ii += b * b;
                    isum += b;
                }
            double v;
            if ( method == TM_CCORR_NORMED )
                v = normed( ti, std::sqrt( tt * ii ) );
            else if ( method == TM_CCOEFF_NORMED )
                v = normed( ti - tsum * isum / n, std::sqrt( std::max( 0.0, ( tt - tsum * tsum / n ) * ( ii - isum * isum / n ) ) ) );
            else
                v = ti;
            result.at( y, x ) = ( float ) v;
        }
}
inline void minMaxLoc( const Mat &m, double *minVal, double *maxVal, Point *minLoc = nullptr, Point *maxLoc = nullptr, const Mat & = Mat() )
{
    double mn = 1e300, mx = -1e300;
    Point  pn, px;
    for ( int i = 0; i < m.rows; i++ )
        for ( int j = 0; j < m.cols; j++ )
        {
            const double v = m.at( i, j );
            if ( v < mn ) mn = v, pn = Point( j, i );
            if ( v > mx ) mx = v, px = Point( j, i );
        }
    if ( minVal ) *minVal = mn;
    if ( maxVal ) *maxVal = mx;
    if ( minLoc ) *minLoc = pn;
    if ( maxLoc ) *maxLoc = px;
}
inline void hconcat( const Mat &a, const Mat &b, Mat &o )
{
    Mat t( a.rows, a.cols + b.cols, CV_32F );
    for ( int i = 0; i < a.rows; i++ )
    {
        for ( int j = 0; j < a.cols; j++ ) t.at( i, j ) = a.at( i, j );
        for ( int j = 0; j < b.cols; j++ ) t.at( i, a.cols + j ) = b.at( i, j );
    }
    o = t;
}
inline void vconcat( const Mat &a, const Mat &b, Mat &o )
{
    Mat t( a.rows + b.rows, a.cols, CV_32F );
    for ( int j = 0; j < a.cols; j++ )
    {
        for ( int i = 0; i < a.rows; i++ ) t.at( i, j ) = a.at( i, j );
        for ( int i = 0; i < b.rows; i++ ) t.at( a.rows + i, j ) = b.at( i, j );
    }
    o = t;
}
inline double compareHist( const Mat &, const Mat &, int ) { return 0.0; }  // (only in a commented-out return)
} // namespace cv
#endif
