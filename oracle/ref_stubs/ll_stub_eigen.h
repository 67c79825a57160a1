/*
 * ll_stub_eigen.h -- OUR minimal stand-in for the parts of Eigen3 that the reference's hot-path headers use
 * (livox_feature_extractor.hpp, tools_eigen_math.hpp, pcl_tools.hpp, ceres_icp.hpp, point_cloud_registration.hpp).
 *
 * TEST INFRASTRUCTURE ONLY (oracle/_ref build, see oracle/Makefile target `ref`).  Eigen3 itself is absent from this
 * image and is not vendored by the reference, so the reference sources are compiled VERBATIM from /root/reference
 * against this header.  What is restated here from Eigen 3.3's documented behaviour (and is therefore still
 * "unpinned" third-party arithmetic):
 *   - fixed-size reductions: a 3-vector of float or of a non-vectorisable scalar (ceres::Jet) reduces as
 *     e0 + (e1 + e2) (redux_novec_unroller); a 3-vector of double reduces as (e0 + e1) + e2 (one SSE2 packet + tail);
 *   - QuaternionBase::_transformVector, operator*, slerp, angularDistance, toRotationMatrix; AngleAxis(quaternion);
 *   - SelfAdjointEigenSolver<Matrix3d>: eigenvalues ascending (cyclic Jacobi here; Eigen uses a closed form + QR).
 * Everything is evaluated eagerly, element by element, in the order Eigen's lazy expressions evaluate coefficients.
 */
#ifndef LL_STUB_EIGEN_H
#define LL_STUB_EIGEN_H
#include <cmath>
#include <cstddef>
#include <limits>
#include <ostream>
#include <type_traits>

namespace Eigen
{
namespace ll_detail
{
template <typename T> struct scalar_info
{
    static double value( const T &x ) { return ( double ) x.a; } // ceres::Jet-like: scalar part .a
    enum { is_double = 0 };
};
template <> struct scalar_info<float>
{
    static double value( float x ) { return x; }
    enum { is_double = 0 };
};
template <> struct scalar_info<double>
{
    static double value( double x ) { return x; }
    enum { is_double = 1 };
};
// redux order of a 3-element sum (see header comment)
template <typename T> inline T sum3( const T &e0, const T &e1, const T &e2 )
{
    if ( scalar_info<T>::is_double )
        return ( e0 + e1 ) + e2;
    return e0 + ( e1 + e2 );
}
// 4 elements: float -> one packet: (e0+e2)+(e1+e3) [predux of a Packet4f]; double -> two packets added, then predux:
// (e0+e2)+(e1+e3); generic scalar -> (e0+e1)+(e2+e3)
template <typename T> inline T sum4( const T &e0, const T &e1, const T &e2, const T &e3 )
{
    if ( std::is_arithmetic<T>::value )
        return ( e0 + e2 ) + ( e1 + e3 );
    return ( e0 + e1 ) + ( e2 + e3 );
}
} // namespace ll_detail

template <typename T, int R, int C> class Matrix;
const int Dynamic = -1;  // (cell_map_keyframe.hpp's direction images: the specialisation lives in ll_stub_eigen_dyn.h)
#define EIGEN_MAKE_ALIGNED_OPERATOR_NEW
template <typename T, int R, int C> class ColRef;
template <typename T, int R, int C> struct ReverseHelper;

template <typename T, int R, int C> class CommaInit
{
    Matrix<T, R, C> &m;
    int              k;

  public:
    CommaInit( Matrix<T, R, C> &mm, const T &first ) : m( mm ), k( 0 ) { put( first ); }
    void put( const T &v )
    {
        // row-major fill order, like Eigen's CommaInitializer
        int r = k / C, c = k % C;
        m( r, c ) = v;
        k++;
    }
    CommaInit &operator,( const T &v )
    {
        put( v );
        return *this;
    }
};

template <typename T, int R, int C> class Matrix
{
  public:
    typedef T Scalar;
    T         d[ R * C ]; // column-major

    Matrix()
    {
        for ( int i = 0; i < R * C; i++ )
            d[ i ] = T();
    }
    Matrix( const T &x, const T &y )
    {
        static_assert( R * C == 2, "2-vector ctor" );
        d[ 0 ] = x;
        d[ 1 ] = y;
    }
    Matrix( const T &x, const T &y, const T &z )
    {
        static_assert( R * C == 3, "3-vector ctor" );
        d[ 0 ] = x;
        d[ 1 ] = y;
        d[ 2 ] = z;
    }
    Matrix( const T &x, const T &y, const T &z, const T &w )
    {
        static_assert( R * C == 4, "4-vector ctor" );
        d[ 0 ] = x;
        d[ 1 ] = y;
        d[ 2 ] = z;
        d[ 3 ] = w;
    }

    static Matrix Zero()
    {
        Matrix m;
        for ( int i = 0; i < R * C; i++ )
            m.d[ i ] = T( 0 );
        return m;
    }
    static Matrix Identity()
    {
        Matrix m = Zero();
        for ( int i = 0; i < ( R < C ? R : C ); i++ )
            m( i, i ) = T( 1 );
        return m;
    }
    void setZero() { *this = Zero(); }
    void setIdentity() { *this = Identity(); }

    int rows() const { return R; }
    int cols() const { return C; }
    int size() const { return R * C; }

    T &      operator()( int i ) { return d[ i ]; }
    const T &operator()( int i ) const { return d[ i ]; }
    T &      operator[]( int i ) { return d[ i ]; }
    const T &operator[]( int i ) const { return d[ i ]; }
    T &      operator()( int r, int c ) { return d[ r + c * R ]; }
    const T &operator()( int r, int c ) const { return d[ r + c * R ]; }
    T &      x() { return d[ 0 ]; }
    T &      y() { return d[ 1 ]; }
    T &      z() { return d[ 2 ]; }
    const T &x() const { return d[ 0 ]; }
    const T &y() const { return d[ 1 ]; }
    const T &z() const { return d[ 2 ]; }
    T *      data() { return d; }
    const T *data() const { return d; }

    CommaInit<T, R, C> operator<<( const T &v ) { return CommaInit<T, R, C>( *this, v ); }

    template <typename U> Matrix<U, R, C> cast() const
    {
        Matrix<U, R, C> o;
        for ( int i = 0; i < R * C; i++ )
            o.d[ i ] = U( d[ i ] );
        return o;
    }

    Matrix<T, C, R> transpose() const
    {
        Matrix<T, C, R> o;
        for ( int r = 0; r < R; r++ )
            for ( int c = 0; c < C; c++ )
                o( c, r ) = ( *this )( r, c );
        return o;
    }

    T dot( const Matrix &o ) const
    {
        static_assert( R * C == 3 || R * C == 4 || R * C == 2, "dot: 2/3/4-vectors only" );
        if ( R * C == 2 )
            return d[ 0 ] * o.d[ 0 ] + d[ 1 ] * o.d[ 1 ];
        if ( R * C == 3 )
            return ll_detail::sum3<T>( d[ 0 ] * o.d[ 0 ], d[ 1 ] * o.d[ 1 ], d[ 2 ] * o.d[ 2 ] );
        return ll_detail::sum4<T>( d[ 0 ] * o.d[ 0 ], d[ 1 ] * o.d[ 1 ], d[ 2 ] * o.d[ 2 ], d[ 3 % ( R * C ) ] * o.d[ 3 % ( R * C ) ] );
    }
    T squaredNorm() const { return dot( *this ); }
    T norm() const
    {
        using std::sqrt;
        return sqrt( squaredNorm() );
    }
    T stableNorm() const { return norm(); }
    Matrix normalized() const { return *this / norm(); }

    Matrix cross( const Matrix &o ) const
    {
        static_assert( R * C == 3, "cross: 3-vectors only" );
        return Matrix( d[ 1 ] * o.d[ 2 ] - d[ 2 ] * o.d[ 1 ], d[ 2 ] * o.d[ 0 ] - d[ 0 ] * o.d[ 2 ], d[ 0 ] * o.d[ 1 ] - d[ 1 ] * o.d[ 0 ] );
    }

    // Euler angles are only printed to log files by the reference (PCR:548-549); not part of any compared output.
    Matrix<T, 3, 1> eulerAngles( int, int, int ) const { return Matrix<T, 3, 1>(); }

    Matrix &operator+=( const Matrix &o )
    {
        for ( int i = 0; i < R * C; i++ )
            d[ i ] = d[ i ] + o.d[ i ];
        return *this;
    }
    Matrix &operator-=( const Matrix &o )
    {
        for ( int i = 0; i < R * C; i++ )
            d[ i ] = d[ i ] - o.d[ i ];
        return *this;
    }

    friend Matrix operator+( const Matrix &a, const Matrix &b )
    {
        Matrix o;
        for ( int i = 0; i < R * C; i++ )
            o.d[ i ] = a.d[ i ] + b.d[ i ];
        return o;
    }
    friend Matrix operator-( const Matrix &a, const Matrix &b )
    {
        Matrix o;
        for ( int i = 0; i < R * C; i++ )
            o.d[ i ] = a.d[ i ] - b.d[ i ];
        return o;
    }
    friend Matrix operator-( const Matrix &a )
    {
        Matrix o;
        for ( int i = 0; i < R * C; i++ )
            o.d[ i ] = -a.d[ i ];
        return o;
    }
    friend Matrix operator*( const Matrix &a, const T &s )
    {
        Matrix o;
        for ( int i = 0; i < R * C; i++ )
            o.d[ i ] = a.d[ i ] * s;
        return o;
    }
    friend Matrix operator*( const T &s, const Matrix &a )
    {
        Matrix o;
        for ( int i = 0; i < R * C; i++ )
            o.d[ i ] = s * a.d[ i ];
        return o;
    }
    friend Matrix operator/( const Matrix &a, const T &s )
    {
        Matrix o;
        for ( int i = 0; i < R * C; i++ )
            o.d[ i ] = a.d[ i ] / s;
        return o;
    }
    friend std::ostream &operator<<( std::ostream &os, const Matrix &m )
    {
        for ( int r = 0; r < R; r++ )
        {
            for ( int c = 0; c < C; c++ )
                os << ( c ? " " : "" ) << ll_detail::scalar_info<T>::value( m( r, c ) );
            if ( r + 1 < R )
                os << "\n";
        }
        return os;
    }

    // ---- what source/cell_map_keyframe.hpp needs on top (Points_cloud_cell's moments, Maps_keyframe's eigen frames) ----------
    template <typename S> Matrix &operator/=( const S &s )
    {
        for ( int i = 0; i < R * C; i++ )
            d[ i ] = d[ i ] / T( s );  // (Eigen converts the scalar to the matrix's scalar type first)
        return *this;
    }
    template <typename S> Matrix &operator*=( const S &s )
    {
        for ( int i = 0; i < R * C; i++ )
            d[ i ] = d[ i ] * T( s );
        return *this;
    }
    Matrix<T, R, R> asDiagonal() const
    {
        static_assert( C == 1, "asDiagonal of a vector" );
        Matrix<T, R, R> o;
        for ( int i = 0; i < R; i++ )
            o( i, i ) = d[ i ];
        return o;
    }
    // 3 x 3 inverse by cofactors (Eigen: compute_inverse_size3_helper -- cofactors over the determinant)
    Matrix inverse() const
    {
        static_assert( R == 3 && C == 3, "inverse: 3 x 3 only" );
        const Matrix &m = *this;
        Matrix        o;
        const T       c00 = m( 1, 1 ) * m( 2, 2 ) - m( 1, 2 ) * m( 2, 1 ), c10 = m( 1, 2 ) * m( 2, 0 ) - m( 1, 0 ) * m( 2, 2 ),
                c20 = m( 1, 0 ) * m( 2, 1 ) - m( 1, 1 ) * m( 2, 0 );
        const T det = ll_detail::sum3<T>( m( 0, 0 ) * c00, m( 0, 1 ) * c10, m( 0, 2 ) * c20 );
        const T inv = T( 1 ) / det;
        o( 0, 0 ) = c00 * inv;
        o( 1, 0 ) = c10 * inv;
        o( 2, 0 ) = c20 * inv;
        o( 0, 1 ) = ( m( 0, 2 ) * m( 2, 1 ) - m( 0, 1 ) * m( 2, 2 ) ) * inv;
        o( 1, 1 ) = ( m( 0, 0 ) * m( 2, 2 ) - m( 0, 2 ) * m( 2, 0 ) ) * inv;
        o( 2, 1 ) = ( m( 0, 1 ) * m( 2, 0 ) - m( 0, 0 ) * m( 2, 1 ) ) * inv;
        o( 0, 2 ) = ( m( 0, 1 ) * m( 1, 2 ) - m( 0, 2 ) * m( 1, 1 ) ) * inv;
        o( 1, 2 ) = ( m( 0, 2 ) * m( 1, 0 ) - m( 0, 0 ) * m( 1, 2 ) ) * inv;
        o( 2, 2 ) = ( m( 0, 0 ) * m( 1, 1 ) - m( 0, 1 ) * m( 1, 0 ) ) * inv;
        return o;
    }
    template <int BR, int BC> Matrix<T, BR, BC> block( int i, int j ) const
    {
        Matrix<T, BR, BC> o;
        for ( int r = 0; r < BR; r++ )
            for ( int c = 0; c < BC; c++ )
                o( r, c ) = ( *this )( i + r, j + c );
        return o;
    }
    Matrix<T, R, 1> col( int j ) const
    {
        Matrix<T, R, 1> o;
        for ( int r = 0; r < R; r++ )
            o( r ) = ( *this )( r, j );
        return o;
    }
    ColRef<T, R, C> col( int j ) { return ColRef<T, R, C>( *this, j ); }
    ReverseHelper<T, R, C> rowwise() const { return ReverseHelper<T, R, C>( *this, 1 ); }  // .reverse(): every row reversed
    ReverseHelper<T, R, C> colwise() const { return ReverseHelper<T, R, C>( *this, 0 ); }  // .reverse(): every column reversed
    Matrix eval() const { return *this; }
    T maxCoeff() const
    {
        T m = d[ 0 ];
        for ( int i = 1; i < R * C; i++ )
            if ( d[ i ] > m ) m = d[ i ];
        return m;
    }
};

// lvalue column of a fixed-size matrix: m.col( 2 ) = m.col( 0 ).cross( m.col( 1 ) )
template <typename T, int R, int C> class ColRef
{
    Matrix<T, R, C> &m;
    int              j;

  public:
    ColRef( Matrix<T, R, C> &mm, int jj ) : m( mm ), j( jj ) {}
    operator Matrix<T, R, 1>() const { return const_cast<const Matrix<T, R, C> &>( m ).col( j ); }
    ColRef &operator=( const Matrix<T, R, 1> &v )
    {
        for ( int r = 0; r < R; r++ )
            m( r, j ) = v( r );
        return *this;
    }
    Matrix<T, R, 1> cross( const Matrix<T, R, 1> &o ) const { return Matrix<T, R, 1>( *this ).cross( o ); }
    Matrix<T, R, 1> cross( const ColRef &o ) const { return Matrix<T, R, 1>( *this ).cross( Matrix<T, R, 1>( o ) ); }
};
template <typename T, int R, int C> struct ReverseHelper
{
    Matrix<T, R, C> m;
    int             rowwise;
    ReverseHelper( const Matrix<T, R, C> &mm, int rw ) : m( mm ), rowwise( rw ) {}
    Matrix<T, R, C> reverse() const
    {
        Matrix<T, R, C> o;
        for ( int r = 0; r < R; r++ )
            for ( int c = 0; c < C; c++ )
                o( r, c ) = rowwise ? m( r, C - 1 - c ) : m( R - 1 - r, c );
        return o;
    }
};

// matrix product (3x3 * 3x1, 3x3 * 3x3, 3x1 * 1x3): coefficient (r,c) = sum_k a(r,k) b(k,c), k ascending, reduced like a
// K-element lazy product (K == 3 -> sum3; K == 1 -> single product)
template <typename T, int R, int K, int C> Matrix<T, R, C> operator*( const Matrix<T, R, K> &a, const Matrix<T, K, C> &b )
{
    Matrix<T, R, C> o;
    for ( int r = 0; r < R; r++ )
        for ( int c = 0; c < C; c++ )
        {
            if ( K == 3 )
                o( r, c ) = ll_detail::sum3<T>( a( r, 0 ) * b( 0, c ), a( r, 1 % K ) * b( 1 % K, c ), a( r, 2 % K ) * b( 2 % K, c ) );
            else
            {
                T s = a( r, 0 ) * b( 0, c );
                for ( int k = 1; k < K; k++ )
                    s = s + a( r, k ) * b( k, c );
                o( r, c ) = s;
            }
        }
    return o;
}

typedef Matrix<double, 3, 1> Vector3d;
typedef Matrix<float, 3, 1>  Vector3f;
typedef Matrix<double, 3, 3> Matrix3d;
typedef Matrix<float, 3, 3>  Matrix3f;
typedef Matrix<double, 4, 1> Vector4d;

template <typename T> class Quaternion
{
  public:
    typedef T Scalar;
    T         c[ 4 ]; // x, y, z, w  (Eigen coefficient order)

    Quaternion()
    {
        c[ 0 ] = c[ 1 ] = c[ 2 ] = T( 0 );
        c[ 3 ] = T( 1 );
    }
    Quaternion( const T &w, const T &x, const T &y, const T &z )
    {
        c[ 0 ] = x;
        c[ 1 ] = y;
        c[ 2 ] = z;
        c[ 3 ] = w;
    }
    explicit Quaternion( const Matrix<T, 4, 1> &coeffs_xyzw )
    {
        for ( int i = 0; i < 4; i++ )
            c[ i ] = coeffs_xyzw( i );
    }
    explicit Quaternion( const T *xyzw )
    {
        for ( int i = 0; i < 4; i++ )
            c[ i ] = xyzw[ i ];
    }
    static Quaternion Identity() { return Quaternion( T( 1 ), T( 0 ), T( 0 ), T( 0 ) ); }
    void              setIdentity() { *this = Identity(); }

    T &      x() { return c[ 0 ]; }
    T &      y() { return c[ 1 ]; }
    T &      z() { return c[ 2 ]; }
    T &      w() { return c[ 3 ]; }
    const T &x() const { return c[ 0 ]; }
    const T &y() const { return c[ 1 ]; }
    const T &z() const { return c[ 2 ]; }
    const T &w() const { return c[ 3 ]; }
    Matrix<T, 3, 1> vec() const { return Matrix<T, 3, 1>( c[ 0 ], c[ 1 ], c[ 2 ] ); }
    Matrix<T, 4, 1> coeffs() const { return Matrix<T, 4, 1>( c[ 0 ], c[ 1 ], c[ 2 ], c[ 3 ] ); }

    template <typename U> Quaternion<U> cast() const { return Quaternion<U>( U( c[ 3 ] ), U( c[ 0 ] ), U( c[ 1 ] ), U( c[ 2 ] ) ); }

    Quaternion conjugate() const { return Quaternion( c[ 3 ], -c[ 0 ], -c[ 1 ], -c[ 2 ] ); }
    T          squaredNorm() const { return coeffs().squaredNorm(); }
    T          norm() const { return coeffs().norm(); }
    void       normalize()
    {
        T n = norm();
        for ( int i = 0; i < 4; i++ )
            c[ i ] = c[ i ] / n;
    }
    Quaternion normalized() const
    {
        Quaternion q = *this;
        q.normalize();
        return q;
    }
    Quaternion inverse() const
    {
        // QuaternionBase::inverse: conjugate / squaredNorm (no unit-norm assumption)
        T          n2 = squaredNorm();
        Quaternion q = conjugate();
        for ( int i = 0; i < 4; i++ )
            q.c[ i ] = q.c[ i ] / n2;
        return q;
    }
    T dot( const Quaternion &o ) const { return coeffs().dot( o.coeffs() ); }

    // QuaternionBase::slerp (Eigen 3.3)
    Quaternion slerp( const T &t, const Quaternion &other ) const
    {
        using std::acos;
        using std::sin;
        const double one = 1.0 - std::numeric_limits<double>::epsilon();
        T            d = this->dot( other );
        T            absD = ll_detail::scalar_info<T>::value( d ) < 0 ? -d : d;
        T            scale0, scale1;
        if ( ll_detail::scalar_info<T>::value( absD ) >= one )
        {
            scale0 = T( 1 ) - t;
            scale1 = t;
        }
        else
        {
            T theta = acos( absD );
            T sinTheta = sin( theta );
            scale0 = sin( ( T( 1 ) - t ) * theta ) / sinTheta;
            scale1 = sin( ( t * theta ) ) / sinTheta;
        }
        if ( ll_detail::scalar_info<T>::value( d ) < 0 )
            scale1 = -scale1;
        Quaternion r;
        for ( int i = 0; i < 4; i++ )
            r.c[ i ] = scale0 * c[ i ] + scale1 * other.c[ i ];
        return r;
    }

    // QuaternionBase::angularDistance (Eigen 3.3): 2 atan2(|vec(d)|, |w(d)|), d = this * conj(other)
    T angularDistance( const Quaternion &other ) const
    {
        using std::atan2;
        Quaternion dq = ( *this ) * other.conjugate();
        T          aw = ll_detail::scalar_info<T>::value( dq.w() ) < 0 ? -dq.w() : dq.w();
        return T( 2 ) * atan2( dq.vec().norm(), aw );
    }

    Matrix<T, 3, 3> toRotationMatrix() const
    {
        Matrix<T, 3, 3> res;
        const T         tx = T( 2 ) * x(), ty = T( 2 ) * y(), tz = T( 2 ) * z();
        const T         twx = tx * w(), twy = ty * w(), twz = tz * w();
        const T         txx = tx * x(), txy = ty * x(), txz = tz * x();
        const T         tyy = ty * y(), tyz = tz * y(), tzz = tz * z();
        res( 0, 0 ) = T( 1 ) - ( tyy + tzz );
        res( 0, 1 ) = txy - twz;
        res( 0, 2 ) = txz + twy;
        res( 1, 0 ) = txy + twz;
        res( 1, 1 ) = T( 1 ) - ( txx + tzz );
        res( 1, 2 ) = tyz - twx;
        res( 2, 0 ) = txz - twy;
        res( 2, 1 ) = tyz + twx;
        res( 2, 2 ) = T( 1 ) - ( txx + tyy );
        return res;
    }

    // quaternion product (internal::quat_product, generic form)
    friend Quaternion operator*( const Quaternion &a, const Quaternion &b )
    {
        return Quaternion( a.w() * b.w() - a.x() * b.x() - a.y() * b.y() - a.z() * b.z(),
                           a.w() * b.x() + a.x() * b.w() + a.y() * b.z() - a.z() * b.y(),
                           a.w() * b.y() + a.y() * b.w() + a.z() * b.x() - a.x() * b.z(),
                           a.w() * b.z() + a.z() * b.w() + a.x() * b.y() - a.y() * b.x() );
    }
    // QuaternionBase::_transformVector: uv = 2 (vec x v); v + w uv + vec x uv
    friend Matrix<T, 3, 1> operator*( const Quaternion &q, const Matrix<T, 3, 1> &v )
    {
        Matrix<T, 3, 1> uv = q.vec().cross( v );
        uv += uv;
        return v + q.w() * uv + q.vec().cross( uv );
    }
};
typedef Quaternion<double> Quaterniond;
typedef Quaternion<float>  Quaternionf;

template <typename T> class AngleAxis
{
    Matrix<T, 3, 1> m_axis;
    T               m_angle;

  public:
    // AngleAxis::operator=(QuaternionBase) (Eigen 3.3)
    explicit AngleAxis( const Quaternion<T> &q )
    {
        using std::atan2;
        T n = q.vec().norm();
        if ( n < std::numeric_limits<T>::epsilon() )
            n = q.vec().stableNorm();
        if ( n != T( 0 ) )
        {
            m_angle = T( 2 ) * atan2( n, std::abs( q.w() ) );
            if ( q.w() < T( 0 ) )
                n = -n;
            m_axis = q.vec() / n;
        }
        else
        {
            m_angle = T( 0 );
            m_axis = Matrix<T, 3, 1>( T( 1 ), T( 0 ), T( 0 ) );
        }
    }
    const Matrix<T, 3, 1> &axis() const { return m_axis; }
    T                      angle() const { return m_angle; }
};
typedef AngleAxis<double> AngleAxisd;

// Map<X>: reference semantics onto caller storage; converts to X and assigns from X.
template <typename X> class Map;
template <typename T> class Map<Quaternion<T>>
{
    T *p;

  public:
    explicit Map( T *ptr ) : p( ptr ) {}
    operator Quaternion<T>() const { return Quaternion<T>( p ); }
    Map &operator=( const Quaternion<T> &q )
    {
        for ( int i = 0; i < 4; i++ )
            p[ i ] = q.c[ i ];
        return *this;
    }
    Map &operator=( const Map &o )
    {
        for ( int i = 0; i < 4; i++ )
            p[ i ] = o.p[ i ];
        return *this;
    }
    Map( const Map &o ) : p( o.p ) {}
    T &x() { return p[ 0 ]; }
    T &y() { return p[ 1 ]; }
    T &z() { return p[ 2 ]; }
    T &w() { return p[ 3 ]; }
    const T &x() const { return p[ 0 ]; }
    const T &y() const { return p[ 1 ]; }
    const T &z() const { return p[ 2 ]; }
    const T &w() const { return p[ 3 ]; }
    Matrix<T, 4, 1> coeffs() const { return Quaternion<T>( p ).coeffs(); }
    Matrix<T, 3, 3> toRotationMatrix() const { return Quaternion<T>( p ).toRotationMatrix(); }
    T angularDistance( const Quaternion<T> &o ) const { return Quaternion<T>( p ).angularDistance( o ); }
    Matrix<T, 3, 1> vec() const { return Quaternion<T>( p ).vec(); }
    void setIdentity() { *this = Quaternion<T>::Identity(); }
};
template <typename T, int R, int C> class Map<Matrix<T, R, C>>
{
    T *p;

  public:
    explicit Map( T *ptr ) : p( ptr ) {}
    Map( const Map &o ) : p( o.p ) {}
    operator Matrix<T, R, C>() const
    {
        Matrix<T, R, C> m;
        for ( int i = 0; i < R * C; i++ )
            m.d[ i ] = p[ i ];
        return m;
    }
    Map &operator=( const Matrix<T, R, C> &m )
    {
        for ( int i = 0; i < R * C; i++ )
            p[ i ] = m.d[ i ];
        return *this;
    }
    Map &operator=( const Map &o )
    {
        for ( int i = 0; i < R * C; i++ )
            p[ i ] = o.p[ i ];
        return *this;
    }
    T &      operator()( int i ) { return p[ i ]; }
    const T &operator()( int i ) const { return p[ i ]; }
    T &x() { return p[ 0 ]; }
    T &y() { return p[ 1 ]; }
    T &z() { return p[ 2 ]; }
    T        norm() const { return Matrix<T, R, C>( *this ).norm(); }
    Matrix<T, C, R> transpose() const { return Matrix<T, R, C>( *this ).transpose(); }
    void setZero()
    {
        for ( int i = 0; i < R * C; i++ )
            p[ i ] = T( 0 );
    }
};

// SelfAdjointEigenSolver<Matrix<S, 3, 3>>, S = double (the optional PCA checks, PCR:259-292, 357-389) or float (the cell moments and
// key-frame frames of cell_map_keyframe.hpp:239-249, 1554-1567): eigenvalues ascending, eigenvectors as columns in the same order.
// Cyclic Jacobi in double on the symmetric input, results rounded to S; Eigen's own tridiagonal-QR path differs in the last bits,
// and in the SIGN of an eigenvector (here: the component of largest magnitude is positive).
template <typename M> class SelfAdjointEigenSolver
{
    typedef typename M::Scalar S;
    Matrix<S, 3, 1>            ev;
    Matrix<S, 3, 3>            evec;

  public:
    SelfAdjointEigenSolver() {}
    explicit SelfAdjointEigenSolver( const Matrix<S, 3, 3> &A ) { compute( A ); }
    SelfAdjointEigenSolver &compute( const Matrix<S, 3, 3> &A )
    {
        double a[ 3 ][ 3 ], v[ 3 ][ 3 ] = { { 1, 0, 0 }, { 0, 1, 0 }, { 0, 0, 1 } };
        for ( int r = 0; r < 3; r++ )
            for ( int c = 0; c < 3; c++ )
                a[ r ][ c ] = 0.5 * ( ( double ) A( r, c ) + ( double ) A( c, r ) );
        for ( int sweep = 0; sweep < 60; sweep++ )
        {
            double off = a[ 0 ][ 1 ] * a[ 0 ][ 1 ] + a[ 0 ][ 2 ] * a[ 0 ][ 2 ] + a[ 1 ][ 2 ] * a[ 1 ][ 2 ];
            if ( off == 0.0 )
                break;
            for ( int p = 0; p < 2; p++ )
                for ( int q = p + 1; q < 3; q++ )
                {
                    if ( a[ p ][ q ] == 0.0 )
                        continue;
                    double theta = ( a[ q ][ q ] - a[ p ][ p ] ) / ( 2.0 * a[ p ][ q ] );
                    double t = ( theta >= 0 ? 1.0 : -1.0 ) / ( std::fabs( theta ) + std::sqrt( theta * theta + 1.0 ) );
                    double cs = 1.0 / std::sqrt( t * t + 1.0 ), sn = t * cs;
                    for ( int k = 0; k < 3; k++ )
                    {
                        double akp = a[ k ][ p ], akq = a[ k ][ q ];
                        a[ k ][ p ] = cs * akp - sn * akq;
                        a[ k ][ q ] = sn * akp + cs * akq;
                    }
                    for ( int k = 0; k < 3; k++ )
                    {
                        double apk = a[ p ][ k ], aqk = a[ q ][ k ];
                        a[ p ][ k ] = cs * apk - sn * aqk;
                        a[ q ][ k ] = sn * apk + cs * aqk;
                    }
                    for ( int k = 0; k < 3; k++ )
                    {
                        double vkp = v[ k ][ p ], vkq = v[ k ][ q ];
                        v[ k ][ p ] = cs * vkp - sn * vkq;
                        v[ k ][ q ] = sn * vkp + cs * vkq;
                    }
                }
        }
        int    idx[ 3 ] = { 0, 1, 2 };
        double e[ 3 ] = { a[ 0 ][ 0 ], a[ 1 ][ 1 ], a[ 2 ][ 2 ] };
        for ( int i = 0; i < 3; i++ )
            for ( int j = i + 1; j < 3; j++ )
                if ( e[ idx[ j ] ] < e[ idx[ i ] ] )
                {
                    int t = idx[ i ];
                    idx[ i ] = idx[ j ];
                    idx[ j ] = t;
                }
        for ( int c = 0; c < 3; c++ )
        {
            ev( c ) = S( e[ idx[ c ] ] );
            int big = 0;
            for ( int r = 1; r < 3; r++ )
                if ( std::fabs( v[ r ][ idx[ c ] ] ) > std::fabs( v[ big ][ idx[ c ] ] ) ) big = r;
            const double sgn = v[ big ][ idx[ c ] ] < 0 ? -1.0 : 1.0;
            for ( int r = 0; r < 3; r++ )
                evec( r, c ) = S( sgn * v[ r ][ idx[ c ] ] );
        }
        return *this;
    }
    const Matrix<S, 3, 1> &eigenvalues() const { return ev; }
    const Matrix<S, 3, 3> &eigenvectors() const { return evec; }
};

} // namespace Eigen
#include "ll_stub_eigen_dyn.h"
#endif
