/*
 * ll_stub_eigen_dyn.h -- OUR stand-in for Eigen::Matrix<float, Dynamic, Dynamic> as source/cell_map_keyframe.hpp uses it (the 60 x 60
 * direction images of Maps_keyframe: generate_feature_img :1385-1427, add_padding_to_feature_image :1321-1356, apply_guassian_blur
 * :1358-1370, refine_feature_img :1127-1142, ratio_of_nonzero_in_img :1144-1154, max_similiarity_of_two_image :1156-1229,
 * similiarity_of_two_image :1569-1582).  Included at the end of ll_stub_eigen.h.  TEST INFRASTRUCTURE ONLY (oracle/_ref build).
 * Column-major, eager, element by element; blocks are index windows onto the owning matrix.
 */
#ifndef LL_STUB_EIGEN_DYN_H
#define LL_STUB_EIGEN_DYN_H
#include <cmath>
#include <vector>

namespace Eigen
{
template <typename T> class DynBlock;
template <typename T> class DynArray;

template <typename T> class Matrix<T, Dynamic, Dynamic>
{
  public:
    typedef T      Scalar;
    int            r_ = 0, c_ = 0;
    std::vector<T> d;

    Matrix() {}
    Matrix( int r, int c ) { resize( r, c ); }
    Matrix( const DynBlock<T> &b ) { *this = b.eval(); }
    void resize( long r, long c )  // (Eigen leaves the new coefficients uninitialised; every caller in the reference fills all of them)
    {
        r_ = ( int ) r;
        c_ = ( int ) c;
        d.assign( ( size_t ) r_ * c_, T( 0 ) );
    }
    void setZero() { std::fill( d.begin(), d.end(), T( 0 ) ); }
    int  rows() const { return r_; }
    int  cols() const { return c_; }
    int  size() const { return r_ * c_; }
    T       &operator()( long i, long j ) { return d[ ( size_t ) i + ( size_t ) j * r_ ]; }
    const T &operator()( long i, long j ) const { return d[ ( size_t ) i + ( size_t ) j * r_ ]; }
    Matrix   eval() const { return *this; }

    DynBlock<T> block( long i, long j, long p, long q ) { return DynBlock<T>( this, ( int ) i, ( int ) j, ( int ) p, ( int ) q ); }
    DynBlock<T> block( long i, long j, long p, long q ) const
    {
        return DynBlock<T>( const_cast<Matrix *>( this ), ( int ) i, ( int ) j, ( int ) p, ( int ) q );
    }
    DynBlock<T> row( long i ) const { return block( i, 0, 1, c_ ); }
    DynBlock<T> col( long j ) const { return block( 0, j, r_, 1 ); }

    struct Wise
    {
        const Matrix *m;
        int           rowwise;
        Matrix        reverse() const
        {
            Matrix o( m->r_, m->c_ );
            for ( int i = 0; i < m->r_; i++ )
                for ( int j = 0; j < m->c_; j++ )
                    o( i, j ) = rowwise ? ( *m )( i, m->c_ - 1 - j ) : ( *m )( m->r_ - 1 - i, j );
            return o;
        }
    };
    Wise rowwise() const { return Wise{ this, 1 }; }  // .reverse(): every row reversed (columns swapped)
    Wise colwise() const { return Wise{ this, 0 }; }  // .reverse(): every column reversed (rows swapped)

    T maxCoeff() const
    {
        T m = d[ 0 ];
        for ( size_t i = 1; i < d.size(); i++ )
            if ( d[ i ] > m ) m = d[ i ];
        return m;
    }
    T sum() const
    {
        T s = T( 0 );
        for ( size_t i = 0; i < d.size(); i++ ) s = s + d[ i ];
        return s;
    }
    T           mean() const { return sum() / T( d.size() ); }
    DynArray<T> array() const { return DynArray<T>( *this ); }

    // m << a, b : blocks stacked top to bottom (the only form the reference writes, in a disabled branch)
    struct Comma
    {
        Matrix *m;
        int     row;
        Comma &operator,( const Matrix &o )
        {
            for ( int i = 0; i < o.r_; i++ )
                for ( int j = 0; j < o.c_; j++ )
                    ( *m )( row + i, j ) = o( i, j );
            row += o.r_;
            return *this;
        }
    };
    Comma operator<<( const Matrix &o )
    {
        Comma c{ this, 0 };
        c, o;
        return c;
    }
};

template <typename T> class DynBlock
{
    typedef Matrix<T, Dynamic, Dynamic> M;
    M                                  *m;
    int                                 i0, j0, p, q;

  public:
    DynBlock( M *mm, int i, int j, int pp, int qq ) : m( mm ), i0( i ), j0( j ), p( pp ), q( qq ) {}
    int rows() const { return p; }
    int cols() const { return q; }
    M   eval() const
    {
        M o( p, q );
        for ( int i = 0; i < p; i++ )
            for ( int j = 0; j < q; j++ )
                o( i, j ) = ( *m )( i0 + i, j0 + j );
        return o;
    }
    T maxCoeff() const { return eval().maxCoeff(); }
    DynBlock &operator=( const M &o )
    {
        for ( int i = 0; i < p; i++ )
            for ( int j = 0; j < q; j++ )
                ( *m )( i0 + i, j0 + j ) = o( i, j );
        return *this;
    }
    DynBlock &operator=( const DynBlock &o ) { return *this = o.eval(); }
};

// coefficient-wise view (similiarity_of_two_image, an unused alternative to the template match)
template <typename T> class DynArray
{
    typedef Matrix<T, Dynamic, Dynamic> M;
    M                                   v;

  public:
    explicit DynArray( const M &m ) : v( m ) {}
    DynArray array() const { return *this; }
    DynArray operator-( const T &s ) const
    {
        DynArray o( v );
        for ( auto &x : o.v.d ) x = x - s;
        return o;
    }
    DynArray cwiseProduct( const DynArray &b ) const
    {
        DynArray o( v );
        for ( size_t i = 0; i < o.v.d.size(); i++ ) o.v.d[ i ] = v.d[ i ] * b.v.d[ i ];
        return o;
    }
    DynArray pow( int e ) const
    {
        DynArray o( v );
        for ( auto &x : o.v.d ) x = ( T ) std::pow( ( double ) x, e );
        return o;
    }
    T sum() const { return v.sum(); }
    T mean() const { return v.mean(); }
};
} // namespace Eigen
#endif
