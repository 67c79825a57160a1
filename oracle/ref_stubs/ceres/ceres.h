/*
 * ceres/ceres.h -- OUR minimal stand-in for the parts of Ceres Solver (< 2.2 API) that the reference's hot path uses
 * (ceres_icp.hpp:81-380, point_cloud_registration.hpp:43,143-161,220-228,323,422,449,460-508).
 *
 * TEST INFRASTRUCTURE ONLY (oracle/_ref build).  Ceres is absent from this image and not vendored by the reference;
 * the reference sources are compiled VERBATIM against this header.  Restated from Ceres' documented behaviour
 * (still "unpinned" third-party arithmetic):
 *   - Jet<double,N> forward-mode duals and AutoDiffCostFunction<F, kNumResiduals, N0, N1>::Evaluate;
 *   - HuberLoss, Corrector (rho'' <= 0 branch), EigenQuaternionParameterization (Plus / ComputeJacobian);
 *   - Problem (parameter blocks, residual blocks, bounds, Evaluate with loss-corrected residuals);
 *   - Solve(): trust-region Levenberg-Marquardt with default Solver::Options (see ll_stub_ceres_solver.h).
 */
#ifndef LL_STUB_CERES_H
#define LL_STUB_CERES_H
#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <map>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

namespace ceres
{
// ------------------------------------------------------------------------------------------------ Jet
template <typename T, int N> struct Jet
{
    T a;
    T v[ N ];
    Jet() : a( 0 )
    {
        for ( int i = 0; i < N; i++ )
            v[ i ] = 0;
    }
    Jet( const T &value ) : a( value ) // NOLINT (implicit, like ceres::Jet)
    {
        for ( int i = 0; i < N; i++ )
            v[ i ] = 0;
    }
    Jet( const T &value, int k ) : a( value )
    {
        for ( int i = 0; i < N; i++ )
            v[ i ] = 0;
        v[ k ] = 1;
    }
    Jet &operator+=( const Jet &y )
    {
        *this = *this + y;
        return *this;
    }
    Jet &operator-=( const Jet &y )
    {
        *this = *this - y;
        return *this;
    }
    Jet &operator*=( const Jet &y )
    {
        *this = *this * y;
        return *this;
    }
    Jet &operator/=( const Jet &y )
    {
        *this = *this / y;
        return *this;
    }
};
#define LL_JET template <typename T, int N> inline
LL_JET Jet<T, N> operator+( const Jet<T, N> &f ) { return f; }
LL_JET Jet<T, N> operator-( const Jet<T, N> &f )
{
    Jet<T, N> h;
    h.a = -f.a;
    for ( int i = 0; i < N; i++ )
        h.v[ i ] = -f.v[ i ];
    return h;
}
LL_JET Jet<T, N> operator+( const Jet<T, N> &f, const Jet<T, N> &g )
{
    Jet<T, N> h;
    h.a = f.a + g.a;
    for ( int i = 0; i < N; i++ )
        h.v[ i ] = f.v[ i ] + g.v[ i ];
    return h;
}
LL_JET Jet<T, N> operator+( const Jet<T, N> &f, T s )
{
    Jet<T, N> h = f;
    h.a = f.a + s;
    return h;
}
LL_JET Jet<T, N> operator+( T s, const Jet<T, N> &f )
{
    Jet<T, N> h = f;
    h.a = s + f.a;
    return h;
}
LL_JET Jet<T, N> operator-( const Jet<T, N> &f, const Jet<T, N> &g )
{
    Jet<T, N> h;
    h.a = f.a - g.a;
    for ( int i = 0; i < N; i++ )
        h.v[ i ] = f.v[ i ] - g.v[ i ];
    return h;
}
LL_JET Jet<T, N> operator-( const Jet<T, N> &f, T s )
{
    Jet<T, N> h = f;
    h.a = f.a - s;
    return h;
}
LL_JET Jet<T, N> operator-( T s, const Jet<T, N> &f )
{
    Jet<T, N> h;
    h.a = s - f.a;
    for ( int i = 0; i < N; i++ )
        h.v[ i ] = -f.v[ i ];
    return h;
}
LL_JET Jet<T, N> operator*( const Jet<T, N> &f, const Jet<T, N> &g )
{
    Jet<T, N> h;
    h.a = f.a * g.a;
    for ( int i = 0; i < N; i++ )
        h.v[ i ] = f.a * g.v[ i ] + f.v[ i ] * g.a;
    return h;
}
LL_JET Jet<T, N> operator*( const Jet<T, N> &f, T s )
{
    Jet<T, N> h;
    h.a = f.a * s;
    for ( int i = 0; i < N; i++ )
        h.v[ i ] = f.v[ i ] * s;
    return h;
}
LL_JET Jet<T, N> operator*( T s, const Jet<T, N> &f )
{
    Jet<T, N> h;
    h.a = f.a * s;
    for ( int i = 0; i < N; i++ )
        h.v[ i ] = f.v[ i ] * s;
    return h;
}
LL_JET Jet<T, N> operator/( const Jet<T, N> &f, const Jet<T, N> &g )
{
    // ceres/jet.h: g_a_inverse = 1/g.a; f_a_by_g_a = f.a * g_a_inverse; v = (f.v - f_a_by_g_a * g.v) * g_a_inverse
    Jet<T, N> h;
    const T   g_a_inverse = T( 1.0 ) / g.a;
    const T   f_a_by_g_a = f.a * g_a_inverse;
    h.a = f_a_by_g_a;
    for ( int i = 0; i < N; i++ )
        h.v[ i ] = ( f.v[ i ] - f_a_by_g_a * g.v[ i ] ) * g_a_inverse;
    return h;
}
LL_JET Jet<T, N> operator/( const Jet<T, N> &f, T s )
{
    Jet<T, N> h;
    const T   s_inverse = T( 1.0 ) / s;
    h.a = f.a * s_inverse;
    for ( int i = 0; i < N; i++ )
        h.v[ i ] = f.v[ i ] * s_inverse;
    return h;
}
LL_JET Jet<T, N> operator/( T s, const Jet<T, N> &g )
{
    Jet<T, N> h;
    const T   minus_s_g_a_inverse2 = -s / ( g.a * g.a );
    h.a = s / g.a;
    for ( int i = 0; i < N; i++ )
        h.v[ i ] = g.v[ i ] * minus_s_g_a_inverse2;
    return h;
}
#define LL_JET_CMP( op )                                                                  \
    LL_JET bool operator op( const Jet<T, N> &f, const Jet<T, N> &g ) { return f.a op g.a; } \
    LL_JET bool operator op( const Jet<T, N> &f, T g ) { return f.a op g; }                \
    LL_JET bool operator op( T f, const Jet<T, N> &g ) { return f op g.a; }
LL_JET_CMP( < )
LL_JET_CMP( <= )
LL_JET_CMP( > )
LL_JET_CMP( >= )
LL_JET_CMP( == )
LL_JET_CMP( != )
#undef LL_JET_CMP
LL_JET Jet<T, N> abs( const Jet<T, N> &f ) { return f.a < T( 0 ) ? -f : f; }
LL_JET Jet<T, N> sqrt( const Jet<T, N> &f )
{
    Jet<T, N> h;
    const T   tmp = std::sqrt( f.a );
    const T   two_a_inverse = T( 1.0 ) / ( T( 2.0 ) * tmp );
    h.a = tmp;
    for ( int i = 0; i < N; i++ )
        h.v[ i ] = f.v[ i ] * two_a_inverse;
    return h;
}
LL_JET Jet<T, N> sin( const Jet<T, N> &f )
{
    Jet<T, N> h;
    const T   c = std::cos( f.a );
    h.a = std::sin( f.a );
    for ( int i = 0; i < N; i++ )
        h.v[ i ] = c * f.v[ i ];
    return h;
}
LL_JET Jet<T, N> cos( const Jet<T, N> &f )
{
    Jet<T, N> h;
    const T   s = -std::sin( f.a );
    h.a = std::cos( f.a );
    for ( int i = 0; i < N; i++ )
        h.v[ i ] = s * f.v[ i ];
    return h;
}
LL_JET Jet<T, N> acos( const Jet<T, N> &f )
{
    Jet<T, N> h;
    const T   tmp = -T( 1.0 ) / std::sqrt( T( 1.0 ) - f.a * f.a );
    h.a = std::acos( f.a );
    for ( int i = 0; i < N; i++ )
        h.v[ i ] = tmp * f.v[ i ];
    return h;
}
LL_JET Jet<T, N> atan2( const Jet<T, N> &g, const Jet<T, N> &f )
{
    // d atan2(g, f) = (f dg - g df) / (f^2 + g^2)
    Jet<T, N> h;
    const T   tmp = T( 1.0 ) / ( f.a * f.a + g.a * g.a );
    h.a = std::atan2( g.a, f.a );
    for ( int i = 0; i < N; i++ )
        h.v[ i ] = tmp * ( -g.a * f.v[ i ] + f.a * g.v[ i ] );
    return h;
}
#undef LL_JET

// ------------------------------------------------------------------------------------------------ cost functions
class CostFunction
{
  public:
    virtual ~CostFunction() {}
    // parameters[i] -> block i; jacobians[i] (may be null) row-major num_residuals x block_size(i)
    virtual bool Evaluate( double const *const *parameters, double *residuals, double **jacobians ) const = 0;
    const std::vector<int> &parameter_block_sizes() const { return sizes_; }
    int                     num_residuals() const { return nres_; }

  protected:
    std::vector<int> sizes_;
    int              nres_ = 0;
};

template <typename Functor, int kNumResiduals, int N0, int N1> class AutoDiffCostFunction : public CostFunction
{
    std::unique_ptr<Functor> f_;

  public:
    explicit AutoDiffCostFunction( Functor *f ) : f_( f )
    {
        sizes_.push_back( N0 );
        sizes_.push_back( N1 );
        nres_ = kNumResiduals;
    }
    bool Evaluate( double const *const *parameters, double *residuals, double **jacobians ) const override
    {
        if ( !jacobians )
            return ( *f_ )( parameters[ 0 ], parameters[ 1 ], residuals );
        typedef Jet<double, N0 + N1> J;
        J x0[ N0 ], x1[ N1 ], out[ kNumResiduals ];
        for ( int i = 0; i < N0; i++ )
            x0[ i ] = J( parameters[ 0 ][ i ], i );
        for ( int i = 0; i < N1; i++ )
            x1[ i ] = J( parameters[ 1 ][ i ], N0 + i );
        if ( !( *f_ )( x0, x1, out ) )
            return false;
        for ( int r = 0; r < kNumResiduals; r++ )
        {
            residuals[ r ] = out[ r ].a;
            if ( jacobians[ 0 ] )
                for ( int c = 0; c < N0; c++ )
                    jacobians[ 0 ][ r * N0 + c ] = out[ r ].v[ c ];
            if ( jacobians[ 1 ] )
                for ( int c = 0; c < N1; c++ )
                    jacobians[ 1 ][ r * N1 + c ] = out[ r ].v[ N0 + c ];
        }
        return true;
    }
};

// ------------------------------------------------------------------------------------------------ loss
class LossFunction
{
  public:
    virtual ~LossFunction() {}
    virtual void Evaluate( double sq_norm, double out[ 3 ] ) const = 0; // rho, rho', rho''
};
class HuberLoss : public LossFunction
{
    double a_, b_;

  public:
    explicit HuberLoss( double a ) : a_( a ), b_( a * a ) {}
    void Evaluate( double s, double rho[ 3 ] ) const override
    {
        if ( s > b_ )
        {
            const double r = std::sqrt( s );
            rho[ 0 ] = 2.0 * a_ * r - b_;
            rho[ 1 ] = std::max( std::numeric_limits<double>::min(), a_ / r );
            rho[ 2 ] = -rho[ 1 ] / ( 2.0 * s );
        }
        else
        {
            rho[ 0 ] = s;
            rho[ 1 ] = 1.0;
            rho[ 2 ] = 0.0;
        }
    }
};

// ------------------------------------------------------------------------------------------------ parameterization
class LocalParameterization
{
  public:
    virtual ~LocalParameterization() {}
    virtual bool Plus( const double *x, const double *delta, double *x_plus_delta ) const = 0;
    virtual bool ComputeJacobian( const double *x, double *jacobian ) const = 0; // row-major GlobalSize x LocalSize
    virtual int  GlobalSize() const = 0;
    virtual int  LocalSize() const = 0;
};
// storage (x, y, z, w); Plus(x, d) = [sin|d| d/|d|, cos|d|] (x) x
class EigenQuaternionParameterization : public LocalParameterization
{
  public:
    bool Plus( const double *x, const double *delta, double *out ) const override
    {
        const double nd = std::sqrt( delta[ 0 ] * delta[ 0 ] + delta[ 1 ] * delta[ 1 ] + delta[ 2 ] * delta[ 2 ] );
        if ( nd > 0.0 )
        {
            const double s = std::sin( nd ) / nd;
            const double qd[ 4 ] = { s * delta[ 0 ], s * delta[ 1 ], s * delta[ 2 ], std::cos( nd ) }; // x y z w
            const double w = qd[ 3 ] * x[ 3 ] - qd[ 0 ] * x[ 0 ] - qd[ 1 ] * x[ 1 ] - qd[ 2 ] * x[ 2 ];
            const double xx = qd[ 3 ] * x[ 0 ] + qd[ 0 ] * x[ 3 ] + qd[ 1 ] * x[ 2 ] - qd[ 2 ] * x[ 1 ];
            const double yy = qd[ 3 ] * x[ 1 ] + qd[ 1 ] * x[ 3 ] + qd[ 2 ] * x[ 0 ] - qd[ 0 ] * x[ 2 ];
            const double zz = qd[ 3 ] * x[ 2 ] + qd[ 2 ] * x[ 3 ] + qd[ 0 ] * x[ 1 ] - qd[ 1 ] * x[ 0 ];
            out[ 0 ] = xx;
            out[ 1 ] = yy;
            out[ 2 ] = zz;
            out[ 3 ] = w;
        }
        else
            for ( int i = 0; i < 4; i++ )
                out[ i ] = x[ i ];
        return true;
    }
    bool ComputeJacobian( const double *x, double *J ) const override
    {
        // clang-format off
        J[0] =  x[3]; J[1]  =  x[2]; J[2]  = -x[1];
        J[3] = -x[2]; J[4]  =  x[3]; J[5]  =  x[0];
        J[6] =  x[1]; J[7]  = -x[0]; J[8]  =  x[3];
        J[9] = -x[0]; J[10] = -x[1]; J[11] = -x[2];
        // clang-format on
        return true;
    }
    int GlobalSize() const override { return 4; }
    int LocalSize() const override { return 3; }
};

enum LinearSolverType
{
    DENSE_NORMAL_CHOLESKY,
    DENSE_QR,
    SPARSE_NORMAL_CHOLESKY,
    DENSE_SCHUR,
    SPARSE_SCHUR,
    ITERATIVE_SCHUR,
    CGNR
};

struct ll_ResidualBlock
{
    CostFunction *cost;
    LossFunction *loss;
    double *      p0;
    double *      p1;
};
typedef ll_ResidualBlock *ResidualBlockId;

class Problem
{
  public:
    struct Options
    {
    };
    struct EvaluateOptions
    {
        std::vector<ResidualBlockId> residual_blocks;
    };
    struct ParamBlock
    {
        double *               ptr;
        int                    size;
        LocalParameterization *lp;
        std::vector<double>    lo, hi;
    };

    Problem() {}
    explicit Problem( const Options & ) {}
    ~Problem()
    {
        // Problem owns cost functions, loss functions and parameterizations by default (each deleted once)
        std::vector<void *> seen;
        for ( auto *b : all_blocks_ )
        {
            if ( std::find( seen.begin(), seen.end(), ( void * ) b->cost ) == seen.end() )
            {
                seen.push_back( b->cost );
                delete b->cost;
            }
            if ( b->loss && std::find( seen.begin(), seen.end(), ( void * ) b->loss ) == seen.end() )
            {
                seen.push_back( b->loss );
                delete b->loss;
            }
            delete b;
        }
        for ( auto &pb : params_ )
            delete pb.lp;
    }
    void AddParameterBlock( double *v, int size, LocalParameterization *lp = nullptr )
    {
        ParamBlock pb;
        pb.ptr = v;
        pb.size = size;
        pb.lp = lp;
        pb.lo.assign( size, -std::numeric_limits<double>::max() );
        pb.hi.assign( size, std::numeric_limits<double>::max() );
        params_.push_back( pb );
    }
    ResidualBlockId AddResidualBlock( CostFunction *c, LossFunction *l, double *x0, double *x1 )
    {
        ll_ResidualBlock *b = new ll_ResidualBlock{ c, l, x0, x1 };
        blocks_.push_back( b );
        all_blocks_.push_back( b );
        return b;
    }
    void RemoveResidualBlock( ResidualBlockId id )
    {
        auto it = std::find( blocks_.begin(), blocks_.end(), id );
        if ( it != blocks_.end() )
            blocks_.erase( it );
    }
    void SetParameterLowerBound( double *v, int index, double lower ) { find( v )->lo[ index ] = lower; }
    void SetParameterUpperBound( double *v, int index, double upper ) { find( v )->hi[ index ] = upper; }
    int  NumResidualBlocks() const { return ( int ) blocks_.size(); }

    // cost = 1/2 sum rho(|r|^2); residuals are loss-corrected (Corrector::CorrectResiduals, rho'' <= 0 branch:
    // r *= sqrt(rho'))
    bool Evaluate( const EvaluateOptions &opt, double *cost, std::vector<double> *residuals, std::vector<double> *gradient, void *jacobian )
    {
        ( void ) gradient;
        ( void ) jacobian;
        const std::vector<ResidualBlockId> &bl = opt.residual_blocks.empty() ? blocks_ : opt.residual_blocks;
        double                              c = 0;
        if ( residuals )
            residuals->clear();
        for ( auto *b : bl )
        {
            double        r[ 16 ];
            const double *pp[ 2 ] = { b->p0, b->p1 };
            b->cost->Evaluate( pp, r, nullptr );
            const int nr = b->cost->num_residuals();
            double    s = 0;
            for ( int i = 0; i < nr; i++ )
                s += r[ i ] * r[ i ];
            double rho[ 3 ] = { s, 1.0, 0.0 };
            if ( b->loss )
                b->loss->Evaluate( s, rho );
            c += 0.5 * rho[ 0 ];
            const double sr = std::sqrt( rho[ 1 ] );
            if ( residuals )
                for ( int i = 0; i < nr; i++ )
                    residuals->push_back( r[ i ] * sr );
        }
        if ( cost )
            *cost = c;
        return true;
    }

    // internals used by the stub solver
    ParamBlock *find( double *v )
    {
        for ( auto &pb : params_ )
            if ( pb.ptr == v )
                return &pb;
        return nullptr;
    }
    std::vector<ParamBlock>         params_;
    std::vector<ll_ResidualBlock *> blocks_;     // live, in insertion order
    std::vector<ll_ResidualBlock *> all_blocks_; // for ownership
};

struct Solver
{
    struct Options
    {
        LinearSolverType linear_solver_type = SPARSE_NORMAL_CHOLESKY;
        int              max_num_iterations = 50;
        bool             minimizer_progress_to_stdout = false;
        bool             check_gradients = false;
        double           gradient_check_relative_precision = 1e-8;
        double           function_tolerance = 1e-6;
        double           gradient_tolerance = 1e-10;
        double           parameter_tolerance = 1e-8;
        double           initial_trust_region_radius = 1e4;
        double           max_trust_region_radius = 1e16;
        double           min_trust_region_radius = 1e-32;
        double           min_relative_decrease = 1e-3;
        double           min_lm_diagonal = 1e-6;
        double           max_lm_diagonal = 1e32;
        int              max_num_consecutive_invalid_steps = 5;
        bool             jacobi_scaling = true;
        bool             use_nonmonotonic_steps = false;
    };
    struct Summary
    {
        double      initial_cost = 0;
        double      final_cost = 0;
        int         num_residual_blocks = 0;
        int         num_successful_steps = 0;
        int         num_unsuccessful_steps = 0;
        int         ll_iterations = 0; // stub extension: LM iterations run (excluding iteration 0)
        std::string message;
        std::string BriefReport() const
        {
            std::ostringstream s;
            s << "stub ceres: iterations " << ll_iterations << " initial_cost " << initial_cost << " final_cost " << final_cost;
            return s.str();
        }
        std::string FullReport() const { return BriefReport(); }
    };
};

inline void Solve( const Solver::Options &options, Problem *problem, Solver::Summary *summary );

} // namespace ceres
#include "ll_stub_ceres_solver.h"
#endif
