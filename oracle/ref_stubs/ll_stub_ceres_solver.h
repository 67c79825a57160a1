/*
 * ll_stub_ceres_solver.h -- ceres::Solve for the stand-in ceres/ceres.h (TEST INFRASTRUCTURE ONLY, oracle/_ref build).
 *
 * A dense restatement of Ceres' TrustRegionMinimizer + LevenbergMarquardtStrategy with default Solver::Options,
 * written against the generic Problem (explicit n x 6 Jacobian, normal equations by Cholesky) -- deliberately a
 * different formulation from oracle/ll_oracle_reg.c (which accumulates H and g block by block), so that the two can be
 * checked against each other through the reference's own driver text (point_cloud_registration.hpp).
 *
 * Sequence (trust_region_minimizer.cc, levenberg_marquardt_strategy.cc, line_search.cc of Ceres 1.14):
 *   IterationZero: project x on the bounds, evaluate, Jacobi scaling 1/(1+|col|), projected gradient norm;
 *   loop: LM step on the scaled system, model cost change, invalid-step handling, projected ARMIJO line search
 *   (problem has bounds), candidate evaluation, parameter / function tolerance, step quality, radius update.
 * A second or later contraction of the line search interpolates through three samples (quintic_min below), as Ceres does
 * (round 2 re-fitted the two-sample cubic there).
 */
#ifndef LL_STUB_CERES_SOLVER_H
#define LL_STUB_CERES_SOLVER_H

namespace ceres
{
namespace ll_solver
{
struct Layout
{
    // global parameter vector = concatenation of the problem's parameter blocks; local (tangent) likewise
    std::vector<int> goff, loff;
    int              nglobal = 0, nlocal = 0;
};

inline Layout make_layout( Problem *p )
{
    Layout L;
    for ( auto &pb : p->params_ )
    {
        L.goff.push_back( L.nglobal );
        L.loff.push_back( L.nlocal );
        L.nglobal += pb.size;
        L.nlocal += pb.lp ? pb.lp->LocalSize() : pb.size;
    }
    return L;
}

inline int block_index( Problem *p, const double *ptr )
{
    for ( size_t i = 0; i < p->params_.size(); i++ )
        if ( p->params_[ i ].ptr == ptr )
            return ( int ) i;
    return -1;
}

// ProgramEvaluator::Plus: per block local-parameterization Plus (or x + d), then clamp to the bounds
inline void plus( Problem *p, const Layout &L, const std::vector<double> &x, const std::vector<double> &d, std::vector<double> &out )
{
    out.resize( L.nglobal );
    for ( size_t b = 0; b < p->params_.size(); b++ )
    {
        auto &        pb = p->params_[ b ];
        const double *xb = &x[ L.goff[ b ] ];
        const double *db = &d[ L.loff[ b ] ];
        double *      ob = &out[ L.goff[ b ] ];
        if ( pb.lp )
            pb.lp->Plus( xb, db, ob );
        else
            for ( int i = 0; i < pb.size; i++ )
                ob[ i ] = xb[ i ] + db[ i ];
        for ( int i = 0; i < pb.size; i++ )
        {
            if ( ob[ i ] < pb.lo[ i ] )
                ob[ i ] = pb.lo[ i ];
            if ( ob[ i ] > pb.hi[ i ] )
                ob[ i ] = pb.hi[ i ];
        }
    }
}

// Evaluate at x: cost, loss-corrected residuals r (3 per block), corrected local Jacobian J (rows x nlocal, row-major),
// gradient g = J' r.  J / g may be skipped.
inline bool evaluate( Problem *p, const Layout &L, const std::vector<double> &x, double *cost, std::vector<double> *r,
                      std::vector<double> *J, std::vector<double> *g )
{
    const int nb = ( int ) p->blocks_.size();
    int       rows = 0;
    for ( auto *b : p->blocks_ )
        rows += b->cost->num_residuals();
    if ( r )
        r->assign( rows, 0.0 );
    if ( J )
        J->assign( ( size_t ) rows * L.nlocal, 0.0 );
    if ( g )
        g->assign( L.nlocal, 0.0 );
    // local-parameterization Jacobians at x
    std::vector<std::vector<double>> lpj( p->params_.size() );
    if ( J || g )
        for ( size_t b = 0; b < p->params_.size(); b++ )
            if ( p->params_[ b ].lp )
            {
                lpj[ b ].resize( p->params_[ b ].lp->GlobalSize() * p->params_[ b ].lp->LocalSize() );
                p->params_[ b ].lp->ComputeJacobian( &x[ L.goff[ b ] ], lpj[ b ].data() );
            }
    double c = 0;
    int    row0 = 0;
    for ( int k = 0; k < nb; k++ )
    {
        ll_ResidualBlock *blk = p->blocks_[ k ];
        const int         nr = blk->cost->num_residuals();
        const int         i0 = block_index( p, blk->p0 ), i1 = block_index( p, blk->p1 );
        const double *    pp[ 2 ] = { &x[ L.goff[ i0 ] ], &x[ L.goff[ i1 ] ] };
        double            res[ 8 ], j0[ 8 * 8 ], j1[ 8 * 8 ];
        double *          jj[ 2 ] = { j0, j1 };
        const bool        want_j = J || g;
        if ( !blk->cost->Evaluate( pp, res, want_j ? jj : nullptr ) )
            return false;
        double s = 0;
        for ( int i = 0; i < nr; i++ )
            s += res[ i ] * res[ i ];
        double rho[ 3 ] = { s, 1.0, 0.0 };
        if ( blk->loss )
            blk->loss->Evaluate( s, rho );
        c += 0.5 * rho[ 0 ];
        // Corrector: Huber has rho'' <= 0 -> residual and Jacobian scaled by sqrt(rho')
        const double sr = std::sqrt( rho[ 1 ] );
        for ( int i = 0; i < nr; i++ )
        {
            if ( r )
                ( *r )[ row0 + i ] = sr * res[ i ];
        }
        if ( want_j )
        {
            const int    idx[ 2 ] = { i0, i1 };
            const double *jg[ 2 ] = { j0, j1 };
            for ( int side = 0; side < 2; side++ )
            {
                auto &    pb = p->params_[ idx[ side ] ];
                const int gs = pb.size, ls = pb.lp ? pb.lp->LocalSize() : pb.size;
                for ( int i = 0; i < nr; i++ )
                    for ( int cidx = 0; cidx < ls; cidx++ )
                    {
                        double v;
                        if ( pb.lp )
                        {
                            v = 0;
                            for ( int m = 0; m < gs; m++ )
                                v += jg[ side ][ i * gs + m ] * lpj[ idx[ side ] ][ m * ls + cidx ];
                        }
                        else
                            v = jg[ side ][ i * gs + cidx ];
                        v *= sr;
                        if ( J )
                            ( *J )[ ( size_t )( row0 + i ) * L.nlocal + L.loff[ idx[ side ] ] + cidx ] = v;
                        if ( g )
                            ( *g )[ L.loff[ idx[ side ] ] + cidx ] += v * sr * res[ i ];
                    }
            }
        }
        row0 += nr;
    }
    *cost = c;
    return true;
}

inline bool cholesky_solve( int n, std::vector<double> A, const std::vector<double> &b, std::vector<double> &x )
{
    for ( int i = 0; i < n; i++ )
        for ( int j = 0; j <= i; j++ )
        {
            double s = A[ i * n + j ];
            for ( int k = 0; k < j; k++ )
                s -= A[ i * n + k ] * A[ j * n + k ];
            if ( i == j )
            {
                if ( !( s > 0.0 ) )
                    return false;
                A[ i * n + i ] = std::sqrt( s );
            }
            else
                A[ i * n + j ] = s / A[ j * n + j ];
        }
    std::vector<double> y( n );
    for ( int i = 0; i < n; i++ )
    {
        double s = b[ i ];
        for ( int k = 0; k < i; k++ )
            s -= A[ i * n + k ] * y[ k ];
        y[ i ] = s / A[ i * n + i ];
    }
    x.assign( n, 0.0 );
    for ( int i = n - 1; i >= 0; i-- )
    {
        double s = y[ i ];
        for ( int k = i + 1; k < n; k++ )
            s -= A[ k * n + i ] * x[ k ];
        x[ i ] = s / A[ i * n + i ];
    }
    for ( int i = 0; i < n; i++ )
        if ( !std::isfinite( x[ i ] ) )
            return false;
    return true;
}

// LineSearch::InterpolatingPolynomialMinimizingStepSize, CUBIC, with a valid `previous` sample (line_search.cc): three samples
// with value and gradient -> the quintic interpolant (polynomial.cc FindInterpolatingPolynomial solves the 6 x 6 system with a
// row [x^5 .. 1] per value and [5 x^4 .. 0] per gradient by fully pivoted LU; here the same polynomial in Newton form by divided
// differences, the arithmetic the oracle and the device share), minimised over [lo, hi] by MinimizePolynomial: the better end
// point, then the real roots of the derivative inside the interval.  (Ceres: companion-matrix eigenvalues, i.e. every real root;
// here every real root as well, isolated by the derivative chain -- p' is monotone between the roots of p'', p'' between the roots of
// the quadratic p''' -- and bisected.  A root without a sign change is no minimum, and the real parts of complex roots that Ceres
// also tries cannot beat the stationary points.)
inline double quintic_poly4( const double k[ 5 ], double x ) { return std::fma( std::fma( std::fma( std::fma( k[ 4 ], x, k[ 3 ] ), x, k[ 2 ] ), x, k[ 1 ] ), x, k[ 0 ] ); }
inline bool quintic_interval_root( const double k[ 5 ], double a, double b, double va, double vb, double *root )
{
    // Illinois regula falsi on the bracket (the same operations as ll_reg_core.h quintic_interval_root / the oracle's qm_interval_root)
    if ( vb == 0.0 )
    {
        *root = b;
        return true;
    }
    if ( !( ( va < 0.0 && vb > 0.0 ) || ( va > 0.0 && vb < 0.0 ) ) ) return false;
    double     l = a, r = b, wl = va, wr = vb;
    const bool neg_left = va < 0.0;
    double     x = b;
    int        side = 0;
    for ( int it = 0; it < 64; it++ )
    {
        double c = ( wl * r - wr * l ) / ( wl - wr );
        if ( !( c > l && c < r ) ) c = 0.5 * ( l + r );
        if ( c == l || c == r || c == x )
        {
            x = c;
            break;
        }
        x = c;
        const double vc = quintic_poly4( k, c );
        if ( vc == 0.0 ) break;
        if ( ( vc < 0.0 ) == neg_left )
        {
            l = c;
            wl = vc;
            if ( side == -1 ) wr *= 0.5;
            side = -1;
        }
        else
        {
            r = c;
            wr = vc;
            if ( side == 1 ) wl *= 0.5;
            side = 1;
        }
    }
    *root = x;
    return true;
}
inline int quintic_roots_between( const double k[ 5 ], double lo, double hi, const double *bp, int nb, double *roots )
{
    int    n = 0;
    double a = lo, va = quintic_poly4( k, lo );
    for ( int i = 0; i <= nb; i++ )
    {
        const double b = ( i == nb ) ? hi : bp[ i ];
        const double vb = quintic_poly4( k, b );
        double       r;
        if ( quintic_interval_root( k, a, b, va, vb, &r ) ) roots[ n++ ] = r;
        a = b;
        va = vb;
    }
    return n;
}
inline double quintic_min( double f0, double g0, double x1, double f1, double g1, double x2, double f2, double g2, double lo, double hi )
{
    /* Newton form on the nodes z = {0, 0, x1, x1, x2} (the sixth, x2 again, closes the table): divided differences with the
     * derivative in place of the quotient at a repeated node */
    const double h1 = x1, h2 = x2, h21 = x2 - x1;
    if (!(h1 != 0.0) || !(h2 != 0.0) || !(h21 != 0.0)) return std::min(std::max(0.5 * x1, lo), hi); /* coincident samples: bisect like an invalid sample */
    const double e01 = g0, e12 = (f1 - f0) / h1, e23 = g1, e34 = (f2 - f1) / h21, e45 = g2;
    const double a0 = (e12 - e01) / h1, a1 = (e23 - e12) / h1, a2 = (e34 - e23) / h21, a3 = (e45 - e34) / h21;
    const double b0 = (a1 - a0) / h1, b1 = (a2 - a1) / h2, b2 = (a3 - a2) / h21;
    const double c0 = (b1 - b0) / h2, c1 = (b2 - b1) / h2;
    const double d0 = (c1 - c0) / h2;
    /* p(x) = f0 + x (e01 + x (a0 + (x - x1) (b0 + (x - x1) (c0 + (x - x2) d0)))); value and derivative by one nested sweep of
     * explicitly fused multiply-adds (IEEE: the same bits in the oracle, the stand-in and on the device) */
#define LL_Q_EVAL(X, PV, DV)                         \
    do {                                             \
        const double x_ = (X);                       \
        const double u1_ = x_ - x1, u2_ = x_ - x2;   \
        double b_ = d0, db_ = 0.0;                   \
        db_ = std::fma(u2_, db_, b_);                     \
        b_ = std::fma(u2_, b_, c0);                       \
        db_ = std::fma(u1_, db_, b_);                     \
        b_ = std::fma(u1_, b_, b0);                       \
        db_ = std::fma(u1_, db_, b_);                     \
        b_ = std::fma(u1_, b_, a0);                       \
        db_ = std::fma(x_, db_, b_);                      \
        b_ = std::fma(x_, b_, e01);                       \
        db_ = std::fma(x_, db_, b_);                      \
        b_ = std::fma(x_, b_, f0);                        \
        (PV) = b_;                                   \
        (DV) = db_;                                  \
    } while (0)
    /* monomial coefficients: p = u3 x^5 + u2 x^4 + u1 x^3 + u0 x^2 + e01 x + f0 */
    const double s1 = d0, s0 = std::fma( -x2, d0, c0 );
    const double t2 = s1, t1 = std::fma( -x1, s1, s0 ), t0 = std::fma( -x1, s0, b0 );
    const double u3 = t2, u2 = std::fma( -x1, t2, t1 ), u1 = std::fma( -x1, t1, t0 ), u0 = std::fma( -x1, t0, a0 );
    const double dq[ 5 ] = { e01, 2.0 * u0, 3.0 * u1, 4.0 * u2, 5.0 * u3 };
    const double d2[ 5 ] = { 2.0 * u0, 6.0 * u1, 12.0 * u2, 20.0 * u3, 0.0 };
    const double A = 60.0 * u3, B = 24.0 * u2, C = 6.0 * u1;
    double r3[ 2 ];
    int    n3 = 0;
    {
        double q0 = 0.0, q1 = 0.0;
        int    n = 0;
        if ( A == 0.0 )
        {
            if ( B != 0.0 ) q0 = -C / B, n = 1;
        }
        else
        {
            const double disc = std::fma( B, B, -4.0 * A * C );
            if ( disc >= 0.0 )
            {
                const double sq = std::sqrt( disc );
                const double qq = -0.5 * ( B + ( B < 0.0 ? -sq : sq ) );
                q0 = qq / A;
                n = 1;
                if ( qq != 0.0 )
                {
                    q1 = C / qq;
                    n = 2;
                    if ( q1 < q0 ) std::swap( q0, q1 );
                }
            }
        }
        if ( n >= 1 && q0 > lo && q0 < hi ) r3[ n3++ ] = q0;
        if ( n >= 2 && q1 > lo && q1 < hi && !( n3 == 1 && q1 == r3[ 0 ] ) ) r3[ n3++ ] = q1;
    }
    double r2[ 3 ], r1[ 4 ];
    int    n2 = quintic_roots_between( d2, lo, hi, r3, n3, r2 );
    if ( n2 > 0 && !( r2[ n2 - 1 ] < hi ) ) n2--;
    const int n1 = quintic_roots_between( dq, lo, hi, r2, n2, r1 );
    double best_x = lo, best_v, vh, da, dh;
    LL_Q_EVAL(lo, best_v, da);
    LL_Q_EVAL(hi, vh, dh);
    (void)da;
    (void)dh;
    if (!(best_v < vh)) { /* MinimizePolynomial: x_min wins only when strictly smaller */
        best_v = vh;
        best_x = hi;
    }
    for ( int i = 0; i < n1; i++ )
    {
        double v, dv;
        LL_Q_EVAL( r1[ i ], v, dv );
        (void)dv;
        if ( v < best_v )
        {
            best_v = v;
            best_x = r1[ i ];
        }
    }
#undef LL_Q_EVAL
    return best_x;
}

// minimiser on [lo, hi] of the cubic through (0, f0, g0), (x1, f1, g1)
inline double cubic_min( double f0, double g0, double x1, double f1, double g1, double lo, double hi )
{
    // p(x) = a x^3 + b x^2 + g0 x + f0
    const double d0 = f1 - f0 - g0 * x1, d1 = g1 - g0;
    const double a = ( x1 * d1 - 2.0 * d0 ) / ( x1 * x1 * x1 );
    const double b = ( 3.0 * d0 - x1 * d1 ) / ( x1 * x1 );
    auto         P = [&]( double x ) { return ( ( a * x + b ) * x + g0 ) * x + f0; };
    double       bx = lo, bv = P( lo );
    if ( P( hi ) < bv )
    {
        bv = P( hi );
        bx = hi;
    }
    double       roots[ 2 ];
    int          nr = 0;
    const double A = 3.0 * a, B = 2.0 * b, C = g0;
    if ( std::fabs( A ) < 1e-300 )
    {
        if ( std::fabs( B ) > 1e-300 )
            roots[ nr++ ] = -C / B;
    }
    else
    {
        const double disc = B * B - 4.0 * A * C;
        if ( disc >= 0 )
        {
            roots[ nr++ ] = ( -B + std::sqrt( disc ) ) / ( 2.0 * A );
            roots[ nr++ ] = ( -B - std::sqrt( disc ) ) / ( 2.0 * A );
        }
    }
    for ( int i = 0; i < nr; i++ )
        if ( roots[ i ] > lo && roots[ i ] < hi && P( roots[ i ] ) < bv )
        {
            bv = P( roots[ i ] );
            bx = roots[ i ];
        }
    return bx;
}
} // namespace ll_solver

inline void Solve( const Solver::Options &opt, Problem *problem, Solver::Summary *summary )
{
    using namespace ll_solver;
    Layout    L = make_layout( problem );
    const int n = L.nlocal;
    bool      constrained = false;
    for ( auto &pb : problem->params_ )
        for ( int i = 0; i < pb.size; i++ )
            if ( pb.lo[ i ] > -std::numeric_limits<double>::max() || pb.hi[ i ] < std::numeric_limits<double>::max() )
                constrained = true;

    std::vector<double> x( L.nglobal ), cand, zero( n, 0.0 );
    for ( size_t b = 0; b < problem->params_.size(); b++ )
        for ( int i = 0; i < problem->params_[ b ].size; i++ )
            x[ L.goff[ b ] + i ] = problem->params_[ b ].ptr[ i ];
    auto norm = []( const std::vector<double> &v ) {
        double s = 0;
        for ( double e : v )
            s += e * e;
        return std::sqrt( s );
    };
    auto write_back = [&]( const std::vector<double> &v ) {
        for ( size_t b = 0; b < problem->params_.size(); b++ )
            for ( int i = 0; i < problem->params_[ b ].size; i++ )
                problem->params_[ b ].ptr[ i ] = v[ L.goff[ b ] + i ];
    };

    *summary = Solver::Summary();
    summary->num_residual_blocks = problem->NumResidualBlocks();

    // IterationZero
    if ( constrained )
    {
        plus( problem, L, x, zero, cand );
        x = cand;
    }
    double              x_norm = norm( x );
    double              cost = 0;
    std::vector<double> r, J, g;
    evaluate( problem, L, x, &cost, &r, &J, &g );
    const int           rows = ( int ) r.size();
    std::vector<double> scale( n, 1.0 );
    if ( opt.jacobi_scaling )
        for ( int j = 0; j < n; j++ )
        {
            double s = 0;
            for ( int i = 0; i < rows; i++ )
                s += J[ ( size_t ) i * n + j ] * J[ ( size_t ) i * n + j ];
            scale[ j ] = 1.0 / ( 1.0 + std::sqrt( s ) );
        }
    auto scale_columns = [&]( std::vector<double> &JJ ) {
        for ( int i = 0; i < rows; i++ )
            for ( int j = 0; j < n; j++ )
                JJ[ ( size_t ) i * n + j ] *= scale[ j ];
    };
    scale_columns( J );
    auto gradient_max_norm = [&]( const std::vector<double> &xx, const std::vector<double> &gg ) {
        std::vector<double> ng( n ), xp;
        for ( int j = 0; j < n; j++ )
            ng[ j ] = -gg[ j ];
        plus( problem, L, xx, ng, xp );
        double m = 0;
        for ( int i = 0; i < L.nglobal; i++ )
            m = std::max( m, std::fabs( xx[ i ] - xp[ i ] ) );
        return m;
    };
    double gmax = gradient_max_norm( x, g );
    summary->initial_cost = cost;
    summary->final_cost = cost;
    write_back( x );
    if ( rows == 0 )
        return;

    double              radius = opt.initial_trust_region_radius, decrease_factor = 2.0;
    bool                reuse_diagonal = false;
    int                 invalid_steps = 0, iteration = 0;
    std::vector<double> diag( n, 0.0 );

    for ( ;; )
    {
        if ( iteration >= opt.max_num_iterations )
            break;
        if ( gmax <= opt.gradient_tolerance )
            break;
        if ( radius <= opt.min_trust_region_radius )
            break;
        iteration++;
        summary->ll_iterations = iteration;

        // LevenbergMarquardtStrategy::ComputeStep: (J'J + D'D) y = J'r on the scaled Jacobian, step = -y
        std::vector<double> A( ( size_t ) n * n, 0.0 ), rhs( n, 0.0 ), y;
        for ( int i = 0; i < rows; i++ )
            for ( int a = 0; a < n; a++ )
            {
                const double ja = J[ ( size_t ) i * n + a ];
                rhs[ a ] += ja * r[ i ];
                for ( int b = 0; b < n; b++ )
                    A[ a * n + b ] += ja * J[ ( size_t ) i * n + b ];
            }
        if ( !reuse_diagonal )
            for ( int j = 0; j < n; j++ )
                diag[ j ] = std::min( std::max( A[ j * n + j ], opt.min_lm_diagonal ), opt.max_lm_diagonal );
        std::vector<double> Ad = A;
        for ( int j = 0; j < n; j++ )
            Ad[ j * n + j ] += diag[ j ] / radius;
        const bool ok = cholesky_solve( n, Ad, rhs, y );
        reuse_diagonal = true;
        double              model_cost_change = 0;
        std::vector<double> step( n, 0.0 ), delta( n, 0.0 );
        if ( ok )
        {
            for ( int j = 0; j < n; j++ )
                step[ j ] = -y[ j ];
            // model_residuals = J step; change = -model_residuals . (r + model_residuals / 2)
            for ( int i = 0; i < rows; i++ )
            {
                double m = 0;
                for ( int j = 0; j < n; j++ )
                    m += J[ ( size_t ) i * n + j ] * step[ j ];
                model_cost_change -= m * ( r[ i ] + m / 2.0 );
            }
        }
        if ( !ok || !( model_cost_change > 0.0 ) )
        {
            if ( ++invalid_steps >= opt.max_num_consecutive_invalid_steps )
                break;
            radius *= 0.5;
            reuse_diagonal = true;
            summary->num_unsuccessful_steps++;
            continue;
        }
        invalid_steps = 0;
        for ( int j = 0; j < n; j++ )
            delta[ j ] = step[ j ] * scale[ j ];

        double              cand_cost = 0;
        std::vector<double> cg;
        if ( constrained )
        {
            // projected ARMIJO line search along delta, first trial step 1
            double gd = 0, dmax = 0;
            for ( int j = 0; j < n; j++ )
            {
                gd += g[ j ] * delta[ j ];
                dmax = std::max( dmax, std::fabs( delta[ j ] ) );
            }
            double alpha = 1.0, f_cur = 0;
            std::vector<double> sd( n ), sx;
            auto                eval_at = [&]( double a, double *f, double *dg ) {
                for ( int j = 0; j < n; j++ )
                    sd[ j ] = delta[ j ] * a;
                plus( problem, L, x, sd, sx );
                std::vector<double> gg;
                bool                okk = evaluate( problem, L, sx, f, nullptr, nullptr, &gg );
                *dg = 0;
                for ( int j = 0; j < n; j++ )
                    *dg += gg[ j ] * delta[ j ];
                return okk && std::isfinite( *f );
            };
            double dg_cur = 0;
            bool   valid = eval_at( alpha, &f_cur, &dg_cur );
            int    it = 0;
            bool   success = true;
            bool   prev_valid = false;  // `previous` of ArmijoLineSearch::DoSearch
            double prev_x = 0, prev_f = 0, prev_g = 0;
            while ( !valid || f_cur > cost + 1e-4 * gd * alpha )
            {
                if ( ++it >= 20 )
                {
                    success = false;
                    break;
                }
                double na;
                if ( !valid )
                    na = std::min( std::max( alpha * 0.5, 1e-3 * alpha ), 0.6 * alpha );
                else if ( prev_valid )
                    na = quintic_min( cost, gd, alpha, f_cur, dg_cur, prev_x, prev_f, prev_g, 1e-3 * alpha, 0.6 * alpha );
                else
                    na = cubic_min( cost, gd, alpha, f_cur, dg_cur, 1e-3 * alpha, 0.6 * alpha );
                if ( na * dmax < 1e-9 )
                {
                    success = false;
                    break;
                }
                prev_valid = valid;
                prev_x = alpha;
                prev_f = f_cur;
                prev_g = dg_cur;
                alpha = na;
                valid = eval_at( alpha, &f_cur, &dg_cur );
            }
            if ( success )
                for ( int j = 0; j < n; j++ )
                    delta[ j ] *= alpha;
        }
        plus( problem, L, x, delta, cand );
        if ( !evaluate( problem, L, cand, &cand_cost, nullptr, nullptr, nullptr ) || !std::isfinite( cand_cost ) )
            cand_cost = std::numeric_limits<double>::max();

        // ParameterToleranceReached
        double step_norm = 0;
        for ( int i = 0; i < L.nglobal; i++ )
            step_norm += ( x[ i ] - cand[ i ] ) * ( x[ i ] - cand[ i ] );
        step_norm = std::sqrt( step_norm );
        if ( step_norm <= opt.parameter_tolerance * ( x_norm + opt.parameter_tolerance ) )
            break;
        // FunctionToleranceReached
        const double cost_change = cost - cand_cost;
        if ( std::fabs( cost_change ) <= opt.function_tolerance * cost )
            break;

        const double relative_decrease = cost_change / model_cost_change;
        if ( relative_decrease > opt.min_relative_decrease )
        {
            x = cand;
            x_norm = norm( x );
            evaluate( problem, L, x, &cost, &r, &J, &g );
            scale_columns( J );
            gmax = gradient_max_norm( x, g );
            if ( cost < summary->final_cost )
            {
                summary->final_cost = cost;
                write_back( x );
            }
            const double t = 2.0 * relative_decrease - 1.0;
            radius = radius / std::max( 1.0 / 3.0, 1.0 - t * t * t );
            radius = std::min( opt.max_trust_region_radius, radius );
            decrease_factor = 2.0;
            reuse_diagonal = false;
            summary->num_successful_steps++;
        }
        else
        {
            radius = radius / decrease_factor;
            decrease_factor *= 2.0;
            reuse_diagonal = true;
            summary->num_unsuccessful_steps++;
        }
    }
}
} // namespace ceres
#endif
