// mgk_kernels3.hip -- round 3: three-sweep passes (2-D independent-wave forms; 3-D forms below).  Same build flags as mgk_kernels.hip
// (-O3 -ffp-contract=off, gfx950 only); shares mgk_dev.hpp with it.
#include "mgk_dev.hpp"
#include <type_traits>

// ------------------------------------------------------------------------------------------
// Round 3: THREE Richardson+Jacobi sweeps in one pass, 2-D (src/solver.c:1531 / :1536 / :1542 with max_it = 3):
//     plain      unew = J(J(J(u)))                               24 B per unknown instead of 24 + 24 (sweep + two-sweep pass)
//     NORM       ... and || b - A u ||^2 of the INPUT field (the first stage forms that residual anyway): closes cycle k (:1545-1546) and
//                makes ALL pre-smoothing sweeps of cycle k+1 (:1531); adopted only if a next cycle runs
//     ZG         from the zero guess: the first sweep is pointwise, u1 = scale * (b * dinv); u is not read: 8 + 8 B (:1536 on a coarse level)
//     PRO        unew = J(J(J(u + P uc))): prolongation, correction and ALL post-smoothing sweeps (:1540-1542): 25 B
// With these a V(3,3) cycle makes three passes over a level -- (N + S1 S2 S3)(R)(P + S1' S2' S3') = 24 + 18 + 25 = 67 B on the fine
// level, (J3 from b)(R)(P + 3 sweeps) = 16 + 18 + 25 = 59 B below -- instead of four (99 / 91 B).
// Structure of k_pj2d / k_rr2d: every WAVE is independent (no LDS, no barrier).  Lane l holds the column pair x0 = 2 (60 tx + l - 2);
// x neighbours come by whole-wavefront DPP shifts.  A pair is two columns but a stage consumes ONE column of halo, so three stages
// need two halo lanes a side: stage 1 is right on the columns c0+1 .. c0+126 of the wave's 128, stage 2 on c0+2 .. c0+125, stage 3 on
// c0+3 .. c0+124: lanes 2 .. 61 store (120 of 128 columns; the tiles overlap by four lanes).  The wave marches along y with the rows
// t+1 .. t+3 of u, t .. t+2 of the first sweep and t-1 .. t+1 of the second in registers: at step t it makes the first sweep of row
// t+2, the second of row t+1, the third of row t.  A chunk [y0, y1) starts four steps early (first sweeps only, then first and
// second: wave-uniform branches) so that its three-row windows are filled; the rows it recomputes come from L2.
// Every stage evaluates the expression of k_stencil<MODE_JACOBI> (ascending-column sum, no FMA) and forces the positions outside
// the grid to zero (the homogeneous Dirichlet ring every sweep sees): the result equals three separate sweeps bit for bit.
// Stretched meshes: per-row coefficient tables (ctab / dtab) instead of the launch constants, as in the other 2-D kernels.
// ------------------------------------------------------------------------------------------
struct J3dArgs {
    const double *u, *b, *uc;
    double *out;
    int nx, ny, nxc, nyc;
    long rs, crs;
    int ntx, yc, nwaves, xcd, bous, plainst;
    double a0, a2, a3, a4, a6, dinv, scale;
    const double *ctab, *dtab;
    double *partials;               // NORM: one partial of || b - A u ||^2 per wave
    double *rout;                   // NORM, optional: b - A u of the input field is also STORED (the drop-in's KSPBuildResidual + VecNorm)
};
// YC = 0: a chunk of a.yc rows per wave, marching with the next row in flight (the big levels: long streams).
// YC > 0 (even): a chunk of exactly YC rows, EVERY load of the chunk issued before the first sweep and the marching loop fully unrolled
// (the levels that fit the caches, 127^2 .. 1023^2: they are short of waves, not of bandwidth -- a wave that waits for one row per
// step spends 0.7 us per step; with all its rows requested at once it pays the latency once).
template <bool PRO, bool ZG, bool NORM, bool TAB, int YC>
__global__ void __launch_bounds__(256) k_jacobi3_2d(const J3dArgs a) {
    using VT = V16<double>;
    const int lane = threadIdx.x & 63;
    // workgroups are dealt round-robin over the 8 XCDs: give every XCD a CONTIGUOUS range of wave tiles, so that the rows and columns
    // neighbouring tiles share are re-read from the L2 they were first brought into (the grid is padded to a multiple of 8 blocks)
    int bid = blockIdx.x;
    if (a.xcd) bid = (bid & 7) * (gridDim.x >> 3) + (bid >> 3);
    // (readfirstlane: the wave number, and with it every row index and row address below, lives on the scalar unit -- the row tables of the
    // stretched meshes are then read by scalar loads, not by 64 lanes loading one address)
    const int wid = __builtin_amdgcn_readfirstlane(bid * 4 + (threadIdx.x >> 6));
    const int tx = wid % a.ntx, cy = wid / a.ntx;
    const int yc = YC > 0 ? YC : a.yc;
    const int y0 = cy * yc, y1 = min(y0 + yc, a.ny);
    if (y0 >= y1) { if (NORM && lane == 0) a.partials[wid] = 0.0; return; }      // whole wave
    const int pidx = tx * 60 + lane - 2;                      // pair index (= coarse column of the pair's odd fine column)
    const int x0 = 2 * pidx;
    const bool xin = (x0 >= 0 && x0 < a.nx);
    const bool lastvec = (x0 + 2 > a.nx);
    const bool store = (lane >= 2 && lane <= 61 && xin);
    const int xc = min(max(x0, 0), a.nx - 1);
    const double *__restrict__ up_ = a.u + xc;
    const double *__restrict__ bp_ = a.b + xc;
    const double *__restrict__ cp_ = PRO ? a.uc + min(max(pidx, -1), a.nxc) : nullptr;
    double *__restrict__ op_ = a.out + x0;
    const VT Z = v16_zero<double>();
    const bool nts = a.plainst == 2 || (a.plainst == 0 && mgk_store_nt_2d(a.ny, a.rs));       // non-temporal stores: big fields only (mgk_dev.hpp)
    // a row of a field as it counts for a sweep: zero outside the grid (rows -1 / ny and everything beyond, columns < 0 and >= nx)
    auto fix = [&](VT v, int y) -> VT {
        if (!xin || y < 0 || y >= a.ny) { v.v[0] = 0.0; v.v[1] = 0.0; }
        if (lastvec) v.v[1] = 0.0;
        return v;
    };
    auto ldraw = [&](int y) -> VT { return *reinterpret_cast<const VT *>(up_ + (long)min(max(y, -1), a.ny) * a.rs); };
    // (b stays a non-temporal load at every size: ordinary loads measured 4 us slower per pass at 4095^2, a wash below)
    auto ldbraw = [&](int y) -> VT { return ldv_stream(bp_ + (long)min(max(y, 0), a.ny - 1) * a.rs, true); };
    auto ldc = [&](int ic) -> double { return PRO ? cp_[(long)min(max(ic, -1), a.nyc) * a.crs] : 0.0; };
    auto pA = [&](int y) { return (y & 1) ? (y - 1) >> 1 : (y >> 1) - 1; };
    auto pB = [&](int y) { return (y & 1) ? (y - 1) >> 1 : (y >> 1); };
    // u + P uc on row y (k_pj2d: terms in the order of the prolongation's row, weights w = wi * wj as there)
    auto correct = [&](VT v, int y, double cA, double cB) -> VT {
        if (PRO) {
            const bool two = ((y & 1) == 0);
            const double wi = two ? 0.5 : 1.0;
            const double wh = wi * 0.5, w1 = wi * 1.0;
            const double mA = lane_up<true>(cA), mB = lane_up<true>(cB);
            double s0 = 0.0, s1 = 0.0;
            s0 += wh * mA; s0 += wh * cA; s1 += w1 * cA;
            if (two) { s0 += wh * mB; s0 += wh * cB; s1 += w1 * cB; }
            v.v[0] = v.v[0] + s0; v.v[1] = v.v[1] + s1;
        }
        return fix(v, y);
    };
    // coefficients of grid row y: the launch constants, or row y of the tables (stretched meshes).  The row index is wave-uniform (readfirstlane
    // above), so these are scalar loads; a step requests the sets of its three stages at its top, before it waits for its vector loads.  (As
    // first written -- loaded inside each stage, with a row index the compiler could not prove uniform -- they were 18 vector loads per step,
    // each waited for where it was used: 4097^2 -mesh 1 138 / 130 us per pass against 85 / 89 us on the uniform mesh.)
    struct K6 { double k0, k2, k3, k4, k6, kd; };
    auto ldk = [&](int y) -> K6 {
        K6 k = {a.a0, a.a2, a.a3, a.a4, a.a6, a.dinv};
        if (TAB) {
            // (the constant address space: the tables are read-only for the life of the kernel, which the compiler cannot see through the
            // pointers of the argument struct -- without it, loads with a uniform address still go through the vector memory path)
            const int yy_ = min(max(y, 0), a.ny - 1);
            const CDBL4 *cr_ = (const CDBL4 *)(a.ctab + 5 * (long)yy_);
            const CDBL4 *dr_ = (const CDBL4 *)(a.dtab + yy_);
            k.k0 = cr_[0]; k.k2 = cr_[1]; k.k3 = cr_[2]; k.k4 = cr_[3]; k.k6 = cr_[4]; k.kd = dr_[0];
        }
        return k;
    };
#define J3_COEFS(kk)                                                                                                  \
    const double k0 = TAB ? (kk).k0 : a.a0, k2 = TAB ? (kk).k2 : a.a2, k3 = TAB ? (kk).k3 : a.a3, k4 = TAB ? (kk).k4 : a.a4,    \
                 k6 = TAB ? (kk).k6 : a.a6, kd = TAB ? (kk).kd : a.dinv;
    // one sweep of row y from the rows lo / c / hi of the previous iterate
    double nacc = 0.0;
    auto sweep = [&](const VT &lo, const VT &c, const VT &hi, const VT &bb, int y, bool norm, const K6 &kk) -> VT {
        J3_COEFS(kk)
        const double Wv = lane_up<true>(c.v[1]), Ev = lane_dn<true>(c.v[0]);
        VT o, rr;
#pragma unroll
        for (int e = 0; e < 2; e++) {
            const double wv = (e == 0) ? Wv : c.v[0];
            const double ev = (e == 1) ? Ev : c.v[1];
            double s = k0 * lo.v[e];
            s = s + k2 * wv;
            s = s + k3 * c.v[e];
            s = s + k4 * ev;
            s = s + k6 * hi.v[e];
            const double res = bb.v[e] - s;
            const double zz = res * kd;
            o.v[e] = c.v[e] + a.scale * zz;
            rr.v[e] = (lastvec && e == 1) ? 0.0 : res;
            if (NORM && norm) nacc += (store && y >= y0 && y < y1 && !(lastvec && e == 1)) ? res * res : 0.0;     // the points this wave owns
        }
        if (NORM && norm && a.rout && store && y >= y0 && y < y1) stv_policy(a.rout + (long)y * a.rs + x0, rr, nts);
        return fix(o, y);
    };
    auto sweep0 = [&](const VT &bb, int y, const K6 &kk) -> VT {            // first sweep from the zero guess (k_jacobi_zero)
        J3_COEFS(kk)
        (void)k0; (void)k2; (void)k3; (void)k4; (void)k6;
        VT o;
#pragma unroll
        for (int e = 0; e < 2; e++) { const double zz = bb.v[e] * kd; o.v[e] = a.scale * zz; }
        return fix(o, y);
    };
    // one marching step: first sweep of row t+2 (from the corrected u rows ua, ub and the raw row ur = row t+3 with its parents), second
    // sweep of row t+1, third sweep of row t.  The rows a step hands to the next one live in rings of two (u, first sweep, second sweep) and
    // three (b) registers whose ROLES rotate with the phase K = step mod 6 instead of their contents being copied: the marching loop is
    // unrolled by six by hand (the compiler declines), every index below is a constant after inlining (round 3: 54 of the 171 vector
    // instructions of a step were register copies; the 2-D passes are bound by instruction issue, not by HBM)
    VT UU[2] = {Z, Z};                                        // phase K: ua = UU[K & 1] (row t+1), ub = UU[(K + 1) & 1] (row t+2)
    VT BB[3] = {Z, Z, Z};                                     // b0 = BB[K % 3] (row t), b1 = BB[(K + 1) % 3], b2 = BB[(K + 2) % 3]
    VT PP[2] = {Z, Z};                                        // first sweep: p0 = PP[K & 1] (row t), p1 = PP[(K + 1) & 1]
    VT QQ[2] = {Z, Z};                                        // second sweep: q0 = QQ[K & 1] (row t-1), q1 = QQ[(K + 1) & 1]
    // REV: the chunk is marched DOWNWARDS (logical step t works on the physical row y0 + y1 - 1 - t): neighbouring chunks then touch the rows
    // they share at the same time (a forward chunk ends where the reversed chunk above it ends, and starts where the one below starts), so
    // that the second reader finds them in L2.  The physical row above (coefficient a0) is then the logically NEXT row: the sweeps take
    // their rows in swapped order, the per-point expression is the same
    auto stepg = [&](auto revc, const int K, int t, const VT &ur, double cA, double cB, const VT &bnext) {
        constexpr bool REV = decltype(revc)::value;
        auto ph = [&](int tt) -> int { return REV ? (y0 + y1 - 1 - tt) : tt; };
        const K6 kc2 = ldk(ph(t + 2)), kc1 = ldk(ph(t + 1)), kc0 = ldk(ph(t));
        VT &ua = UU[K & 1], &ub = UU[(K + 1) & 1];
        VT &p0 = PP[K & 1], &p1 = PP[(K + 1) & 1];
        VT &q0 = QQ[K & 1], &q1 = QQ[(K + 1) & 1];
        VT &b0 = BB[K % 3], &b1 = BB[(K + 1) % 3], &b2 = BB[(K + 2) % 3];
        VT p2;
        if (ZG) p2 = sweep0(b2, ph(t + 2), kc2);
        else {
            const VT uc = correct(ur, ph(t + 3), cA, cB);
            p2 = REV ? sweep(uc, ub, ua, b2, ph(t + 2), true, kc2) : sweep(ua, ub, uc, b2, ph(t + 2), true, kc2);
            ua = uc;                                          // (the next phase's ub)
        }
        if (t >= y0 - 2) {                                    // wave-uniform
            const VT q2 = REV ? sweep(p2, p1, p0, b1, ph(t + 1), false, kc1) : sweep(p0, p1, p2, b1, ph(t + 1), false, kc1);
            if (t >= y0 && t < y1) {
                const VT o = REV ? sweep(q2, q1, q0, b0, ph(t), false, kc0) : sweep(q0, q1, q2, b0, ph(t), false, kc0);
                if (store) stv_policy(op_ + (long)ph(t) * a.rs, o, nts);
            }
            q0 = q2;                                          // (the next phase's q1; before the first second sweep both are zero)
        }
        p0 = p2;
        b0 = fix(bnext, ph(t + 3));                           // (the next phase's b2)
    };
    const int t0 = y0 - 4;
    if constexpr (YC > 0) {
        // rows t0+1 .. t0+YC+6 of u, t0+2 .. t0+YC+6 of b, and the coarse rows that are their parents: requested together
        constexpr int NU = YC + 6, NC = NU / 2 + 2;
        VT U[NU], B[NU];
        double Cc[NC];
        const int c0 = pA(t0 + 1);                            // first coarse row needed
#pragma unroll
        for (int q = 0; q < NU; q++) { if (!ZG) U[q] = ldraw(t0 + 1 + q); B[q] = ldbraw(t0 + 1 + q); }
#pragma unroll
        for (int q = 0; q < NC; q++) Cc[q] = ldc(c0 + q);
        // t0 = cy * YC - 4 is even, so the parents of row t0 + k are the coarse rows c0 + (k - 1) / 2 (k odd) or c0 + k / 2 - 1 and
        // c0 + k / 2 (k even) with c0 = t0 / 2: every index below is a compile-time constant once the loop is unrolled
        if (!ZG) {
            UU[0] = correct(U[0], t0 + 1, Cc[0], Cc[0]);
            UU[1] = correct(U[1], t0 + 2, Cc[0], Cc[1]);
        }
        BB[2] = fix(B[1], t0 + 2);
#pragma unroll
        for (int sidx = 0; sidx < YC + 4; sidx++) {
            const int k = sidx + 3;                           // the raw row of this step is row t0 + k
            const int ia = (k & 1) ? (k - 1) / 2 : k / 2 - 1;
            const int ib = (k & 1) ? (k - 1) / 2 : k / 2;
            stepg(std::false_type{}, sidx % 6, t0 + sidx, ZG ? Z : U[sidx + 2], Cc[ia], Cc[ib], B[sidx + 2]);
        }
    } else {
        auto march = [&](auto revc) {
            constexpr bool REV = decltype(revc)::value;
            auto ph = [&](int tt) -> int { return REV ? (y0 + y1 - 1 - tt) : tt; };
            VT UR[2] = {Z, Z}, BN[2];                         // the loads a step consumes are requested a step ahead: two sets, roles by phase
            double CA[2] = {0.0, 0.0}, CB[2] = {0.0, 0.0};
            if (!ZG) {
                UU[0] = correct(ldraw(ph(t0 + 1)), ph(t0 + 1), ldc(pA(ph(t0 + 1))), ldc(pB(ph(t0 + 1))));
                UU[1] = correct(ldraw(ph(t0 + 2)), ph(t0 + 2), ldc(pA(ph(t0 + 2))), ldc(pB(ph(t0 + 2))));
                UR[0] = ldraw(ph(t0 + 3));
                CA[0] = ldc(pA(ph(t0 + 3))); CB[0] = ldc(pB(ph(t0 + 3)));
            }
            BB[2] = fix(ldbraw(ph(t0 + 2)), ph(t0 + 2));
            BN[0] = ldbraw(ph(t0 + 3));
            auto S = [&](const int K, int t) {
                if (!ZG) { UR[(K + 1) & 1] = ldraw(ph(t + 4)); CA[(K + 1) & 1] = ldc(pA(ph(t + 4))); CB[(K + 1) & 1] = ldc(pB(ph(t + 4))); }
                BN[(K + 1) & 1] = ldbraw(ph(t + 4));
                stepg(revc, K, t, UR[K & 1], CA[K & 1], CB[K & 1], BN[K & 1]);
            };
            int t = t0;
            for (; t + 6 <= y1; t += 6) { S(0, t); S(1, t + 1); S(2, t + 2); S(3, t + 3); S(4, t + 4); S(5, t + 5); }
            if (t < y1) { S(0, t); t++; }
            if (t < y1) { S(1, t); t++; }
            if (t < y1) { S(2, t); t++; }
            if (t < y1) { S(3, t); t++; }
            if (t < y1) { S(4, t); t++; }
        };
        if (a.bous && ((cy & 1) || a.bous == 2)) march(std::true_type{}); else march(std::false_type{});
    }
    if (NORM) {
        const double sw = wave_sum(nacc);
        if (lane == 0) a.partials[wid] = sw;
    }
#undef J3_COEFS
}
// shapes: any 2-D grid (nx odd, as everywhere)
template <bool PRO, bool ZG, bool NORM>
static int jacobi3_2d(mgk_ctx *c, const mgk_geom *g, const mgk_geom *gc, const double *coef, double dinv, double scale, const double *ctab, const double *dtab,
                      const double *b, const double *uc, const double *u, double *unew, void *stream, int *norm_parts, double *rout = nullptr) {
    if (!c || !g || g->dim != 2 || (!coef && !ctab) || (ctab && !dtab) || !b || !unew || (!ZG && (!u || u == unew)) || b == unew)
        return fail(MGK_EINVAL, "mgk_jacobi3_2d: bad arguments (2-D)");
    J3dArgs a; memset(&a, 0, sizeof(a));
    a.u = ZG ? nullptr : u + g->org; a.b = b + g->org; a.out = unew + g->org;
    a.nx = g->nx; a.ny = g->ny; a.rs = g->pitch;
    if (PRO) {
        if (!gc || !uc || gc->dim != 2 || g->nx != 2 * gc->nx + 1 || g->ny != 2 * gc->ny + 1) return fail(MGK_EINVAL, "mgk_prolong_jacobi3_2d: coarse grid does not match");
        a.uc = uc + gc->org; a.nxc = gc->nx; a.nyc = gc->ny; a.crs = gc->pitch;
    }
    if (coef) { a.a0 = coef[0]; a.a2 = coef[1]; a.a3 = coef[2]; a.a4 = coef[3]; a.a6 = coef[4]; }
    a.dinv = dinv; a.scale = scale; a.ctab = ctab; a.dtab = dtab;
    if (rout) { if (!NORM || rout == unew || rout == u || rout == b) return fail(MGK_EINVAL, "mgk_jacobi3_2d_sumsq_store_f64: bad arguments"); a.rout = rout + g->org; }
    a.ntx = ((g->nx + 1) / 2 + 59) / 60;                      // pairs 0 .. (nx-1)/2
    // levels that fit the caches (rows of <= 1024): short chunks with every load up front (they lack waves, not bandwidth); the big
    // levels: ~4096 waves (16 per CU) marching over long chunks -- a chunk pays four warm-up steps (two sweep-equivalents) and re-reads
    // six rows.  Tuning variants 50 / 51 / 52 force the marching form / chunks of 4 / chunks of 8 rows.
    int ycs = (g->nx + 1 <= 1024) ? 4 : 0;
    if (g_variant == 50) ycs = 0; else if (g_variant == 51) ycs = 4; else if (g_variant == 52) ycs = 8;
    if (g_zchunk > 0 && g_variant != 51 && g_variant != 52) ycs = 0;      // an explicit chunk length: the marching form
    long nch = (4096 + a.ntx - 1) / a.ntx;
    if (g_zchunk > 0) nch = (g->ny + g_zchunk - 1) / g_zchunk;
    int yc = (int)((g->ny + nch - 1) / nch);
    if (yc < 12 && g_zchunk <= 0) yc = 12;
    if (yc > g->ny) yc = g->ny;
    if (ycs) yc = ycs;
    long waves = (long)a.ntx * ((g->ny + yc - 1) / yc);
    if (NORM && ((waves + 3) / 4 + 7) * 4 > c->max_partials) {    // one partial per wave: longer chunks where that would overflow the slots
        if (ycs) { ycs = 0; }
        const long maxch = c->max_partials / a.ntx - 1;
        if (maxch < 1) return fail(MGK_EINVAL, "mgk_jacobi3_2d_sumsq_f64: more waves than partial slots");
        yc = (int)((g->ny + maxch - 1) / maxch);
        waves = (long)a.ntx * ((g->ny + yc - 1) / yc);
    }
    a.yc = yc;
    unsigned nblk = (unsigned)((waves + 3) / 4);
    // XCD-aware tile order where it paid (MI355X, HIP events): the short-chunk form (1023^2: 10.5 -> 9.3 us) and the marching form from
    // 4095^2 on (69 -> 66 us); at 2047^2 the dispatch order was quicker (23.6 against 25.3 us).  53 / 54 force dispatch / XCD order
    a.xcd = (g_variant == 54) || (g_variant != 53 && nblk >= 64 && (ycs != 0 || g->nx >= 4095));
    if (a.xcd) nblk = (nblk + 7u) & ~7u;
    a.nwaves = (int)waves;
    // odd chunks marched downwards (tuning variant 58 only: bit-identical, but at 4095^2 it measured 67-68 us against 64 us with every chunk
    // marching upwards -- the shared rows are served by the Infinity Cache either way)
    a.bous = (ycs == 0 && g_variant == 58) ? 1 : (ycs == 0 && g_variant == 59) ? 2 : 0;      // 59: EVERY chunk downwards
    a.plainst = (g_variant == 60) ? 1 : (g_variant == 61) ? 2 : 0;      // tuning: 60 forces ordinary stores, 61 non-temporal ones; default by field size
    if (NORM) {
        if (!norm_parts || 4L * nblk > c->max_partials) return fail(MGK_EINVAL, "mgk_jacobi3_2d_sumsq_f64: more waves than partial slots");
        a.partials = c->partials;
        *norm_parts = (int)(4 * nblk);
    }
    hipStream_t st = S(c, stream);
#define J3_LAUNCH(TABV, YCV) hipLaunchKernelGGL((k_jacobi3_2d<PRO, ZG, NORM, TABV, YCV>), dim3(nblk), dim3(256), 0, st, a)
    if (ctab) { if (ycs == 4) J3_LAUNCH(true, 4); else if (ycs == 8) J3_LAUNCH(true, 8); else J3_LAUNCH(true, 0); }
    else { if (ycs == 4) J3_LAUNCH(false, 4); else if (ycs == 8) J3_LAUNCH(false, 8); else J3_LAUNCH(false, 0); }
#undef J3_LAUNCH
    HIPCHK(hipGetLastError());
    return 0;
}
extern "C" int mgk_jacobi3_2d_f64(mgk_ctx *c, const mgk_geom *g, const double *coef, double dinv, double scale, const double *ctab, const double *dtab,
                                  const double *b, const double *u, double *unew, void *stream) {
    return jacobi3_2d<false, false, false>(c, g, nullptr, coef, dinv, scale, ctab, dtab, b, nullptr, u, unew, stream, nullptr);
}
extern "C" int mgk_jacobi3_2d_sumsq_f64(mgk_ctx *c, const mgk_geom *g, const double *coef, double dinv, double scale, const double *ctab, const double *dtab,
                                        const double *b, const double *u, double *unew, double *sumsq_host, void *stream) {
    if (!sumsq_host) return fail(MGK_EINVAL, "mgk_jacobi3_2d_sumsq_f64: bad arguments");
    int nparts = 0;
    int rc = jacobi3_2d<false, false, true>(c, g, nullptr, coef, dinv, scale, ctab, dtab, b, nullptr, u, unew, stream, &nparts);
    if (rc) return rc;
    return finish_to_host(c, nparts, 1, S(c, stream), sumsq_host);
}
extern "C" int mgk_jacobi3_2d_sumsq_store_f64(mgk_ctx *c, const mgk_geom *g, const double *coef, double dinv, double scale, const double *ctab, const double *dtab,
                                              const double *b, const double *u, double *unew, double *r, double *sumsq_host, void *stream) {
    if (!sumsq_host || !r) return fail(MGK_EINVAL, "mgk_jacobi3_2d_sumsq_store_f64: bad arguments");
    int nparts = 0;
    int rc = jacobi3_2d<false, false, true>(c, g, nullptr, coef, dinv, scale, ctab, dtab, b, nullptr, u, unew, stream, &nparts, r);
    if (rc) return rc;
    return finish_to_host(c, nparts, 1, S(c, stream), sumsq_host);
}
extern "C" int mgk_jacobi3_2d_zero_f64(mgk_ctx *c, const mgk_geom *g, const double *coef, double dinv, double scale, const double *ctab, const double *dtab,
                                       const double *b, double *unew, void *stream) {
    return jacobi3_2d<false, true, false>(c, g, nullptr, coef, dinv, scale, ctab, dtab, b, nullptr, nullptr, unew, stream, nullptr);
}
extern "C" int mgk_prolong_jacobi3_2d_f64(mgk_ctx *c, const mgk_geom *gf, const mgk_geom *gc, const double *coef, double dinv, double scale,
                                          const double *ctab, const double *dtab, const double *b, const double *uc, const double *u, double *unew, void *stream) {
    return jacobi3_2d<true, false, false>(c, gf, gc, coef, dinv, scale, ctab, dtab, b, uc, u, unew, stream, nullptr);
}

// ------------------------------------------------------------------------------------------
// THREE sweeps in one pass, 3-D (prototype of DESIGN.md section 10.2): independent waves as above, one wave per SIMD so that a lane may
// hold up to 512 registers (VGPR + AGPR).  A wave owns 120 of its 128 columns (two halo lanes a side) x TY rows and marches along z;
// the lane keeps, for its column pair: u on TY+6 rows of the planes t+1 .. t+3 (+ the plane in flight), the first sweep on TY+4 rows of
// the planes t .. t+2, the second sweep on TY+2 rows of the planes t-1 .. t+1, and b of the three planes the stages work on.  y
// neighbours are the lane's own registers, x neighbours DPP shifts, z neighbours the other planes: no LDS, no barrier.  At step t:
// first sweep of plane t+2, second sweep of plane t+1, third sweep of plane t (stored).  The halo rows are recomputed by the
// neighbouring tiles: (TY+4 + TY+2 + TY) / TY row-sweeps per three output sweeps.
// ------------------------------------------------------------------------------------------
struct J33Args {
    const double *u, *b;
    double *out;
    int nx, ny, nz;
    long rs, ms;
    int ntx, nty, ntz, zc, xcd;
    double a0, a1, a2, a3, a4, a5, a6, dinv, scale;
    double *partials;
};
template <int TY, bool NORM, int g_form>
__global__ void __launch_bounds__(256, 1) k_jacobi3_3d(const J33Args a) {
    constexpr int R0 = TY + 6, R1 = TY + 4, R2 = TY + 2;
    using VT = V16<double>;
    const int lane = threadIdx.x & 63;
    int bid = blockIdx.x;
    if (a.xcd) bid = (bid & 7) * (gridDim.x >> 3) + (bid >> 3);
    // everything derived from the wave's number is wave-uniform: keep it on the scalar unit (row / plane offsets in SGPRs)
    const int wid = bid * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int tx = wid % a.ntx, ty = (wid / a.ntx) % a.nty, tz = wid / (a.ntx * a.nty);
    const int z0 = tz * a.zc, z1 = min(z0 + a.zc, a.nz);
    if (tz >= a.ntz || z0 >= z1) { if (NORM && lane == 0) a.partials[wid] = 0.0; return; }
    const int yb = ty * TY;
    const int x0 = 2 * (tx * 60 + lane - 2);
    const bool xin = (x0 >= 0 && x0 < a.nx);
    const bool lastvec = (x0 + 2 > a.nx);
    const bool store = (lane >= 2 && lane <= 61 && xin);
    // lanes left / right of the grid read the zero padding of the row (columns -4, -2 / nx + 1: the pitch leaves room, mgk_geom_init):
    // with the zero ghost rows and planes of a whole grid no loaded value needs a select
    const unsigned lo = (unsigned)(8 * min(x0, a.nx + 1));    // byte offset of the lane's pair in its row (may be negative: -32, -16)
    const char *__restrict__ ub_ = reinterpret_cast<const char *>(a.u);
    const char *__restrict__ bb_ = reinterpret_cast<const char *>(a.b);
    char *__restrict__ ob_ = reinterpret_cast<char *>(a.out);
    const long loff = (long)(int)lo;
    const VT Z = v16_zero<double>();
    // wave-uniform row offsets in bytes (rows outside -1 .. ny are clamped to the zero ghost rows; b has no ghost data: clamped into the grid,
    // the sweeps of rows outside the grid are discarded)
    long uro[R0], bro[R1];
#pragma unroll
    for (int r = 0; r < R0; r++) uro[r] = 8 * (long)min(max(yb - 3 + r, -1), a.ny) * a.rs;
#pragma unroll
    for (int q = 0; q < R1; q++) bro[q] = 8 * (long)min(max(yb - 2 + q, 0), a.ny - 1) * a.rs;
    bool rowok[R1];                                           // rows yb-2 .. yb+TY+1 inside the grid (wave-uniform)
#pragma unroll
    for (int q = 0; q < R1; q++) rowok[q] = (yb - 2 + q >= 0 && yb - 2 + q < a.ny);
    double nacc = 0.0;
    // positions outside the grid count as zero for the next sweep.  Bit masks, not selects: a select on a condition with wave-uniform
    // parts becomes an exec-mask region per value (30 branch regions in the first build: the sweeps of different rows could not be
    // interleaved any more, and with one wave per SIMD nothing else covers the latency of a dependent chain)
    const unsigned long long Mx0 = xin ? ~0ull : 0ull, Mx1 = (xin && !lastvec) ? ~0ull : 0ull;
    auto andm = [](double v, unsigned long long m) -> double { return __longlong_as_double((long long)((unsigned long long)__double_as_longlong(v) & m)); };
    // one sweep of a lane vector; mu: all ones if the row and the plane are inside the grid (wave-uniform), else zero
    auto jac = [&](const VT &dn, const VT &sv, const VT &c, const VT &nv, const VT &upv, const VT &bb, unsigned long long mu, bool own, bool norm) -> VT {
        const double Wv = lane_up<true>(c.v[1]), Ev = lane_dn<true>(c.v[0]);
        VT o;
#pragma unroll
        for (int e = 0; e < 2; e++) {
            const double wv = (e == 0) ? Wv : c.v[0];
            const double ev = (e == 1) ? Ev : c.v[1];
            double s = a.a0 * dn.v[e];
            s = s + a.a1 * sv.v[e];
            s = s + a.a2 * wv;
            s = s + a.a3 * c.v[e];
            s = s + a.a4 * ev;
            s = s + a.a5 * nv.v[e];
            s = s + a.a6 * upv.v[e];
            const double res = bb.v[e] - s;
            const double zz = res * a.dinv;
            const double val = c.v[e] + a.scale * zz;
            const unsigned long long m = (e == 0 ? Mx0 : Mx1) & mu;
            o.v[e] = andm(val, m);
            if (NORM && norm) { const double rm = andm(res, (own && store) ? m : 0ull); nacc += rm * rm; }
        }
        return o;
    };
    auto ldplane = [&](VT (&P)[R0], int z) {
        const char *pz = ub_ + 8 * (long)min(max(z, -1), a.nz) * a.ms + loff;
#pragma unroll
        for (int r = 0; r < R0; r++) P[r] = *reinterpret_cast<const VT *>(pz + uro[r]);
    };
    auto ldb = [&](VT (&P)[R1], int z) {
        const char *pz = bb_ + 8 * (long)min(max(z, 0), a.nz - 1) * a.ms + loff;
#pragma unroll
        for (int q = 0; q < R1; q++) P[q] = ldv_stream(reinterpret_cast<const double *>(pz + bro[q]), true);
    };
    // one sweep of NR rows at once, TERM BY TERM over all rows (the same per-point expression and order as jac): 2 NR independent
    // accumulators, so that a wave that is alone on its SIMD never waits for the previous add of the same chain.
    // rows: C has NR + 2 rows (row q + 1 is the centre of output row q); dn / up / bb start at their offsets o_dn / o_up / o_b
#define J33_STAGE(NR, OUT, DN, o_dn, C, UP, o_up, BB, o_b, MASK, ...)                                            \
    {                                                                                                                  \
        double s_[NR][2], w_[NR], e_[NR];                                                                              \
        _Pragma("unroll") for (int q = 0; q < NR; q++) { w_[q] = lane_up<true>(C[q + 1].v[1]); e_[q] = lane_dn<true>(C[q + 1].v[0]); } \
        _Pragma("unroll") for (int q = 0; q < NR; q++) { s_[q][0] = a.a0 * DN[q + o_dn].v[0]; s_[q][1] = a.a0 * DN[q + o_dn].v[1]; } \
        _Pragma("unroll") for (int q = 0; q < NR; q++) { s_[q][0] = s_[q][0] + a.a1 * C[q].v[0]; s_[q][1] = s_[q][1] + a.a1 * C[q].v[1]; } \
        _Pragma("unroll") for (int q = 0; q < NR; q++) { s_[q][0] = s_[q][0] + a.a2 * w_[q]; s_[q][1] = s_[q][1] + a.a2 * C[q + 1].v[0]; } \
        _Pragma("unroll") for (int q = 0; q < NR; q++) { s_[q][0] = s_[q][0] + a.a3 * C[q + 1].v[0]; s_[q][1] = s_[q][1] + a.a3 * C[q + 1].v[1]; } \
        _Pragma("unroll") for (int q = 0; q < NR; q++) { s_[q][0] = s_[q][0] + a.a4 * C[q + 1].v[1]; s_[q][1] = s_[q][1] + a.a4 * e_[q]; } \
        _Pragma("unroll") for (int q = 0; q < NR; q++) { s_[q][0] = s_[q][0] + a.a5 * C[q + 2].v[0]; s_[q][1] = s_[q][1] + a.a5 * C[q + 2].v[1]; } \
        _Pragma("unroll") for (int q = 0; q < NR; q++) { s_[q][0] = s_[q][0] + a.a6 * UP[q + o_up].v[0]; s_[q][1] = s_[q][1] + a.a6 * UP[q + o_up].v[1]; } \
        _Pragma("unroll") for (int q = 0; q < NR; q++) {                                                               \
            const double r0_ = BB[q + o_b].v[0] - s_[q][0], r1_ = BB[q + o_b].v[1] - s_[q][1];                          \
            const double z0_ = r0_ * a.dinv, z1_ = r1_ * a.dinv;                                                       \
            const double v0_ = C[q + 1].v[0] + a.scale * z0_, v1_ = C[q + 1].v[1] + a.scale * z1_;                     \
            const unsigned long long mu_ = MASK;                                                                       \
            OUT[q].v[0] = andm(v0_, Mx0 & mu_); OUT[q].v[1] = andm(v1_, Mx1 & mu_);                                   \
            __VA_ARGS__                                                                                                \
        }                                                                                                              \
    }
    unsigned long long rowm[R1];                              // all ones for the rows yb-2 .. yb+TY+1 that lie inside the grid
#pragma unroll
    for (int q = 0; q < R1; q++) rowm[q] = rowok[q] ? ~0ull : 0ull;
    const int t0 = z0 - 4;
    // rings of three planes, indexed by the step's phase (t - t0) mod 3 -- the marching loop is unrolled by three, every index below is a
    // compile-time constant, and no plane is ever copied except the two that arrive from memory (they land in Un / Bn: a full step of
    // latency tolerance, then 18 register-pair moves)
    VT U[3][R0], Un[R0];                                      // u:            planes t+1, t+2, t+3 in the slots PH, PH+1, PH+2 (mod 3)
    VT P[3][R1];                                              // first sweep:  planes t, t+1, (new) t+2
    VT Q[3][R2];                                              // second sweep: planes t-1, t, (new) t+1
    VT B[3][R1], Bn[R1];                                      // b:            planes t, t+1, t+2
    ldplane(U[0], t0 + 1); ldplane(U[1], t0 + 2); ldplane(U[2], t0 + 3);
    ldb(B[2], t0 + 2);
#pragma unroll
    for (int k = 0; k < 3; k++) {
#pragma unroll
        for (int q = 0; q < R1; q++) { P[k][q] = Z; if (k < 2) B[k][q] = Z; }
#pragma unroll
        for (int q = 0; q < R2; q++) Q[k][q] = Z;
    }
    auto step = [&](auto phc, int t) {
        constexpr int PH = decltype(phc)::value, S0 = PH % 3, S1 = (PH + 1) % 3, S2 = (PH + 2) % 3;
        ldplane(Un, t + 4);
        ldb(Bn, t + 3);
        const unsigned long long z2m = (t + 2 >= 0 && t + 2 < a.nz) ? ~0ull : 0ull, z1m = (t + 1 >= 0 && t + 1 < a.nz) ? ~0ull : 0ull;
        const bool z2own = (t + 2 >= z0 && t + 2 < z1);
        // first sweep of plane t+2 (rows yb-2 .. yb+TY+1) into the slot the dead plane t-1 held
        if (g_form == 0) {
#pragma unroll
        for (int q = 0; q < R1; q++)
            P[S2][q] = jac(U[S0][q + 1], U[S1][q], U[S1][q + 1], U[S1][q + 2], U[S2][q + 1], B[S2][q], rowm[q] & z2m, z2own && q >= 2 && q < 2 + TY, true);
        if (t >= z0 - 2) {
            // second sweep of plane t+1 (rows yb-1 .. yb+TY)
#pragma unroll
            for (int q = 0; q < R2; q++)
                Q[S2][q] = jac(P[S0][q + 1], P[S1][q], P[S1][q + 1], P[S1][q + 2], P[S2][q + 1], B[S1][q + 1], rowm[q + 1] & z1m, false, false);
            if (t >= z0) {
                // third sweep of plane t (rows yb .. yb+TY-1)
#pragma unroll
                for (int j = 0; j < TY; j++) {
                    const VT o = jac(Q[S0][j + 1], Q[S1][j], Q[S1][j + 1], Q[S1][j + 2], Q[S2][j + 1], B[S0][j + 2], ~0ull, false, false);
                    if (store && rowok[j + 2]) stv_stream(reinterpret_cast<double *>(ob_ + 8 * ((long)t * a.ms + (long)(yb + j) * a.rs) + loff), o);
                }
            }
        }
        } else {
        J33_STAGE(R1, P[S2], U[S0], 1, U[S1], U[S2], 1, B[S2], 0, (rowm[q] & z2m),
                  if (NORM) { const unsigned long long mn_ = (z2own && q >= 2 && q < 2 + TY && store) ? mu_ : 0ull;
                              const double n0_ = andm(r0_, Mx0 & mn_), n1_ = andm(r1_, Mx1 & mn_); nacc += n0_ * n0_; nacc += n1_ * n1_; })
        if (t >= z0 - 2) {
            J33_STAGE(R2, Q[S2], P[S0], 1, P[S1], P[S2], 1, B[S1], 1, (rowm[q + 1] & z1m), ;)
            if (t >= z0) {
                VT O[TY];
                J33_STAGE(TY, O, Q[S0], 1, Q[S1], Q[S2], 1, B[S0], 2, (~0ull), ;)
#pragma unroll
                for (int j = 0; j < TY; j++)
                    if (store && rowok[j + 2]) stv_stream(reinterpret_cast<double *>(ob_ + 8 * ((long)t * a.ms + (long)(yb + j) * a.rs) + loff), O[j]);
            }
        }
        }
        // the planes requested at the top of the step are first touched here: a whole step of latency tolerance
#pragma unroll
        for (int r = 0; r < R0; r++) U[S0][r] = Un[r];         // plane t+4 takes the place of plane t+1 (dead since the first sweep)
#pragma unroll
        for (int q = 0; q < R1; q++) B[S0][q] = Bn[q];         // b of plane t+3 takes the place of b of plane t
    };
    for (int t = t0; t < z1; t += 3) {
        step(std::integral_constant<int, 0>{}, t);
        if (t + 1 < z1) step(std::integral_constant<int, 1>{}, t + 1);
        if (t + 2 < z1) step(std::integral_constant<int, 2>{}, t + 2);
    }
    if (NORM) {
        const double sw = wave_sum(nacc);
        if (lane == 0) a.partials[wid] = sw;
    }
#undef J33_STAGE
}
template <int TY, bool NORM>
static int jacobi3_3d_launch(mgk_ctx *c, J33Args &a, const mgk_geom *g, int *norm_parts, void *stream) {
    a.ntx = ((g->nx + 1) / 2 + 59) / 60;
    a.nty = (g->ny + TY - 1) / TY;
    // one wave per SIMD: 1024 run at a time; cut z so that the wave tiles are a whole number of rounds (>= 4), chunks of >= 32 planes
    const long per = (long)a.ntx * a.nty;
    int ntz = 1;
    if (g_zchunk > 0) ntz = (g->nz + g_zchunk - 1) / g_zchunk;
    else { while (per * ntz < 4096 && g->nz / (ntz + 1) >= 32) ntz++; }
    if (NORM) while (ntz > 1 && per * ntz + 32 > c->max_partials) ntz--;      // one partial per wave
    a.zc = (g->nz + ntz - 1) / ntz;
    a.ntz = (g->nz + a.zc - 1) / a.zc;
    const long waves = per * a.ntz;
    unsigned nblk = (unsigned)((waves + 3) / 4);
    a.xcd = (g_variant != 53 && nblk >= 64) ? 1 : 0;
    if (a.xcd) nblk = (nblk + 7u) & ~7u;
    if (NORM) {
        if (!norm_parts || 4L * nblk > c->max_partials) return fail(MGK_EINVAL, "mgk_jacobi3_sumsq_f64: more waves than partial slots");
        a.partials = c->partials;
        *norm_parts = (int)(4 * nblk);
    }
    if (g_variant == 64) hipLaunchKernelGGL((k_jacobi3_3d<TY, NORM, 0>), dim3(nblk), dim3(256), 0, S(c, stream), a);      // row by row (64)
    else hipLaunchKernelGGL((k_jacobi3_3d<TY, NORM, 1>), dim3(nblk), dim3(256), 0, S(c, stream), a);                       // term by term over the rows
    HIPCHK(hipGetLastError());
    return 0;
}
template <bool NORM>
static int jacobi3_3d(mgk_ctx *c, const mgk_geom *g, const double *coef, double dinv, double scale, const double *b, const double *u, double *unew,
                      void *stream, int *norm_parts) {
    if (!c || !g || g->dim != 3 || !coef || !b || !u || !unew || u == unew || b == unew) return fail(MGK_EINVAL, "mgk_jacobi3_f64: bad arguments (3-D)");
    J33Args a; memset(&a, 0, sizeof(a));
    a.u = u + g->org; a.b = b + g->org; a.out = unew + g->org;
    a.nx = g->nx; a.ny = g->ny; a.nz = g->nz; a.rs = g->pitch; a.ms = g->plane;
    a.a0 = coef[0]; a.a1 = coef[1]; a.a2 = coef[2]; a.a3 = coef[3]; a.a4 = coef[4]; a.a5 = coef[5]; a.a6 = coef[6];
    a.dinv = dinv; a.scale = scale;
    // rows per wave tile: 4 (480-498 registers per lane, none in scratch); with the norm 3 (456): the 4-row form would spill 28-38 registers
    // to scratch.  Tuning variants 62 / 63 force 2 / 3 rows
    const int ty = (g_variant == 62) ? 2 : (g_variant == 63 || NORM) ? 3 : 4;
    if (ty == 2) return jacobi3_3d_launch<2, NORM>(c, a, g, norm_parts, stream);
    if (ty == 3) return jacobi3_3d_launch<3, NORM>(c, a, g, norm_parts, stream);
    return jacobi3_3d_launch<4, NORM>(c, a, g, norm_parts, stream);
}
extern "C" int mgk_jacobi3_f64(mgk_ctx *c, const mgk_geom *g, const double *coef, double dinv, double scale,
                               const double *b, const double *u, double *unew, void *stream) {
    return jacobi3_3d<false>(c, g, coef, dinv, scale, b, u, unew, stream, nullptr);
}
extern "C" int mgk_jacobi3_sumsq_f64(mgk_ctx *c, const mgk_geom *g, const double *coef, double dinv, double scale,
                                     const double *b, const double *u, double *unew, double *sumsq_host, void *stream) {
    if (!sumsq_host) return fail(MGK_EINVAL, "mgk_jacobi3_sumsq_f64: bad arguments");
    int nparts = 0;
    int rc = jacobi3_3d<true>(c, g, coef, dinv, scale, b, u, unew, stream, &nparts);
    if (rc) return rc;
    return finish_to_host(c, nparts, 1, S(c, stream), sumsq_host);
}

// ------------------------------------------------------------------------------------------
// Round 3: primitives of the PEER halo transport (include/mg_comm.h, csrc/mg_comm.c: mg_comm_peer_*).  One process per GPU; every rank
// allocates a mailbox and a block of flag words in FINE-GRAINED device memory (not cached in L2: what a neighbour writes over xGMI is what the
// next load sees), exports them with hipIpc*, and maps its neighbours'.  An exchange is: peer copies of the boundary planes into the
// neighbour's mailbox (hipMemcpyAsync between devices: the copy engines, no workgroup), a ONE-WAVE kernel that stores the exchange's
// sequence number into the neighbour's flag word, a one-wave kernel that waits for the neighbour's number in the own flag word, copies
// mailbox -> ghost planes.  The two flag kernels use ~10 registers: they find a slot beside the one-block-per-CU marching kernels (which
// leave 32 VGPRs per SIMD), where RCCL's send/recv kernel does not.
// ------------------------------------------------------------------------------------------
extern "C" int mgk_ipc_alloc(mgk_ctx *c, size_t bytes, void **ptr, void *handle64) {
    if (!c || !ptr || !handle64 || !bytes) return fail(MGK_EINVAL, "mgk_ipc_alloc: bad arguments");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipExtMallocWithFlags(ptr, bytes, hipDeviceMallocFinegrained));
    HIPCHK(hipMemsetAsync(*ptr, 0, bytes, c->compute));
    HIPCHK(hipStreamSynchronize(c->compute));
    hipIpcMemHandle_t h;
    HIPCHK(hipIpcGetMemHandle(&h, *ptr));
    static_assert(sizeof(h) == MGK_IPC_HANDLE_BYTES, "handle size");
    memcpy(handle64, &h, sizeof(h));
    return 0;
}
extern "C" int mgk_ipc_open(mgk_ctx *c, const void *handle64, void **ptr) {
    if (!c || !ptr || !handle64) return fail(MGK_EINVAL, "mgk_ipc_open: bad arguments");
    HIPCHK(hipSetDevice(c->device));
    hipIpcMemHandle_t h;
    memcpy(&h, handle64, sizeof(h));
    HIPCHK(hipIpcOpenMemHandle(ptr, h, hipIpcMemLazyEnablePeerAccess));
    return 0;
}
extern "C" int mgk_ipc_close(mgk_ctx *c, void *ptr) { (void)c; if (ptr) HIPCHK(hipIpcCloseMemHandle(ptr)); return 0; }
// copy between two device allocations that may live on different GPUs (peer copy: copy engines), queued on `stream`
extern "C" int mgk_peer_copy(mgk_ctx *c, void *dst, const void *src, size_t bytes, void *stream) {
    if (!c || !dst || !src) return fail(MGK_EINVAL, "mgk_peer_copy: bad arguments");
    if (bytes) HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, S(c, stream)));
    return 0;
}
struct FlagArgs { unsigned long long *flag[MGK_PEER_MAX]; int n; unsigned long long value, ticks; unsigned int *status; };
__global__ void __launch_bounds__(64) k_flag_set(const FlagArgs a) {
    if ((int)threadIdx.x < a.n) __hip_atomic_store(a.flag[threadIdx.x], a.value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// waits until every *flag[q] >= value; gives up after `ticks` of the 100 MHz clock and reports in *status (every lane reaches the exit)
__global__ void __launch_bounds__(64) k_flag_wait(const FlagArgs a) {
    if ((int)threadIdx.x >= a.n) return;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__hip_atomic_load(a.flag[threadIdx.x], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < a.value) {
        if (__builtin_amdgcn_s_memrealtime() - t0 > a.ticks) { __hip_atomic_store(a.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); return; }
        __builtin_amdgcn_s_sleep(16);
    }
}
static int flag_args(FlagArgs &a, void *const *flags, int n, unsigned long long value) {
    memset(&a, 0, sizeof(a));
    if (!flags || n < 1 || n > MGK_PEER_MAX) return 1;
    for (int q = 0; q < n; q++) { if (!flags[q]) return 1; a.flag[q] = (unsigned long long *)flags[q]; }
    a.n = n; a.value = value;
    return 0;
}
extern "C" int mgk_flags_set(mgk_ctx *c, void *const *flags, int n, unsigned long long value, void *stream) {
    FlagArgs a;
    if (!c || flag_args(a, flags, n, value)) return fail(MGK_EINVAL, "mgk_flags_set: bad arguments");
    hipLaunchKernelGGL(k_flag_set, dim3(1), dim3(64), 0, S(c, stream), a);
    HIPCHK(hipGetLastError());
    return 0;
}
extern "C" int mgk_flags_wait(mgk_ctx *c, void *const *flags, int n, unsigned long long value, double timeout_s, void *status_u32, void *stream) {
    FlagArgs a;
    if (!c || !status_u32 || !(timeout_s > 0.0) || flag_args(a, flags, n, value)) return fail(MGK_EINVAL, "mgk_flags_wait: bad arguments");
    a.ticks = (unsigned long long)(timeout_s * 1.0e8); a.status = (unsigned int *)status_u32;
    hipLaunchKernelGGL(k_flag_wait, dim3(1), dim3(64), 0, S(c, stream), a);
    HIPCHK(hipGetLastError());
    return 0;
}
int mgk_preload_kernels3() {
    hipFuncAttributes fa;
    HIPCHK(hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(k_flag_set)));
    return 0;
}
extern "C" int mgk_flag_set(mgk_ctx *c, void *flag, unsigned long long value, void *stream) { void *f[1] = {flag}; return mgk_flags_set(c, f, 1, value, stream); }
extern "C" int mgk_flag_wait(mgk_ctx *c, const void *flag, unsigned long long value, double timeout_s, void *status_u32, void *stream) {
    void *f[1] = {const_cast<void *>(flag)};
    return mgk_flags_wait(c, f, 1, value, timeout_s, status_u32, stream);
}
// all-reduce (sum) of n <= 64 doubles between `nranks` processes through their fine-grained slot blocks: rank me stores its values and the
// sequence number into slot `me` of EVERY rank's block (peers[r]: that rank's block as mapped here; one 8-byte store per value),
// waits until all slots of its own block carry the number, and sums them in rank order -- the same bits on every rank.
// Block layout: two buffers (parity of the sequence number: a rank may be one all-reduce ahead of a peer that still sums the last one,
// never two) of nranks slots [seq, v0 .. v63] (65 x 8 bytes each).  One wave.
struct ARArgs { unsigned long long *peers[MGK_PEER_MAX]; int nranks, me, n; unsigned long long seq, ticks; double *vals; unsigned int *status; };
__global__ void __launch_bounds__(64) k_peer_allreduce(const ARArgs a) {
    const int lane = threadIdx.x;
    const int buf = (int)(a.seq & 1ull);
    const double mine = lane < a.n ? a.vals[lane] : 0.0;
    for (int r = 0; r < a.nranks; r++) {
        unsigned long long *slot = a.peers[r] + 65 * (buf * a.nranks + a.me);
        if (lane < a.n) __hip_atomic_store(reinterpret_cast<double *>(slot + 1 + lane), mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    __builtin_amdgcn_s_waitcnt(0);
    if (lane == 0) for (int r = 0; r < a.nranks; r++) __hip_atomic_store(a.peers[r] + 65 * (buf * a.nranks + a.me), a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    const unsigned long long *own = a.peers[a.me] + 65 * buf * a.nranks;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    bool ok = true;
    if (lane < a.nranks) {
        while (__hip_atomic_load(own + 65 * lane, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < a.seq) {
            if (__builtin_amdgcn_s_memrealtime() - t0 > a.ticks) { ok = false; break; }
            __builtin_amdgcn_s_sleep(16);
        }
    }
    if (!ok) __hip_atomic_store(a.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    // every lane's poll has ended (one wave: the loop above is a divergent region that reconverges here)
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    if (lane < a.n) {
        double s = 0.0;
        for (int r = 0; r < a.nranks; r++) s += __hip_atomic_load(reinterpret_cast<const double *>(own + 65 * r + 1 + lane), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        a.vals[lane] = s;
    }
}
extern "C" int mgk_peer_allreduce(mgk_ctx *c, void *const *peer_blocks, int nranks, int me, unsigned long long seq, double *vals_dev, int n,
                                  double timeout_s, void *status_u32, void *stream) {
    if (!c || !peer_blocks || nranks < 1 || nranks > MGK_PEER_MAX || me < 0 || me >= nranks || !vals_dev || n < 1 || n > 64 || !status_u32)
        return fail(MGK_EINVAL, "mgk_peer_allreduce: bad arguments");
    ARArgs a; memset(&a, 0, sizeof(a));
    for (int r = 0; r < nranks; r++) { if (!peer_blocks[r]) return fail(MGK_EINVAL, "mgk_peer_allreduce: unmapped peer"); a.peers[r] = (unsigned long long *)peer_blocks[r]; }
    a.nranks = nranks; a.me = me; a.n = n; a.seq = seq; a.ticks = (unsigned long long)(timeout_s * 1.0e8); a.vals = vals_dev; a.status = (unsigned int *)status_u32;
    hipLaunchKernelGGL(k_peer_allreduce, dim3(1), dim3(64), 0, S(c, stream), a);
    HIPCHK(hipGetLastError());
    return 0;
}
