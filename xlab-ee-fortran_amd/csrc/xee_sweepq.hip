// xee_sweepq.hip -- the temporally blocked Jacobi sweep with the update's division strength-reduced,
// optimistically, and validated per tile.
//
// The reference update (xtt-lib-fortran/elliptic_tools.f90:238) divides by -coe(5): a quantity that does
// not change between iterations.  The IEEE quotient the compiler emits for `a / d` on gfx950 is
//     d' = div_scale(d)   y0 = rcp(d')   e = fma(-d', y0, 1)   y1 = fma(e, y0, y0)        <- depends on d only
//     a' = div_scale(a)   q0 = a'*y1   r0 = fma(-d', q0, a')   q1 = fma(r0, y1, q0)
//     r1 = fma(-d', q1, a')   q2 = div_fmas(r1, y1, q1)   q = div_fixup(q2, d, a)
// (15 issue slots, two of them quarter rate: 41 cycles per point against 2.7 for a multiply -- 38 % of
// the blocked loop, tools/ubench_valu.hip).  When neither div_scale rescales (both operands comfortably
// inside the normal range) d' = d, a' = a, div_fmas is a plain fma and div_fixup returns q2, so
//     y1 once per launch;   q0 = a*y1, r0 = fma(-d,q0,a), q1 = fma(r0,y1,q0), r1 = fma(-d,q1,a), q2 = fma(r1,y1,q1)
// are THE SAME INSTRUCTIONS ON THE SAME OPERANDS: the same bits, at 5 slots per point.  "Comfortably
// inside" is checked, not assumed.  Per launch, tile and lane, with dmin/dmax the extremes of |coe5| over
// the lane's interior points:
//     static   2^-120 <= dmin, dmax <= 2^120
//     dynamic  max(2^-100, dmax*2^-120) <= |alpha*R| <= min(2^100, dmin*2^94) at every interior point
//              and every one of the T steps (running max/min of |alpha*R|: two instructions per point),
//              and every value of the tile finite at the end (non-finite values are sticky under the update)
// which keeps the exponent difference below the 96 where div_scale starts to rescale and quotient,
// reciprocal and remainders normal and exactly representable (tests/test_gpu_quotient.py checks the
// identity over 10^9 operand pairs drawn over the whole admitted range).  A tile that fails any of it (a
// zero residual, an exploding iterate, exotic coefficients) is recomputed from its unchanged inputs by
// the exact body -- same kernel, same workgroup -- so the result is bit-identical to k_sweep_tb4 always.
//
// The same validation covers two more savings:
//  * BAROTROPIC operator (b = 0): its four corner weights are +-0 (elliptic_tools.f90:46,48,52,54 with
//    BX = BY = 0), and `s + (+-0)*x` differs from s only when x is not finite or the partial sum is itself
//    a zero -- in which case either a non-finite value is in the tile at the end, or every later term
//    vanished too and alpha*R = 0 fails the dynamic bound.  The optimistic body sums the five-point star
//    only (5 mul + 5 add instead of 9 + 9).
//  * no select for the points that never update (rim, outside the domain): their weights are set to 0, f
//    to -1 and y1 to the zero whose sign makes q2 = -0, so `psi + q2` returns psi bit for bit and their
//    "residual" alpha*1 stays inside the dynamic bound.
#include "xee_device.h"

#include <cfloat>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#pragma clang fp contract(off)

namespace xee {

namespace {

__device__ __forceinline__ float qshr1(float src)   // lane l <- lane l-1 (lane 0 gets 0)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, src), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float qshl1(float src)   // lane l <- lane l+1 (lane 63 gets 0)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, src), 0x130, 0xf, 0xf, true));
}

// the loop-invariant half of the quotient by d (see the header)
__device__ __forceinline__ float quotient_y1(float d)
{
    const float y0 = __builtin_amdgcn_rcpf(d);
    const float e = __builtin_fmaf(-d, y0, 1.0f);
    return __builtin_fmaf(e, y0, y0);
}
// a / d given y1 = quotient_y1(d) and nd = -d, valid inside the bounds of the header
__device__ __forceinline__ float quotient_q2(float a, float nd, float y1)
{
    const float q0 = a * y1;
    const float r0 = __builtin_fmaf(nd, q0, a);
    const float q1 = __builtin_fmaf(r0, y1, q0);
    const float r1 = __builtin_fmaf(nd, q1, a);
    return __builtin_fmaf(r1, y1, q1);
}

struct TileGeom {
    int lane, w, x0, y0, gx, gy0;
    bool col_in, col_int, vx;
    size_t mo, ks;
    unsigned long long* stamps;     // developer timeline (tools/probe_q.hip) or nullptr
};

// s_memtime at a phase boundary of the tile (STAMP builds only: tools/probe_q.hip)
template <bool STAMP>
__device__ __forceinline__ void stamp(const TileGeom& tg, int slot)
{
    if constexpr (STAMP) {
        unsigned long long t;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : : "memory");
        if (threadIdx.x == 0) tg.stamps[slot] = t;
    }
}

// One tile, T steps.  FAST: the optimistic body; returns whether the WHOLE workgroup validated, and only
// then stores.  !FAST: the exact body (the arithmetic of k_sweep_tb4); always stores.
template <int NW, int R, bool WRITE_R, bool BARO, bool FAST, bool STAMP = false>
__device__ __forceinline__ bool sweep_tile(const Grid& g, const TileGeom& tg, float (*edge)[2][NW][64], int* vote,
        const float* __restrict__ psi_in, float* __restrict__ psi_out, const float* __restrict__ base,
        const float* __restrict__ f, float* __restrict__ res, float alpha, int T,
        const MemberState* __restrict__ member = nullptr /* if given: return at once when it has stopped */)
{
    constexpr int H = NW * R;
    constexpr bool STAR = FAST && BARO;
    constexpr int NC = STAR ? 5 : 9;                    // weights kept: the star (2,4,5,6,8) or all nine
    constexpr int K5 = STAR ? 2 : 4;                    // where coe5 sits in c[]
    const int lane = tg.lane, w = tg.w, gx = tg.gx, gy0 = tg.gy0;
    const size_t mo = tg.mo, ks = tg.ks;

    float c[NC][R], ff[R], p[R], y1[FAST ? R : 1];
    float dmin = 0x1p120f, dmax = 0x1p-120f;            // extremes of |coe5| over this lane's interior points
    {
        // Every load of the tile is issued before the first value is used, unconditionally and from a
        // clamped (always valid) address; what lies outside the domain is replaced afterwards.  (Loading
        // under the `in` predicate makes the compiler branch around each load and wait row by row: eight
        // memory latencies in sequence instead of one.)
        const float* __restrict__ pm = psi_in + mo;
        const float* __restrict__ fm = f + mo;
        const float* __restrict__ apm = base + mo;
        const float* __restrict__ cqm = base + ks + mo;
        const float* __restrict__ bxm = base + 2 * ks + mo;
        const float* __restrict__ bym = base + 3 * ks + mo;
        // (the wave index is scalar -- readfirstlane in the kernel -- so a row's offset is scalar arithmetic and
        // the only per-lane part of an address is the clamped column, shared by all 4R+1 loads)
        const int cx = gx < 0 ? 0 : (gx > g.nx - 1 ? g.nx - 1 : gx);
        auto rowoff = [&](int gy) -> size_t {
            const int cy = gy < 0 ? 0 : (gy > g.ny - 1 ? g.ny - 1 : gy);
            return (size_t)cy * (size_t)g.pitch;
        };
        float rp[R], rf[R], rap[R], rcq[R], rbx[BARO ? 1 : R], rby[BARO ? 1 : R];
        const size_t ob = rowoff(gy0 - 1);
        const float rcq_b = (cqm + ob)[cx];
        const float rby_b = BARO ? 0.0f : (bym + ob)[cx];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const size_t o = rowoff(gy0 + r);
            rp[r] = (pm + o)[cx];
            rf[r] = (fm + o)[cx];
            rap[r] = (apm + o)[cx];
            rcq[r] = (cqm + o)[cx];
            if constexpr (!BARO) { rbx[r] = (bxm + o)[cx]; rby[r] = (bym + o)[cx]; }
        }
        stamp<STAMP>(tg, 1);                                // loads issued
        // (only now the member's `active` flag is looked at: testing it first would put a scalar-memory round
        // trip in front of the tile's loads; for a stopped member the loads above are harmless)
        if (member && !member->active) return true;         // workgroup-uniform
        const float y_idle = alpha > 0.0f ? -0.0f : 0.0f;   // (alpha*1) * y_idle = -0
        const int gyb = gy0 - 1;
        const bool inb = tg.col_in && gyb >= 0 && gyb < g.ny;
        float cq_m = inb ? rcq_b : 0.0f;
        float by_m = (!BARO && inb) ? rby_b : 0.0f;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int gy = gy0 + r;
            const bool in = tg.col_in && gy >= 0 && gy < g.ny;
            const bool interior = tg.col_int && gy >= 1 && gy <= g.ny - 2;
            p[r] = in ? rp[r] : 0.0f;
            ff[r] = interior ? rf[r] : (FAST ? -1.0f : 0.0f);
            const float Ap = in ? rap[r] : 0.0f;
            const float Cp = in ? rcq[r] : 0.0f;
            const float BXp = (!BARO && in) ? rbx[BARO ? 0 : r] : 0.0f;
            const float BYp = (!BARO && in) ? rby[BARO ? 0 : r] : 0.0f;
            const float Am = qshr1(Ap);
            const float BXm = BARO ? 0.0f : qshr1(BXp);
            const float Cm = cq_m, BYm = by_m;
            // elliptic_tools.f90:46-54
            const float w1 = -(BXm + BYp);
            const float w2 = Cp + (BXp - BXm);
            const float w3 = BXp + BYp;
            const float w4 = Am - (BYp - BYm);
            const float w5 = -(Am + Ap + Cm + Cp);
            const float w6 = Ap + (BYp - BYm);
            const float w7 = BXm + BYm;
            const float w8 = Cm - (BXp - BXm);
            const float w9 = -(BXp + BYm);
            if constexpr (FAST) {
                if constexpr (STAR) {
                    c[0][r] = interior ? w2 : 0.0f; c[1][r] = interior ? w4 : 0.0f; c[2][r] = interior ? w5 : 0.0f;
                    c[3][r] = interior ? w6 : 0.0f; c[4][r] = interior ? w8 : 0.0f;
                } else {
                    c[0][r] = interior ? w1 : 0.0f; c[1][r] = interior ? w2 : 0.0f; c[2][r] = interior ? w3 : 0.0f;
                    c[3][r] = interior ? w4 : 0.0f; c[4][r] = interior ? w5 : 0.0f; c[5][r] = interior ? w6 : 0.0f;
                    c[6][r] = interior ? w7 : 0.0f; c[7][r] = interior ? w8 : 0.0f; c[8][r] = interior ? w9 : 0.0f;
                }
                const float d = -w5;
                y1[r] = interior ? quotient_y1(d) : y_idle;
                // (a NaN d passes through fmin/fmax unnoticed, but then y1 and every iterate of the point are NaN)
                dmin = interior ? __builtin_fminf(dmin, __builtin_fabsf(d)) : dmin;
                dmax = interior ? __builtin_fmaxf(dmax, __builtin_fabsf(d)) : dmax;
            } else {
                c[0][r] = w1; c[1][r] = w2; c[2][r] = w3; c[3][r] = w4; c[4][r] = w5;
                c[5][r] = w6; c[6][r] = w7; c[7][r] = w8; c[8][r] = w9;
            }
            cq_m = Cp;
            by_m = BYp;
        }
    }

    stamp<STAMP>(tg, 2);                                  // weights built
    float amax = 0.0f, amin = __builtin_inff();           // running extremes of |alpha * R| (FAST)
    float q[R], rl[(FAST && WRITE_R) ? R : 1];
    // One Jacobi step src -> dst.  The edge rows of src are already in LDS (written by the tail of the previous
    // step, or before the loop); after the barrier the neighbours' rows are requested and, in the star body, the
    // rows that do not need them (1..R-2) are computed first, so the LDS round trip hides behind the step's work.
    // (Measured alternatives, tools/probe_q.hip: the barrier costs about as much as a step's arithmetic -- 9.6k
    // ticks per 6 steps with it, 5.7k without -- but per-neighbour flags in LDS instead are worse still, 13.9k:
    // waves that spin on a flag take issue slots from the wave they wait for.)
    auto step = [&](const float (&src)[R], float (&dst)[R], int t) {
        const int par = t & 1;
        __syncthreads();
        const float below = (w > 0) ? edge[par][0][w - 1][lane] : 0.0f;
        const float above = (w < NW - 1) ? edge[par][1][w + 1][lane] : 0.0f;
        if constexpr (STAR) {
            auto row = [&](int r) {
                const float om = (r > 0) ? src[r - 1] : below;
                const float oc = src[r];
                const float on = (r + 1 < R) ? src[r + 1] : above;
                // the star of elliptic_tools.f90:77-85, in its order (terms 2, 4, 5, 6, 8), then :190
                float Rv = c[0][r] * on;
                Rv = Rv + c[1][r] * qshr1(oc);
                Rv = Rv + c[2][r] * oc;
                Rv = Rv + c[3][r] * qshl1(oc);
                Rv = Rv + c[4][r] * om;
                Rv = Rv - ff[r];
                const float a = alpha * Rv;
                dst[r] = oc + quotient_q2(a, c[2][r], y1[r]);              // :238, -d = coe5
                amax = __builtin_fmaxf(amax, __builtin_fabsf(a));
                amin = __builtin_fminf(amin, __builtin_fabsf(a));
                if (WRITE_R && t == T - 1) rl[WRITE_R ? r : 0] = Rv;
            };
#pragma unroll
            for (int r = 1; r < R - 1; ++r) row(r);
            row(0);
            row(R - 1);
        } else {
            // the i-1 / i+1 views of rows -1..R of the old iterate, one DPP move each
            float lft[R + 2], rgt[R + 2];
            lft[0] = qshr1(below); rgt[0] = qshl1(below);
#pragma unroll
            for (int r = 0; r < R; ++r) { lft[r + 1] = qshr1(src[r]); rgt[r + 1] = qshl1(src[r]); }
            lft[R + 1] = qshr1(above); rgt[R + 1] = qshl1(above);
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const float om = (r > 0) ? src[r - 1] : below;
                const float oc = src[r];
                const float on = (r + 1 < R) ? src[r + 1] : above;
                float Rv = c[0][r] * lft[r + 2];              // elliptic_tools.f90:77-85, left to right
                Rv = Rv + c[1][r] * on;
                Rv = Rv + c[2][r] * rgt[r + 2];
                Rv = Rv + c[3][r] * lft[r + 1];
                Rv = Rv + c[4][r] * oc;
                Rv = Rv + c[5][r] * rgt[r + 1];
                Rv = Rv + c[6][r] * lft[r];
                Rv = Rv + c[7][r] * om;
                Rv = Rv + c[8][r] * rgt[r];
                Rv = Rv - ff[r];                              // :190
                if constexpr (FAST) {
                    const float a = alpha * Rv;
                    dst[r] = oc + quotient_q2(a, c[K5][r], y1[r]);
                    amax = __builtin_fmaxf(amax, __builtin_fabsf(a));
                    amin = __builtin_fminf(amin, __builtin_fabsf(a));
                    if (WRITE_R && t == T - 1) rl[WRITE_R ? r : 0] = Rv;
                } else {
                    const float upd = oc + alpha * Rv / (-c[K5][r]);   // :238
                    const int gy = gy0 + r;
                    const bool interior = tg.col_int && gy >= 1 && gy <= g.ny - 2;
                    dst[r] = interior ? upd : oc;
                    if (WRITE_R && t == T - 1) {
                        const int ly = w * R + r;
                        const bool vy = (ly >= T || tg.y0 <= 0) && (ly < H - T || tg.y0 + H - 1 >= g.ny - 1);
                        if (tg.vx && vy && interior) res[mo + (size_t)gy * g.pitch + gx] = Rv;
                    }
                }
            }
        }
        if (t + 1 < T) {                                      // the next step's exchange
            edge[par ^ 1][0][w][lane] = dst[R - 1];
            edge[par ^ 1][1][w][lane] = dst[0];
        }
    };
    edge[0][0][w][lane] = p[R - 1];
    edge[0][1][w][lane] = p[0];
    for (int t = 0; t < T; t += 2) {
        step(p, q, t);
        if (t + 1 < T) step(q, p, t + 1);
    }
    const bool in_q = (T & 1) != 0;                       // odd T: the last step wrote q
    stamp<STAMP>(tg, 3);                                  // T steps done

    if constexpr (FAST) {
        bool ok = dmin >= 0x1p-120f && dmax <= 0x1p120f;
        const float a_lo = __builtin_fmaxf(0x1p-100f, dmax * 0x1p-120f);
        const float a_hi = __builtin_fminf(0x1p100f, dmin * 0x1p94f);
        ok = ok && amin >= a_lo && amax <= a_hi;
#pragma unroll
        for (int r = 0; r < R; ++r) ok = ok && __builtin_fabsf(in_q ? q[r] : p[r]) <= FLT_MAX;   // false for NaN too
        // one-barrier vote (__syncthreads_and takes three): a flag per wave, then every wave looks at all of them
        const bool wave_bad = __builtin_amdgcn_ballot_w64(!ok) != 0;
        if (lane == 0) vote[w] = wave_bad ? 1 : 0;
        __syncthreads();
        const int v = vote[lane % NW];
        if (__builtin_amdgcn_ballot_w64(v != 0) != 0) return false;
    }
    stamp<STAMP>(tg, 4);                                  // validated
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int ly = w * R + r;
        const int gy = gy0 + r;
        const bool vy = (ly >= T || tg.y0 <= 0) && (ly < H - T || tg.y0 + H - 1 >= g.ny - 1);
        const bool interior = tg.col_int && gy >= 1 && gy <= g.ny - 2;
        if (tg.vx && vy && interior) {
            const size_t o = mo + (size_t)gy * g.pitch + gx;
            psi_out[o] = in_q ? q[r] : p[r];
            if constexpr (FAST && WRITE_R) res[o] = rl[r];
        }
    }
    stamp<STAMP>(tg, 5);                                  // stores issued
    return true;
}

}  // namespace

// (HIP's second launch bound is waves per SIMD: 4 keeps the 8x8 BAROTROPIC kernels at 128 VGPRs, i.e. two
// workgroups per CU, also for the residual-writing variant)
template <int NW, int R, bool WRITE_R, bool BARO, bool STAMP = false>
__global__ __launch_bounds__(NW * 64, (BARO && NW * R <= 64) ? 4 : 1) void k_sweep_q(Grid g, const float* __restrict__ psi_in,
        float* __restrict__ psi_out, const float* __restrict__ base, const float* __restrict__ f,
        float* __restrict__ res, const MemberState* __restrict__ st, float alpha, int T, int exact_only,
        unsigned long long* __restrict__ stamps)
{
    __shared__ float edge[2][2][NW][64];
    __shared__ int vote[NW];
    const int m = blockIdx.z;
    const MemberState* member = st ? st + m : nullptr;
    constexpr int H = NW * R;
    TileGeom tg;
    tg.lane = threadIdx.x & 63;
    tg.w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform, and known to be
    tg.x0 = (int)blockIdx.x * (64 - 2 * T) - T;
    tg.y0 = (int)blockIdx.y * (H - 2 * T) - T;
    tg.gx = tg.x0 + tg.lane;
    tg.gy0 = tg.y0 + tg.w * R;
    tg.col_in = tg.gx >= 0 && tg.gx < g.nx;
    tg.col_int = tg.gx >= 1 && tg.gx <= g.nx - 2;
    tg.vx = (tg.lane >= T || tg.x0 <= 0) && (tg.lane < 64 - T || tg.x0 + 63 >= g.nx - 1);
    tg.mo = (size_t)m * g.plane;
    tg.ks = (size_t)g.members * g.plane;
    tg.stamps = nullptr;
    if constexpr (STAMP) {
        tg.stamps = stamps + 32 * ((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x);
        stamp<STAMP>(tg, 0);                              // workgroup started
        if (threadIdx.x == 0)                             // HW_ID (cu/sh/se in bits 8..15) and XCC_ID
            tg.stamps[7] = (unsigned long long)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) |
                           ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) << 32);
    }
    if (!exact_only && sweep_tile<NW, R, WRITE_R, BARO, true, STAMP>(g, tg, edge, vote, psi_in, psi_out, base, f, res, alpha, T, member)) {
        stamp<STAMP>(tg, 6);
        return;
    }
    sweep_tile<NW, R, WRITE_R, BARO, false, STAMP>(g, tg, edge, vote, psi_in, psi_out, base, f, res, alpha, T, member);
    stamp<STAMP>(tg, 6);
}

// validation kernel for the quotient identity: n pairs per thread drawn from a counter-based generator
// over the admitted range; counts the pairs where quotient_q2 and the IEEE division differ
__global__ __launch_bounds__(256) void k_quotient_check(unsigned long long seed, int per_thread, int d_exp_lo, int d_exp_hi,
        unsigned long long* __restrict__ mismatches, float* __restrict__ first_bad)
{
    unsigned long long s = seed + (unsigned long long)(blockIdx.x * 256u + threadIdx.x) * 0x9E3779B97F4A7C15ull;
    auto next = [&]() {                                   // splitmix64
        s += 0x9E3779B97F4A7C15ull;
        unsigned long long z = s;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    };
    auto mant = [&](unsigned long long z) -> unsigned {   // a third of the draws from the awkward significands
        const unsigned m = (unsigned)(z >> 8) & 0x7fffffu;
        switch ((z >> 40) % 6) {
            case 0: return m & 0x7u;                      // just above a power of two
            case 1: return 0x7fffffu - (m & 0x7u);        // just below
            default: return m;
        }
    };
    unsigned long long bad = 0;
    for (int k = 0; k < per_thread; ++k) {
        const unsigned long long z1 = next(), z2 = next();
        const int ed = d_exp_lo + (int)((z1 >> 44) % (unsigned)(d_exp_hi - d_exp_lo + 1));
        const float d = __builtin_bit_cast(float, (unsigned)((z1 >> 63) << 31) | (unsigned)(ed + 127) << 23 | mant(z1));
        // |a| in [max(2^-100, |d| 2^-120), min(2^100, |d| 2^94)], the dynamic bound of the sweep
        const float ad = __builtin_fabsf(d);
        const float lo = __builtin_fmaxf(0x1p-100f, ad * 0x1p-120f), hi = __builtin_fminf(0x1p100f, ad * 0x1p94f);
        const int elo = (int)((__builtin_bit_cast(unsigned, lo) >> 23) & 0xff) - 127 + 1;
        const int ehi = (int)((__builtin_bit_cast(unsigned, hi) >> 23) & 0xff) - 127 - 1;
        const int ea = elo + (int)((z2 >> 44) % (unsigned)(ehi - elo + 1));
        float a = __builtin_bit_cast(float, (unsigned)((z2 >> 63) << 31) | (unsigned)(ea + 127) << 23 | mant(z2));
        if (k & 1) {                                      // every other pair: a next to a rounding boundary of q*d
            const float qq = __builtin_bit_cast(float, (unsigned)(127 + ea - ed) << 23 | mant(z2 >> 3));
            const float t = qq * d;
            if (__builtin_fabsf(t) >= lo && __builtin_fabsf(t) <= hi) a = t;
        }
        const float want = a / d;
        const float got = quotient_q2(a, -d, quotient_y1(d));
        if (__builtin_bit_cast(unsigned, want) != __builtin_bit_cast(unsigned, got)) {
            if (bad == 0 && atomicAdd(mismatches + 1, 1ull) == 0) { first_bad[0] = a; first_bad[1] = d; }
            ++bad;
        }
    }
    if (bad) atomicAdd(mismatches, bad);
}

template <int NW, int R, bool BARO>
static void launch_q_cfg(const Grid& g, const float* psi_in, float* psi_out, const float* base, const float* f,
                         float* res, const MemberState* st, float alpha, int T, int exact_only, hipStream_t s)
{
    constexpr int H = NW * R;
    dim3 grid((g.nx - 1 + (64 - 2 * T) - 1) / (64 - 2 * T), (g.ny - 1 + (H - 2 * T) - 1) / (H - 2 * T), g.members);
    if (res)
        hipLaunchKernelGGL((k_sweep_q<NW, R, true, BARO>), grid, dim3(NW * 64), 0, s, g, psi_in, psi_out, base, f, res, st, alpha, T, exact_only, (unsigned long long*)nullptr);
    else
        hipLaunchKernelGGL((k_sweep_q<NW, R, false, BARO>), grid, dim3(NW * 64), 0, s, g, psi_in, psi_out, base, f, res, st, alpha, T, exact_only, (unsigned long long*)nullptr);
}

void launch_sweep_q(const Grid& g, const float* psi_in, float* psi_out, const float* base, const float* f,
                    float* res, const MemberState* st, float alpha, int T, bool barotropic, const SweepOpts& o,
                    SweepDesc* desc, hipStream_t s)
{
    const int exact_only = o.q_exact_only ? 1 : 0;
    // (the nine-weight body needs 156 VGPRs at eight rows per lane -- one workgroup per CU; at six it fits
    // two: 8.9 against 7.1 cases/s on the baroclinic bench workload)
    int cfg = o.tile;
    if (cfg != 808 && cfg != 806 && cfg != 416) cfg = barotropic ? 808 : 806;
    if (desc) { desc->name = barotropic ? "k_sweep_q<star>" : "k_sweep_q<nine>"; desc->T = T; desc->tile_w = 64; desc->tile_h = (cfg / 100) * (cfg % 100); }
#define Q_CASE(NW, R) case NW * 100 + R: \
        if (barotropic) launch_q_cfg<NW, R, true>(g, psi_in, psi_out, base, f, res, st, alpha, T, exact_only, s); \
        else launch_q_cfg<NW, R, false>(g, psi_in, psi_out, base, f, res, st, alpha, T, exact_only, s); \
        break;
    switch (cfg) {
        Q_CASE(8, 8)
        Q_CASE(8, 6)
        Q_CASE(4, 16)
    }
#undef Q_CASE
}

void launch_quotient_check(unsigned long long seed, int blocks, int per_thread, int d_exp_lo, int d_exp_hi,
                           unsigned long long* mismatches, float* first_bad, hipStream_t s)
{
    hipLaunchKernelGGL(k_quotient_check, dim3(blocks), dim3(256), 0, s, seed, per_thread, d_exp_lo, d_exp_hi, mismatches, first_bad);
}

}  // namespace xee
