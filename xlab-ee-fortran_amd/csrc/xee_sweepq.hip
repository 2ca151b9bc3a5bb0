// xee_sweepq.hip -- the temporally blocked Jacobi sweep with the update's division strength-reduced,
// optimistically, and validated per tile.  (Compiled with -fno-slp-vectorize: the packed arithmetic below is
// written out, and lane-crossing multiplies must stay scalar so that they fuse with their DPP move.)
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
//              and every one of the T steps (running max/min of |alpha*R|: one v_max3/v_min3 per two points),
//              and every value of the tile finite at the end (non-finite values are sticky under the update)
//              (a residual that is EXACTLY zero is admitted -- see "Zero residuals" below; converged regions produce them
//              by the thousand late in a long run, zero right-hand sides from the first iteration on)
// which keeps the exponent difference below the 96 where div_scale starts to rescale and quotient,
// reciprocal and remainders normal and exactly representable (tests/test_gpu_quotient.py checks the
// identity over 10^9 operand pairs drawn over the whole admitted range).  A tile that fails any of it (a
// zero residual, an exploding iterate, exotic coefficients) is recomputed from its unchanged inputs by
// the exact body -- same kernel, same workgroup -- so the result is bit-identical to k_sweep_tb4 always.
//
// The same validation covers two more savings:
//  * BAROTROPIC operator (b = 0): its four corner weights are +-0 (elliptic_tools.f90:46,48,52,54 with
//    BX = BY = 0).  The optimistic body sums the five-point star only (5 mul + 5 add instead of 9 + 9).  With finite
//    iterates the four dropped products are zeros, and adding a zero changes a partial sum only if that sum is itself a
//    zero (its sign): by induction over the nine terms the star's sum S5 and the reference's S9 are the same bits unless
//    both are zeros of different sign -- and S5 is -0 only if all five products are -0 (a cancellation x + (-x) and any
//    +0 term give +0), S9 only if all nine are.  So S5 != S9 needs S5 = -0; then R = S - f is the same for f != 0, and for
//    f = +-0 a zero whose sign may differ, which cannot change psi + a/d unless psi is -0 (excluded, see below).  The one
//    place where the sign is visible is the residual plane a residual-writing launch stores: a lane that sees S5 = -0 in
//    that launch's last step fails the tile.  (A non-finite iterate makes a dropped product NaN: it is in the tile at the
//    end, non-finite values being sticky under the update, and fails the tile too.)
//  * no select for the points that never update (rim, outside the domain): their weights are set to 0, f
//    to -1 and y1 to the zero whose sign makes q2 = -0, so `psi + q2` returns psi bit for bit and their
//    "residual" alpha*1 stays inside the dynamic bound.
//
// Zero residuals.  A point whose residual is exactly zero has a = alpha*R = +-0 and both forms of the quotient give a
// zero; psi + (+-0) is psi for either sign unless psi is -0.  An update cannot create a -0 (x + y is -0 only for
// -0 + -0), so a lane whose loaded interior psi hold no -0 may take zeros (a lane that loaded one fails the tile) --
// PROVIDED nothing in (0, a_lo) hides behind them.  Where f is not tiny nothing can: R = fl(s - f) of two
// floats is either zero or at least |f| * 2^-25 (opposite signs or a factor two apart: more than |f|/2; within a factor
// two: an exact, non-zero multiple of the smaller one's ulp), so with fmin = min |f| over the lane's interior points and
// |alpha| * fmin * 2^-26 >= a_lo every non-zero |a| is above the lower bound by construction and no minimum is kept
// (waves of such lanes run a loop instance without it).  Elsewhere -- f = 0: the boundary-driven solves of the legacy
// sweep, old-diagnose/diagnose.f90:548-568 -- the running minimum is taken over the NON-ZERO |a| only (an unsigned
// minimum of (bits << 1) - 1: three instructions per pair and step).  Round 2 rejected every zero where f = 0: a
// zero right-hand side then ran 13 % of its tiles twice (58 instead of 35 us per launch of 8 cases, profiles/r03).
// Without the first rule, 2-3 % of the tiles of the bench's ensemble fell to the exact body from ~5,000 iterations on,
// and the stragglers stretched every launch by 12 % (profiles/r02).
//
// Round 2: the optimistic body is written in PACKED fp32 (v_pk_mul/add/fma_f32: two points per instruction, each
// half rounded separately -- the same bits as two scalar operations).  A lane's R rows are paired (k, k + R/2), so the
// j+-1 neighbours of a pair are again a register pair (no moves except at the two ends); the i+-1 terms of the star are
// scalar multiplies fused with their DPP lane shift (v_mul_f32_dpp: one instruction instead of shift + multiply).
// 10 VALU instructions per point and step instead of 13.3, and for alpha == 1 (the protocol's value; 1.0f*x is x bit
// for bit) the multiply by alpha is not issued.  Tiles that lie wholly inside the interior (most) run a prologue
// without clamps and selects.
#include "xee_device.h"

#include <cfloat>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#pragma clang fp contract(off)

namespace xee {

namespace {

typedef float f2 __attribute__((ext_vector_type(2)));

// old = 0 with bound_ctrl: the lane without a source reads 0, and the move folds into a following v_mul_f32 (DPP)
__device__ __forceinline__ float qshr1(float src)   // lane l <- lane l-1 (lane 0 gets 0)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, src), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float qshl1(float src)   // lane l <- lane l+1 (lane 63 gets 0)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, src), 0x130, 0xf, 0xf, true));
}
__device__ __forceinline__ f2 qshr1(f2 v) { return f2{qshr1(v.x), qshr1(v.y)}; }
__device__ __forceinline__ f2 qshl1(f2 v) { return f2{qshl1(v.x), qshl1(v.y)}; }

// Raw buffer access (buffer_load/store_dword ... offen): the row offset is a scalar operand and the column offset ONE
// vector register shared by every access of the lane -- no per-access 64-bit address arithmetic on the VALU (the
// compiler turns `rowbase[col]` into a v_lshl_add_u64 per load).  num_records is the byte size of one member's plane;
// a store whose column offset is ST_DROP is dropped by the range check (no exec-mask juggling per store).
typedef __amdgpu_buffer_rsrc_t rsrc_t;
constexpr int ST_DROP = (int)0x80000000;
__device__ __forceinline__ rsrc_t plane_rsrc(const float* plane_of_member, const Grid& g)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(plane_of_member), 0, (int)(g.plane * sizeof(float)), 0x00020000);
}
__device__ __forceinline__ float bload(rsrc_t r, int col_bytes, int row_bytes)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, col_bytes, row_bytes, 0));
}
__device__ __forceinline__ void bstore(rsrc_t r, int col_bytes, int row_bytes, float v)
{
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), r, col_bytes, row_bytes, 0);
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
__device__ __forceinline__ f2 quotient_q2(f2 a, f2 nd, f2 y1)   // two points per instruction
{
    const f2 q0 = a * y1;
    const f2 r0 = __builtin_elementwise_fma(nd, q0, a);
    const f2 q1 = __builtin_elementwise_fma(r0, y1, q0);
    const f2 r1 = __builtin_elementwise_fma(nd, q1, a);
    return __builtin_elementwise_fma(r1, y1, q1);
}

struct TileGeom {
    int lane, w, x0, y0, gx, gy0;
    bool col_in, col_int, vx;
    bool inner;                     // the whole tile (and the row below it) lies inside the interior: workgroup-uniform
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

// ---- the optimistic body: packed arithmetic, hoisted reciprocal, validated -------------------------------------
// One tile, T steps.  Returns whether the WHOLE workgroup validated, and only then stores.
// A lane holds rows gy0 .. gy0+R-1 of its column as R/2 pairs: pair k = (row k, row k + R/2).
template <int NW, int R, bool WRITE_R, bool BARO, bool ALPHA1, bool INNER, bool STAMP>
__device__ __forceinline__ bool fast_tile(const Grid& g, const TileGeom& tg, float (*edge)[2][NW][64], int* vote,
        const float* __restrict__ psi_in, float* __restrict__ psi_out, const float* __restrict__ base,
        const float* __restrict__ f, float* __restrict__ res, float* __restrict__ snap, float alpha, int T,
        const MemberState* __restrict__ member)
{
    static_assert(R % 2 == 0 && R >= 4, "rows are processed in pairs (k, k + R/2)");
    constexpr int H = NW * R;
    constexpr int H2 = R / 2;
    constexpr int NC = BARO ? 5 : 9;                    // weights kept: the star (2,4,5,6,8) or all nine
    constexpr int K5 = BARO ? 2 : 4;                    // where coe5 sits in c[]
    const int lane = tg.lane, w = tg.w, gx = tg.gx, gy0 = tg.gy0;
    const size_t mo = tg.mo, ks = tg.ks;

    f2 c[NC][H2], ff[H2], y1[H2], p[H2];
    float dmin = 0x1p120f, dmax = 0x1p-120f;            // extremes of |coe5| over this lane's interior points
    float fmin = __builtin_inff();                      // min |f| over them, and whether one of their psi is -0
    bool neg_zero_psi = false;
    unsigned nonzero_bits = 0u;                         // OR of every loaded psi's bits and every interior |f|'s: 0 = "all +0 / +-0"
    {
        // Every load of the tile is issued before the first value is used, unconditionally and from a
        // clamped (always valid) address; what lies outside the domain is replaced afterwards.  Row offsets are
        // scalar (the wave index is), the only per-lane part of an address is the column.
        const rsrc_t pr = plane_rsrc(psi_in + mo, g), fr = plane_rsrc(f + mo, g);
        // (generating planes that all members share are read from member 0's copy: Grid.base_shared)
        const size_t mo_a = (g.base_shared & 1) ? 0 : mo, mo_c = (g.base_shared & 2) ? 0 : mo, mo_x = (g.base_shared & 4) ? 0 : mo, mo_y = (g.base_shared & 8) ? 0 : mo;
        (void)mo_x; (void)mo_y;
        const rsrc_t apr = plane_rsrc(base + mo_a, g), cqr = plane_rsrc(base + ks + mo_c, g);
        const int cx = (INNER ? gx : (gx < 0 ? 0 : (gx > g.nx - 1 ? g.nx - 1 : gx))) * 4;
        auto rowoff = [&](int gy) -> int {
            const int cy = INNER ? gy : (gy < 0 ? 0 : (gy > g.ny - 1 ? g.ny - 1 : gy));
            return cy * g.pitch * 4;
        };
        float rp[R], rf[R], rap[R], rcq[R], rbx[BARO ? 1 : R], rby[BARO ? 1 : R];
        const int ob = rowoff(gy0 - 1);
        const float rcq_b = bload(cqr, cx, ob);
        float rby_b = 0.0f;
        if constexpr (!BARO) rby_b = bload(plane_rsrc(base + 3 * ks + mo_y, g), cx, ob);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int o = rowoff(gy0 + r);
            rp[r] = bload(pr, cx, o);
            rf[r] = bload(fr, cx, o);
            rap[r] = bload(apr, cx, o);
            rcq[r] = bload(cqr, cx, o);
            if constexpr (!BARO) {
                rbx[r] = bload(plane_rsrc(base + 2 * ks + mo_x, g), cx, o);
                rby[r] = bload(plane_rsrc(base + 3 * ks + mo_y, g), cx, o);
            }
        }
        stamp<STAMP>(tg, 1);                                // loads issued
        // (only now the member's `active` flag is looked at: testing it first would put a scalar-memory round
        // trip in front of the tile's loads; for a stopped member the loads above are harmless.  The flag a sweep
        // sees only changes between launches -- xee_api.hip keeps a per-stream view -- so the exit is uniform)
        if (member && !member->active) return true;
        const float y_idle = alpha > 0.0f ? -0.0f : 0.0f;   // (alpha*1) * y_idle = -0
        const int gyb = gy0 - 1;
        const bool inb = INNER || (tg.col_in && gyb >= 0 && gyb < g.ny);
        float cq_m = inb ? rcq_b : 0.0f;
        float by_m = (!BARO && inb) ? rby_b : 0.0f;
        float cw[NC][R], fw[R], yw[R], pw[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int gy = gy0 + r;
            const bool in = INNER || (tg.col_in && gy >= 0 && gy < g.ny);
            const bool interior = INNER || (tg.col_int && gy >= 1 && gy <= g.ny - 2);
            pw[r] = in ? rp[r] : 0.0f;
            fw[r] = interior ? rf[r] : -1.0f;
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
            if constexpr (BARO) {
                cw[0][r] = interior ? w2 : 0.0f; cw[1][r] = interior ? w4 : 0.0f; cw[2][r] = interior ? w5 : 0.0f;
                cw[3][r] = interior ? w6 : 0.0f; cw[4][r] = interior ? w8 : 0.0f;
            } else {
                cw[0][r] = interior ? w1 : 0.0f; cw[1][r] = interior ? w2 : 0.0f; cw[2][r] = interior ? w3 : 0.0f;
                cw[3][r] = interior ? w4 : 0.0f; cw[4][r] = interior ? w5 : 0.0f; cw[5][r] = interior ? w6 : 0.0f;
                cw[6][r] = interior ? w7 : 0.0f; cw[7][r] = interior ? w8 : 0.0f; cw[8][r] = interior ? w9 : 0.0f;
            }
            const float d = -w5;
            yw[r] = interior ? quotient_y1(d) : y_idle;
            // (a NaN d passes through fmin/fmax unnoticed, but then y1 and every iterate of the point are NaN)
            dmin = interior ? __builtin_fminf(dmin, __builtin_fabsf(d)) : dmin;
            dmax = interior ? __builtin_fmaxf(dmax, __builtin_fabsf(d)) : dmax;
            fmin = interior ? __builtin_fminf(fmin, __builtin_fabsf(rf[r])) : fmin;
            neg_zero_psi = neg_zero_psi || (interior && __builtin_bit_cast(unsigned, rp[r]) == 0x80000000u);
            nonzero_bits |= __builtin_bit_cast(unsigned, pw[r]) | (interior ? (__builtin_bit_cast(unsigned, rf[r]) & 0x7fffffffu) : 0u);
            cq_m = Cp;
            by_m = BYp;
        }
        // rows (k, k + R/2) into register pairs
#pragma unroll
        for (int k = 0; k < H2; ++k) {
            p[k] = f2{pw[k], pw[k + H2]};
            ff[k] = f2{fw[k], fw[k + H2]};
            y1[k] = f2{yw[k], yw[k + H2]};
#pragma unroll
            for (int n = 0; n < NC; ++n) c[n][k] = f2{cw[n][k], cw[n][k + H2]};
        }
    }

    stamp<STAMP>(tg, 2);                                  // weights built
    // running extremes of |alpha * R|.  The minimum is taken over the NON-ZERO values only (see "Zero residuals"): as an
    // unsigned minimum of (bits << 1) - 1, which drops the sign, keeps the order of the magnitudes and sends both zeros to
    // 0xffffffff (v_lshl_add_u32 + v_min3_u32: three instructions per pair)
    float amax = 0.0f;
    unsigned amin_nz = 0xffffffffu;
    bool zero_at_last = false;                            // residual-writing star body: a sum that is -0 in the launch's last step
    f2 q[H2], rl[WRITE_R ? H2 : 1];
    const f2 alpha2 = f2{alpha, alpha};
    // One Jacobi step src -> dst.  The edge rows of src are already in LDS (written by the tail of the previous
    // step, or before the loop); after the barrier the neighbours' rows are requested and the pairs that do not
    // need them (1..H2-2) are computed first, so the LDS round trip hides behind the step's work.
    // (Measured alternatives, tools/probe_q.hip: the barrier costs about as much as a step's arithmetic, but
    // per-neighbour flags in LDS instead are worse still: waves that spin on a flag take issue slots from the
    // wave they wait for.)
    auto step = [&](const f2 (&src)[H2], f2 (&dst)[H2], int t, auto last_tag, auto track_tag) {
        constexpr bool LAST = decltype(last_tag)::value;      // the launch's final step (keeps the residual)
        constexpr bool TRACK_MIN = decltype(track_tag)::value; // the wave needs the running minimum of |a| (see below)
        const int par = t & 1;
        __syncthreads();
        // (read unconditionally from a clamped slot, then select: a conditional read is a branch, and with a branch at its
        // head the step is no longer ONE basic block -- the compiler then hoists the first pairs' lane shifts in front of
        // it, where they cannot fold into their multiplies any more: 4 extra moves per step)
        // (the nine-weight body shifts with separate moves anyway, and the two extra live values made it spill: it keeps
        // the conditional reads)
        float below, above;
        if constexpr (BARO) {
            const float below_raw = edge[par][0][w > 0 ? w - 1 : 0][lane];
            const float above_raw = edge[par][1][w < NW - 1 ? w + 1 : NW - 1][lane];
            below = (w > 0) ? below_raw : 0.0f;
            above = (w < NW - 1) ? above_raw : 0.0f;
        } else {
            below = (w > 0) ? edge[par][0][w - 1][lane] : 0.0f;
            above = (w < NW - 1) ? edge[par][1][w + 1][lane] : 0.0f;
        }
        auto finish = [&](int k, f2 Rv) {
            // (the residual-writing star body: its sum and the reference's nine-term sum can differ in one way only, the sign
            // of a zero, and only when the star's own sum is -0, i.e. all five products are -0 -- see "Zero residuals")
            if constexpr (WRITE_R && LAST && BARO)
                zero_at_last = zero_at_last || __builtin_bit_cast(unsigned, Rv.x) == 0x80000000u || __builtin_bit_cast(unsigned, Rv.y) == 0x80000000u;
            Rv = Rv - ff[k];                                           // :190
            const f2 a = ALPHA1 ? Rv : alpha2 * Rv;
            dst[k] = src[k] + quotient_q2(a, c[K5][k], y1[k]);         // :238, -d = coe5
            amax = __builtin_fmaxf(__builtin_fmaxf(amax, __builtin_fabsf(a.x)), __builtin_fabsf(a.y));
            if constexpr (TRACK_MIN) {
                const unsigned ux = (__builtin_bit_cast(unsigned, a.x) << 1) + 0xffffffffu, uy = (__builtin_bit_cast(unsigned, a.y) << 1) + 0xffffffffu;
                amin_nz = __builtin_elementwise_min(__builtin_elementwise_min(amin_nz, ux), uy);
            }
            if constexpr (WRITE_R && LAST) rl[k] = Rv;
        };
        if constexpr (BARO) {
            auto pair = [&](int k) {
                const f2 oc = src[k];
                const f2 on = (k + 1 < H2) ? src[k + 1] : f2{src[0].y, above};       // rows k+1, k+R/2+1
                const f2 om = (k > 0) ? src[k - 1] : f2{below, src[H2 - 1].x};       // rows k-1, k+R/2-1
                // the star of elliptic_tools.f90:77-85, in its order (terms 2, 4, 5, 6, 8)
                f2 Rv = c[0][k] * on;
                Rv = Rv + f2{c[1][k].x * qshr1(oc.x), c[1][k].y * qshr1(oc.y)};      // v_mul_f32_dpp
                Rv = Rv + c[2][k] * oc;
                Rv = Rv + f2{c[3][k].x * qshl1(oc.x), c[3][k].y * qshl1(oc.y)};
                Rv = Rv + c[4][k] * om;
                finish(k, Rv);
            };
#pragma unroll
            for (int k = 1; k < H2 - 1; ++k) pair(k);
            pair(0);
            pair(H2 - 1);
        } else {
            // the i-1 / i+1 views of the old iterate: one DPP move per value, each used by three rows
            f2 lft[H2], rgt[H2];
#pragma unroll
            for (int k = 0; k < H2; ++k) { lft[k] = qshr1(src[k]); rgt[k] = qshl1(src[k]); }
            const float lb = qshr1(below), rb = qshl1(below), la = qshr1(above), ra = qshl1(above);
#pragma unroll
            for (int k = 0; k < H2; ++k) {
                const f2 oc = src[k];
                const f2 on = (k + 1 < H2) ? src[k + 1] : f2{src[0].y, above};
                const f2 om = (k > 0) ? src[k - 1] : f2{below, src[H2 - 1].x};
                const f2 ln = (k + 1 < H2) ? lft[k + 1] : f2{lft[0].y, la};
                const f2 rn = (k + 1 < H2) ? rgt[k + 1] : f2{rgt[0].y, ra};
                const f2 lm = (k > 0) ? lft[k - 1] : f2{lb, lft[H2 - 1].x};
                const f2 rm = (k > 0) ? rgt[k - 1] : f2{rb, rgt[H2 - 1].x};
                f2 Rv = c[0][k] * ln;                         // elliptic_tools.f90:77-85, left to right
                Rv = Rv + c[1][k] * on;
                Rv = Rv + c[2][k] * rn;
                Rv = Rv + c[3][k] * lft[k];
                Rv = Rv + c[4][k] * oc;
                Rv = Rv + c[5][k] * rgt[k];
                Rv = Rv + c[6][k] * lm;
                Rv = Rv + c[7][k] * om;
                Rv = Rv + c[8][k] * rm;
                finish(k, Rv);
            }
        }
        // the next step's exchange (written after the last step too: nobody reads it, and a step body without a
        // branch is one basic block -- the DPP moves only fold into their multiplies inside a block)
        edge[par ^ 1][0][w][lane] = dst[H2 - 1].y;            // row R-1, for the wave above
        edge[par ^ 1][1][w][lane] = dst[0].x;                 // row 0, for the wave below
    };
    edge[0][0][w][lane] = p[H2 - 1].y;
    edge[0][1][w][lane] = p[0].x;
    // Where every non-zero |a| is above the lower bound by construction (see "Zero residuals" in the header) the running
    // minimum is never looked at: a WAVE whose lanes all are in that position runs the instance of the loop that does not
    // keep it (three instructions per pair and step less).  The choice is wave-uniform; both instances meet the
    // workgroup's barriers the same number of times.
    const float a_lo = __builtin_fmaxf(0x1p-100f, dmax * 0x1p-120f);
    const bool zero_ok = __builtin_fabsf(alpha) * fmin * 0x1p-26f >= a_lo;
    bool in_q;

    // All-zero tiles.  A zero right-hand side with a boundary-driven solution (old-diagnose/diagnose.f90:548-568: f = 0,
    // rchi = rchi_bc) leaves most of the plane at the all-zero first guess for thousands of iterations.  If EVERY psi the
    // tile loaded (halo and rim included) is +0 and every interior f is +-0, one step decides all T of them: a Jacobi step
    // is a function of the tile's values, so if it returns +0 everywhere again, so does every later one.  The step is
    // computed for real (a non-finite weight, 0 * inf, would show as a NaN) and checked; its residual is the residual of
    // every step.  One extra barrier per tile for the vote (the waves are in step here: it costs its own latency only).
    bool zero_tile = false;
    {
        const bool wave_zero = __builtin_amdgcn_ballot_w64(nonzero_bits != 0u) == 0;
        if (lane == 0) vote[w] = wave_zero ? 0 : 1;
        __syncthreads();
        const int v0 = vote[lane % NW];
        if (__builtin_amdgcn_ballot_w64(v0 != 0) == 0) {
            step(p, q, 0, std::integral_constant<bool, WRITE_R>{}, std::false_type{});
            bool z = true;
#pragma unroll
            for (int k = 0; k < H2; ++k)
                z = z && __builtin_bit_cast(unsigned, q[k].x) == 0u && __builtin_bit_cast(unsigned, q[k].y) == 0u;
            // (vote[] again: every wave has read it before the barrier at the head of the step)
            const bool wave_bad = __builtin_amdgcn_ballot_w64(!z) != 0;
            if (lane == 0) vote[w] = wave_bad ? 1 : 0;
            __syncthreads();
            const int v1 = vote[lane % NW];
            if (__builtin_amdgcn_ballot_w64(v1 != 0) != 0) return false;   // exotic weights: the exact body decides
            if constexpr (WRITE_R && BARO) {
                // the star's sum of signed zeros need not have the sign of the reference's nine-term sum, whose third term
                // (+0) * (+0) makes it +0, and +0 - (+-0) = +0 (elliptic_tools.f90:77-85,190 with :46-54 for b = 0)
#pragma unroll
                for (int k = 0; k < H2; ++k) rl[k] = f2{0.0f, 0.0f};
            }
            zero_tile = true;                             // p holds the result already: all +0
        }
    }

    if (!zero_tile) {
    // (the loop is written out once per instance -- a macro, not a lambda around `step`: with one more closure level the
    // iterate's register arrays were left in scratch memory)
#define XEE_Q_RUN(TRACK)                                                                                           \
    {                                                                                                              \
        int t = 0;                                                                                                 \
        if constexpr (WRITE_R) {                                                                                   \
            /* the final step is its own instance (it keeps the residual): T - 1 or T - 2 steps in pairs first */  \
            for (; t + 2 < T; t += 2) {                                                                            \
                step(p, q, t, std::false_type{}, TRACK);                                                           \
                step(q, p, t + 1, std::false_type{}, TRACK);                                                       \
            }                                                                                                      \
            in_q = t + 1 >= T;                                                                                     \
            if (in_q) step(p, q, t, std::true_type{}, TRACK);                                                      \
            else { step(p, q, t, std::false_type{}, TRACK); step(q, p, t + 1, std::true_type{}, TRACK); }          \
        } else {                                                                                                   \
            for (; t + 1 < T; t += 2) {                                                                            \
                step(p, q, t, std::false_type{}, TRACK);                                                           \
                step(q, p, t + 1, std::false_type{}, TRACK);                                                       \
            }                                                                                                      \
            in_q = t < T;                                     /* odd T: one more step, which writes q */           \
            if (in_q) step(p, q, t, std::false_type{}, TRACK);                                                     \
        }                                                                                                          \
    }
    if (__builtin_amdgcn_ballot_w64(!zero_ok) != 0) XEE_Q_RUN(std::true_type{})
    else XEE_Q_RUN(std::false_type{})
#undef XEE_Q_RUN
    // the result into p whichever array the last step wrote (R/2 moves once per launch: with a choice between the two
    // ARRAYS further down, and two instances of the loop above, the compiler kept both arrays in scratch memory)
    if (in_q) {
#pragma unroll
        for (int k = 0; k < H2; ++k) p[k] = q[k];
    }
    stamp<STAMP>(tg, 3);                                  // T steps done

    {
        bool ok = dmin >= 0x1p-120f && dmax <= 0x1p120f;
        const float a_hi = __builtin_fminf(0x1p100f, dmin * 0x1p94f);
        // Lower bound: by construction where zero_ok, else the smallest NON-ZERO |a| seen (a wave of zero_ok lanes did not
        // keep it: still "none").  Zeros pass unless a psi the lane loaded is -0 (see "Zero residuals"); in the residual-
        // writing star body a sum of -0 in the last step is left to the exact body too, for the sign of the stored residual.
        const bool none = amin_nz == 0xffffffffu;
        const bool lo_ok = zero_ok || none || amin_nz + 1u >= (__builtin_bit_cast(unsigned, a_lo) << 1);
        ok = ok && !neg_zero_psi && lo_ok && !zero_at_last && amax <= a_hi;
#pragma unroll
        for (int k = 0; k < H2; ++k) {
            const f2 v = p[k];
            ok = ok && __builtin_fabsf(v.x) <= FLT_MAX && __builtin_fabsf(v.y) <= FLT_MAX;   // false for NaN too
        }
        // one-barrier vote (__syncthreads_and takes three): a flag per wave, then every wave looks at all of them
        const bool wave_bad = __builtin_amdgcn_ballot_w64(!ok) != 0;
        if (lane == 0) vote[w] = wave_bad ? 1 : 0;
        __syncthreads();
        const int v = vote[lane % NW];
        if (__builtin_amdgcn_ballot_w64(v != 0) != 0) return false;
    }
    }
    stamp<STAMP>(tg, 4);                                  // validated
    {
        // a row is stored or not as a whole (scalar test); lanes outside the valid columns get a dropped offset
        const rsrc_t outr = plane_rsrc(psi_out + mo, g);
        const int sx = (tg.vx && (INNER || tg.col_int)) ? gx * 4 : ST_DROP;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int k = r < H2 ? r : r - H2;
            const int ly = w * R + r;
            const int gy = gy0 + r;
            const bool vy = (ly >= T || tg.y0 <= 0) && (ly < H - T || tg.y0 + H - 1 >= g.ny - 1);
            const bool row_int = INNER || (gy >= 1 && gy <= g.ny - 2);
            if (vy && row_int) {
                const int o = gy * g.pitch * 4;
                const f2 v = p[k];
                bstore(outr, sx, o, r < H2 ? v.x : v.y);
                if constexpr (WRITE_R) {
                    bstore(plane_rsrc(res + mo, g), sx, o, r < H2 ? rl[k].x : rl[k].y);
                    // the pipelined driver's snapshot of the iterate this check belongs to (xee_api.hip), when asked for
                    if (snap) bstore(plane_rsrc(snap + mo, g), sx, o, r < H2 ? v.x : v.y);
                }
            }
        }
    }
    stamp<STAMP>(tg, 5);                                  // stores issued
    return true;
}

// ---- the exact body: the arithmetic of k_sweep_tb4 (all nine terms, IEEE division, a select for the points that
// do not update).  Runs only for tiles the optimistic body rejected (or under the option "q_exact_only").
template <int NW, int R, bool WRITE_R, bool BARO>
__device__ __forceinline__ void exact_tile(const Grid& g, const TileGeom& tg, float (*edge)[2][NW][64],
        const float* __restrict__ psi_in, float* __restrict__ psi_out, const float* __restrict__ base,
        const float* __restrict__ f, float* __restrict__ res, float* __restrict__ snap, float alpha, int T,
        const MemberState* __restrict__ member)
{
    constexpr int H = NW * R;
    const int lane = tg.lane, w = tg.w, gx = tg.gx, gy0 = tg.gy0;
    const size_t mo = tg.mo, ks = tg.ks;

    float c[9][R], ff[R], p[R];
    {
        const float* __restrict__ pm = psi_in + mo;
        const float* __restrict__ fm = f + mo;
        const float* __restrict__ apm = base + ((g.base_shared & 1) ? 0 : mo);
        const float* __restrict__ cqm = base + ks + ((g.base_shared & 2) ? 0 : mo);
        const float* __restrict__ bxm = base + 2 * ks + ((g.base_shared & 4) ? 0 : mo);
        const float* __restrict__ bym = base + 3 * ks + ((g.base_shared & 8) ? 0 : mo);
        const unsigned cx = (unsigned)(gx < 0 ? 0 : (gx > g.nx - 1 ? g.nx - 1 : gx));
        auto rowoff = [&](int gy) -> size_t {
            const int cy = gy < 0 ? 0 : (gy > g.ny - 1 ? g.ny - 1 : gy);
            return (size_t)cy * (size_t)g.pitch;
        };
        if (member && !member->active) return;
        const int gyb = gy0 - 1;
        const bool inb = tg.col_in && gyb >= 0 && gyb < g.ny;
        const size_t ob = rowoff(gyb);
        float cq_m = inb ? (cqm + ob)[cx] : 0.0f;
        float by_m = (!BARO && inb) ? (bym + ob)[cx] : 0.0f;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int gy = gy0 + r;
            const size_t o = rowoff(gy);
            const bool in = tg.col_in && gy >= 0 && gy < g.ny;
            const bool interior = tg.col_int && gy >= 1 && gy <= g.ny - 2;
            p[r] = in ? (pm + o)[cx] : 0.0f;
            ff[r] = interior ? (fm + o)[cx] : 0.0f;
            const float Ap = in ? (apm + o)[cx] : 0.0f;
            const float Cp = in ? (cqm + o)[cx] : 0.0f;
            const float BXp = (!BARO && in) ? (bxm + o)[cx] : 0.0f;
            const float BYp = (!BARO && in) ? (bym + o)[cx] : 0.0f;
            const float Am = qshr1(Ap);
            const float BXm = BARO ? 0.0f : qshr1(BXp);
            const float Cm = cq_m, BYm = by_m;
            // elliptic_tools.f90:46-54
            c[0][r] = -(BXm + BYp);
            c[1][r] = Cp + (BXp - BXm);
            c[2][r] = BXp + BYp;
            c[3][r] = Am - (BYp - BYm);
            c[4][r] = -(Am + Ap + Cm + Cp);
            c[5][r] = Ap + (BYp - BYm);
            c[6][r] = BXm + BYm;
            c[7][r] = Cm - (BXp - BXm);
            c[8][r] = -(BXp + BYm);
            cq_m = Cp;
            by_m = BYp;
        }
    }
    float q[R];
    auto step = [&](const float (&src)[R], float (&dst)[R], int t) {
        const int par = t & 1;
        __syncthreads();
        const float below = (w > 0) ? edge[par][0][w - 1][lane] : 0.0f;
        const float above = (w < NW - 1) ? edge[par][1][w + 1][lane] : 0.0f;
        // (the lane shifts are taken where they are used -- six per row instead of two: this body runs for the few tiles the
        // optimistic one rejects, and the 2 (R + 2) registers of shifted copies were what made the residual-writing
        // kernel, which inlines both bodies, spill)
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const float om = (r > 0) ? src[r - 1] : below;
            const float oc = src[r];
            const float on = (r + 1 < R) ? src[r + 1] : above;
            float Rv = c[0][r] * qshr1(on);               // elliptic_tools.f90:77-85, left to right
            Rv = Rv + c[1][r] * on;
            Rv = Rv + c[2][r] * qshl1(on);
            Rv = Rv + c[3][r] * qshr1(oc);
            Rv = Rv + c[4][r] * oc;
            Rv = Rv + c[5][r] * qshl1(oc);
            Rv = Rv + c[6][r] * qshr1(om);
            Rv = Rv + c[7][r] * om;
            Rv = Rv + c[8][r] * qshl1(om);
            Rv = Rv - ff[r];                              // :190
            const float upd = oc + alpha * Rv / (-c[4][r]);   // :238
            const int gy = gy0 + r;
            const bool interior = tg.col_int && gy >= 1 && gy <= g.ny - 2;
            dst[r] = interior ? upd : oc;
            if (WRITE_R && t == T - 1) {
                const int ly = w * R + r;
                const bool vy = (ly >= T || tg.y0 <= 0) && (ly < H - T || tg.y0 + H - 1 >= g.ny - 1);
                if (tg.vx && vy && interior) res[mo + (size_t)gy * g.pitch + gx] = Rv;
            }
        }
        if (t + 1 < T) {
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
    const bool in_q = (T & 1) != 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int ly = w * R + r;
        const int gy = gy0 + r;
        const bool vy = (ly >= T || tg.y0 <= 0) && (ly < H - T || tg.y0 + H - 1 >= g.ny - 1);
        const bool interior = tg.col_int && gy >= 1 && gy <= g.ny - 2;
        if (tg.vx && vy && interior) {
            psi_out[mo + (size_t)gy * g.pitch + gx] = in_q ? q[r] : p[r];
            if (WRITE_R && snap) snap[mo + (size_t)gy * g.pitch + gx] = in_q ? q[r] : p[r];
        }
    }
}

}  // namespace

template <int NW, int R>
__device__ __forceinline__ TileGeom make_geom(const Grid& g, int T, int bx, int by, int m)
{
    constexpr int H = NW * R;
    TileGeom tg;
    tg.lane = threadIdx.x & 63;
    tg.w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform, and known to be
    tg.x0 = bx * (64 - 2 * T) - T;
    tg.y0 = by * (H - 2 * T) - T;
    tg.gx = tg.x0 + tg.lane;
    tg.gy0 = tg.y0 + tg.w * R;
    tg.col_in = tg.gx >= 0 && tg.gx < g.nx;
    tg.col_int = tg.gx >= 1 && tg.gx <= g.nx - 2;
    tg.vx = (tg.lane >= T || tg.x0 <= 0) && (tg.lane < 64 - T || tg.x0 + 63 >= g.nx - 1);
    tg.inner = tg.x0 >= 1 && tg.x0 + 63 <= g.nx - 2 && tg.y0 >= 1 && tg.y0 + H - 1 <= g.ny - 2;
    tg.mo = (size_t)m * g.plane;
    tg.ks = g.kstride ? g.kstride : (size_t)g.members * g.plane;
    tg.stamps = nullptr;
    return tg;
}

// (HIP's second launch bound is waves per SIMD: 4 keeps the kernels at 128 VGPRs, 2 at 256, 1 at 512.)
// Rows per lane decide the register budget: the star keeps 9 values per row (two iterates, five weights, f, y1), the
// nine-weight body 13.
template <int NW, int R, bool BARO>
constexpr int q_waves_per_simd()
{
    const int regs = R * (BARO ? 9 : 13);
    return regs <= 78 ? 4 : (regs <= 208 ? 2 : 1);       // star: R <= 8 | R <= 16(+) ; nine weights: R <= 6 | R <= 16
}
// A tile the optimistic body rejects is redone by the exact body in this same workgroup, from its unchanged inputs.
// (Round 2 kept a device-side list and a second kernel for the residual-writing variant: with both bodies inlined it
// spilled to scratch, and a kernel that uses scratch is throttled at dispatch.  The exact body now takes its lane shifts
// where it uses them instead of holding 2 (R + 2) shifted copies, both bodies fit 128 VGPRs, and the list, its kernel and
// one launch per check -- 4.4 us of a single case's 136-us block -- are gone.)
// diag[0] / diag[1] count the tiles redone by residual-writing / plain launches (xee_timing_t).
template <int NW, int R, bool WRITE_R, bool BARO, bool ALPHA1, bool STAMP = false>
__global__ __launch_bounds__(NW * 64, (q_waves_per_simd<NW, R, BARO>())) void k_sweep_q(Grid g, const float* __restrict__ psi_in,
        float* __restrict__ psi_out, const float* __restrict__ base, const float* __restrict__ f,
        float* __restrict__ res, float* __restrict__ snap, const MemberState* __restrict__ st, float alpha, int T, int exact_only,
        int* __restrict__ diag, unsigned long long* __restrict__ stamps, int ntx, int nty)
{
    __shared__ float edge[2][2][NW][64];
    __shared__ int vote[NW];
    // XCD-aware tile order.  The grid is one-dimensional; the dispatcher deals workgroups to the eight XCDs round-robin
    // (workgroup b runs on XCD b % 8), each with its own L2.  The tile list (column, row, member order) is cut into eight
    // contiguous parts and XCD x walks part x, so the tiles in flight on one L2 at a time are neighbours whose halos
    // overlap: the overlap is then read from HBM once instead of once per XCD that touches it.
    const int ntile = ntx * nty * g.members;
    const int part = (ntile + 7) >> 3;
    const int tile = ((int)blockIdx.x & 7) * part + ((int)blockIdx.x >> 3);
    if (tile >= ntile) return;                            // (the grid is padded to a multiple of eight)
    const int m = tile / (ntx * nty);
    const MemberState* member = st ? st + m : nullptr;
    TileGeom tg = make_geom<NW, R>(g, T, tile % ntx, (tile / ntx) % nty, m);
    if constexpr (STAMP) {
        tg.stamps = stamps + 32 * (size_t)tile;
        stamp<STAMP>(tg, 0);                              // workgroup started
        if (threadIdx.x == 0)                             // HW_ID (cu/sh/se in bits 8..15) and XCC_ID
            tg.stamps[7] = (unsigned long long)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) |
                           ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) << 32);
    }
    if (!exact_only) {
        const bool done = tg.inner
            ? fast_tile<NW, R, WRITE_R, BARO, ALPHA1, true, STAMP>(g, tg, edge, vote, psi_in, psi_out, base, f, res, snap, alpha, T, member)
            : fast_tile<NW, R, WRITE_R, BARO, ALPHA1, false, STAMP>(g, tg, edge, vote, psi_in, psi_out, base, f, res, snap, alpha, T, member);
        if (done) {
            stamp<STAMP>(tg, 6);
            return;
        }
    }
    if (threadIdx.x == 0 && diag && !exact_only) atomicAdd(diag + (WRITE_R ? 0 : 1), 1);
    exact_tile<NW, R, WRITE_R, BARO>(g, tg, edge, psi_in, psi_out, base, f, res, snap, alpha, T, member);
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
        // the scalar form, and the packed form (v_pk_mul/fma_f32) in either half of the register pair
        float got;
        if (k & 2) {
            const f2 g2 = quotient_q2(f2{a, (k & 4) ? a : 1.5f}, f2{-d, -d}, f2{quotient_y1(d), quotient_y1(d)});
            const f2 h2 = quotient_q2(f2{1.25f, a}, f2{-d, -d}, f2{quotient_y1(d), quotient_y1(d)});
            got = (k & 4) ? g2.x : h2.y;
        } else {
            got = quotient_q2(a, -d, quotient_y1(d));
        }
        if (__builtin_bit_cast(unsigned, want) != __builtin_bit_cast(unsigned, got)) {
            if (bad == 0 && atomicAdd(mismatches + 1, 1ull) == 0) { first_bad[0] = a; first_bad[1] = d; }
            ++bad;
        }
    }
    if (bad) atomicAdd(mismatches, bad);
}

template <int NW, int R, bool BARO>
static void launch_q_cfg(const Grid& g, const float* psi_in, float* psi_out, const float* base, const float* f,
                         float* res, float* snap, const MemberState* st, float alpha, int T, int exact_only, int* diag, SweepDesc* desc, hipStream_t s)
{
    constexpr int H = NW * R;
    const int ntx = (g.nx - 1 + (64 - 2 * T) - 1) / (64 - 2 * T), nty = (g.ny - 1 + (H - 2 * T) - 1) / (H - 2 * T);
    const long long ntile = (long long)ntx * nty * g.members;
    const dim3 grid((unsigned)(8 * ((ntile + 7) / 8)));    // one-dimensional, a multiple of eight: see the kernel's tile order
    const bool a1 = alpha == 1.0f;     // the protocol's damping: 1.0f * x is x, the multiply is not issued
#define Q_LAUNCH(WR, A1) hipLaunchKernelGGL((k_sweep_q<NW, R, WR, BARO, A1>), grid, dim3(NW * 64), 0, s, g, psi_in, psi_out, base, f, res, snap, st, alpha, T, exact_only, diag, (unsigned long long*)nullptr, ntx, nty)
    if (desc) desc->tiles = ntile;
    if (res) { if (a1) Q_LAUNCH(true, true); else Q_LAUNCH(true, false); }
    else { if (a1) Q_LAUNCH(false, true); else Q_LAUNCH(false, false); }
#undef Q_LAUNCH
}

bool launch_sweep_q(const Grid& g, const float* psi_in, float* psi_out, const float* base, const float* f,
                    float* res, float* snap, const MemberState* st, float alpha, int T, bool barotropic, const SweepOpts& o,
                    int* diag, SweepDesc* desc, hipStream_t s)
{
    // the buffer descriptors address one member's plane with 32-bit byte offsets: a plane of 2 GiB or more (a 23,000 x 23,000
    // grid) goes to the kernel with 64-bit addressing
    if (g.plane * sizeof(float) >= ((size_t)1 << 31)) {
        launch_sweep_tb4(g, psi_in, psi_out, base, f, res, st, alpha, T, barotropic, o, desc, s);
        return false;
    }
    const int exact_only = o.q_exact_only ? 1 : 0;
    // (the nine-weight body needs too many registers at eight rows per lane for two workgroups per CU; at six it fits)
    int cfg = o.tile;
    // Shapes measured and NOT kept (round 3, profiles/r03/tile_shapes.txt; all bit-identical): 4 waves x 16 rows (the same
    // 64x64 tile, 256 VGPRs, two workgroups per CU = 2 waves per SIMD) waits less at the barrier (SQ_WAIT_ANY 26 % of the
    // wave cycles against 38 %) but leaves the VALU idle more often -- with two waves per SIMD a dependent-issue stall is
    // not covered: 18.3 against 16.9 us per iteration of 32 cases; 8 x 16, 8 x 12, 12 x 8 and 16 x 8 (64x96 / 64x128 tiles,
    // 66 % instead of 56 % valid points) run ONE workgroup per CU, whose prologue, vote and stores overlap with nothing:
    // 21-24 us.  The nine-weight body: 4 x 12 28.1, 8 x 12 34.1, 16 x 6 32.3, 8 x 8 36.8 against 27.1 us for 8 x 6.
    if (cfg != 808 && cfg != 806 && cfg != 804) cfg = barotropic ? 808 : 806;
    if ((cfg / 100) * (cfg % 100) - 2 * T < 8) cfg = 808;           // the tile must keep a useful core
    if (desc) { desc->name = barotropic ? "k_sweep_q<star>" : "k_sweep_q<nine>"; desc->T = T; desc->tile_w = 64; desc->tile_h = (cfg / 100) * (cfg % 100); }
#define Q_CASE(NW, R) case NW * 100 + R: \
        if (barotropic) launch_q_cfg<NW, R, true>(g, psi_in, psi_out, base, f, res, snap, st, alpha, T, exact_only, diag, desc, s); \
        else launch_q_cfg<NW, R, false>(g, psi_in, psi_out, base, f, res, snap, st, alpha, T, exact_only, diag, desc, s); \
        break;
    switch (cfg) {
        Q_CASE(8, 8)
        Q_CASE(8, 6)
        Q_CASE(8, 4)
    }
#undef Q_CASE
    return res && snap;            // the residual-writing launch also filled the snapshot
}

void launch_quotient_check(unsigned long long seed, int blocks, int per_thread, int d_exp_lo, int d_exp_hi,
                           unsigned long long* mismatches, float* first_bad, hipStream_t s)
{
    hipLaunchKernelGGL(k_quotient_check, dim3(blocks), dim3(256), 0, s, seed, per_thread, d_exp_lo, d_exp_hi, mismatches, first_bad);
}

}  // namespace xee
