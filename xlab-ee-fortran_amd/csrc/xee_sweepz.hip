// xee_sweepz.hip -- k_sweep_z: the temporally blocked Jacobi sweep STREAMED along z, for large batches.
//
// k_sweep_q (xee_sweepq.hip) blocks time in a 64x64 tile that a workgroup of eight waves steps T times in place: every
// step ends in a barrier, and both directions pay a halo of T points per side (56 % of the computed points are stored at
// T = 8).  Here ONE WAVE owns a strip of 64 columns and walks it along z: at tick j it takes row j of the old iterate and
// computes time level t of row j - t for t = 1..T (each level needs the three rows of the level below that the previous
// two ticks and this one produced), and level T of row j - T leaves for memory.  No barrier; no halo in z except the T
// rows of run-in at the two ends of a chunk of rows; the halo in r stays (lane shifts).  Two chunks of the same strip are
// walked at once, one in each half of a packed register pair, so the arithmetic is the packed fp32 of k_sweep_q.
// 35 % fewer instructions per iteration than k_sweep_q (4.6 M against 7.0 M wave instructions for 32 cases of 1024x512).
//
// Memory: a wave-wide access occupies the CU's address path for its 64 lanes whatever it moves, and ten dword accesses per
// tick cost a third of the walk (ablations: profiles/r03/z_ablation.txt).  So a fetch moves FOUR rows of a plane per
// instruction -- LDS-DMA of 16 bytes per lane, lanes 16 r .. 16 r + 15 carrying the 64 columns of row r, which land as
// [4 rows][64 columns] in LDS without passing a register -- one group of four rows ahead of the walk, and the results of four
// ticks go through LDS into one 16-byte store per chunk: 2.5 memory instructions per tick.  20 KB of LDS per wave, one wave
// per workgroup: a CU holds eight (two per SIMD), so that a launch of up to 2048 waves spreads over all 256 CUs whatever
// order the dispatcher takes them in (four-wave workgroups were packed four to a CU while other CUs idled).
//
// With T = 6 the walk is bound by HBM: 20 bytes per point and launch (psi, f, AP, CQ in, psi out), 4.9 TB/s measured with the
// arithmetic taken out.  A larger T trades that for run-in and halo (T = 8: 48 own columns of 64, and 22 strips x 3 chunk
// pairs x 32 cases = 2112 waves do not fit the 2048 places); DESIGN.md "The streaming sweep" has the numbers.
//
// BAROTROPIC operator (five-point star from the two generating planes AP, CQ), alpha = 1, plain launches (no residual
// plane): the host (xee_api.hip) gives the residual-writing launch at the end of a check block and any remainder to
// k_sweep_q.  Results are bit-identical to k_sweep_q's and the reference's:
//   * the division is k_sweep_q's hoisted reciprocal (y1 once per row, five operations per point), exact for every quotient
//     whose operands pass the bounds that k_sweep_q checks per tile.  Here the HOST checks them per member before a solve
//     (k_z_extremes: |coe5| in [2^-120, 2^120] and min |f| 2^-26 >= max(2^-100, max |coe5| 2^-120), which puts every non-zero
//     |R - f| above the lower bound by construction -- xee_sweepq.hip "Zero residuals"); the upper bound only guards
//     against an overflow inside the quotient, and an overflow there ends as a NaN or an infinity in the point's own result,
//     which the wave looks for in what it stores (and in what enters its last level: see there);
//   * points that never update (rim, rows and columns outside the domain -- clamped copies) are kept idle by y1 = -0 alone:
//     their quotient is a zero whatever their "residual", and psi + (+-0) = psi unless psi is -0;
//   * a wave that loaded a -0 or stored a non-finite value walks its chunks again with the exact body: the reference's
//     nine-term sum in its order, the IEEE division, idle points by selection -- k_sweep_q's exact_tile, streamed.
#include "xee_device.h"

#include <cfloat>
#include <type_traits>
#include <utility>

#pragma clang fp contract(off)

namespace xee {

namespace {

typedef float f2 __attribute__((ext_vector_type(2)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float zshr1(float src)   // lane l <- lane l-1 (lane 0 gets 0)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, src), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float zshl1(float src)   // lane l <- lane l+1 (lane 63 gets 0)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, src), 0x130, 0xf, 0xf, true));
}
__device__ __forceinline__ f2 zshr2(f2 v) { return f2{zshr1(v.x), zshr1(v.y)}; }
__device__ __forceinline__ f2 zshl2(f2 v) { return f2{zshl1(v.x), zshl1(v.y)}; }

typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t zrsrc(const float* plane_of_member, const Grid& g, int more = 0)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(plane_of_member), 0, (int)(g.plane * sizeof(float)) + more, 0x00020000);
}

__device__ __forceinline__ float quotient_y1(float d)   // the loop-invariant half of the IEEE quotient by d (xee_sweepq.hip)
{
    const float y0 = __builtin_amdgcn_rcpf(d);
    const float e = __builtin_fmaf(-d, y0, 1.0f);
    return __builtin_fmaf(e, y0, y0);
}
__device__ __forceinline__ f2 z_q2(f2 a, f2 nd, f2 y1)   // a / d with y1 = quotient_y1(d), nd = -d (xee_sweepq.hip)
{
    const f2 q0 = a * y1;
    const f2 r0 = __builtin_elementwise_fma(nd, q0, a);
    const f2 q1 = __builtin_elementwise_fma(r0, y1, q0);
    const f2 r1 = __builtin_elementwise_fma(nd, q1, a);
    return __builtin_elementwise_fma(r1, y1, q1);
}

// compile-time loops: body(std::integral_constant<int, i>{}) for i = 0 .. N-1 (indices into the register arrays must be
// constants: with a run-time index, or a loop the compiler does not unroll, the arrays live in scratch memory)
template <class F, int... Is>
__device__ __forceinline__ void z_for_impl(F&& body, std::integer_sequence<int, Is...>) { (body(std::integral_constant<int, Is>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void z_for(F&& body) { z_for_impl(body, std::make_integer_sequence<int, N>{}); }

}  // namespace

// One wave: strip s (own columns s W + 1 .. s W + W, W = 64 - 2 T), member m, chunks c and c + P of 2 P: chunk c covers the
// interior rows [1 + c Lc, 1 + (c + 1) Lc), clipped to ny - 2, and is walked from T + 2 rows before its first to T rows past
// its last so that level T of its own rows is complete.
template <int T>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_sweep_z(Grid g, const float* __restrict__ psi_in, float* __restrict__ psi_out,
        const float* __restrict__ base, const float* __restrict__ f, const MemberState* __restrict__ st, int nstrip, int P, int Lc, int nunits,
        int exact_only, int* __restrict__ diag)
{
    constexpr int S = (T + 1 + 3) & ~3;                   // rows whose weights are in flight (j - T .. j), rounded to whole fetch groups
    constexpr int U = S;                                  // ticks per turn of the unrolled loop: the slots' rotation, the history's parity
    constexpr int W = 64 - 2 * T;                         // and the fetch groups all come round
    static_assert(W % 4 == 0 && U % 4 == 0, "the 16-byte stores take the strip's own columns in fours");
    constexpr int NG = 2, NGP = 3;                        // groups of four rows in LDS: the weights' planes / psi (two rows of history)
    __shared__ __attribute__((aligned(16))) float fring[NG][6][4][64];   // f a, f b, AP a, AP b, CQ a, CQ b
    __shared__ __attribute__((aligned(16))) float pring[NGP][2][4][64];
    __shared__ __attribute__((aligned(16))) float obuf[2][4][64];
    const int lane = threadIdx.x;
    // XCD-aware order (as k_sweep_q): workgroup b runs on XCD b % 8; the list of units (strip fastest, then chunk pair, then
    // member) is cut into eight contiguous parts, one per XCD, so that strips whose columns overlap meet in one L2
    const int part = (int)gridDim.x >> 3;                 // (the grid is a multiple of eight)
    const int wv = ((int)blockIdx.x & 7) * part + ((int)blockIdx.x >> 3);
    if (wv >= nunits) return;
    const int per_member = nstrip * P;
    const int m = wv / per_member;
    if (st && !st[m].active) return;
    const int rem = wv - m * per_member;
    const int s = rem % nstrip, c = rem / nstrip;
    // the strip's own columns are carried by lanes lo .. lo + W - 1: T lanes of halo to the left, except in strip 0, whose left
    // neighbour is the rim column (a fetch must not start left of the plane: a 16-byte access that starts out of range is
    // dropped whole, the columns of it that exist included)
    const int lo = s == 0 ? 1 : T;
    const int x0 = s * W + 1 - lo;                        // column of lane 0
    const int gx = x0 + lane;
    const bool col_int = gx >= 1 && gx <= g.nx - 2;
    const int nvalid = (g.nx - 2 - s * W) < W ? (g.nx - 2 - s * W) : W;   // own columns that are interior (the last strip: fewer)
    const size_t mo = (size_t)m * g.plane, ks = g.kstride ? g.kstride : (size_t)g.members * g.plane;
    // (read descriptors reach 16 bytes past the plane: a 16-byte fetch that straddles the end of the last row must not be
    // dropped -- what it reads there is the next member's first columns or the allocation's slack (kPlaneSlack), and only
    // lanes right of the domain get it)
    const rsrc_t pr = zrsrc(psi_in + mo, g, 16), fr = zrsrc(f + mo, g, 16), apr = zrsrc(base + mo, g, 16), cqr = zrsrc(base + ks + mo, g, 16);
    const rsrc_t outr = zrsrc(psi_out + mo, g);
    const int ya = 1 + c * Lc, yb = 1 + (c + P) * Lc;     // first rows of the two chunks (.x and .y of every pair)
    const int yend_a = ya + Lc < g.ny - 1 ? ya + Lc : g.ny - 1, yend_b = yb + Lc < g.ny - 1 ? yb + Lc : g.ny - 1;
    const int pitch4 = g.pitch * 4;

    typedef __attribute__((address_space(3))) void* lds_t;
    const int frow = lane >> 4, fcol4 = (x0 + 4 * (lane & 15)) * 4;   // this lane's part of a fetch: row frow of the group, 16 bytes from fcol4
    // rows r0 .. r0 + 3 (relative to the chunks' first rows) of every plane, both chunks: eight LDS-DMA loads.  Rows outside
    // the plane are clamped copies (finite, and idle: see the weights)
    auto fetch = [&](int r0, int fs, int ps) {
        int ra = ya + r0 + frow, rb = yb + r0 + frow;
        ra = ra < 0 ? 0 : (ra > g.ny - 1 ? g.ny - 1 : ra);
        rb = rb < 0 ? 0 : (rb > g.ny - 1 ? g.ny - 1 : rb);
        const int oa = ra * pitch4 + fcol4, ob = rb * pitch4 + fcol4;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(pr, (lds_t)&pring[ps][0][0][0], 16, oa, 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(pr, (lds_t)&pring[ps][1][0][0], 16, ob, 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(fr, (lds_t)&fring[fs][0][0][0], 16, oa, 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(fr, (lds_t)&fring[fs][1][0][0], 16, ob, 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(apr, (lds_t)&fring[fs][2][0][0], 16, oa, 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(apr, (lds_t)&fring[fs][3][0][0], 16, ob, 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(cqr, (lds_t)&fring[fs][4][0][0], 16, oa, 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(cqr, (lds_t)&fring[fs][5][0][0], 16, ob, 0, 0, 0);
    };

    // stores: lane l < W carries columns 4 (l % (W/4)) .. + 3 of row l / (W/4) of a group of four result rows
    constexpr int GC = W / 4;
    const int srow = lane / GC, sgc = lane - srow * GC;
    const bool s_lane = lane < 4 * GC;
    const bool s_full = s_lane && 4 * sgc + 4 <= nvalid;
    const int s_part = s_lane && !s_full && 4 * sgc < nvalid ? nvalid - 4 * sgc : 0;   // 1..3 columns of a last, partial four
    const int scol4 = (s * W + 1 + 4 * sgc) * 4;
    const int o_w = ((lane - lo) & 63);                   // a tick's row goes to LDS shifted: own column k at [k]

    const int j_first = -T - 2;                           // relative row of the first tick (row = chunk start + j)
    const int n_ticks = Lc + 2 * T + 4;

    // ---- one walk of the two chunks; EXACT: the fallback body.  Returns whether the results may stand. ------------------
    auto walk = [&](auto exact_tag) -> bool {
        constexpr bool EXACT = decltype(exact_tag)::value;
        // Weights of the rows in flight, slot = (tick index) mod S, UNMASKED.  y1: the optimistic body's reciprocal, -0 where
        // the point never updates; the exact body keeps 1 / 0 ("updates" / "idle") there instead.
        f2 wA[S], wC[S], w5[S], y1[S], ff[S];
        f2 hist[T][2];                                    // hist[l][q]: level l's row of the tick with parity q (level 0 lives in LDS)
#pragma unroll
        for (int k = 0; k < S; ++k) { wA[k] = wC[k] = w5[k] = y1[k] = ff[k] = f2{0.f, 0.f}; }
#pragma unroll
        for (int l = 0; l < T; ++l) hist[l][0] = hist[l][1] = f2{0.f, 0.f};
        bool bad = false;                                 // (optimistic body) a -0 among the loaded psi, a non-finite result
        // the psi rows "before the first": zeros (they only reach values that are never stored)
#pragma unroll
        for (int r = 0; r < 4; ++r) { pring[NGP - 1][0][r][lane] = 0.0f; pring[NGP - 1][1][r][lane] = 0.0f; }
        int fs = 0, ps = 0;                               // ring slots of the group being consumed
        fetch(j_first, 0, 0);
        for (int j0 = 0; j0 < n_ticks; j0 += U) {
            z_for<U>([&](auto uc) {
                constexpr int u = decltype(uc)::value;
                constexpr int sub = u & 3;
                const int jr = j_first + j0 + u;          // relative row consumed at this tick
                const int rowa = ya + jr, rowb = yb + jr;
                constexpr int slot = u % S, q = u & 1;
                if constexpr (sub == 0) {
                    // the next group goes out, then this one must have landed: at most the eight younger fetches may be in
                    // flight (stores may retire out of order with loads, but loads retire in order among themselves: with an
                    // older fetch outstanding the count is above eight whatever the stores do)
                    const int fn = fs + 1 >= NG ? 0 : fs + 1, pn = ps + 1 >= NGP ? 0 : ps + 1;
                    fetch(jr + 4, fn, pn);
                    __builtin_amdgcn_s_waitcnt((8 & 0xf) | (0x7 << 4) | (0xf << 8));   // vmcnt(8)
                }
                // ---- level 0: row jr of the old iterate, the two rows before it, and what row jr's weights are made of.  Read
                // with inline assembly: the compiler cannot tell the slot an LDS-DMA fetch fills from the slots read here and
                // would drain vmcnt(0) in front of every read of these arrays that it can see (measured: 15.7 -> 22.6 us).
                f2 p0, p1, p2, fa, ap, cq;
                {
                    const int pm = ps >= 1 ? ps - 1 : NGP - 1;    // the group before
                    const unsigned aP = (unsigned)(size_t)(lds_t)&pring[ps][0][0][lane], aM = (unsigned)(size_t)(lds_t)&pring[pm][0][0][lane];
                    const unsigned aF = (unsigned)(size_t)(lds_t)&fring[fs][0][0][lane];
                    // [chunk][row][64]: row r of chunk a at offset r (x 64 dwords), of chunk b at r + 4; plane k of fring at 8 k
                    constexpr int r1 = (sub + 3) & 3, r2 = (sub + 2) & 3;
                    asm volatile("ds_read2st64_b32 %0, %6 offset0:%9 offset1:%10\n\t"
                                 "ds_read2st64_b32 %1, %7 offset0:%11 offset1:%12\n\t"
                                 "ds_read2st64_b32 %2, %8 offset0:%13 offset1:%14\n\t"
                                 "ds_read2st64_b32 %3, %15 offset0:%9 offset1:%10\n\t"
                                 "ds_read2st64_b32 %4, %15 offset0:%16 offset1:%17\n\t"
                                 "ds_read2st64_b32 %5, %15 offset0:%18 offset1:%19\n\t"
                                 "s_waitcnt lgkmcnt(0)"
                                 : "=&v"(p0), "=&v"(p1), "=&v"(p2), "=&v"(fa), "=&v"(ap), "=&v"(cq)
                                 : "v"(aP), "v"(sub >= 1 ? aP : aM), "v"(sub >= 2 ? aP : aM),
                                   "n"(sub), "n"(sub + 4), "n"(r1), "n"(r1 + 4), "n"(r2), "n"(r2 + 4),
                                   "v"(aF), "n"(sub + 8), "n"(sub + 12), "n"(sub + 16), "n"(sub + 20) : "memory");
                }
                if constexpr (sub == 3) { fs = fs + 1 >= NG ? 0 : fs + 1; ps = ps + 1 >= NGP ? 0 : ps + 1; }
                if constexpr (!EXACT)
                    bad = bad || __builtin_bit_cast(unsigned, p0.x) == 0x80000000u || __builtin_bit_cast(unsigned, p0.y) == 0x80000000u;
                // ---- levels 1 .. T, each on row jr - t, each from the level below: this tick's row (`up`), the previous tick's
                // (`oc`) and the one before (`om`)
                f2 up = p0;
                z_for<T>([&](auto tc) {
                    constexpr int t = decltype(tc)::value + 1;
                    constexpr int sl = ((u - t) % S + S) % S;          // slot of row jr - t
                    constexpr int slm = ((u - t - 1) % S + S) % S;     // ... of the row below it: its CQ is this row's coe8
                    f2 oc, om;
                    if constexpr (t == 1) { oc = p1; om = p2; }
                    else {
                        oc = hist[t - 1][q ^ 1];          // level t-1, row jr - t
                        om = hist[t - 1][q];              // level t-1, row jr - t - 1
                        hist[t - 1][q] = up;              // level t-1, row jr - t + 1, for the next two ticks
                    }
                    if constexpr (!EXACT) {
                        // The reference's sum has nine terms; the four this star leaves out are (+-0) x a diagonal neighbour: zero
                        // unless that neighbour is not finite, when they make the sum a NaN.  A non-finite value stays one at
                        // its own point from level to level, so it is enough to look at what enters the LAST level (every lane:
                        // the diagonal neighbours of the own points are among them) -- and at what is stored, below.
                        if constexpr (t == T)
                            bad = bad || (__builtin_bit_cast(unsigned, up.x) & 0x7f800000u) == 0x7f800000u ||
                                  (__builtin_bit_cast(unsigned, up.y) & 0x7f800000u) == 0x7f800000u;
                        // the star of elliptic_tools.f90:77-85 in its order (terms 2, 4, 5, 6, 8); coe4 * psi(i-1) = (AP * psi)(i-1)
                        const f2 Pm = wA[sl] * oc;
                        f2 Rv = wC[sl] * up;
                        Rv = f2{Rv.x + zshr1(Pm.x), Rv.y + zshr1(Pm.y)};   // (two v_add_f32_dpp: a packed add would need the shifts as moves first)
                        Rv = Rv + w5[sl] * oc;
                        Rv = Rv + f2{wA[sl].x * zshl1(oc.x), wA[sl].y * zshl1(oc.y)};
                        Rv = Rv + wC[slm] * om;
                        Rv = Rv - ff[sl];                 // :190
                        up = oc + z_q2(Rv, w5[sl], y1[sl]);   // :238 (alpha = 1; -d = coe5)
                    } else {
                        // elliptic_tools.f90:46-54 with b = 0, then :77-85 left to right, :190, :238 -- as exact_tile (xee_sweepq.hip)
                        const f2 zero = f2{0.0f, 0.0f};
                        const f2 Am = zshr2(wA[sl]);
                        const f2 c0 = -(zero + zero), c1 = wC[sl] + (zero - zero), c2 = zero + zero, c3 = Am - (zero - zero);
                        const f2 c5 = wA[sl] + (zero - zero), c6 = zero + zero, c7 = wC[slm] - (zero - zero), c8 = -(zero + zero);
                        f2 Rv = c0 * zshr2(up);
                        Rv = Rv + c1 * up;
                        Rv = Rv + c2 * zshl2(up);
                        Rv = Rv + c3 * zshr2(oc);
                        Rv = Rv + w5[sl] * oc;
                        Rv = Rv + c5 * zshl2(oc);
                        Rv = Rv + c6 * zshr2(om);
                        Rv = Rv + c7 * om;
                        Rv = Rv + c8 * zshl2(om);
                        Rv = Rv - ff[sl];
                        const f2 upd = f2{oc.x + Rv.x / (-w5[sl].x), oc.y + Rv.y / (-w5[sl].y)};
                        up = f2{y1[sl].x != 0.0f ? upd.x : oc.x, y1[sl].y != 0.0f ? upd.y : oc.y};
                    }
                });
                // ---- level T of row jr - T: into the group of four result rows; the fourth sends the group to memory
                obuf[0][sub][o_w] = up.x;
                obuf[1][sub][o_w] = up.y;
                if constexpr (sub == 3) {
                    const u4 va = *(const u4*)&obuf[0][srow][4 * sgc], vb = *(const u4*)&obuf[1][srow][4 * sgc];
                    const int ra_ = rowa - T - 3 + srow, rb_ = rowb - T - 3 + srow;
                    const bool oka = ra_ >= ya && ra_ < yend_a, okb = rb_ >= yb && rb_ < yend_b;
                    const int offa = ra_ * pitch4 + scol4, offb = rb_ * pitch4 + scol4;
                    // (a lane with nothing to store aims out of range: the access is dropped)
                    __builtin_amdgcn_raw_buffer_store_b128(va, outr, (oka && s_full) ? offa : (int)0x80000000, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b128(vb, outr, (okb && s_full) ? offb : (int)0x80000000, 0, 0);
                    if (nvalid & 3) {                     // (the last strip only) the columns of a partial four, one by one
#pragma unroll
                        for (int k = 0; k < 3; ++k) {
                            __builtin_amdgcn_raw_buffer_store_b32(va[k], outr, (oka && k < s_part) ? offa + 4 * k : (int)0x80000000, 0, 0);
                            __builtin_amdgcn_raw_buffer_store_b32(vb[k], outr, (okb && k < s_part) ? offb + 4 * k : (int)0x80000000, 0, 0);
                        }
                    }
                    if constexpr (!EXACT) {
                        // an exponent of all ones in anything a lane stores (or would: spare lanes and rows carry finite values too)
#pragma unroll
                        for (int k = 0; k < 4; ++k)
                            bad = bad || (va[k] & 0x7f800000u) == 0x7f800000u || (vb[k] & 0x7f800000u) == 0x7f800000u;
                    }
                }
                // ---- the new row's weights take a free slot (row jr - T - 1's, or one nobody holds)
                const bool ia = col_int && rowa >= 1 && rowa <= g.ny - 2, ib = col_int && rowb >= 1 && rowb <= g.ny - 2;
                const f2 am = zshr2(ap);
                const f2 c5 = -(am + ap + wC[(slot + S - 1) % S] + cq);   // elliptic_tools.f90:50, left to right; Cm = the previous row's CQ
                wA[slot] = ap;
                wC[slot] = cq;
                w5[slot] = c5;
                if constexpr (EXACT) y1[slot] = f2{ia ? 1.0f : 0.0f, ib ? 1.0f : 0.0f};
                else y1[slot] = f2{ia ? quotient_y1(-c5.x) : -0.0f, ib ? quotient_y1(-c5.y) : -0.0f};
                ff[slot] = fa;
            });
        }
        __builtin_amdgcn_s_waitcnt((0x7 << 4) | (0xf << 8));   // vmcnt(0): the fetch past the end has landed before LDS is used again
        return __builtin_amdgcn_ballot_w64(bad) == 0;
    };

    bool ok = false;
    if (!exact_only) ok = walk(std::false_type{});
    if (!ok) {
        if (lane == 0 && diag) atomicAdd(&diag[1], 1);
        walk(std::true_type{});
    }
}

// ---- per member: extremes of |coe5| and |f| over the interior (the host's test of k_sweep_z's preconditions) ------------------
// zstat[m] = {min |coe5|, max |coe5|, min |f|} as the bits of non-negative floats (whose order is the order of the bits as
// unsigned integers; a NaN sorts above infinity and so fails the test of the maximum).  Set to {~0, 0, ~0} before.
__global__ __launch_bounds__(256) void k_z_extremes(Grid g, const float* __restrict__ base, const float* __restrict__ f, unsigned* __restrict__ zstat)
{
    const int m = blockIdx.y;
    const size_t mo = (size_t)m * g.plane, ks = g.kstride ? g.kstride : (size_t)g.members * g.plane;
    const float* __restrict__ ap = base + mo;
    const float* __restrict__ cq = base + ks + mo;
    const float* __restrict__ fm = f + mo;
    unsigned dmin = 0xffffffffu, dmax = 0u, fmin = 0xffffffffu;
    for (int j = 1 + (int)blockIdx.x; j <= g.ny - 2; j += (int)gridDim.x)
        for (int i = 1 + (int)threadIdx.x; i <= g.nx - 2; i += 256) {
            const size_t o = (size_t)j * g.pitch + i;
            const float c5 = -(ap[o - 1] + ap[o] + cq[o - g.pitch] + cq[o]);   // elliptic_tools.f90:50 (b = 0)
            const unsigned d = __builtin_bit_cast(unsigned, c5) & 0x7fffffffu, a = __builtin_bit_cast(unsigned, fm[o]) & 0x7fffffffu;
            dmin = d < dmin ? d : dmin; dmax = d > dmax ? d : dmax; fmin = a < fmin ? a : fmin;
        }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const unsigned a = __shfl_xor(dmin, o), b = __shfl_xor(dmax, o), c = __shfl_xor(fmin, o);
        dmin = a < dmin ? a : dmin; dmax = b > dmax ? b : dmax; fmin = c < fmin ? c : fmin;
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMin(&zstat[3 * m + 0], dmin);
        atomicMax(&zstat[3 * m + 1], dmax);
        atomicMin(&zstat[3 * m + 2], fmin);
    }
}

__global__ void k_z_extremes_init(unsigned* __restrict__ zstat, int members)
{
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m < members) { zstat[3 * m + 0] = 0xffffffffu; zstat[3 * m + 1] = 0u; zstat[3 * m + 2] = 0xffffffffu; }
}

void launch_z_extremes(const Grid& g, const float* base, const float* f, unsigned* zstat, hipStream_t s)
{
    hipLaunchKernelGGL(k_z_extremes_init, dim3((g.members + 63) / 64), dim3(64), 0, s, zstat, g.members);
    const int by = g.ny - 2 < 64 ? (g.ny - 2 < 1 ? 1 : g.ny - 2) : 64;
    hipLaunchKernelGGL(k_z_extremes, dim3(by, g.members), dim3(256), 0, s, g, base, f, zstat);
}

// the host's reading of k_z_extremes for one member (see the header of this file)
bool z_member_ok(const unsigned* zstat3)
{
    const float dmin = __builtin_bit_cast(float, zstat3[0]), dmax = __builtin_bit_cast(float, zstat3[1]), fmin = __builtin_bit_cast(float, zstat3[2]);
    if (!(dmin >= 0x1p-120f) || !(dmax <= 0x1p120f)) return false;       // (false for a NaN too)
    const float a_lo = dmax * 0x1p-120f > 0x1p-100f ? dmax * 0x1p-120f : 0x1p-100f;
    return fmin * 0x1p-26f >= a_lo;
}


// Chunk pairs per strip for a launch over `members` cases that shares the device with `streams` - 1 launches like it.
// The device holds 2048 waves of this kernel (eight per CU); a launch of n waves takes ceil(n / places) rounds, and a
// chunk of Lc rows costs Lc + 2 T + 4 ticks: the P in 1 .. (ny - 2) / 64 with the largest product of the rounds' filling
// and Lc / (Lc + 2 T + 4).  Returns 0 where that product is below `min_eff_percent`: k_sweep_q's tiles use the device better
// there (measured break-even near 70: 32 cases of 1024x512 reach 79 and gain 10 %, 16 cases reach 39 and lose 9 %).
int sweep_z_T() { return 6; }

int sweep_z_pairs(const Grid& g, int members, int streams, int min_eff_percent)
{
    constexpr int T = 6, W = 64 - 2 * T;
    if (g.nx < 8 || g.ny < 8 || members < 1) return 0;
    const long long strips = (g.nx - 2 + W - 1) / W;
    const long long places = 2048 / (streams > 0 ? streams : 1);
    const int pmax = (g.ny - 2) / 64 > 1 ? (g.ny - 2) / 64 : 1;
    int best = 0;
    double best_eff = -1.0;
    for (int P = 1; P <= pmax && P <= 64; ++P) {
        const long long n = strips * members * P;
        const long long rounds = (n + places - 1) / places;
        const int Lc = (g.ny - 2 + 2 * P - 1) / (2 * P);
        const double eff = (double)n / (double)(rounds * places) * (double)Lc / (double)(Lc + 2 * T + 4);
        if (eff > best_eff) { best_eff = eff; best = P; }
    }
    return best_eff * 100.0 >= (double)min_eff_percent ? best : 0;
}

void launch_sweep_z(const Grid& g, const float* psi_in, float* psi_out, const float* base, const float* f,
                    const MemberState* st, int T, int P, int exact_only, int* diag, SweepDesc* desc, hipStream_t s)
{
    constexpr int W = 64 - 2 * 6;
    const int nstrip = (g.nx - 2 + W - 1) / W;
    const int Lc = (g.ny - 2 + 2 * P - 1) / (2 * P);
    const long long units = (long long)nstrip * P * g.members;
    if (desc) { desc->name = "k_sweep_z<star>"; desc->T = T; desc->tile_w = 64; desc->tile_h = Lc; desc->tiles = units; desc->streamed = true; }
    // (T = 8 was measured too: 17.3 against 17.6 us per iteration at equal P -- what it saves in bytes it pays in halo and run-in,
    // and its 22 strips leave no P that fills the 2048 places; profiles/r03/z_ablation.txt)
    hipLaunchKernelGGL((k_sweep_z<6>), dim3((unsigned)(8 * ((units + 7) / 8))), dim3(64), 0, s, g, psi_in, psi_out, base, f, st, nstrip, P, Lc,
                       (int)units, exact_only, diag);
}

}  // namespace xee
