// xee_sweepz.hip -- k_sweep_z: the temporally blocked Jacobi sweep STREAMED along z, for large batches.
//
// k_sweep_q (xee_sweepq.hip) blocks time in a 64x64 tile that a workgroup of eight waves steps T times in place: every
// step ends in a barrier, and both directions pay a halo of T points per side (56 % of the computed points are stored at
// T = 8).  Here ONE WAVE owns a strip of 64 columns and walks it along z: at tick j it takes row j of the old iterate and
// computes time level t of row j - t for t = 1..T (each level needs the three rows of the level below that the previous
// two ticks and this one produced), and level T of row j - T leaves for memory.  No barrier; no halo in z except the T
// rows of run-in at the two ends of a chunk of rows; the halo in r stays (lane shifts).  Two chunks of the same strip are
// walked at once, one in each half of a packed register pair, so the arithmetic is the packed fp32 of k_sweep_q.
// 37 % fewer instructions per iteration than k_sweep_q (4.4 M against 7.0 M wave instructions for 32 cases of 1024x512).
//
// Memory: ten dword accesses per tick cost a third of the walk, and so do 2.5 wide ones -- the cost follows the bytes
// (ablations: profiles/r03/z_ablation.txt).  A fetch moves two rows of a plane for both halves per instruction -- LDS-DMA of
// 16 bytes per lane, which lands as [plane][half][row][64 columns] in LDS without passing a register -- three groups (member
// pairs: four) ahead of the one being read; the next tick's row is read from LDS a tick ahead; the results of two ticks go through
// LDS into one 16-byte store: 2.5 memory instructions per tick.  17 KB of LDS per wave (member pairs: 16), one wave per
// workgroup, at most two per SIMD (amdgpu_waves_per_eu): a launch of up to 2048 waves spreads over all 256 CUs whatever order
// the dispatcher takes them in (four-wave workgroups, four waves per SIMD allowed, were packed four to a CU while other CUs
// idled: 14.3 us per iteration with or without their memory operations).
//
// Where every member has its own operator the walk is bound by HBM at T = 6: 20 bytes per point and launch (psi, f, AP, CQ in,
// psi out), 4.9 TB/s measured with the arithmetic taken out.  A larger T trades that for run-in and halo (T = 8: 48 own columns
// of 64, and 22 strips x 3 chunk pairs x 32 cases = 2112 waves do not fit the 2048 places).  Members that share their operator
// read ONE copy of AP and CQ (Grid.base_shared; 12.5 bytes per point and launch for 16 members) and are walked in pairs (MP,
// below): the VALU then bounds the walk (10.4 us per iteration of 32 cases with the memory operations removed, 11 - 12 with
// them).  DESIGN.md "The streaming sweep" has the numbers.
//
// BAROTROPIC operator (five-point star from the two generating planes AP, CQ) or, for member pairs that share AP and CQ, the
// nine-weight BAROCLINIC one (B9 below); alpha = 1; plain launches and the residual-writing launch at the end of a check block
// (WR below); the host (xee_api.hip) gives a remainder that is no whole launch to k_sweep_q.  Results are bit-identical to
// k_sweep_q's and the reference's:
//   * the division is k_sweep_q's hoisted reciprocal (y1 once per row, five operations per point), exact for every quotient
//     whose operands pass the bounds that k_sweep_q checks per tile.  Here the HOST checks them per member before a solve
//     (k_z_extremes: |coe5| in [2^-120, 2^120] and min |f| 2^-26 >= max(2^-100, max |coe5| 2^-120), which puts every non-zero
//     |R - f| above the lower bound by construction -- xee_sweepq.hip "Zero residuals"); the upper bound only guards
//     against an overflow inside the quotient, and an overflow there ends as a NaN or an infinity in the point's own result,
//     which the wave looks for in what it stores (and in what enters its last level: see there);
//   * points that never update (the rim; rows outside the domain -- clamped copies; lanes right of it -- whatever follows the row
//     in memory) are kept idle by y1 = -0 alone:
//     their quotient is a zero whatever their "residual", and psi + (+-0) = psi unless psi is -0;
//   * a wave that loaded a -0 or stored a non-finite value walks its chunks again with the exact body: the reference's
//     nine-term sum in its order, the IEEE division, idle points by selection -- k_sweep_q's exact_tile, streamed.
// The nine-weight operator (BAROCLINIC): 16 registers of weights per row in flight (eight pairs; coe5 and y1 are one value for a
// pair of members that share AP and CQ) fit at T = 4 (236 VGPRs, no spill; at T = 6 they spill, and a spill's reload drains the
// prefetch queue: the first attempt, tools/experiments/sweepz_two_kernels_and_nine_weight.hip.txt, lost to k_sweep_q).  20 bytes per
// point and launch (psi, f, BX, BY in, psi out) for four iterations: HBM-bound, 22.2 against k_sweep_q<nine>'s 27.2 us per iteration
// of 32 cases (profiles/r03/z_ablation.txt).
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

// a weight is a pair (one value per chunk) or, where both halves of the register pair work on the SAME rows of two members that
// share their operator, one value for both
__device__ __forceinline__ f2 b2(f2 v) { return v; }
__device__ __forceinline__ f2 b2(float v) { return f2{v, v}; }
__device__ __forceinline__ f2 zshrW(f2 v) { return zshr2(v); }
__device__ __forceinline__ float zshrW(float v) { return zshr1(v); }

// compile-time loops: body(std::integral_constant<int, i>{}) for i = 0 .. N-1 (indices into the register arrays must be
// constants: with a run-time index, or a loop the compiler does not unroll, the arrays live in scratch memory)
template <class F, int... Is>
__device__ __forceinline__ void z_for_impl(F&& body, std::integer_sequence<int, Is...>) { (body(std::integral_constant<int, Is>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void z_for(F&& body) { z_for_impl(body, std::make_integer_sequence<int, N>{}); }

}  // namespace

// One wave: strip s (own columns s W + 1 .. s W + W, W = 64 - 2 T), member m, chunks c and c + P of 2 P: chunk c covers the
// interior rows [1 + c Lc, 1 + (c + 1) Lc), clipped to ny - 2, and is walked from T rows before its first to T rows past its
// last so that level T of its own rows is complete.
// Fetch groups of TWO rows: one 16-byte LDS-DMA instruction per plane and group -- lanes 0..31 carry rows r, r + 1 of chunk a,
// lanes 32..63 those of chunk b (16 lanes x 16 bytes = the 64 columns of a row) -- landing as [chunk][row][64] in LDS.
// (Groups of four rows, one group ahead, 20 KB: the same speed.)
// MP ("member pairs"; the launcher chooses it where all members share AP and CQ): the two halves of the register pair are the SAME
// chunk of TWO members instead of two chunks of one -- the weights of a row (AP, CQ, coe5, y1) then are one value for both halves:
// half the registers and half the arithmetic to make them, three fetches per group instead of four (AP and CQ share one: lanes
// 0..31 AP's two rows, lanes 32..63 CQ's), 3 KB of LDS per group and so one more group in flight.
// WR: the launch at the end of a check block -- the residual of its LAST iteration (do_elliptic(psi) - f, elliptic_tools.f90:189-190)
// goes to `res` and a second copy of the new iterate to `snap` (may be null), for the members that are active (a stopped member's
// snapshot is its result: nothing of it may be written).
// B9: the nine-weight operator (BAROCLINIC) for member pairs that share AP and CQ -- coe5 and y1 stay one value for both halves, the
// other eight weights are a pair; T = 4 (at T = 6 the weights of the eight rows in flight do not fit 256 registers).  The walk is
// HBM-bound then, so it reads the solver coefficient b (k_base_planes keeps a copy as the fifth generating plane) instead of BX and
// BY and makes the two itself, by cal_coe's own divisions (elliptic_tools.f90:41-44): 16 instead of 20 bytes per point and launch.
template <int T, bool MP, bool WR, bool B9 = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_sweep_z(Grid g, const float* __restrict__ psi_in, float* __restrict__ psi_out,
        const float* __restrict__ base, const float* __restrict__ f, const MemberState* __restrict__ st, int nstrip, int P, int Lc, int nunits,
        int exact_only, int* __restrict__ diag, float* __restrict__ res, float* __restrict__ snap)
{
    // rows whose weights are in flight: j - T .. j, and a spare slot so that one turn of the unrolled loop (the slots' rotation
    // x the two-row groups and the history's parity) is S ticks
    constexpr int S = (T + 2) & ~1;
    constexpr int U = (S % 4 == 0) ? S : 2 * S;           // (... and of the four places the psi rows rotate through)
    constexpr int W = 64 - 2 * T;
    static_assert(W % 4 == 0 && U % 2 == 0 && U % S == 0, "the 16-byte stores take the strip's own columns in fours");
    static_assert(!B9 || MP, "the nine-weight walk is built for member pairs");
    constexpr int NF = B9 ? 4 : (MP ? 3 : 4);             // fetch instructions per group: psi, f, AP, CQ (MP: AP and CQ in one; B9: and b)
    constexpr int DG = (MP && !B9) ? 4 : 3;               // groups in flight ahead of the one being read (20 KB of LDS per wave at most)
    constexpr int NG = DG + 1;                            // groups in LDS
    __shared__ __attribute__((aligned(16))) float ring[NG][NF][2][2][64];   // [group][plane][chunk or member][row][column]: 4 KB (MP: 3 KB) per group
    __shared__ __attribute__((aligned(16))) float obuf[2][2][64];
    __shared__ __attribute__((aligned(16))) float robuf[WR ? 2 : 1][2][64];   // (WR) the residual rows, as obuf
    const int lane = threadIdx.x;
    const int part = (int)gridDim.x >> 3;                 // (the grid is a multiple of eight: workgroup b runs on XCD b % 8)
    const int wv = ((int)blockIdx.x & 7) * part + ((int)blockIdx.x >> 3);
    if (wv >= nunits) return;
    // units: strip fastest, then chunk pair (MP: chunk), then member (MP: pair of members; an odd last member pairs with itself)
    const int per_member = nstrip * (MP ? 2 * P : P);
    const int mu = wv / per_member;
    const int m = MP ? 2 * mu : mu, mb = MP ? (m + 1 < g.members ? m + 1 : m) : m;
    const bool act_a = !st || st[m].active, act_b = !st || st[mb].active;
    if (!act_a && !act_b) return;                         // (MP: a stopped member beside a running one is iterated on: its result is its snapshot's)
    const int rem = wv - mu * per_member;
    const int s = rem % nstrip, c = rem / nstrip;
    const int lo = s == 0 ? 1 : T;
    const int x0 = s * W + 1 - lo;                        // column of lane 0
    const int gx = x0 + lane;
    const bool col_int = gx >= 1 && gx <= g.nx - 2;
    const unsigned cmask = ~(unsigned)(((gx - 1) | (g.nx - 2 - gx)) >> 31);   // all ones where the column is interior (see `pick` below)
    const int nvalid = (g.nx - 2 - s * W) < W ? (g.nx - 2 - s * W) : W;
    const size_t mo = (size_t)m * g.plane, ks = g.kstride ? g.kstride : (size_t)g.members * g.plane;
    // (generating planes that all members share are read from member 0's copy: Grid.base_shared)
    // (MP: one descriptor spans both members' planes -- the second half's lanes add dmo4 bytes -- and AP's descriptor spans up to CQ)
    const int dmo4 = (int)((size_t)(mb - m) * g.plane * 4);
    const int dcq4 = MP ? (int)(ks * 4) : 0;
    const rsrc_t pr = zrsrc(psi_in + mo, g, 16 + dmo4), fr = zrsrc(f + mo, g, 16 + dmo4);
    const rsrc_t apr = zrsrc(base + ((g.base_shared & 1) ? 0 : mo), g, 16 + dcq4), cqr = zrsrc(base + ks + ((g.base_shared & 2) ? 0 : mo), g, 16);
    // (B9) b, the fifth generating plane: each member's own unless all share it
    const int dbx4 = (g.base_shared & 16) ? 0 : dmo4;
    const rsrc_t bxr = zrsrc(base + (B9 ? 4 * ks + ((g.base_shared & 16) ? 0 : mo) : 0), g, 16 + dbx4);
    const rsrc_t outr = zrsrc(psi_out + mo, g, dmo4);
    const rsrc_t resr = zrsrc((WR ? res : psi_out) + mo, g, dmo4), snapr = zrsrc((WR && snap ? snap : psi_out) + mo, g, dmo4);
    const int ya = 1 + c * Lc, yb = MP ? ya : 1 + (c + P) * Lc;
    const int yend_a = ya + Lc < g.ny - 1 ? ya + Lc : g.ny - 1, yend_b = yb + Lc < g.ny - 1 ? yb + Lc : g.ny - 1;
    const int pitch4 = g.pitch * 4;

    typedef __attribute__((address_space(3))) void* lds_t;
    // this lane's part of a fetch: row (lane >> 4) & 1 of the group, of chunk lane >> 5, 16 bytes from column x0 + 4 (lane & 15)
    const int fy = (lane >= 32 ? yb : ya) + ((lane >> 4) & 1), fcol4 = (x0 + 4 * (lane & 15)) * 4;
    const int fhalf = lane >= 32 ? dmo4 : 0, fac4 = lane >= 32 ? dcq4 : 0;   // MP: the second half's member / CQ behind AP
    const int fbx = lane >= 32 ? dbx4 : 0;
    auto fetch = [&](int r0, int gs) {
        int ry = fy + r0;
        ry = __builtin_elementwise_min(__builtin_elementwise_max(ry, 0), g.ny - 1);   // (v_max / v_min: a select costs five times as much)
        const int o = ry * pitch4 + fcol4;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(pr, (lds_t)&ring[gs][0][0][0][0], 16, o + fhalf, 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(fr, (lds_t)&ring[gs][1][0][0][0], 16, o + fhalf, 0, 0, 0);
        if constexpr (B9) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(apr, (lds_t)&ring[gs][2][0][0][0], 16, o + fac4, 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(bxr, (lds_t)&ring[gs][3][0][0][0], 16, o + fbx, 0, 0, 0);
        } else if constexpr (MP) __builtin_amdgcn_raw_ptr_buffer_load_lds(apr, (lds_t)&ring[gs][2][0][0][0], 16, o + fac4, 0, 0, 0);
        else {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(apr, (lds_t)&ring[gs][2][0][0][0], 16, o, 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(cqr, (lds_t)&ring[gs][NF - 1][0][0][0], 16, o, 0, 0, 0);
        }
    };
    // One tick's row of the four planes from LDS: issued a tick ahead, into `nx`, and waited for at that tick's end, so that the
    // LDS latency hides behind a tick of arithmetic.  Inline assembly: the compiler cannot tell the slot an LDS-DMA fetch
    // fills from the slot read here and would drain vmcnt(0) in front of every read of this array that it can see (measured:
    // 15.7 -> 22.6 us).  [chunk][row][64]: row r of chunk a at offset r (x 64 dwords), of chunk b at 2 + r; plane k at 4 k.
    struct Fac { f2 f, a, c; };                           // (MP: a = {AP, CQ} of the row, c unused -- B9: the two members' b)
    auto lds_issue = [&](f2& np, Fac& nx, int gs, auto sub_tag) {
        constexpr int r = decltype(sub_tag)::value;
        const unsigned ad = (unsigned)(size_t)(lds_t)&ring[gs][0][0][0][lane];
        if constexpr (MP && !B9)
            asm volatile("ds_read2st64_b32 %0, %3 offset0:%4 offset1:%5\n\t"
                         "ds_read2st64_b32 %1, %3 offset0:%6 offset1:%7\n\t"
                         "ds_read2st64_b32 %2, %3 offset0:%8 offset1:%9"
                         : "=&v"(np), "=&v"(nx.f), "=&v"(nx.a)
                         : "v"(ad), "n"(r), "n"(r + 2), "n"(r + 4), "n"(r + 6), "n"(r + 8), "n"(r + 10) : "memory");
        else
        asm volatile("ds_read2st64_b32 %0, %4 offset0:%5 offset1:%6\n\t"
                     "ds_read2st64_b32 %1, %4 offset0:%7 offset1:%8\n\t"
                     "ds_read2st64_b32 %2, %4 offset0:%9 offset1:%10\n\t"
                     "ds_read2st64_b32 %3, %4 offset0:%11 offset1:%12"
                     : "=&v"(np), "=&v"(nx.f), "=&v"(nx.a), "=&v"(nx.c)
                     : "v"(ad), "n"(r), "n"(r + 2), "n"(r + 4), "n"(r + 6), "n"(r + 8), "n"(r + 10), "n"(r + 12), "n"(r + 14) : "memory");
    };
    // (the pending registers are INPUTS of the waiting statement: that keeps them reserved until the data is there -- with no
    // operands the compiler reused the registers of a tick's unused last read as temporaries while the read was in flight -- and
    // asks for no copy -- as tied in/out operands they were copied, still pending, to wherever the "results" were wanted.  Nothing may
    // be scheduled across the wait; tests/test_sweepz_isa.py checks what came out.)
    auto lds_wait = [&](const f2& np, const Fac& nx) {
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (MP && !B9) asm volatile("s_waitcnt lgkmcnt(0)" : : "v"(np), "v"(nx.f), "v"(nx.a) : "memory");
        else asm volatile("s_waitcnt lgkmcnt(0)" : : "v"(np), "v"(nx.f), "v"(nx.a), "v"(nx.c) : "memory");
        __builtin_amdgcn_sched_barrier(0);
    };

    // stores: lane l < 4 GC carries columns 4 (l % GC) .. + 3 of row (l / GC) & 1 of chunk l / (2 GC) of a group of two result rows
    constexpr int GC = W / 4;
    const int sch = lane >= 2 * GC ? 1 : 0, srow = (lane / GC) & 1, sgc = lane % GC;
    const bool s_lane = lane < 4 * GC;
    const bool s_full = s_lane && 4 * sgc + 4 <= nvalid;
    const int s_part = s_lane && !s_full && 4 * sgc < nvalid ? nvalid - 4 * sgc : 0;
    const int scol4 = (s * W + 1 + 4 * sgc) * 4;
    const int sy0 = sch ? yb : ya, sy1 = sch ? yend_b : yend_a;
    const int o_w = ((lane - lo) & 63);
    const int s_drop = (s_full && (!WR || (sch ? act_b : act_a))) ? 0 : (int)0x80000000;   // bit 31: this lane never stores a full four

    // Level T of the chunk's rows 0 .. Lc - 1 needs level t of rows -(T - t) .. Lc - 1 + (T - t): the first row to take is -T (its
    // CQ is row -T + 1's coe8), the last tick is the one of row Lc - 1 + T; whole groups of two rows (the second sends the results)
    const int j_first = -T;
    const int n_ticks = (Lc + 2 * T + 1) & ~1;

    auto walk = [&](auto exact_tag) -> bool {
        constexpr bool EXACT = decltype(exact_tag)::value;
        constexpr bool NINE = EXACT || B9;                // the exact body keeps all nine weights of a row (b = 0 unless B9); else the star's AP, CQ, coe5
        constexpr int NW9 = NINE ? 9 : 3;
        // w[k][slot]: NINE: coe1 .. coe9 (w[4] = coe5); else w[0] = AP, w[1] = CQ, w[2] = coe5.  y1: the optimistic body's
        // reciprocal, -0 where the point never updates; the exact body keeps 1 / 0 ("updates" / "idle") there instead.
        using WT = std::conditional_t<MP, float, f2>;    // a row's weights: per chunk, or one for both members
        using WW = std::conditional_t<B9, f2, WT>;        // ... except coe5 and y1, a pair where the members have their own BX, BY
        const WT zeroW = WT{};
        const WW zeroWW = WW{};
        auto ww = [](auto v) -> WW { if constexpr (std::is_same_v<decltype(v), WW>) return v; else return WW{v, v}; };
        WW w[NW9][S];
        WT y1[S];
        f2 ff[S];
        f2 hist[T][2];
        WT cq_prev = zeroW;                               // the row before the newest: its CQ and BY make the new row's weights
        WW by_prev = zeroWW;
        f2 b_prev = f2{0.f, 0.f};                         // (B9) b of the row before the newest
        const float yd_b = quotient_y1(g.two_pq4);
        const f2 y1_b = f2{yd_b, yd_b}, nd_b = f2{-g.two_pq4, -g.two_pq4};   // (B9) BX, BY = (b + b') / (2 PQ4)
#pragma unroll
        for (int k = 0; k < S; ++k) {
            y1[k] = zeroW; ff[k] = f2{0.f, 0.f};
#pragma unroll
            for (int q = 0; q < NW9; ++q) w[q][k] = zeroWW;
        }
#pragma unroll
        for (int l = 0; l < T; ++l) hist[l][0] = hist[l][1] = f2{0.f, 0.f};
        bool bad = false;
        int cs = 0;                                       // ring slot of the group the current tick's row is in
#pragma unroll
        for (int d = 0; d < DG; ++d) fetch(j_first + 2 * d, d);
        // The rows in registers rotate through fixed places -- psi of ticks u, u - 1, u - 2 in psr[u & 3], psr[(u + 3) & 3],
        // psr[(u + 2) & 3], the next tick's into psr[(u + 1) & 3]; f, AP, CQ of tick u in fac[u & 1] -- so that no tick moves a value
        // (a turn of the unrolled loop is a multiple of four ticks).  The rows "before the first": zeros (they only reach values
        // that are never stored).
        f2 psr[4] = {f2{0.f, 0.f}, f2{0.f, 0.f}, f2{0.f, 0.f}, f2{0.f, 0.f}};
        Fac fac[2];
        __builtin_amdgcn_s_waitcnt((((DG - 1) * NF) & 0xf) | (0x7 << 4) | (0xf << 8));   // the first group has landed
        lds_issue(psr[0], fac[0], 0, std::integral_constant<int, 0>{});
        lds_wait(psr[0], fac[0]);
        int j0 = 0;
        auto tick = [&](auto uc, auto guard_tag) {
                constexpr int u = decltype(uc)::value;
                constexpr int sub = u & 1;
                if constexpr (decltype(guard_tag)::value) { if (j0 + u >= n_ticks) return; }   // (the last, partial turn ends where the walk does)
                const int jr = j_first + j0 + u;
                const int rowa = ya + jr, rowb = yb + jr;
                constexpr int slot = u % S, q = u & 1;
                const f2 p0 = psr[u & 3], p1 = psr[(u + 3) & 3], p2 = psr[(u + 2) & 3];
                const f2 fa = fac[u & 1].f;
                WT ap, cq;
                if constexpr (MP) { ap = fac[u & 1].a.x; cq = fac[u & 1].a.y; } else { ap = fac[u & 1].a; cq = fac[u & 1].c; }
                f2& np = psr[(u + 1) & 3];                // where the next tick's row goes
                Fac& nx = fac[(u + 1) & 1];
                WW bx = zeroWW, by = zeroWW;              // (b = 0 unless B9: what the nine weights are made of besides AP, CQ)
                if constexpr (B9) {
                    // elliptic_tools.f90:41-44: BX = (b(i,j) + b(i,j-1)) / (2 PQ4), BY = (b(i-1,j) + b(i,j)) / (2 PQ4) -- IEEE divisions, as k_base_planes makes them
                    const f2 bc = fac[u & 1].c, sx = bc + b_prev, sy = zshr2(bc) + bc;
                    if constexpr (EXACT) {
                        bx = f2{sx.x / g.two_pq4, sx.y / g.two_pq4};
                        by = f2{sy.x / g.two_pq4, sy.y / g.two_pq4};
                    } else {
                        // (the hoisted reciprocal again, 10 instead of 44 instructions per row: exact for the sums the host let
                        // through -- k_z_extremes: +0, or dividend and quotient within [2^-100, 2^100], 2 PQ4 itself within [2^-60, 2^60])
                        bx = z_q2(sx, nd_b, y1_b);
                        by = z_q2(sy, nd_b, y1_b);
                    }
                    b_prev = bc;
                }
                if constexpr (sub == 0) {
                    // a group starts: the slot of the group before it is free (its last row was read two ticks ago) and takes the
                    // group DG ahead; the next tick's row is this group's second, in LDS since this group's first was read
                    fetch(jr + 2 * DG, cs >= 1 ? cs - 1 : NG - 1);
                    lds_issue(np, nx, cs, std::integral_constant<int, 1>{});
                } else {
                    // the next tick's row is the first of the next group: that group must have landed -- at most the DG - 1 younger
                    // ones may be in flight (loads retire in order among themselves; stores only make the count larger)
                    __builtin_amdgcn_s_waitcnt((((DG - 1) * NF) & 0xf) | (0x7 << 4) | (0xf << 8));
                    cs = cs + 1 >= NG ? 0 : cs + 1;
                    lds_issue(np, nx, cs, std::integral_constant<int, 0>{});
                }
                // (between the issue and the wait nothing may touch nx's registers -- the hardware does not interlock them.  The
                // optimistic body keeps them apart for a whole tick; tests/test_sweepz_isa.py reads the compiled code to see that no
                // instruction got in between.  The exact body, whose register pressure invites copies, waits at once.)
                if constexpr (EXACT) lds_wait(np, nx);
                if constexpr (!EXACT)
                    bad = bad || __builtin_bit_cast(unsigned, p0.x) == 0x80000000u || __builtin_bit_cast(unsigned, p0.y) == 0x80000000u;
                f2 up = p0;
                f2 rlast = f2{0.f, 0.f};                  // (WR) the residual of the last level's row
                z_for<T>([&](auto tc) {
                    constexpr int t = decltype(tc)::value + 1;
                    constexpr int sl = ((u - t) % S + S) % S;
                    constexpr int slm = ((u - t - 1) % S + S) % S;
                    f2 oc, om;
                    if constexpr (t == 1) { oc = p1; om = p2; }
                    else {
                        oc = hist[t - 1][q ^ 1];
                        om = hist[t - 1][q];
                        hist[t - 1][q] = up;
                    }
                    if constexpr (!NINE) {
                        // (the four terms of the reference's sum that this star leaves out are (+-0) x a diagonal neighbour: zero
                        // unless that neighbour is not finite.  A non-finite value stays one at its own point from level to level:
                        // it is enough to look at what enters the LAST level, in every lane, and at what is stored)
                        if constexpr (t == T)
                            bad = bad || (__builtin_bit_cast(unsigned, up.x) & 0x7f800000u) == 0x7f800000u ||
                                  (__builtin_bit_cast(unsigned, up.y) & 0x7f800000u) == 0x7f800000u;
                        // the star of elliptic_tools.f90:77-85 in its order (terms 2, 4, 5, 6, 8); coe4 * psi(i-1) = (AP * psi)(i-1)
                        const f2 wa = b2(w[0][sl]);
                        const f2 Pm = wa * oc;
                        f2 Rv = b2(w[1][sl]) * up;
                        Rv = f2{Rv.x + zshr1(Pm.x), Rv.y + zshr1(Pm.y)};
                        Rv = Rv + b2(w[2][sl]) * oc;
                        Rv = Rv + f2{wa.x * zshl1(oc.x), wa.y * zshl1(oc.y)};
                        Rv = Rv + b2(w[1][slm]) * om;
                        Rv = Rv - ff[sl];                 // :190
                        if constexpr (WR && t == T) rlast = Rv;
                        up = oc + z_q2(Rv, b2(w[2][sl]), b2(y1[sl]));   // :238 (alpha = 1; -d = coe5)
                    } else {
                        // elliptic_tools.f90:77-85 left to right, :190, :238
                        const f2 w0 = b2(w[0][sl]), w2 = b2(w[2][sl]), w3 = b2(w[3][sl]), w4 = b2(w[4][sl]), w5_ = b2(w[5][sl]),
                                 w6 = b2(w[6][sl]), w8 = b2(w[8][sl]), yy = b2(y1[sl]);
                        f2 Rv = f2{w0.x * zshr1(up.x), w0.y * zshr1(up.y)};
                        Rv = Rv + b2(w[1][sl]) * up;
                        Rv = Rv + f2{w2.x * zshl1(up.x), w2.y * zshl1(up.y)};
                        Rv = Rv + f2{w3.x * zshr1(oc.x), w3.y * zshr1(oc.y)};
                        Rv = Rv + w4 * oc;
                        Rv = Rv + f2{w5_.x * zshl1(oc.x), w5_.y * zshl1(oc.y)};
                        Rv = Rv + f2{w6.x * zshr1(om.x), w6.y * zshr1(om.y)};
                        Rv = Rv + b2(w[7][sl]) * om;
                        Rv = Rv + f2{w8.x * zshl1(om.x), w8.y * zshl1(om.y)};
                        Rv = Rv - ff[sl];
                        if constexpr (WR && t == T) rlast = Rv;
                        if constexpr (!EXACT) up = oc + z_q2(Rv, w4, yy);
                        else {
                            const f2 upd = f2{oc.x + Rv.x / (-w4.x), oc.y + Rv.y / (-w4.y)};
                            up = f2{yy.x != 0.0f ? upd.x : oc.x, yy.y != 0.0f ? upd.y : oc.y};
                        }
                    }
                    (void)slm;
                });
                // ---- level T of row jr - T: into the group of two result rows; the second sends the group to memory
                obuf[0][sub][o_w] = up.x;
                obuf[1][sub][o_w] = up.y;
                if constexpr (WR) { robuf[0][sub][o_w] = rlast.x; robuf[1][sub][o_w] = rlast.y; }
                if constexpr (sub == 1) {
                    const u4 v = *(const u4*)&obuf[sch][srow][4 * sgc];
                    const int r_ = (sch ? rowb : rowa) - T - 1 + srow;
                    // (WR: nothing of a stopped member is written, as k_sweep_q's residual-writing launch skips it)
                    const bool okr = r_ >= sy0 && r_ < sy1 && (!WR || (sch ? act_b : act_a));
                    const int off = r_ * pitch4 + scol4 + (sch ? dmo4 : 0);
                    // a lane with nothing to store aims out of range (bit 31): its row outside the chunk -- the sign of
                    // (r_ - sy0) | (sy1 - 1 - r_) -- or its four columns not all its strip's, or (WR) its member stopped (s_drop).
                    // Arithmetic, not a select: v_cndmask_b32 costs 23 cycles on this chip against 3 for an add (profiles/r01 ubench)
                    const int off_full = off | ((((r_ - sy0) | (sy1 - 1 - r_)) | s_drop) & (int)0x80000000);
                    __builtin_amdgcn_raw_buffer_store_b128(v, outr, off_full, 0, 0);
                    if (nvalid & 3) {                     // (the last strip only) the columns of a partial four, one by one
#pragma unroll
                        for (int k = 0; k < 3; ++k)
                            __builtin_amdgcn_raw_buffer_store_b32(v[k], outr, (okr && k < s_part) ? off + 4 * k : (int)0x80000000, 0, 0);
                    }
                    if constexpr (WR) {
                        const u4 vr = *(const u4*)&robuf[sch][srow][4 * sgc];
                        __builtin_amdgcn_raw_buffer_store_b128(vr, resr, off_full, 0, 0);
                        if (snap) __builtin_amdgcn_raw_buffer_store_b128(v, snapr, off_full, 0, 0);
                        if (nvalid & 3) {
#pragma unroll
                            for (int k = 0; k < 3; ++k) {
                                __builtin_amdgcn_raw_buffer_store_b32(vr[k], resr, (okr && k < s_part) ? off + 4 * k : (int)0x80000000, 0, 0);
                                if (snap) __builtin_amdgcn_raw_buffer_store_b32(v[k], snapr, (okr && k < s_part) ? off + 4 * k : (int)0x80000000, 0, 0);
                            }
                        }
                    }
                    if constexpr (!EXACT) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) bad = bad || (v[k] & 0x7f800000u) == 0x7f800000u;
                    }
                }
                // ---- the new row's weights take a free slot
                const bool ia = col_int && rowa >= 1 && rowa <= g.ny - 2, ib = col_int && rowb >= 1 && rowb <= g.ny - 2;
                // y1 = interior ? q : -0 without a select (v_cndmask_b32 costs 23 cycles on this chip, an and / or 3: profiles/r01 ubench):
                // (q & m) | z with m = all ones, z = 0 where the point is interior, else m = 0, z = the bits of -0.  The masks come from the
                // signs of (x - lo) | (hi - x) and pass an empty asm statement, or the compiler turns them back into selects.
                auto pick = [&](float qv, int row) -> float {
                    int m = ~(((row - 1) | (g.ny - 2 - row)) >> 31) & (int)cmask;   // all ones: row and column interior
                    asm volatile("" : "+v"(m));
                    return __builtin_bit_cast(float, (__builtin_bit_cast(int, qv) & m) | (~m & (int)0x80000000));
                };
                const WT am = zshrW(ap);
                WT c5;
                if constexpr (!NINE) {
                    c5 = -(am + ap + cq_prev + cq);       // elliptic_tools.f90:50, left to right
                    w[0][slot] = ww(ap);
                    w[1][slot] = ww(cq);
                    w[2][slot] = ww(c5);
                } else {
                    // elliptic_tools.f90:46-54 with b = 0, every term kept so that the zeros' signs are the reference's (as exact_tile, xee_sweepq.hip)
                    const WW bxm = zshrW(bx), bym = by_prev;
                    const WT cm = cq_prev;
                    c5 = -(am + ap + cm + cq);
                    w[0][slot] = -(bxm + by);
                    w[1][slot] = ww(cq) + (bx - bxm);
                    w[2][slot] = bx + by;
                    w[3][slot] = ww(am) - (by - bym);
                    w[4][slot] = ww(c5);
                    w[5][slot] = ww(ap) + (by - bym);
                    w[6][slot] = bxm + bym;
                    w[7][slot] = ww(cm) - (bx - bxm);
                    w[8][slot] = -(bx + bym);
                }
                cq_prev = cq; by_prev = by;
                if constexpr (MP) {                       // (the two halves are the same points of two members: ia = ib)
                    if constexpr (EXACT) y1[slot] = ia ? 1.0f : 0.0f;
                    else y1[slot] = pick(quotient_y1(-c5), rowa);
                    (void)ib;
                } else {
                    if constexpr (EXACT) y1[slot] = f2{ia ? 1.0f : 0.0f, ib ? 1.0f : 0.0f};
                    else y1[slot] = f2{pick(quotient_y1(-c5.x), rowa), pick(quotient_y1(-c5.y), rowb)};
                }
                ff[slot] = fa;
                if constexpr (!EXACT) lds_wait(np, nx);   // the next tick's row has arrived (LDS returns in order)
        };
        // whole turns of the unrolled loop run without a test per tick (a test makes every tick a basic block of its own, and the
        // values that live across ticks were shuffled at each boundary: 19 moves per tick), the last, partial turn with one
        for (; j0 + U <= n_ticks; j0 += U) z_for<U>([&](auto uc) { tick(uc, std::false_type{}); });
        if (j0 < n_ticks) z_for<U>([&](auto uc) { tick(uc, std::true_type{}); });
        __builtin_amdgcn_s_waitcnt((0x7 << 4) | (0xf << 8));   // vmcnt(0): the fetches past the end have landed before LDS is used again
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
// NINE: and {min, max} over the NON-ZERO |b + b'| that make BX and BY at the points the interior's weights take them from
// (elliptic_tools.f90:41-44), and whether one of those sums is -0 -- k_sweep_z<nine> divides them by 2 PQ4 itself, by the hoisted
// reciprocal, which turns a -0 into +0 and is not exact where the quotient leaves the normal range.
template <bool NINE>
__global__ __launch_bounds__(256) void k_z_extremes(Grid g, const float* __restrict__ base, const float* __restrict__ f, unsigned* __restrict__ zstat)
{
    const int m = blockIdx.y;
    const size_t mo = (size_t)m * g.plane, ks = g.kstride ? g.kstride : (size_t)g.members * g.plane;
    const float* __restrict__ ap = base + mo;
    const float* __restrict__ cq = base + ks + mo;
    const float* __restrict__ fm = f + mo;
    const float* __restrict__ bm = base + 4 * ks + ((g.base_shared & 16) ? 0 : mo);
    unsigned dmin = 0xffffffffu, dmax = 0u, fmin = 0xffffffffu, smin = 0xffffffffu, smax = 0u, negz = 0u;
    for (int j = 1 + (int)blockIdx.x; j <= g.ny - 2; j += (int)gridDim.x)
        for (int i = 1 + (int)threadIdx.x; i <= g.nx - 2; i += 256) {
            const size_t o = (size_t)j * g.pitch + i;
            const float c5 = -(ap[o - 1] + ap[o] + cq[o - g.pitch] + cq[o]);   // elliptic_tools.f90:50 (b = 0)
            const unsigned d = __builtin_bit_cast(unsigned, c5) & 0x7fffffffu, a = __builtin_bit_cast(unsigned, fm[o]) & 0x7fffffffu;
            dmin = d < dmin ? d : dmin; dmax = d > dmax ? d : dmax; fmin = a < fmin ? a : fmin;
            if constexpr (NINE) {
                // BXp, BXm, BYp, BYm of this point: the sums behind BX(i,j), BX(i-1,j), BY(i,j), BY(i,j-1)
                const float b00 = bm[o], bw = bm[o - 1], bs = bm[o - g.pitch], bsw = bm[o - g.pitch - 1];
                const float sums[4] = {b00 + bs, bw + bsw, bw + b00, bsw + bs};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const unsigned u = __builtin_bit_cast(unsigned, sums[k]), v = u & 0x7fffffffu;
                    negz |= (u == 0x80000000u) ? 1u : 0u;
                    if (v) { smin = v < smin ? v : smin; smax = v > smax ? v : smax; }
                }
            }
        }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const unsigned a = __shfl_xor(dmin, o), b = __shfl_xor(dmax, o), c = __shfl_xor(fmin, o);
        dmin = a < dmin ? a : dmin; dmax = b > dmax ? b : dmax; fmin = c < fmin ? c : fmin;
        if constexpr (NINE) {
            const unsigned p = __shfl_xor(smin, o), q = __shfl_xor(smax, o);
            smin = p < smin ? p : smin; smax = q > smax ? q : smax; negz |= __shfl_xor(negz, o);
        }
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMin(&zstat[6 * m + 0], dmin);
        atomicMax(&zstat[6 * m + 1], dmax);
        atomicMin(&zstat[6 * m + 2], fmin);
        if constexpr (NINE) {
            atomicMin(&zstat[6 * m + 3], smin);
            atomicMax(&zstat[6 * m + 4], smax);
            if (negz) atomicOr(&zstat[6 * m + 5], 1u);
        }
    }
}

__global__ void k_z_extremes_init(unsigned* __restrict__ zstat, int members)
{
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m < members) { zstat[6 * m + 0] = 0xffffffffu; zstat[6 * m + 1] = 0u; zstat[6 * m + 2] = 0xffffffffu;
                       zstat[6 * m + 3] = 0xffffffffu; zstat[6 * m + 4] = 0u; zstat[6 * m + 5] = 0u; }
}

void launch_z_extremes(const Grid& g, const float* base, const float* f, bool barotropic, unsigned* zstat, hipStream_t s)
{
    hipLaunchKernelGGL(k_z_extremes_init, dim3((g.members + 63) / 64), dim3(64), 0, s, zstat, g.members);
    const int by = g.ny - 2 < 64 ? (g.ny - 2 < 1 ? 1 : g.ny - 2) : 64;
    if (barotropic) hipLaunchKernelGGL(k_z_extremes<false>, dim3(by, g.members), dim3(256), 0, s, g, base, f, zstat);
    else hipLaunchKernelGGL(k_z_extremes<true>, dim3(by, g.members), dim3(256), 0, s, g, base, f, zstat);
}

// the host's reading of k_z_extremes for one member (see the header of this file)
bool z_member_ok(const unsigned* zstat3, bool barotropic, float two_pq4)
{
    const float dmin = __builtin_bit_cast(float, zstat3[0]), dmax = __builtin_bit_cast(float, zstat3[1]), fmin = __builtin_bit_cast(float, zstat3[2]);
    if (!(dmin >= 0x1p-120f) || !(dmax <= 0x1p120f)) return false;       // (false for a NaN too)
    const float a_lo = dmax * 0x1p-120f > 0x1p-100f ? dmax * 0x1p-120f : 0x1p-100f;
    if (!(fmin * 0x1p-26f >= a_lo)) return false;
    if (barotropic) return true;
    // the nine-weight walk's own BX, BY = (b + b') / (2 PQ4): divisor, dividends and quotients well inside the normal range, no -0
    if (!(two_pq4 >= 0x1p-60f) || !(two_pq4 <= 0x1p60f) || zstat3[5]) return false;
    if (zstat3[4] == 0u) return true;                                     // every sum is +0
    const float smin = __builtin_bit_cast(float, zstat3[3]), smax = __builtin_bit_cast(float, zstat3[4]);
    // (dividend and quotient at least 2^-100: the remainders a - d q of the two correction steps are exact, nothing is subnormal)
    const float s_lo = two_pq4 > 1.0f ? two_pq4 * 0x1p-100f : 0x1p-100f, s_hi = two_pq4 < 1.0f ? two_pq4 * 0x1p100f : 0x1p100f;
    return smin >= s_lo && smax <= s_hi;
}


int sweep_z_T(bool barotropic) { return barotropic ? 6 : 4; }

// Chunk pairs per strip for a launch over `members` cases that shares the device with `streams` - 1 launches like it.
// The device holds 2048 waves of this kernel (eight per CU); a launch of n waves takes ceil(n / places) rounds, and a
// chunk of Lc rows costs Lc + 2 T ticks: the P in 1 .. (ny - 2) / 64 with the largest product of the rounds' filling
// and Lc / (Lc + 2 T).  Returns 0 where that product is below `min_eff_percent`: k_sweep_q's tiles use the device better
// there (the caller's thresholds are measured end to end, profiles/r03/auto_vs_tiled.txt: 32 cases of 1024x512 reach 82, 24 cases 79,
// 16 cases and 4 cases of 2048x1024 73, one case of 4096x2048 75, 8 cases of 1024x512 41; T = 4, the nine-weight operator: 81, 75
// for 32, 16 cases, 71 for 12, 43 for 8).
int sweep_z_pairs(const Grid& g, int members, int streams, int min_eff_percent, int T)
{
    const int W = 64 - 2 * T;
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
        const double eff = (double)n / (double)(rounds * places) * (double)Lc / (double)(Lc + 2 * T);
        if (eff > best_eff) { best_eff = eff; best = P; }
    }
    return best_eff * 100.0 >= (double)min_eff_percent ? best : 0;
}

bool sweep_z_can(int T, bool with_residual, bool barotropic) { return barotropic ? (T == 6 || (with_residual && T == 4)) : T == 4; }

void launch_sweep_z(const Grid& g, const float* psi_in, float* psi_out, const float* base, const float* f, float* res, float* snap,
                    const MemberState* st, int T, bool barotropic, int P, int exact_only, int* diag, SweepDesc* desc, hipStream_t s)
{
    const int W = 64 - 2 * T;
    const int nstrip = (g.nx - 2 + W - 1) / W;
    const int Lc = (g.ny - 2 + 2 * P - 1) / (2 * P);
    // members in pairs where they all share AP and CQ (and the spans of the paired descriptors stay below 2^31 bytes)
    const size_t ks = g.kstride ? g.kstride : (size_t)g.members * g.plane;
    const bool pairs = (g.base_shared & 3) == 3 && g.members >= 2 && (ks + g.plane) * 4 + 64 < ((size_t)1 << 31);
    const long long units = pairs ? (long long)nstrip * 2 * P * ((g.members + 1) / 2) : (long long)nstrip * P * g.members;
    if (desc) { desc->name = !barotropic ? "k_sweep_z<nine, member pairs>" : pairs ? "k_sweep_z<star, member pairs>" : "k_sweep_z<star>"; desc->T = T; desc->tile_w = 64; desc->tile_h = Lc; desc->tiles = units; desc->streamed = true; }
    // (T = 8 was measured too: 17.3 against 17.6 us per iteration at equal P -- what it saves in bytes it pays in halo and run-in,
    // and its 22 strips leave no P that fills the 2048 places; profiles/r03/z_ablation.txt)
    const dim3 grid((unsigned)(8 * ((units + 7) / 8)));
#define XEE_Z_LAUNCH(TT, MPAIR, WRES) hipLaunchKernelGGL((k_sweep_z<TT, MPAIR, WRES>), grid, dim3(64), 0, s, g, psi_in, psi_out, base, f, st, nstrip, P, Lc, (int)units, exact_only, diag, res, snap)
    if (!barotropic) {                                    // (z_decide sends the nine-weight operator here for member pairs only)
        if (!pairs || T != 4) return;
        if (res) hipLaunchKernelGGL((k_sweep_z<4, true, true, true>), grid, dim3(64), 0, s, g, psi_in, psi_out, base, f, st, nstrip, P, Lc, (int)units, exact_only, diag, res, snap);
        else hipLaunchKernelGGL((k_sweep_z<4, true, false, true>), grid, dim3(64), 0, s, g, psi_in, psi_out, base, f, st, nstrip, P, Lc, (int)units, exact_only, diag, res, snap);
    }
    else if (!res) { if (pairs) XEE_Z_LAUNCH(6, true, false); else XEE_Z_LAUNCH(6, false, false); }
    else if (T == 4) { if (pairs) XEE_Z_LAUNCH(4, true, true); else XEE_Z_LAUNCH(4, false, true); }
    else { if (pairs) XEE_Z_LAUNCH(6, true, true); else XEE_Z_LAUNCH(6, false, true); }
#undef XEE_Z_LAUNCH
}

}  // namespace xee
