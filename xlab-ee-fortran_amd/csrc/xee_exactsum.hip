// xee_exactsum.hip -- the residual norm of solve_elliptic (xtt-lib-fortran/elliptic_tools.f90:193-199)
// reproduced BIT FOR BIT in parallel.
//
// The reference accumulates  err_now = err_now + to_dat(i,j)**2.0  sequentially in fp32, i outer /
// j inner; the value decides when the solver stops, so it must be reproduced exactly (a pairwise
// sum differs at the 1e-6 level and flips iteration counts, SURVEY.md section 7 hard part 2).  A
// single lane walking 521,220 values in order costs milliseconds; this file gets the same bits from
// mostly parallel work by using what sequential fp32 addition of NON-NEGATIVE terms really does:
//
//   While the running sum s stays inside one binade [2^e, 2^(e+1)) its bit pattern is base + S with
//   S an integer counting ulps u = 2^(e-23), and  fl(s + t) = (S + rne(t/u)) * u : every term just adds
//   an integer number of ulps, round-to-nearest with ties decided by the parity of S.  The number of
//   ulps a run of terms adds therefore depends on the incoming sum only through (a) the binade and
//   (b) the parity of S.  So:
//     A. per chunk of 32 consecutive terms (in reference order): fp64 sum              (parallel)
//     B. exclusive scan of the chunk sums (per column, then over the column totals) -> predicted
//        running sum, hence predicted binade, at every chunk start                      (parallel)
//     C. per chunk, for its predicted binade: the ulp counts K_even / K_odd it adds for an even /
//        odd incoming S (exact integer arithmetic on the fp32 fields; ties handled)   (parallel)
//     D. per column: compose its chunks' (K_even, K_odd) when they share one binade    (parallel)
//     E. one wave per member walks the columns in reference order.  Composition of the (K_even,
//        K_odd) maps is associative, so an inclusive WAVE SCAN over the next 64 summaries gives all
//        64 run lengths at once; the longest run that is valid for the ACTUAL binade and keeps the
//        sum inside it is applied with one integer add on the bit pattern.  Everything is checked
//        on the real bits, so a wrong prediction can only cost time.  A column that does not fit
//        is walked over its chunks the same way (their summaries were staged in LDS up front for
//        the columns known to be composite-invalid); a chunk that still does not fit -- binade
//        crossing, a term above half the sum, NaN/Inf -- gets its 32 genuine fp32 additions.
//   Typically ~25 of the 16,352 chunks of a 1024x512 plane take the slow path.
#include "xee_device.h"

#include <type_traits>

#pragma clang fp contract(off)

namespace xee {

namespace {

constexpr int SEG = EXACT_SEG;
constexpr int MAXREP = 48;      // columns whose chunk summaries are staged in LDS for the walk
constexpr int MAXT = 64;        // chunks whose 32 terms are staged in LDS as well (the ones known to need genuine adds)
constexpr int MAXSEG = 64;      // segments per column the term map is sized for (2048 rows)

__device__ __forceinline__ unsigned f2u(float x) { return __builtin_bit_cast(unsigned, x); }
__device__ __forceinline__ float u2f(unsigned x) { return __builtin_bit_cast(float, x); }

// exponent of the ulp of a non-negative fp32 bit pattern: ulp = 2^eu; subnormals share E=1's ulp
__device__ __forceinline__ int ulp_exp(unsigned bits)
{
    const int E = (int)((bits >> 23) & 0xffu);
    return (E > 0 ? E : 1) - 150;
}

// chunk (column, segment) of one member: terms R(1+col, 1 + seg*SEG + r)^2, r = 0..n-1
struct ChunkAddr {
    const float* p;
    int n;
};

__device__ __forceinline__ ChunkAddr chunk_addr(const Grid& g, const float* Rm, int col, int seg, int nrows)
{
    ChunkAddr a;
    const int r0 = seg * SEG;
    a.n = min(SEG, nrows - r0);
    a.p = Rm + (size_t)(1 + r0) * g.pitch + (1 + col);
    return a;
}

// exact ulp counts a chunk adds to a sum whose ulp is 2^eu (valid while the sum stays in that binade)
__device__ __forceinline__ ExactSum chunk_summary(const ChunkAddr& a, int pitch, int eu)
{
    unsigned ke = 0, ko = 0;
    bool ok = true;
    // (all of the chunk's terms are requested before the first one is looked at: a load per trip of the loop below made
    // the kernel wait for memory 32 times in a row)
    float xs[SEG];
#pragma unroll
    for (int r = 0; r < SEG; ++r) xs[r] = (r < a.n) ? a.p[(size_t)r * pitch] : 0.0f;
#pragma unroll
    for (int r = 0; r < SEG; ++r) {
        // (no break / continue: the loop must unroll for xs[] to stay in registers; `ok` going false ends the work)
        const float x = xs[r];
        const unsigned tb = f2u(x * x);                      // to_dat(i,j)**2.0, >= +0
        const unsigned Et = (tb >> 23) & 0xffu;
        const unsigned fr = tb & 0x7fffffu;
        const unsigned Mt = Et ? (fr | 0x800000u) : fr;
        const int shift = eu - ((int)(Et ? Et : 1u) - 150);
        if (r < a.n && ok) {
            if (Et == 255u) ok = false;                      // Inf / NaN term: genuine adds only
            else if (Mt != 0u) {                             // (+0 adds nothing)
                if (shift <= 0) ok = false;                  // term >= 2^23 ulps: leaves the binade
                else if (shift < 25) {                       // (>= 25: below half an ulp, absorbed)
                    const unsigned q = Mt >> shift;
                    const unsigned rem = Mt & ((1u << shift) - 1u);
                    const unsigned half = 1u << (shift - 1);
                    const unsigned up_e = (rem > half) || (rem == half && ((ke + q) & 1u));
                    const unsigned up_o = (rem > half) || (rem == half && (((ko + q) & 1u) ^ 1u));
                    ke += q + up_e;
                    ko += q + up_o;
                }
            }
        }
    }
    ExactSum out;
    out.eu = eu;
    out.valid = (ok && ke < (1u << 24) && ko < (1u << 24)) ? 1 : 0;
    out.ke = out.valid ? ke : 0u;
    out.ko = out.valid ? ko : 0u;
    return out;
}

// run `a` followed by run `b`
__device__ __forceinline__ ExactSum compose(const ExactSum& a, const ExactSum& b)
{
    const unsigned ke = a.ke + ((a.ke & 1u) ? b.ko : b.ke);                 // parity after a, incoming even
    const unsigned ko = a.ko + (((a.ko & 1u) ^ 1u) ? b.ko : b.ke);          // parity after a, incoming odd
    const bool ok = a.valid && b.valid && a.eu == b.eu && ke < (1u << 24) && ko < (1u << 24);
    ExactSum o;
    o.eu = a.eu;
    o.valid = ok ? 1 : 0;
    o.ke = ok ? ke : 0u;
    o.ko = ok ? ko : 0u;
    return o;
}

}  // namespace

// A. fp64 sum of each chunk's terms (prediction only; accuracy irrelevant for correctness)
__global__ __launch_bounds__(256) void k_ex_chunk_sums(Grid g, const float* __restrict__ R, int ncol, int nseg,
        int nrows, double* __restrict__ sigma, const MemberState* __restrict__ st)
{
    const int m = blockIdx.z;
    if (st && !st[m].active) return;
    const int col = blockIdx.x * 64 + threadIdx.x;
    const int seg = blockIdx.y * 4 + threadIdx.y;
    if (col >= ncol || seg >= nseg) return;
    const ChunkAddr a = chunk_addr(g, R + (size_t)m * g.plane, col, seg, nrows);
    // (the loads first, all of them in flight together; then the sum in the reference's order)
    float xs[SEG];
#pragma unroll
    for (int r = 0; r < SEG; ++r) xs[r] = (r < a.n) ? a.p[(size_t)r * g.pitch] : 0.0f;
    double acc = 0.0;
#pragma unroll
    for (int r = 0; r < SEG; ++r) {
        const float t = xs[r] * xs[r];
        if (r < a.n) acc += (double)t;
    }
    sigma[(size_t)m * ncol * nseg + (size_t)col * nseg + seg] = acc;
}

// B1. per column: exclusive prefix of its chunk sums (in place) and the column total
__global__ __launch_bounds__(256) void k_ex_col_prefix(int ncol, int nseg, double* __restrict__ sigma,
        double* __restrict__ coltot, const MemberState* __restrict__ st)
{
    const int m = blockIdx.y;
    if (st && !st[m].active) return;
    const int col = blockIdx.x * 256 + threadIdx.x;
    if (col >= ncol) return;
    double* sg = sigma + ((size_t)m * ncol + col) * nseg;
    double run = 0.0;
    for (int k = 0; k < nseg; ++k) { const double v = sg[k]; sg[k] = run; run += v; }
    coltot[(size_t)m * ncol + col] = run;
}

// B2. exclusive scan of the column totals of one member
__global__ __launch_bounds__(1024) void k_ex_scan_cols(int ncol, const double* __restrict__ coltot,
        double* __restrict__ coloff, const MemberState* __restrict__ st)
{
    __shared__ double part[1024];
    const int m = blockIdx.x;
    if (st && !st[m].active) return;
    const double* ct = coltot + (size_t)m * ncol;
    double* co = coloff + (size_t)m * ncol;
    const int per = (ncol + 1023) / 1024;
    const int lo = min(ncol, (int)threadIdx.x * per), hi = min(ncol, lo + per);
    double loc = 0.0;
    for (int k = lo; k < hi; ++k) loc += ct[k];
    part[threadIdx.x] = loc;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {               // Hillis-Steele inclusive scan
        double v = 0.0;
        if ((int)threadIdx.x >= off) v = part[threadIdx.x - off];
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    double run = part[threadIdx.x] - loc;
    for (int k = lo; k < hi; ++k) { co[k] = run; run += ct[k]; }
}

// C. exact ulp counts of each chunk for its predicted binade
__global__ __launch_bounds__(256) void k_ex_chunk_summaries(Grid g, const float* __restrict__ R, int ncol, int nseg,
        int nrows, const double* __restrict__ prefix, const double* __restrict__ coloff,
        ExactSum* __restrict__ chunk, const MemberState* __restrict__ st)
{
    const int m = blockIdx.z;
    if (st && !st[m].active) return;
    const int col = blockIdx.x * 64 + threadIdx.x;
    const int seg = blockIdx.y * 4 + threadIdx.y;
    if (col >= ncol || seg >= nseg) return;
    const size_t c = (size_t)m * ncol * nseg + (size_t)col * nseg + seg;
    const double P = coloff[(size_t)m * ncol + col] + prefix[c];
    ExactSum out;
    if (!(P >= 0.0) || !(P < 3.0e38)) {                      // NaN / Inf / overflowing prediction
        out.ke = 0; out.ko = 0; out.eu = -100000; out.valid = 0;
    } else {
        out = chunk_summary(chunk_addr(g, R + (size_t)m * g.plane, col, seg, nrows), g.pitch, ulp_exp(f2u((float)P)));
    }
    chunk[c] = out;
}

// D. compose the chunks of a column when they are all valid for one and the same binade
__global__ __launch_bounds__(256) void k_ex_col_summaries(int ncol, int nseg, const ExactSum* __restrict__ chunk,
        ExactSum* __restrict__ colsum, const MemberState* __restrict__ st)
{
    const int m = blockIdx.y;
    if (st && !st[m].active) return;
    const int col = blockIdx.x * 256 + threadIdx.x;
    if (col >= ncol) return;
    const ExactSum* cs = chunk + ((size_t)m * ncol + col) * nseg;
    ExactSum out = cs[0];
    for (int k = 1; k < nseg && out.valid; ++k) out = compose(out, cs[k]);
    colsum[(size_t)m * ncol + col] = out;
}

// E. the walk: one wave per member; the state `bits` is wave-uniform
__global__ __launch_bounds__(64) void k_ex_walk(Grid g, const float* __restrict__ R, int ncol, int nseg, int nrows,
        const ExactSum* __restrict__ chunk, const ExactSum* __restrict__ colsum, float* __restrict__ out,
        int* __restrict__ slow_chunks, const MemberState* __restrict__ st, int cols_in_lds)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ int rep_of[MAXREP];       // staged column ids
    __shared__ int nrep_s, nterm_s;
    __shared__ float terms[MAXT][SEG];   // squared residuals of the chunks staged for the slow path
    __shared__ short term_of[MAXREP * MAXSEG];   // staged chunk (rep slot, segment) -> row of `terms`, or -1
    __shared__ short term_src[MAXT];     // ... and back
    const int m = blockIdx.x;
    if (st && !st[m].active) return;
    const int lane = threadIdx.x;
    const float* Rm = R + (size_t)m * g.plane;
    const ExactSum* cols_g = colsum + (size_t)m * ncol;
    ExactSum* cols_l = reinterpret_cast<ExactSum*>(smem);                         // [cols_in_lds ? ncol : 0]
    ExactSum* rep_l = cols_l + (cols_in_lds ? ncol : 0);                          // [MAXREP][nseg]

    // stage the column summaries, and the chunk summaries of the columns that will need a descent
    if (lane == 0) { nrep_s = 0; nterm_s = 0; }
    __syncthreads();
    // (loads are issued in batches of 8 per lane so that their latencies overlap)
    if (cols_in_lds) {
        for (int base = 0; base < ncol; base += 64 * 8) {
            ExactSum tmp[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int c = base + u * 64 + lane; tmp[u] = c < ncol ? cols_g[c] : ExactSum{0u, 0u, 0, 1}; }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int c = base + u * 64 + lane;
                if (c < ncol) {
                    cols_l[c] = tmp[u];
                    if (!tmp[u].valid) { const int slot = atomicAdd(&nrep_s, 1); if (slot < MAXREP) rep_of[slot] = c; }
                }
            }
        }
    }
    __syncthreads();
    const int nrep = cols_in_lds ? min(nrep_s, MAXREP) : 0;
    const int nstage = nrep * nseg;
    for (int base = 0; base < nstage; base += 64 * 8) {
        ExactSum tmp[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = base + u * 64 + lane;
            if (e < nstage) tmp[u] = chunk[((size_t)m * ncol + rep_of[e / nseg]) * nseg + e % nseg];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = base + u * 64 + lane;
            if (e < nstage) {
                rep_l[e] = tmp[u];
                // a chunk without a valid summary will get its genuine additions in the walk: stage its terms too,
                // so that the walk does not wait for global memory at every such chunk
                if (nseg <= MAXSEG) {
                    int slot = -1;
                    if (!tmp[u].valid) { slot = atomicAdd(&nterm_s, 1); if (slot >= MAXT) slot = -1; else term_src[slot] = (short)e; }
                    term_of[e] = (short)slot;
                }
            }
        }
    }
    __syncthreads();
    const int nterm = (nseg <= MAXSEG) ? min(nterm_s, MAXT) : 0;
    for (int base = 0; base < nterm * SEG; base += 64 * 8) {
        float tmp[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int idx = base + u * 64 + lane;
            tmp[u] = 0.0f;
            if (idx < nterm * SEG) {
                const int e = term_src[idx / SEG], r = idx % SEG;
                const ChunkAddr a = chunk_addr(g, Rm, rep_of[e / nseg], e % nseg, nrows);
                if (r < a.n) { const float x = a.p[(size_t)r * g.pitch]; tmp[u] = x * x; }
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int idx = base + u * 64 + lane; if (idx < nterm * SEG) terms[idx / SEG][idx % SEG] = tmp[u]; }
    }
    __syncthreads();
    const ExactSum* cols = cols_in_lds ? cols_l : cols_g;

    unsigned bits = 0u;                                       // err_now = 0
    int slow = 0;

    // lane l holds summary number l of a window; after the inclusive scan it holds the run [0..l].
    // Runs that are valid for the current binade and keep the sum inside it form a prefix (K only
    // grows); the longest is applied.  Returns how many summaries were consumed.
    // The scan runs on DPP moves (row_shr 1, 2, 4, 8 inside each row of 16 lanes, then row_bcast:15 and
    // row_bcast:31 across rows) -- register-to-register, a few cycles each -- instead of ds_bpermute shuffles,
    // which go through the LDS crossbar: this kernel is one wave per member and a chain of ~90 dependent scans,
    // so the scan's latency is its run time.  compose() is associative but not commutative: every step puts the
    // EARLIER run on the left.
    auto scan_take = [&](ExactSum me) -> int {
        // Only summaries counted for the binade the sum is in NOW can be applied, and only as a prefix: find the
        // first one that cannot, then scan the bare ulp counts (two words, three operations per composition)
        // of all 64 lanes -- what lies at or beyond that lane cannot influence the lanes before it.
        const int eu_now = ulp_exp(bits);
        const unsigned long long good = __ballot(me.valid && me.eu == eu_now);
        const int first_bad = (good == ~0ull) ? 64 : __builtin_ctzll(~good);
        if (first_bad == 0) return 0;
        unsigned ke = me.ke, ko = me.ko;
        auto stepd = [&](auto ctrl_tag, auto rmask_tag, bool has) {
            constexpr int CTRL = decltype(ctrl_tag)::value, RMASK = decltype(rmask_tag)::value;
            const unsigned a_ke = (unsigned)__builtin_amdgcn_update_dpp(0, (int)ke, CTRL, RMASK, 0xf, false);
            const unsigned a_ko = (unsigned)__builtin_amdgcn_update_dpp(0, (int)ko, CTRL, RMASK, 0xf, false);
            // compose(a, me): the earlier run a on the left
            const unsigned nke = a_ke + ((a_ke & 1u) ? ko : ke);
            const unsigned nko = a_ko + ((a_ko & 1u) ? ke : ko);
            if (has) { ke = nke; ko = nko; }
        };
        using std::integral_constant;
        const int rl = lane & 15;
        stepd(integral_constant<int, 0x111>{}, integral_constant<int, 0xf>{}, rl >= 1);       // row_shr:1
        stepd(integral_constant<int, 0x112>{}, integral_constant<int, 0xf>{}, rl >= 2);       // row_shr:2
        stepd(integral_constant<int, 0x114>{}, integral_constant<int, 0xf>{}, rl >= 4);       // row_shr:4
        stepd(integral_constant<int, 0x118>{}, integral_constant<int, 0xf>{}, rl >= 8);       // row_shr:8
        stepd(integral_constant<int, 0x142>{}, integral_constant<int, 0xa>{}, (lane & 16) != 0);   // row_bcast:15 -> rows 1, 3
        stepd(integral_constant<int, 0x143>{}, integral_constant<int, 0xc>{}, lane >= 32);         // row_bcast:31 -> rows 2, 3
        const unsigned nb = bits + ((bits & 1u) ? ko : ke);
        const unsigned e0 = bits >> 23, e1 = nb >> 23;
        // K ulps keep the sum in its binade iff the exponent field (0 and 1 merged) is unchanged; the counts are
        // non-decreasing along the lanes, so the runs that fit form a prefix as well
        const bool fits = lane < first_bad && ke < (1u << 24) && ko < (1u << 24) &&
                          (e0 > 1u ? e0 : 1u) == (e1 > 1u ? e1 : 1u) && e1 < 255u;
        const unsigned long long mask = __ballot(fits);
        const int n = (mask == ~0ull) ? 64 : __builtin_ctzll(~mask);
        if (n > 0) bits = (unsigned)__builtin_amdgcn_readlane((int)nb, __builtin_amdgcn_readfirstlane(n - 1));
        return n;
    };
    const ExactSum none = {0u, 0u, 0, 0};

    for (int col = 0; col < ncol;) {
        const int n = scan_take(col + lane < ncol ? cols[col + lane] : none);
        col += n;
        if (n == 64 || col >= ncol) continue;
        // column `col` is the first of the window that does not fit as a whole: the same game over its chunks
        const ExactSum* ch = chunk + ((size_t)m * ncol + col) * nseg;
        int krep = -1;                                        // staged copy?  (one compare per lane instead of a search)
        {
            const unsigned long long hit = __ballot(lane < nrep && rep_of[lane < MAXREP ? lane : 0] == col);
            if (hit) { krep = __builtin_ctzll(hit); ch = rep_l + (size_t)krep * nseg; }
        }
        for (int seg = 0; seg < nseg;) {
            const int eu_now = ulp_exp(bits);
            ExactSum me = none;
            if (seg + lane < nseg) me = ch[seg + lane];
            // A summary stored for another binade cannot be used now.  If that already holds for the
            // FIRST chunk of the window the prediction was off here: recount the window for the real
            // binade.  Otherwise the later chunks simply lie beyond a crossing that is still ahead.
            const bool stale = (seg + lane < nseg) && me.eu != eu_now;
            if (__shfl((int)stale, 0, 64)) {
                if (stale) me = chunk_summary(chunk_addr(g, Rm, col, seg + lane, nrows), g.pitch, eu_now);
            } else if (stale) {
                me = none;
            }
            seg += scan_take(me);
            if (seg >= nseg) break;
            // chunk `seg` does not fit this binade (crossing, a term above half the sum, Inf/NaN):
            // genuine sequential fp32 additions -- lanes fetch, the sum runs in reference order
            const ChunkAddr a = chunk_addr(g, Rm, col, seg, nrows);
            float t = 0.0f;
            const int ts = (krep >= 0 && nseg <= MAXSEG) ? (int)term_of[krep * nseg + seg] : -1;
            if (ts >= 0) { if (lane < SEG) t = terms[ts][lane]; }          // (rows beyond a.n were staged as 0 and are not added)
            else if (lane < a.n) { const float x = a.p[(size_t)lane * g.pitch]; t = x * x; }
            float sacc = u2f(bits);
            const int tb = (int)f2u(t);
#pragma unroll
            for (int r = 0; r < SEG; ++r)                      // v_readlane with a constant lane: no LDS round trip
                if (r < a.n) sacc = sacc + u2f((unsigned)__builtin_amdgcn_readlane(tb, r));
            bits = f2u(sacc);
            ++slow;
            ++seg;
        }
        ++col;
    }
    if (lane == 0) {
        out[m] = u2f(bits);
        if (slow_chunks) slow_chunks[m] = slow;
    }
}

// ---------------------------------------------------------------------------------------------

size_t exact_ws_bytes(const Grid& g)
{
    const int ncol = g.nx - 2, nrows = g.ny - 2;
    const int nseg = (nrows + SEG - 1) / SEG;
    const size_t nchunk = (size_t)ncol * nseg;
    return (size_t)g.members * (nchunk * (sizeof(double) + sizeof(ExactSum)) + (size_t)ncol * (2 * sizeof(double) + sizeof(ExactSum))) + 256;
}

int exact_kernel_setup()
{
    return (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ex_walk),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
}

namespace {
struct ExactWs { double *sigma, *coltot, *coloff; ExactSum *chunk, *colsum; int ncol, nrows, nseg; };
ExactWs exact_ws(const Grid& g, void* workspace)
{
    ExactWs x;
    x.ncol = g.nx - 2; x.nrows = g.ny - 2;
    x.nseg = (x.nrows + SEG - 1) / SEG;
    const size_t nchunk = (size_t)x.ncol * x.nseg;
    char* w = static_cast<char*>(workspace);
    x.sigma = reinterpret_cast<double*>(w);               w += (size_t)g.members * nchunk * sizeof(double);
    x.coltot = reinterpret_cast<double*>(w);              w += (size_t)g.members * x.ncol * sizeof(double);
    x.coloff = reinterpret_cast<double*>(w);              w += (size_t)g.members * x.ncol * sizeof(double);
    x.chunk = reinterpret_cast<ExactSum*>(w);             w += (size_t)g.members * nchunk * sizeof(ExactSum);
    x.colsum = reinterpret_cast<ExactSum*>(w);
    return x;
}
}  // namespace

// the wide, bandwidth-shaped half: chunk sums, scans, chunk and column summaries (about 75 us for 32 cases of 1024x512)
void launch_exact_sumsq_prepare(const Grid& g, const float* R, void* workspace, const MemberState* st, hipStream_t s)
{
    const ExactWs x = exact_ws(g, workspace);
    const dim3 gc((x.ncol + 63) / 64, (x.nseg + 3) / 4, g.members), bc(64, 4, 1);
    const dim3 gcol((x.ncol + 255) / 256, g.members);
    hipLaunchKernelGGL(k_ex_chunk_sums, gc, bc, 0, s, g, R, x.ncol, x.nseg, x.nrows, x.sigma, st);
    hipLaunchKernelGGL(k_ex_col_prefix, gcol, dim3(256), 0, s, x.ncol, x.nseg, x.sigma, x.coltot, st);
    hipLaunchKernelGGL(k_ex_scan_cols, dim3(g.members), dim3(1024), 0, s, x.ncol, x.coltot, x.coloff, st);
    hipLaunchKernelGGL(k_ex_chunk_summaries, gc, bc, 0, s, g, R, x.ncol, x.nseg, x.nrows, x.sigma, x.coloff, x.chunk, st);
    hipLaunchKernelGGL(k_ex_col_summaries, gcol, dim3(256), 0, s, x.ncol, x.nseg, x.chunk, x.colsum, st);
}

// the narrow, serial half: one wave per member walks the summaries (100-160 us, but only `members` waves)
void launch_exact_sumsq_walk(const Grid& g, const float* R, void* workspace, float* out_sums, int* slow_chunks,
                             const MemberState* st, hipStream_t s)
{
    const ExactWs x = exact_ws(g, workspace);
    // LDS: all column summaries + the chunk summaries of up to MAXREP columns, when that fits
    size_t lds = ((size_t)x.ncol + (size_t)MAXREP * x.nseg) * sizeof(ExactSum);
    int cols_in_lds = 1;
    if (lds > 128 * 1024) { lds = 0; cols_in_lds = 0; }         // (+ ~15 KiB of static LDS in the walk)
    hipLaunchKernelGGL(k_ex_walk, dim3(g.members), dim3(64), lds, s, g, R, x.ncol, x.nseg, x.nrows, x.chunk, x.colsum, out_sums,
                       slow_chunks, st, cols_in_lds);
}

void launch_exact_sumsq(const Grid& g, const float* R, void* workspace, float* out_sums, int* slow_chunks,
                        const MemberState* st, hipStream_t s)
{
    launch_exact_sumsq_prepare(g, R, workspace, st, s);
    launch_exact_sumsq_walk(g, R, workspace, out_sums, slow_chunks, st, s);
}

// =============================================================================================
// The same idea for terms of EITHER SIGN (the EE integrals: sum Q, sum Q*eta of
// old-diagnose/diagnose.f90:1029-1092 are sequential fp32 sums, i outer / j inner, over (nr-1) x (nz-1)
// terms of either sign; a single lane needs 3 ms at 1024x512 and 44 ms at 4096x2048).
//
// Write the running sum as a SIGNED integer V of ulps, |V| in [2^23, 2^24): while it stays in its binade
// and keeps its sign, fl(s + t) = rne_even(V + t/u) * u for a term of either sign, so a chunk still maps
// the parity of the incoming V to the signed ulp count it adds -- (K_even, K_odd) -- but the counts are
// no longer monotone: the summary also carries the extremes (kmin, kmax) over all its prefixes, taken
// over both parities (conservative), and a run may be applied only if V + kmin .. V + kmax stays inside
// [2^23 + 1, 2^24 - 1] in magnitude (at |V| = 2^23 exactly the spacing below is u/2 and the model stops
// holding for a term that reduces |s|).  compose() of the counts is associative as before; the extremes
// compose conservatively, which is all validity needs.  Everything that does not fit -- the start from 0,
// sign changes, binade crossings, cancellation down into lower binades -- is added term by term.
// =============================================================================================
namespace {

struct SSum {
    int ke, ko;        // signed ulps added for an even / odd incoming V
    int kmin, kmax;    // extremes of the prefixes' counts (both parities), including the empty prefix (0)
    int eu;            // log2(ulp) of the binade the counts are valid for
    int valid;
};

constexpr int KLIM = 1 << 24;

__device__ __forceinline__ SSum schunk_summary(const ChunkAddr& a, int pitch, int eu)
{
    int ke = 0, ko = 0, kmin = 0, kmax = 0;
    bool ok = true;
    for (int r = 0; r < a.n; ++r) {
        const unsigned tb = f2u(a.p[(size_t)r * pitch]);
        const unsigned Et = (tb >> 23) & 0xffu;
        const unsigned fr = tb & 0x7fffffu;
        if (Et == 255u) { ok = false; break; }               // Inf / NaN term: genuine adds only
        const unsigned Mt = Et ? (fr | 0x800000u) : fr;
        if (Mt == 0u) continue;                              // +-0
        const int shift = eu - ((int)(Et ? Et : 1u) - 150);
        if (shift <= 0) { ok = false; break; }               // |term| >= 2^23 ulps: leaves the binade
        if (shift >= 25) continue;                           // below half an ulp: absorbed (either sign)
        const unsigned q = Mt >> shift;
        const unsigned rem = Mt & ((1u << shift) - 1u);
        const unsigned half = 1u << (shift - 1);
        // V +- q and its neighbour one further out: above half the farther one, a tie goes to the even one
        const unsigned pe = (unsigned)ke & 1u, po = ((unsigned)ko & 1u) ^ 1u;          // parity of V so far
        const int de = (int)(q + ((rem > half) || (rem == half && ((pe ^ q) & 1u))));
        const int dd = (int)(q + ((rem > half) || (rem == half && ((po ^ q) & 1u))));
        if (tb >> 31) { ke -= de; ko -= dd; } else { ke += de; ko += dd; }
        kmin = min(kmin, min(ke, ko));
        kmax = max(kmax, max(ke, ko));
        if (kmin < -KLIM || kmax > KLIM) { ok = false; break; }
    }
    SSum out;
    out.eu = eu;
    out.valid = ok ? 1 : 0;
    out.ke = ok ? ke : 0; out.ko = ok ? ko : 0; out.kmin = ok ? kmin : 0; out.kmax = ok ? kmax : 0;
    return out;
}

__device__ __forceinline__ SSum scompose(const SSum& a, const SSum& b)
{
    SSum o;
    o.ke = a.ke + ((a.ke & 1) ? b.ko : b.ke);
    o.ko = a.ko + ((a.ko & 1) ? b.ke : b.ko);
    o.kmin = min(a.kmin, min(a.ke, a.ko) + b.kmin);
    o.kmax = max(a.kmax, max(a.ke, a.ko) + b.kmax);
    o.eu = a.eu;
    o.valid = (a.valid && b.valid && a.eu == b.eu && o.kmin >= -KLIM && o.kmax <= KLIM) ? 1 : 0;
    if (!o.valid) { o.ke = 0; o.ko = 0; o.kmin = 0; o.kmax = 0; }
    return o;
}

// chunk (column, segment) of a rectangle that starts at (i0, j0): terms T(i0+col, j0 + seg*SEG + r)
__device__ __forceinline__ ChunkAddr schunk_addr(const Grid& g, const float* Tm, int i0, int j0, int col, int seg, int nrows)
{
    ChunkAddr a;
    const int r0 = seg * SEG;
    a.n = min(SEG, nrows - r0);
    a.p = Tm + (size_t)(j0 + r0) * g.pitch + (i0 + col);
    return a;
}

}  // namespace

__global__ __launch_bounds__(256) void k_sx_chunk_sums(Grid g, const float* __restrict__ T, int i0, int j0, int ncol, int nseg,
        int nrows, double* __restrict__ sigma, const MemberState* __restrict__ st)
{
    const int m = blockIdx.z;
    if (st && !st[m].active) return;
    const int col = blockIdx.x * 64 + threadIdx.x;
    const int seg = blockIdx.y * 4 + threadIdx.y;
    if (col >= ncol || seg >= nseg) return;
    const ChunkAddr a = schunk_addr(g, T + (size_t)m * g.plane, i0, j0, col, seg, nrows);
    double acc = 0.0;
    for (int r = 0; r < a.n; ++r) acc += (double)a.p[(size_t)r * g.pitch];
    sigma[(size_t)m * ncol * nseg + (size_t)col * nseg + seg] = acc;
}

__global__ __launch_bounds__(256) void k_sx_chunk_summaries(Grid g, const float* __restrict__ T, int i0, int j0, int ncol,
        int nseg, int nrows, const double* __restrict__ prefix, const double* __restrict__ coloff,
        SSum* __restrict__ chunk, const MemberState* __restrict__ st)
{
    const int m = blockIdx.z;
    if (st && !st[m].active) return;
    const int col = blockIdx.x * 64 + threadIdx.x;
    const int seg = blockIdx.y * 4 + threadIdx.y;
    if (col >= ncol || seg >= nseg) return;
    const size_t c = (size_t)m * ncol * nseg + (size_t)col * nseg + seg;
    const double P = fabs(coloff[(size_t)m * ncol + col] + prefix[c]);
    SSum out;
    if (!(P >= 1.2e-38) || !(P < 3.0e38)) {                  // zero / subnormal / NaN / Inf / overflowing prediction
        out.ke = 0; out.ko = 0; out.kmin = 0; out.kmax = 0; out.eu = -100000; out.valid = 0;
    } else {
        out = schunk_summary(schunk_addr(g, T + (size_t)m * g.plane, i0, j0, col, seg, nrows), g.pitch, ulp_exp(f2u((float)P)));
    }
    chunk[c] = out;
}

__global__ __launch_bounds__(256) void k_sx_col_summaries(int ncol, int nseg, const SSum* __restrict__ chunk,
        SSum* __restrict__ colsum, const MemberState* __restrict__ st)
{
    const int m = blockIdx.y;
    if (st && !st[m].active) return;
    const int col = blockIdx.x * 256 + threadIdx.x;
    if (col >= ncol) return;
    const SSum* cs = chunk + ((size_t)m * ncol + col) * nseg;
    SSum out = cs[0];
    for (int k = 1; k < nseg && out.valid; ++k) out = scompose(out, cs[k]);
    colsum[(size_t)m * ncol + col] = out;
}

// the walk: one wave per member
__global__ __launch_bounds__(64) void k_sx_walk(Grid g, const float* __restrict__ T, int i0, int j0, int ncol, int nseg, int nrows,
        const SSum* __restrict__ chunk, const SSum* __restrict__ colsum, float* __restrict__ out,
        int* __restrict__ slow_chunks, const MemberState* __restrict__ st)
{
    const int m = blockIdx.x;
    if (st && !st[m].active) return;
    const int lane = threadIdx.x;
    const float* Tm = T + (size_t)m * g.plane;
    const SSum* cols = colsum + (size_t)m * ncol;
    unsigned bits = 0u;                                       // the sum starts at +0
    int slow = 0;                                             // chunks added term by term (diagnostic)

    auto scan_take = [&](SSum me) -> int {
        const unsigned E = (bits >> 23) & 0xffu;
        if (E == 0u || E == 255u) return 0;                   // zero / subnormal / non-finite sum: term by term
        const int eu_now = (int)E - 150;
        const unsigned long long good = __ballot(me.valid && me.eu == eu_now);
        const int first_bad = (good == ~0ull) ? 64 : __builtin_ctzll(~good);
        if (first_bad == 0) return 0;
        int ke = me.ke, ko = me.ko, kmin = me.kmin, kmax = me.kmax;
        auto stepd = [&](auto ctrl_tag, auto rmask_tag, bool has) {
            constexpr int CTRL = decltype(ctrl_tag)::value, RMASK = decltype(rmask_tag)::value;
            const int a_ke = __builtin_amdgcn_update_dpp(0, ke, CTRL, RMASK, 0xf, false);
            const int a_ko = __builtin_amdgcn_update_dpp(0, ko, CTRL, RMASK, 0xf, false);
            const int a_mn = __builtin_amdgcn_update_dpp(0, kmin, CTRL, RMASK, 0xf, false);
            const int a_mx = __builtin_amdgcn_update_dpp(0, kmax, CTRL, RMASK, 0xf, false);
            // scompose(a, me): the earlier run a on the left (valid runs only: |counts| <= 64 * 2^24)
            const int nke = a_ke + ((a_ke & 1) ? ko : ke);
            const int nko = a_ko + ((a_ko & 1) ? ke : ko);
            const int nmn = min(a_mn, min(a_ke, a_ko) + kmin);
            const int nmx = max(a_mx, max(a_ke, a_ko) + kmax);
            if (has) { ke = nke; ko = nko; kmin = nmn; kmax = nmx; }
        };
        using std::integral_constant;
        const int rl = lane & 15;
        stepd(integral_constant<int, 0x111>{}, integral_constant<int, 0xf>{}, rl >= 1);
        stepd(integral_constant<int, 0x112>{}, integral_constant<int, 0xf>{}, rl >= 2);
        stepd(integral_constant<int, 0x114>{}, integral_constant<int, 0xf>{}, rl >= 4);
        stepd(integral_constant<int, 0x118>{}, integral_constant<int, 0xf>{}, rl >= 8);
        stepd(integral_constant<int, 0x142>{}, integral_constant<int, 0xa>{}, (lane & 16) != 0);
        stepd(integral_constant<int, 0x143>{}, integral_constant<int, 0xc>{}, lane >= 32);
        // apply to the magnitude M of the sum (V = +-M; the parity of V is the parity of M)
        const int M = (int)((bits & 0x7fffffu) | 0x800000u);
        const bool neg = (bits >> 31) != 0u;
        const int K = (M & 1) ? ko : ke;
        const int newM = neg ? M - K : M + K;
        const int lo = neg ? M - kmax : M + kmin;
        const int hi = neg ? M - kmin : M + kmax;
        const bool fits = lane < first_bad && lo >= 0x800001 && hi <= 0xffffff;      // (kmin falls, kmax grows along the lanes)
        const unsigned long long mask = __ballot(fits);
        const int n = (mask == ~0ull) ? 64 : __builtin_ctzll(~mask);
        const unsigned nb = (bits & 0xff800000u) | ((unsigned)newM & 0x7fffffu);
        if (n > 0) bits = (unsigned)__builtin_amdgcn_readlane((int)nb, __builtin_amdgcn_readfirstlane(n - 1));
        return n;
    };
    const SSum none = {0, 0, 0, 0, 0, 0};

    for (int col = 0; col < ncol;) {
        const int n = scan_take(col + lane < ncol ? cols[col + lane] : none);
        col += n;
        if (n == 64 || col >= ncol) continue;
        const SSum* ch = chunk + ((size_t)m * ncol + col) * nseg;
        for (int seg = 0; seg < nseg;) {
            const unsigned E = (bits >> 23) & 0xffu;
            const int eu_now = (int)E - 150;
            SSum me = none;
            if (seg + lane < nseg) me = ch[seg + lane];
            // (see k_ex_walk) a first chunk counted for another binade: the prediction was off, recount the window
            const bool stale = (seg + lane < nseg) && me.eu != eu_now;
            if (E != 0u && E != 255u && __shfl((int)stale, 0, 64)) {
                if (stale) me = schunk_summary(schunk_addr(g, Tm, i0, j0, col, seg + lane, nrows), g.pitch, eu_now);
            } else if (stale) {
                me = none;
            }
            seg += scan_take(me);
            if (seg >= nseg) break;
            // chunk `seg` does not fit: its genuine fp32 additions, in reference order
            const ChunkAddr a = schunk_addr(g, Tm, i0, j0, col, seg, nrows);
            float t = 0.0f;
            if (lane < a.n) t = a.p[(size_t)lane * g.pitch];
            float sacc = u2f(bits);
            const int tb = (int)f2u(t);
#pragma unroll
            for (int r = 0; r < SEG; ++r)
                if (r < a.n) sacc = sacc + u2f((unsigned)__builtin_amdgcn_readlane(tb, r));
            bits = f2u(sacc);
            ++seg;
            ++slow;
        }
        ++col;
    }
    if (lane == 0) { out[m] = u2f(bits); if (slow_chunks) slow_chunks[m] = slow; }
}

size_t exact_signed_ws_bytes(const Grid& g)
{
    const int ncol = g.nx, nrows = g.ny;                      // any sub-rectangle of a plane
    const int nseg = (nrows + SEG - 1) / SEG;
    const size_t nchunk = (size_t)ncol * nseg;
    return (size_t)g.members * (nchunk * (sizeof(double) + sizeof(SSum)) + (size_t)ncol * (2 * sizeof(double) + sizeof(SSum))) + 256;
}

void launch_exact_sum_signed(const Grid& g, const float* T, int i0, int i1, int j0, int j1, void* workspace,
                             float* out_sums, int* slow_chunks, const MemberState* st, hipStream_t s)
{
    const int ncol = i1 - i0, nrows = j1 - j0;
    const int nseg = (nrows + SEG - 1) / SEG;
    const size_t nchunk = (size_t)ncol * nseg;
    char* w = static_cast<char*>(workspace);
    double* sigma = reinterpret_cast<double*>(w);               w += (size_t)g.members * nchunk * sizeof(double);
    double* coltot = reinterpret_cast<double*>(w);              w += (size_t)g.members * ncol * sizeof(double);
    double* coloff = reinterpret_cast<double*>(w);              w += (size_t)g.members * ncol * sizeof(double);
    SSum* chunk = reinterpret_cast<SSum*>(w);                   w += (size_t)g.members * nchunk * sizeof(SSum);
    SSum* colsum = reinterpret_cast<SSum*>(w);
    const dim3 gc((ncol + 63) / 64, (nseg + 3) / 4, g.members), bc(64, 4, 1);
    const dim3 gcol((ncol + 255) / 256, g.members);
    hipLaunchKernelGGL(k_sx_chunk_sums, gc, bc, 0, s, g, T, i0, j0, ncol, nseg, nrows, sigma, st);
    hipLaunchKernelGGL(k_ex_col_prefix, gcol, dim3(256), 0, s, ncol, nseg, sigma, coltot, st);
    hipLaunchKernelGGL(k_ex_scan_cols, dim3(g.members), dim3(1024), 0, s, ncol, coltot, coloff, st);
    hipLaunchKernelGGL(k_sx_chunk_summaries, gc, bc, 0, s, g, T, i0, j0, ncol, nseg, nrows, sigma, coloff, chunk, st);
    hipLaunchKernelGGL(k_sx_col_summaries, gcol, dim3(256), 0, s, ncol, nseg, chunk, colsum, st);
    hipLaunchKernelGGL(k_sx_walk, dim3(g.members), dim3(64), 0, s, g, T, i0, j0, ncol, nseg, nrows, chunk, colsum, out_sums, slow_chunks, st);
}

}  // namespace xee
