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
//     A. per chunk of 32 consecutive terms (in reference order): fp64 sum            (parallel)
//     B. exclusive scan of the chunk sums -> predicted running sum, hence predicted binade, at
//        every chunk start                                                            (parallel)
//     C. per chunk, for its predicted binade: the ulp counts K_even / K_odd it adds for an even /
//        odd incoming S (exact integer arithmetic on the fp32 fields; ties handled) (parallel)
//     D. per column: compose its chunks' (K_even, K_odd) when they share one binade  (parallel)
//     E. one wave per member walks the columns in reference order: bits += K[parity] whenever the
//        actual binade equals the predicted one and the sum stays inside it -- checked on the real
//        bit pattern, so a wrong prediction can only cost time -- else it descends to the column's
//        chunks, and for a chunk that still does not fit (binade crossing, a term larger than half
//        the sum, NaN/Inf) performs the 32 genuine fp32 additions.
//   Typically ~25 of the 16,352 chunks of a 1024x512 plane take the slow path.
#include "xee_device.h"

#pragma clang fp contract(off)

namespace xee {

namespace {

constexpr int SEG = EXACT_SEG;

__device__ __forceinline__ unsigned f2u(float x) { return __builtin_bit_cast(unsigned, x); }
__device__ __forceinline__ float u2f(unsigned x) { return __builtin_bit_cast(float, x); }

// exponent of the ulp of a non-negative fp32 bit pattern: ulp = 2^eu; subnormals share E=1's ulp
__device__ __forceinline__ int ulp_exp(unsigned bits)
{
    const int E = (int)((bits >> 23) & 0xffu);
    return (E > 0 ? E : 1) - 150;
}

// chunk c = (column, segment) of member m: terms R(i, j0 + seg*SEG + r)^2, r = 0..SEG-1
struct ChunkAddr {
    const float* p;   // first term's x
    int n;            // number of terms in this chunk
};

__device__ __forceinline__ ChunkAddr chunk_addr(const Grid& g, const float* Rm, int col, int seg, int nrows)
{
    ChunkAddr a;
    const int r0 = seg * SEG;
    a.n = min(SEG, nrows - r0);
    a.p = Rm + (size_t)(1 + r0) * g.pitch + (1 + col);     // interior starts at (1,1)
    return a;
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
    double acc = 0.0;
    for (int r = 0; r < a.n; ++r) {
        const float x = a.p[(size_t)r * g.pitch];
        const float t = x * x;
        acc += (double)t;
    }
    sigma[(size_t)m * ncol * nseg + (size_t)col * nseg + seg] = acc;
}

// B. exclusive scan over the chunks of one member, reference order (column-major over chunks)
__global__ __launch_bounds__(1024) void k_ex_scan(int nchunk, const double* __restrict__ sigma,
        double* __restrict__ prefix, const MemberState* __restrict__ st)
{
    __shared__ double part[1024];
    const int m = blockIdx.x;
    if (st && !st[m].active) return;
    const double* sg = sigma + (size_t)m * nchunk;
    double* pf = prefix + (size_t)m * nchunk;
    const int per = (nchunk + 1023) / 1024;
    const int lo = min(nchunk, (int)threadIdx.x * per), hi = min(nchunk, lo + per);
    double loc = 0.0;
    for (int k = lo; k < hi; ++k) loc += sg[k];
    part[threadIdx.x] = loc;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {               // Hillis-Steele inclusive scan
        double v = 0.0;
        if ((int)threadIdx.x >= off) v = part[threadIdx.x - off];
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    double run = part[threadIdx.x] - loc;                    // exclusive prefix of this thread's range
    for (int k = lo; k < hi; ++k) { pf[k] = run; run += sg[k]; }
}

// C. exact ulp counts of each chunk for its predicted binade
__global__ __launch_bounds__(256) void k_ex_chunk_summaries(Grid g, const float* __restrict__ R, int ncol, int nseg,
        int nrows, const double* __restrict__ prefix, ExactSum* __restrict__ chunk,
        const MemberState* __restrict__ st)
{
    const int m = blockIdx.z;
    if (st && !st[m].active) return;
    const int col = blockIdx.x * 64 + threadIdx.x;
    const int seg = blockIdx.y * 4 + threadIdx.y;
    if (col >= ncol || seg >= nseg) return;
    const size_t c = (size_t)m * ncol * nseg + (size_t)col * nseg + seg;
    const ChunkAddr a = chunk_addr(g, R + (size_t)m * g.plane, col, seg, nrows);
    ExactSum out;
    out.ke = 0; out.ko = 0; out.valid = 1;
    const double P = prefix[c];
    if (!(P >= 0.0) || !(P < 3.0e38)) {                      // NaN / Inf / overflowing prediction
        out.eu = -100000; out.valid = 0;
        chunk[c] = out;
        return;
    }
    const int eu = ulp_exp(f2u((float)P));
    out.eu = eu;
    unsigned ke = 0, ko = 0;
    bool ok = true;
    for (int r = 0; r < a.n; ++r) {
        const float x = a.p[(size_t)r * g.pitch];
        const unsigned tb = f2u(x * x);                      // to_dat(i,j)**2.0, >= +0
        const unsigned Et = (tb >> 23) & 0xffu;
        const unsigned fr = tb & 0x7fffffu;
        if (Et == 255u) { ok = false; break; }               // Inf / NaN term: genuine adds only
        const unsigned Mt = Et ? (fr | 0x800000u) : fr;
        if (Mt == 0u) continue;                              // + 0
        const int shift = eu - ((int)(Et ? Et : 1u) - 150);
        if (shift <= 0) { ok = false; break; }               // term >= 2^23 ulps: leaves the binade
        if (shift >= 25) continue;                           // < half an ulp: absorbed
        const unsigned q = Mt >> shift;
        const unsigned rem = Mt & ((1u << shift) - 1u);
        const unsigned half = 1u << (shift - 1);
        const unsigned up_e = (rem > half) || (rem == half && ((ke + q) & 1u));
        const unsigned up_o = (rem > half) || (rem == half && (((ko + q) & 1u) ^ 1u));
        ke += q + up_e;
        ko += q + up_o;
    }
    if (!ok || ke >= (1u << 24) || ko >= (1u << 24)) { out.valid = 0; ke = 0; ko = 0; }
    out.ke = ke; out.ko = ko;
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
    unsigned ke = out.ke, ko = out.ko;
    bool ok = out.valid != 0;
    for (int k = 1; k < nseg && ok; ++k) {
        const ExactSum b = cs[k];
        if (!b.valid || b.eu != out.eu) { ok = false; break; }
        // incoming parity p, after the prefix the parity is p ^ (K_prefix & 1)
        ke += ((ke & 1u) ? b.ko : b.ke);
        ko += (((ko & 1u) ^ 1u) ? b.ko : b.ke);
        if (ke >= (1u << 24) || ko >= (1u << 24)) ok = false;
    }
    out.valid = ok ? 1 : 0;
    out.ke = ok ? ke : 0;
    out.ko = ok ? ko : 0;
    colsum[(size_t)m * ncol + col] = out;
}

// E. the walk: one wave per member, every lane carries the same (wave-uniform) state
__global__ __launch_bounds__(64) void k_ex_walk(Grid g, const float* __restrict__ R, int ncol, int nseg, int nrows,
        const ExactSum* __restrict__ chunk, const ExactSum* __restrict__ colsum, float* __restrict__ out,
        int* __restrict__ slow_chunks, const MemberState* __restrict__ st)
{
    const int m = blockIdx.x;
    if (st && !st[m].active) return;
    const int lane = threadIdx.x;
    const float* Rm = R + (size_t)m * g.plane;
    const ExactSum* cols = colsum + (size_t)m * ncol;
    unsigned bits = 0u;                                       // err_now = 0
    int slow = 0;

    // Take as many of the next (up to 64) summaries as fit, in ONE step: lane l loads summary base+l,
    // an inclusive wave scan composes them (composition of (K_even,K_odd) maps is associative), so
    // lane l holds the summary of the run [base, base+l]; runs that are valid for the current binade
    // and keep the sum inside it form a prefix (K only grows); the longest one is applied.
    // Returns how many summaries were consumed (0: the first one itself does not fit).
    auto take_run = [&](const ExactSum* src, int base, int count) -> int {
        ExactSum me;
        if (base + lane < count) me = src[base + lane];
        else { me.ke = 0; me.ko = 0; me.eu = 0; me.valid = 0; }
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            ExactSum a;                                       // the run that ends `off` lanes earlier
            a.ke = __shfl_up(me.ke, off, 64); a.ko = __shfl_up(me.ko, off, 64);
            a.eu = __shfl_up(me.eu, off, 64); a.valid = __shfl_up(me.valid, off, 64);
            if (lane >= off) {
                const unsigned ke = a.ke + ((a.ke & 1u) ? me.ko : me.ke);
                const unsigned ko = a.ko + (((a.ko & 1u) ^ 1u) ? me.ko : me.ke);
                const bool ok = a.valid && me.valid && a.eu == me.eu && ke < (1u << 24) && ko < (1u << 24);
                me.ke = ok ? ke : 0u; me.ko = ok ? ko : 0u; me.valid = ok ? 1 : 0;
            }
        }
        const unsigned nb = bits + ((bits & 1u) ? me.ko : me.ke);
        const unsigned e0 = bits >> 23, e1 = nb >> 23;
        // K ulps keep the sum in its binade iff the exponent field (0 and 1 merged) is unchanged
        const bool fits = me.valid && me.eu == ulp_exp(bits) && (e0 > 1u ? e0 : 1u) == (e1 > 1u ? e1 : 1u) && e1 < 255u;
        const unsigned long long mask = __ballot(fits);
        const int n = (mask == ~0ull) ? 64 : __builtin_ctzll(~mask);
        if (n > 0) bits = __shfl(nb, n - 1, 64);
        return n;
    };

    for (int col = 0; col < ncol;) {
        const int n = take_run(cols, col, ncol);
        col += n;
        if (n > 0 || col >= ncol) continue;
        // column `col` does not fit as a whole: same game over its chunks
        const ExactSum* ch = chunk + ((size_t)m * ncol + col) * nseg;
        for (int seg = 0; seg < nseg;) {
            const int k = take_run(ch, seg, nseg);
            seg += k;
            if (k > 0 || seg >= nseg) continue;
            // chunk `seg` does not fit (binade crossing, a term above half the sum, Inf/NaN):
            // genuine sequential fp32 additions -- lanes fetch, the sum runs in reference order
            const ChunkAddr a = chunk_addr(g, Rm, col, seg, nrows);
            float t = 0.0f;
            if (lane < a.n) { const float x = a.p[(size_t)lane * g.pitch]; t = x * x; }
            float sacc = u2f(bits);
            for (int r = 0; r < a.n; ++r) sacc = sacc + __shfl(t, r, 64);
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
    return (size_t)g.members * (nchunk * (2 * sizeof(double) + sizeof(ExactSum)) + (size_t)ncol * sizeof(ExactSum)) + 256;
}

void launch_exact_sumsq(const Grid& g, const float* R, void* workspace, float* out_sums, int* slow_chunks,
                        const MemberState* st, hipStream_t s)
{
    const int ncol = g.nx - 2, nrows = g.ny - 2;
    const int nseg = (nrows + SEG - 1) / SEG;
    const size_t nchunk = (size_t)ncol * nseg;
    char* w = static_cast<char*>(workspace);
    double* sigma = reinterpret_cast<double*>(w);               w += (size_t)g.members * nchunk * sizeof(double);
    double* prefix = reinterpret_cast<double*>(w);              w += (size_t)g.members * nchunk * sizeof(double);
    ExactSum* chunk = reinterpret_cast<ExactSum*>(w);           w += (size_t)g.members * nchunk * sizeof(ExactSum);
    ExactSum* colsum = reinterpret_cast<ExactSum*>(w);
    const dim3 gc((ncol + 63) / 64, (nseg + 3) / 4, g.members), bc(64, 4, 1);
    hipLaunchKernelGGL(k_ex_chunk_sums, gc, bc, 0, s, g, R, ncol, nseg, nrows, sigma, st);
    hipLaunchKernelGGL(k_ex_scan, dim3(g.members), dim3(1024), 0, s, (int)nchunk, sigma, prefix, st);
    hipLaunchKernelGGL(k_ex_chunk_summaries, gc, bc, 0, s, g, R, ncol, nseg, nrows, prefix, chunk, st);
    hipLaunchKernelGGL(k_ex_col_summaries, dim3((ncol + 255) / 256, g.members), dim3(256), 0, s, ncol, nseg, chunk, colsum, st);
    hipLaunchKernelGGL(k_ex_walk, dim3(g.members), dim3(64), 0, s, g, R, ncol, nseg, nrows, chunk, colsum, out_sums,
                       slow_chunks, st);
}

}  // namespace xee
