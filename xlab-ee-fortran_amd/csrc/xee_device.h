// xee_device.h -- internal types shared by the HIP kernels and the C-ABI layer of libxee.
//
// Device data layout (see DESIGN.md "Data layout in HBM"):
//   every 2-D field of one ensemble member is a *plane* of ny rows x pitch floats, r (i) contiguous,
//   pitch = nx rounded up to 64 floats (256-byte rows, every row 16-byte aligned for dwordx4 access);
//   member m of a field lives at base + m*plane (plane = pitch*ny floats); the nine stencil
//   coefficients are nine such fields (SoA), coefficient k of member m at coe + (k*members+m)*plane.
//   The reference's AoS coe(9,nx,ny) (xtt-lib-fortran/elliptic_tools.f90:12) only exists at the
//   Tier-1 host boundary.
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace xee {

struct Grid {
    int nx, ny;      // logical O-grid extents (nr, nz)
    int pitch;       // floats per row, multiple of 64
    int members;     // ensemble members resident on this device
    size_t plane;    // floats per member per field = pitch * ny
    size_t kstride;  // floats between planes k and k + 1 of the SoA fields (coe, base); 0 = members * plane.  Non-zero in the
                     // view of a member sub-range that xee_ensemble_solve hands to one of its two sweep streams (k_sweep_q only)
    float two_pq4;   // 2 (4 dx dy), as cal_coe rounds it (elliptic_tools.f90:31-32): BX, BY = (b + b') / two_pq4 (k_base_planes; k_sweep_z<nine> divides itself)
    int base_shared; // bit k: generating plane k (AP, CQ, BX, BY; 4: the solver coefficient b they are made of) holds the same bits for every member -- members that share
                     // their operator, e.g. an ensemble that varies the forcing only -- so k_sweep_q / k_sweep_z read member 0's
                     // copy for all of them (4 MB that stay in the caches instead of 8 B per point and pass from HBM)
};

// per-member solver state, owned by the device between checks
// (the locals of solve_elliptic, xtt-lib-fortran/elliptic_tools.f90:102-109,160-164)
struct MemberState {
    int active;        // 1 while the member still iterates
    int iters;         // iterations used (max_iter on return, :253)
    int err_bits;      // :244
    int converge_cnt;  // :160
    int lose_cnt;      // :161
    float err_before;  // :163
    float err_now;     // strategy_r1 on return
    float ratio;       // strategy_r2 on return
};

// what a run of terms adds to a running fp32 sum that stays inside one binade (xee_exactsum.hip)
struct ExactSum {
    unsigned ke, ko;   // ulps added when the incoming ulp count is even / odd
    int eu;            // log2(ulp) of the binade the counts are valid for
    int valid;
};
constexpr int EXACT_SEG = 32;   // terms per chunk
size_t exact_ws_bytes(const Grid& g);
int    exact_kernel_setup();    // one-time function attributes; returns hipError_t as int
// out_sums[m] = the sequential fp32 sum (i outer, j inner) of T(i,j), i0 <= i < i1, j0 <= j < j1, terms of either
// sign, bit for bit (the EE integrals); workspace of exact_signed_ws_bytes(g) bytes
size_t exact_signed_ws_bytes(const Grid& g);
void launch_exact_sum_signed(const Grid& g, const float* T, int i0, int i1, int j0, int j1, void* workspace,
                             float* out_sums, int* slow_chunks /* optional, per member: chunks added term by term */,
                             const MemberState* st, hipStream_t s);
// out_sums[m] = the reference's sequential fp32 sum of R(i,j)^2 over the interior, bit for bit
void launch_exact_sumsq(const Grid& g, const float* R, void* workspace, float* out_sums,
                        int* slow_chunks /* optional: chunks summed term by term */,
                        const MemberState* st, hipStream_t s);

// ... in two halves, for drivers that keep the wide kernels on the sweep stream and only the serial walk beside it
void launch_exact_sumsq_prepare(const Grid& g, const float* R, void* workspace, const MemberState* st, hipStream_t s);
void launch_exact_sumsq_walk(const Grid& g, const float* R, void* workspace, float* out_sums, int* slow_chunks,
                             const MemberState* st, hipStream_t s);

struct ConvParams {
    float r1, r2;          // after the <=0 -> HUGE substitution (:112-124)
    int converge_time;     // :136-139
    int lost_rate;         // :141-144
    int cnt;               // iteration number of this check
    int is_last;           // cnt == max_iter (:242-248)
    int n_interior;        // (nx-2)*(ny-2), integer product converted to real (:199)
    int debug;             // 2: per-check trace (:202-204)
};

// ---- launch wrappers (xee_kernels.hip) ------------------------------------------------------
void launch_solver_coef(const Grid& g, const float* A, const float* B, const float* C,
                        const float* rcuva, const float* rho, float* sA, float* sB, float* sC,
                        hipStream_t s);
void launch_cal_coe(const Grid& g, const float* sA, const float* sB /* nullptr: B = 0 */,
                    const float* sC, float* coe, float dx, float dy, hipStream_t s);
void launch_apply(const Grid& g, const float* psi, const float* coe, float* out, hipStream_t s);
// one damped-Jacobi iteration (streaming kernel): psi_out = psi_in + alpha*R/(-c5), R optional
void launch_sweep(const Grid& g, const float* psi_in, float* psi_out, const float* coe,
                  const float* f, float* res /* nullptr: not written */, const MemberState* st,
                  float alpha, hipStream_t s);
// run-time options of the sweep (xee_set_default_option / xee_ensemble_set_option; no environment variables)
struct SweepOpts {
    int T = 8;             // "tb_T": iterations per temporally blocked launch, 1..tb_max_T() (8: best on all four bench
                           //         workloads once the prologue got cheaper, profiles/r02; round 1 used 6)
    int kernel = 0;        // "kernel": 0 auto | 1 nine coefficient planes (k_sweep_tb) | 2 generating planes, IEEE
                           //           division (k_sweep_tb4) | 3 generating planes, hoisted reciprocal (k_sweep_q)
                           //           | 4 as 3, with the streaming k_sweep_z wherever it applies, whatever the batch size
    int z_pairs = 0;       // "z_pairs": chunk pairs per strip of k_sweep_z, 0 = chosen from the batch size
    int share_planes = 1;  // "share_planes": generating planes that are the same for all members are read from member 0's copy
    int q_exact_only = 0;  // "q_exact_only": k_sweep_q / k_sweep_z run their exact body on every tile / wave
    int elision = 1;       // "elision": k_sweep_tb skips coefficient planes that are uniform over the batch
    int tile = 0;          // "tile": NW*100 + R (waves x rows per lane), 0 = the kernel's default
    int graph_blocks = 8;  // "graph_blocks": check blocks per graph for small batches (solve_graphed), 1..8; 0: one block at a time
    int split_halves = 1;  // "split_halves": a large batch is solved as two halves of its member range side by side (two sweep
                           //                 streams, two host threads), so that one half's launches fill the other's tails
    int graph_events = 0;  // "graph_events": 1 = an external event record per block inside the graph (its check starts at once);
                           //                 0 = one record behind the graph (the checks of a graph start when it has ended): default,
                           //                 121.8 against 126.7 ms per 100,000 iterations of one 1024x512 case (profiles/r03)
};
// what the last blocked launch was (xee_ensemble_sweep_info)
constexpr size_t kPlaneSlack = 256;   // bytes allocated behind every array of planes (see alloc_planes)

struct SweepDesc {
    const char* name = "none";
    int T = 0, tile_w = 0, tile_h = 0;
    long long tiles = 0;   // tiles (workgroups with work) of the last launch
    bool streamed = false; // it was k_sweep_z
};
// coefficient planes that hold one bit pattern over the interior of every member (set by assemble)
struct UniformVals {
    unsigned mask;   // bit k: plane k is uniform
    float v[9];      // its value
};
void launch_uniform_check(const Grid& g, const float* coe, unsigned* differs /* device, 1 word */, hipStream_t s);
void launch_coe_compare(const Grid& g, const float* x, const float* y, unsigned* differs /* device, 1 word */, hipStream_t s);
void launch_base_shared_check(const Grid& g, const float* base, int nplanes, unsigned* differs /* device, 1 word: bit k = plane k differs between members */, hipStream_t s);
// T damped-Jacobi iterations in one launch (temporally blocked kernel), 2 <= T <= tb_max_T()
void launch_sweep_tb(const Grid& g, const float* psi_in, float* psi_out, const float* coe,
                     const float* f, float* res /* residual of the LAST iteration, or nullptr */,
                     const MemberState* st, float alpha, int T, const UniformVals& uv, const SweepOpts& o,
                     SweepDesc* desc, hipStream_t s);
// the same blocked sweep fed by the operator's four generating planes AP, CQ, BX, BY (xee_sweep4.hip):
// base + k*members*plane, k = 0..3; the nine weights are rebuilt in the kernel prologue
void launch_base_planes(const Grid& g, const float* sA, const float* sB /* nullptr: B = 0 */, const float* sC,
                        float* base, float dx, float dy, hipStream_t s);
void launch_sweep_tb4(const Grid& g, const float* psi_in, float* psi_out, const float* base, const float* f,
                      float* res, const MemberState* st, float alpha, int T, bool barotropic, const SweepOpts& o,
                      SweepDesc* desc, hipStream_t s);
// the same, with the loop-invariant half of the update's division hoisted and a validated optimistic body
// per tile (xee_sweepq.hip); bit-identical to launch_sweep_tb4
// (returns whether a residual-writing launch also wrote `snap`)
bool launch_sweep_q(const Grid& g, const float* psi_in, float* psi_out, const float* base, const float* f,
                    float* res, float* snap /* residual-writing launches: a second copy of the new iterate's valid points, or nullptr */,
                    const MemberState* st, float alpha, int T, bool barotropic, const SweepOpts& o,
                    int* diag /* device, 2 ints or nullptr: tiles redone by the exact body in residual-writing / plain launches */,
                    SweepDesc* desc, hipStream_t s);
void launch_quotient_check(unsigned long long seed, int blocks, int per_thread, int d_exp_lo, int d_exp_hi,
                           unsigned long long* mismatches /* device, 2 words */, float* first_bad /* device, 2 */,
                           hipStream_t s);
// large batches, star operator, alpha = 1, plain launches: the blocked sweep streamed along z by single waves
// (xee_sweepz.hip); bit-identical to launch_sweep_q where z_member_ok holds for every member
int  sweep_z_T(bool barotropic);                          // iterations per launch: 6, nine-weight operator (member pairs that share AP, CQ) 4
int  sweep_z_pairs(const Grid& g, int members, int streams, int min_eff_percent, int T);   // chunk pairs per strip, 0 = "use k_sweep_q"
void launch_z_extremes(const Grid& g, const float* base, const float* f, bool barotropic, unsigned* zstat /* device, 6 per member */, hipStream_t s);
bool z_member_ok(const unsigned* zstat6 /* host copy of one member's six words */, bool barotropic, float two_pq4);
bool sweep_z_can(int T, bool with_residual, bool barotropic);   // star: 6 iterations plain, 6 or 4 with the residual of the last one; nine weights: 4
void launch_sweep_z(const Grid& g, const float* psi_in, float* psi_out, const float* base, const float* f,
                    float* res /* nullptr, or the residual plane of the launch's last iteration ... */,
                    float* snap /* ... and (or nullptr) a second copy of the new iterate's interior, active members only */,
                    const MemberState* st, int T, bool barotropic, int chunk_pairs, int exact_only,
                    int* diag /* device, 2 ints or nullptr: [1] counts the waves that fell back to the exact body */,
                    SweepDesc* desc, hipStream_t s);
int  tb_max_T();
// opt-in, non-parity: one 4-colour SOR iteration in place; residual plane of the current iterate
void launch_sor_iteration(const Grid& g, float* psi, const float* coe, const float* f, const MemberState* st,
                          float omega, hipStream_t s);
void launch_residual(const Grid& g, const float* psi, const float* coe, const float* f, float* res,
                     const MemberState* st, hipStream_t s);
// ... the same from the operator's generating planes; BAROTROPIC: red/black (two half sweeps per iteration)
void launch_sor_iteration_base(const Grid& g, float* psi, const float* base, const float* f, const MemberState* st,
                               float omega, bool barotropic, hipStream_t s);
void launch_residual_base(const Grid& g, const float* psi, const float* base, const float* f, float* res,
                          const MemberState* st, bool barotropic, hipStream_t s);
// sequential fp32 sum, reference loop order (i outer, j inner), of T or T*T over a sub-rectangle
void launch_seqsum(const Grid& g, const float* T, int i0, int i1, int j0, int j1, bool square,
                   float* out_sums, const MemberState* st, hipStream_t s);
void launch_conv_update(const Grid& g, const float* sums, MemberState* st, ConvParams p, hipStream_t s,
                        MemberState* view = nullptr /* device */, MemberState* host_out = nullptr /* pinned host */);
void launch_finalize(const Grid& g, MemberState* st, int max_iter, hipStream_t s);
void launch_state_init(const Grid& g, MemberState* st, hipStream_t s);
void launch_copy2(const Grid& g, const float* src, float* dst0, float* dst1, hipStream_t s);
void launch_negate(const Grid& g, const float* src, float* dst, hipStream_t s);
void launch_copy_active(const Grid& g, const float* src, float* dst, const MemberState* st, hipStream_t s);
void launch_cal_eta(const Grid& g, const float* psi, float* eta, const float* ra, const float* rcuva,
                    const float* rho, const float* exner, hipStream_t s);
void launch_cal_uw(const Grid& g, const float* psi, float* u, float* w, const float* ra,
                   const float* za, const float* rcuva, const float* rho, hipStream_t s);
// term planes of the EE integrals (then summed by launch_seqsum over [0,nx-1) x [0,ny-1))
void launch_terms_weight(const Grid& g, const float* weight, float* terms, const float* ra,
                         const float* za, const float* rcuva, const float* rho, hipStream_t s);
void launch_terms_Qeta(const Grid& g, const float* Q, const float* eta, float* terms, const float* ra,
                       const float* za, const float* rcuva, const float* rho, hipStream_t s);
void launch_wtheta(const Grid& g, const float* w, const float* theta, float* out, hipStream_t s);
void launch_rhs_thermal(const Grid& g, const float* Q, const float* ra, const float* exner,
                        float* tmpC, float* rhs, hipStream_t s);
void launch_rhs_from_B(const Grid& g, const float* b, float* f, hipStream_t s);
void launch_coe_aos_to_soa(const Grid& g, const float* aos /* (9,nx,ny) packed */, float* coe, hipStream_t s);
void launch_coe_soa_to_aos(const Grid& g, const float* coe, float* aos, hipStream_t s);
int    record_failure(int code, const char* msg);   // sets xee_last_error() of this thread, returns code
size_t seqsum_lds_bytes(const Grid& g, int nrows);
int    sweep_kernel_setup();   // one-time function attributes; returns hipError_t as int

}  // namespace xee
