// xee_api.hip -- the C-ABI of libxee (include/xee.h): ensemble bookkeeping, the solve_elliptic
// driver loop, and the Tier-1 drop-in entry points built on top of a one-member ensemble.
// No CPU fallback anywhere: without a HIP device every compute entry returns XEE_ERR_NO_DEVICE.
#include "../../include/xee.h"
#include "xee_device.h"

#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

using namespace xee;

namespace {

thread_local std::string g_err;
thread_local unsigned long long g_failures = 0;     // how many errors this thread has reported (see Tier1 below)

int fail(int code, const std::string& msg)
{
    g_err = msg;
    ++g_failures;
    return code;
}

#define HIPCHK(expr)                                                                           \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return fail(e_ == hipErrorNoDevice ? XEE_ERR_NO_DEVICE : XEE_ERR_HIP,              \
                        std::string(#expr) + ": " + hipGetErrorString(e_));                    \
    } while (0)

#define XCHK(expr)                       \
    do {                                 \
        int s_ = (expr);                 \
        if (s_ != XEE_OK) return s_;     \
    } while (0)

}  // namespace

// for the other translation units of the library (xee_stencils.hip): record a failure text
namespace xee {
int record_failure(int code, const char* msg) { return fail(code, msg ? msg : ""); }
}

namespace {

int require_device()
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(XEE_ERR_NO_DEVICE, "no HIP device visible (libxee has no CPU fallback)");
    return XEE_OK;
}

std::once_flag g_setup_once;
int g_setup_status = 0;

// what tests count instead of timing (xee_debug_counters): creations of device ensembles, device allocations made by the
// Tier-1 entry points for host-layout staging, thread-exit releases of Tier-1 caches
std::atomic<long long> g_n_ensembles{0}, g_n_tier1_stage_allocs{0}, g_n_thread_exit_releases{0};
std::atomic<bool> g_process_exiting{false};
struct ExitFlag { ~ExitFlag() { g_process_exiting.store(true); } } g_exit_flag;   // (statics die after the main thread's thread_locals)

}  // namespace

namespace {
// process-wide defaults of the run-time options: every ensemble created afterwards (the Tier-1 entry points
// create theirs internally) starts from them
struct Defaults { SweepOpts sweep{}; int sync_checks = 0; int exact_sum_serial = 0; } g_defaults;
std::mutex g_defaults_mutex;                 // (any thread may set them while another creates an ensemble)
Defaults read_defaults()
{
    std::lock_guard<std::mutex> lk(g_defaults_mutex);
    return g_defaults;
}

// one table for both setters: key -> the int it controls and its admitted range
int set_option_in(SweepOpts& o, int& sync_checks, int& exact_serial, const char* key, int value)
{
    struct Ent { const char* key; int* p; int lo, hi; };
    const Ent tab[] = {{"tb_T", &o.T, 1, tb_max_T()}, {"kernel", &o.kernel, 0, 4}, {"z_pairs", &o.z_pairs, 0, 64}, {"share_planes", &o.share_planes, 0, 1}, {"q_exact_only", &o.q_exact_only, 0, 1},
                       {"elision", &o.elision, 0, 1}, {"tile", &o.tile, 0, 9999}, {"graph_blocks", &o.graph_blocks, 0, 8}, {"split_halves", &o.split_halves, 0, 1}, {"graph_events", &o.graph_events, 0, 1}, {"sync_checks", &sync_checks, 0, 1},
                       {"exact_sum_serial", &exact_serial, 0, 1}};
    if (!key) return fail(XEE_ERR_INVALID_ARGUMENT, "null option key");
    for (const Ent& t : tab)
        if (strcmp(t.key, key) == 0) {
            if (value < t.lo || value > t.hi) return fail(XEE_ERR_INVALID_ARGUMENT, std::string("option ") + key + ": value out of range");
            *t.p = value;
            return XEE_OK;
        }
    return fail(XEE_ERR_INVALID_ARGUMENT, std::string("unknown option: ") + key);
}
}  // namespace

struct xee_ensemble {
    Grid g{};
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int first_case_id = 0;
    // per-member planes (members * plane floats each); optional ones are allocated on first use
    float *A = nullptr, *B = nullptr, *C = nullptr, *bc = nullptr, *f = nullptr;
    float* psi[2] = {nullptr, nullptr};
    float *sA = nullptr, *sB = nullptr, *sC = nullptr, *coe = nullptr, *res = nullptr;
    float *eta = nullptr, *u = nullptr, *w = nullptr, *Q = nullptr, *terms = nullptr;
    int cur = 0;   // psi[cur] holds the solution of every member after a solve
    // shared 1-D profiles
    float *ra = nullptr, *za = nullptr, *rcuva = nullptr, *rho = nullptr, *exner = nullptr;
    float dr = 0.f, dz = 0.f;
    bool have_profiles = false;
    MemberState* st = nullptr;
    MemberState* st_host = nullptr;   // pinned
    float* sums = nullptr;
    float* sums_host = nullptr;       // pinned
    std::vector<xee_case_record_t> rec;
    std::vector<int> sx_slow;     // [2][members]: chunks the last ee_scalars summed term by term (sum_Q, sum_Qeta)
    xee_timing_t timing{};
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    SweepOpts opts{};            // sweep options (xee_ensemble_set_option; initialised from the process defaults)
    SweepDesc last_sweep{};      // the blocked kernel the last launch used (xee_ensemble_sweep_info)
    bool sync_checks = false;    // option "sync_checks": the plain synchronous driver instead of the pipelined one
    void* ex_ws = nullptr;       // workspace of the parallel order-exact residual norm
    void* sx_ws = nullptr;       // ... and of the order-exact sum of signed terms (EE integrals), allocated on first use
    int* ex_slow = nullptr;      // per member: chunks the last check summed term by term
    bool exact_serial = false;   // option "exact_sum_serial": single-lane reference-order walk instead
    float* base = nullptr;       // the operator's four generating planes AP, CQ, BX, BY (xee_sweep4.hip)
    bool have_base = false;      // they describe the current coefficient planes (set by assemble)
    bool base_baro = false;      // ... of the BAROTROPIC operator (BX = BY = +0: not loaded)
    UniformVals uv{};            // coefficient planes found uniform by the last assemble / coe upload
    unsigned* uv_differs = nullptr;
    // pipelined residual checks (solve_pipelined): the check of iteration k runs on a check stream while the sweeps of
    // the next block(s) already run on `stream`.  Resources are per SLOT (check k uses slot k % (depth + 1)) or per check
    // stream (check k runs on cstream[k % depth]); depth is 1 for large batches and 2 for small ones, whose check (a
    // chain of latency-bound kernels) takes longer than a block of sweeps.
    hipStream_t cstream[2] = {nullptr, nullptr};
    hipEvent_t evE[3] = {nullptr, nullptr, nullptr}, evF[3] = {nullptr, nullptr, nullptr};   // sweep-side / check-side, by slot
    hipEvent_t evB[3][2] = {{nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}};     // sweep block begin/end, by slot
    hipEvent_t evZ[3][2] = {{nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}};     // ... its run of k_sweep_z launches
    float* res2 = nullptr;       // residual planes of slots 1 and 2 (slot 0 is `res`); res3 only at depth 2
    float* res3 = nullptr;
    float* snap = nullptr;       // (depth + 1) x members planes: the iterate right after each check iteration, by slot
    int snap_planes = 0;
    MemberState* st_host2 = nullptr;   // pinned, 3 x members: the member states after a check, by slot (written by the device)
    void* ex_ws2 = nullptr;      // second check stream's norm workspace, sums and slow counters (depth 2 only)
    float* sums2 = nullptr;
    int* ex_slow2 = nullptr;
    int* redo = nullptr;               // two diagnostic counters of k_sweep_q: tiles redone by the exact body (xee_sweepq.hip)
    std::map<int, float*> stash;       // device-resident copies of fields (xee_ensemble_field_save): key = field * 64 + slot
    // solve_graphed (small batches): a ring of GS slots (residual plane, snapshot, event pair, pinned host copy of the states)
    // and GV views of the member states
    float* gres = nullptr;
    float* gsnap = nullptr;
    MemberState* gst_host = nullptr;   // pinned
    MemberState* gst_view = nullptr;
    hipEvent_t gevE[24] = {}, gevF[24] = {};
    int g_slots = 0;                   // slots the ring was allocated for
    // Large batches are solved as TWO halves of the member range, each on its own sweep / check streams and host thread
    // (xee_ensemble_solve): the tail of one half's launch -- 7,744 tiles are 15.1 rounds of the 512 workgroups the GPU holds --
    // is filled by the other half's.  A half is a VIEW: its plane pointers point into this ensemble's (non-owned), its
    // streams, events, slots and norm workspaces are its own.
    bool is_view = false;
    xee_ensemble* half[2] = {nullptr, nullptr};
    // k_sweep_z (xee_sweepz.hip): chunk pairs per strip while this solve's plain launches of sweep_z_T() iterations go to it
    // (0: they do not); the members' extremes of |coe5| and |f| that decide it (device, 3 words per member)
    int z_pairs_now = 0;
    int z_T_now = 6;             // ... and their length: sweep_z_T(base_baro)
    unsigned* zstat = nullptr;
    bool base_shared_known = false;    // Grid.base_shared describes the current generating planes (else: tested at the next solve)
    unsigned* shared_differs = nullptr; // device word of the test
    float* aos_stage = nullptr;        // Tier 1: device staging of the reference's AoS coe(9,nx,ny), kept with the ensemble
    int win_first = 0, win_count = 0;  // member window of the forcing_* / field_save / field_restore calls (0 = all members)
    MemberState* st_view = nullptr;    // 4 x members: the sweeps' VIEWS of the member states in the pipelined driver.  The
                                       // state machine of check k writes view k % 4 (on a check stream); the sweeps read the
                                       // view of a check the host has seen finish and switch between launches only, so that
                                       // a workgroup's early exit on `active` is uniform although flags are cleared concurrently
};

namespace {

int alloc_planes(xee_ensemble* e, float** p, size_t nplanes = 1)
{
    if (*p) return XEE_OK;
    // (kPlaneSlack bytes more than the planes: the streaming sweep's 16-byte row fetches may run up to 12 bytes past the last
    // plane's end -- xee_sweepz.hip)
    const size_t bytes = e->g.plane * e->g.members * nplanes * sizeof(float) + kPlaneSlack;
    hipError_t err = hipMalloc(reinterpret_cast<void**>(p), bytes);
    if (err != hipSuccess) return fail(XEE_ERR_ALLOC, std::string("hipMalloc: ") + hipGetErrorString(err));
    HIPCHK(hipMemsetAsync(*p, 0, bytes, e->stream));
    return XEE_OK;
}

int use(xee_ensemble* e)
{
    if (!e) return fail(XEE_ERR_INVALID_ARGUMENT, "null ensemble");
    HIPCHK(hipSetDevice(e->device));
    return XEE_OK;
}

struct FieldRef { float** ptr; int n1, n2; size_t planes_off; };

// field id -> device pointer (allocated on demand) and logical shape
int field_ref(xee_ensemble* e, int field, float** base, int* n1, int* n2)
{
    const int nr = e->g.nx, nz = e->g.ny;
    float** p = nullptr;
    switch (field) {
        case XEE_F_A: p = &e->A; *n1 = nr; *n2 = nz; break;
        case XEE_F_B: p = &e->B; *n1 = nr; *n2 = nz; break;
        case XEE_F_C: p = &e->C; *n1 = nr; *n2 = nz; break;
        case XEE_F_BC_INIT: p = &e->bc; *n1 = nr; *n2 = nz; break;
        case XEE_F_FORCING: p = &e->f; *n1 = nr; *n2 = nz; break;
        case XEE_F_PSI: p = &e->psi[e->cur]; *n1 = nr; *n2 = nz; break;
        case XEE_F_SOLVER_A: p = &e->sA; *n1 = nr - 1; *n2 = nz - 2; break;
        case XEE_F_SOLVER_B: p = &e->sB; *n1 = nr - 1; *n2 = nz - 1; break;
        case XEE_F_SOLVER_C: p = &e->sC; *n1 = nr - 2; *n2 = nz - 1; break;
        case XEE_F_ETA: p = &e->eta; *n1 = nr - 1; *n2 = nz; break;
        case XEE_F_U: p = &e->u; *n1 = nr; *n2 = nz - 1; break;
        case XEE_F_W: p = &e->w; *n1 = nr - 1; *n2 = nz; break;
        case XEE_F_Q: p = &e->Q; *n1 = nr - 1; *n2 = nz - 1; break;
        case XEE_F_RESIDUAL: p = &e->res; *n1 = nr; *n2 = nz; break;
        default:
            if (field >= XEE_F_COE0 && field < XEE_F_COE0 + 9) {
                XCHK(alloc_planes(e, &e->coe, 9));
                *base = e->coe + (size_t)(field - XEE_F_COE0) * e->g.members * e->g.plane;
                *n1 = nr; *n2 = nz;
                return XEE_OK;
            }
            return fail(XEE_ERR_INVALID_ARGUMENT, "unknown field id");
    }
    XCHK(alloc_planes(e, p));
    *base = *p;
    return XEE_OK;
}

int check_launch(const char* what);

// The sweep stream gets the device's highest priority, the residual-check stream the lowest: checks run in the
// shadow of the next block's sweeps (solve_pipelined) and have a whole block of slack, so the dispatcher should hand
// compute units to sweep workgroups first (measured: sweep launches 116 -> see profiles/r02 with the checks co-running).
hipError_t create_stream(hipStream_t* s, bool high)
{
    int least = 0, greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) { least = greatest = 0; }
    return hipStreamCreateWithPriority(s, hipStreamNonBlocking, high ? greatest : least);
}

// after the coefficient planes changed: find the ones that are uniform over the whole batch
int detect_uniform_planes(xee_ensemble* e)
{
    e->uv = UniformVals{};
    e->have_base = false;        // callers that also build the generating planes set it afterwards
    if (!e->uv_differs) HIPCHK(hipMalloc(reinterpret_cast<void**>(&e->uv_differs), sizeof(unsigned)));
    launch_uniform_check(e->g, e->coe, e->uv_differs, e->stream);
    unsigned differs = 0x1ffu;
    HIPCHK(hipMemcpyAsync(&differs, e->uv_differs, sizeof(unsigned), hipMemcpyDeviceToHost, e->stream));
    float v[9];
    const size_t ks = (size_t)e->g.members * e->g.plane;
    for (int k = 0; k < 9; ++k)
        HIPCHK(hipMemcpyAsync(&v[k], e->coe + k * ks + e->g.pitch + 1, sizeof(float), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    e->uv.mask = (~differs) & 0x1ffu;
    for (int k = 0; k < 9; ++k) e->uv.v[k] = v[k];
    return XEE_OK;
}

// Which generating planes do all members share?  (Grid.base_shared; tested once per set of planes: one pass over them and a
// 4-byte copy.)  Members that share their operator -- an ensemble that varies the forcing only, the kinds of a source-term
// sweep -- then read ONE copy of AP, CQ (, BX, BY), which stays in the caches: 12 instead of 20 bytes per point and pass from HBM.
int base_shared_update(xee_ensemble* e)
{
    if (!e->have_base || e->base_shared_known) return XEE_OK;
    e->g.base_shared = 0;
    e->base_shared_known = true;
    if (!e->opts.share_planes) return XEE_OK;
    const int nplanes = e->base_baro ? 2 : 5;             // (AP, CQ | BX, BY and the copy of b)
    const unsigned all = (1u << nplanes) - 1u;
    if (e->g.members < 2) { e->g.base_shared = (int)all; return XEE_OK; }
    if (!e->shared_differs) HIPCHK(hipMalloc(reinterpret_cast<void**>(&e->shared_differs), sizeof(unsigned)));
    launch_base_shared_check(e->g, e->base, nplanes, e->shared_differs, e->stream);
    unsigned differs = all;
    HIPCHK(hipMemcpyAsync(&differs, e->shared_differs, sizeof(unsigned), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    e->g.base_shared = (int)(~differs & all);
    return XEE_OK;
}

// Whether this solve's plain launches go to the streaming k_sweep_z, and with how many chunk pairs per strip
// (e->z_pairs_now): the operator from its generating planes (the star; nine weights where the members share AP and CQ), alpha = 1, a launch of `members` cases beside `streams` - 1
// like it that fills the device (option "kernel" = 4: whatever it fills), and every member within the bounds under which the
// kernel's hoisted division is the IEEE one (xee_sweepz.hip).  One pass over f, AP, CQ and a 12-byte-per-member copy.
// *first_bad (if given): -1, or the first member outside those bounds when that alone keeps the batch off the stream -- the
// solve may then stream the members before it and leave the others to the tiled sweep, side by side (xee_ensemble_solve)
int z_threshold(const xee_ensemble* e, int members)
{
    // (measured end to end, tools/auto_vs_tiled.py, profiles/r03/auto_vs_tiled.txt.  Members that each read their own operator planes:
    // at 73 .. 75 the streamed solve is level with the tiled one or, for one 4096x2048 case, 4 - 8 % behind it, at 79 .. 82 it is
    // 10 % ahead: 78.  Members that share AP and CQ (one copy, walked in pairs): 8 - 13 % ahead at 73 already, 20 % behind at 41: 70.
    // The nine-weight operator, profiles/r03/auto_vs_tiled_baroclinic.txt: 24 % ahead at 81, 8 % at 75, level at 71, 30 % behind at 43: 72)
    if (e->opts.kernel == 4) return 0;
    const bool shared = (e->g.base_shared & 3) == 3 && members >= 2;
    return !e->base_baro ? 72 : shared ? 70 : 78;
}

int z_decide(xee_ensemble* e, float alpha, int tb_T, int streams, int members, int* first_bad = nullptr)
{
    e->z_pairs_now = 0;
    if (first_bad) *first_bad = -1;
    const int k = e->opts.kernel;
    const int zT = sweep_z_T(e->base_baro);
    e->z_T_now = zT;
    if (!(e->have_base && alpha == 1.0f && (k == 0 || k == 4) && tb_T >= zT)) return XEE_OK;
    // (the nine-weight operator is streamed for member pairs that share AP and CQ only)
    if (!e->base_baro && !((e->g.base_shared & 3) == 3 && members >= 2)) return XEE_OK;
    if (e->g.plane * sizeof(float) >= ((size_t)1 << 31) - 64) return XEE_OK;       // (byte offsets inside a plane are 32-bit)
    int P = sweep_z_pairs(e->g, members, streams, z_threshold(e, members), zT);
    if (P > 0 && e->opts.z_pairs > 0) P = e->opts.z_pairs;
    if (P <= 0) return XEE_OK;
    if (!e->zstat) HIPCHK(hipMalloc(reinterpret_cast<void**>(&e->zstat), 6 * sizeof(unsigned) * e->g.members));
    launch_z_extremes(e->g, e->base, e->f, e->base_baro, e->zstat, e->stream);
    std::vector<unsigned> h(6 * (size_t)e->g.members);
    HIPCHK(hipMemcpyAsync(h.data(), e->zstat, h.size() * sizeof(unsigned), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    for (int m = 0; m < e->g.members; ++m)
        if (!z_member_ok(&h[6 * (size_t)m], e->base_baro, e->g.two_pq4)) { if (first_bad) *first_bad = m; return XEE_OK; }
    e->z_pairs_now = P;
    return XEE_OK;
}

// The launches of `total` iterations up to a residual check (or the end): their lengths, first to last.  The last one writes
// the residual when `to_check`.  A block that is not a whole number of launches starts with the short one: the residual-
// writing launch at its end then is a full-length one (at 4 iterations it was HBM-bound by its three stores per point: 113 us
// for 32 cases against 126 us for a plain 8-iteration launch; profiles/r02).
// With k_sweep_z (zT > 0): its launches of zT, then k_sweep_q's residual-writing launch, which takes the remainder along --
// every launch is a pass over all planes, ~75 us of memory traffic for 32 cases whatever its length (k_sweep_q: 82 us for
// 2 iterations, 126 for 8), so 15 x 6 + 10 beats 2 + 15 x 6 + 8 (profiles/r03/final_kernel_stats.csv).  Without a check at
// the end (the last block of a solve) the remainder goes first, never as a single iteration.
void plan_block(int total, bool to_check, int tb_T, int zT, std::vector<int>* plan)
{
    plan->clear();
    if (zT > 0 && tb_T > 1 && total >= (to_check ? tb_T : 0) + zT) {
        if (to_check && (total % zT == 0 || total % zT == 4)) {
            // k_sweep_z writes the residual too, after 6 or 4 iterations: the whole block is its (100 = 16 x 6 + 4)
            const int wr = total % zT == 0 ? zT : 4;
            for (int i = 0; i < (total - wr) / zT; ++i) plan->push_back(zT);
            plan->push_back(wr);
            return;
        }
        if (to_check) {
            int nz = (total - tb_T) / zT;
            int wr = total - nz * zT;                     // tb_T .. tb_T + zT - 1
            if (wr > tb_max_T()) { wr -= zT; nz += 1; }   // (tb_T above tb_max_T() - zT + 1: a shorter last launch instead, >= 2)
            if (wr >= 2) {
                for (int i = 0; i < nz; ++i) plan->push_back(zT);
                plan->push_back(wr);
                return;
            }
        } else {
            int nz = total / zT, rem = total % zT;
            if (rem == 1) { rem += zT; nz -= 1; }
            if (rem) plan->push_back(rem);
            for (int i = 0; i < nz; ++i) plan->push_back(zT);
            return;
        }
    }
    int left = total;
    while (left > 0) {
        const int rem = left % tb_T;
        const int n = rem ? rem : tb_T;
        plan->push_back(n);
        left -= n;
    }
}

// T iterations of the blocked sweep: from the four generating planes when assemble built them
// (fewer bytes), else from the nine coefficient planes (any coefficients, e.g. a Tier-1 caller's)
// returns true when the kernel itself also wrote the new iterate's interior into `snap` (k_sweep_q's residual-writing
// launch can; the other kernels leave the snapshot to a copy)
bool launch_blocked(xee_ensemble* e, const float* psi_in, float* psi_out, float* res, float alpha, int T,
                    const MemberState* st = nullptr /* default: e->st */, float* snap = nullptr)
{
    const int k = e->opts.kernel;
    if (!st) st = e->st;
    if (e->z_pairs_now > 0 && alpha == 1.0f && sweep_z_can(T, res != nullptr, e->base_baro)) {
        launch_sweep_z(e->g, psi_in, psi_out, e->base, e->f, res, res ? snap : nullptr, st, T, e->base_baro, e->z_pairs_now, e->opts.q_exact_only, e->redo,
                       &e->last_sweep, e->stream);
        return res != nullptr && snap != nullptr;           // (the residual-writing form also wrote the snapshot)
    }
    if (e->have_base && k != 1) {
        if (k == 2) launch_sweep_tb4(e->g, psi_in, psi_out, e->base, e->f, res, st, alpha, T, e->base_baro, e->opts, &e->last_sweep, e->stream);
        else {
            return launch_sweep_q(e->g, psi_in, psi_out, e->base, e->f, res, res ? snap : nullptr, st, alpha, T, e->base_baro, e->opts, e->redo, &e->last_sweep, e->stream);
        }
    }
    else
        launch_sweep_tb(e->g, psi_in, psi_out, e->coe, e->f, res, st, alpha, T, e->uv, e->opts, &e->last_sweep, e->stream);
    return false;
}

int check_launch(const char* what)
{
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) return fail(XEE_ERR_HIP, std::string(what) + ": " + hipGetErrorString(err));
    return XEE_OK;
}

}  // namespace

extern "C" {

int xee_version(void) { return XEE_VERSION; }

const char* xee_last_error(void) { return g_err.c_str(); }

int xee_device_count(int* count)
{
    if (!count) return fail(XEE_ERR_INVALID_ARGUMENT, "null count");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
    *count = n;
    return XEE_OK;
}

void xee_judge_error(int err)
{
    // xtt-lib-fortran/elliptic_tools.f90:333-358
    bool known = false;
    if (err == 0) { printf(" Elliptic Tools: Iteration success.\n"); known = true; }
    if (err & XEE_ERRBIT_OVER_MAX_ITERATION) { printf(" Elliptic Tools: [Error] Max iteration reached.\n"); known = true; }
    if (err & XEE_ERRBIT_EXPLODE) { printf(" Elliptic Tools: [Error] Iteration explodes.\n"); known = true; }
    if (!known) printf(" Elliptic Tools: Unknown error code %d\n", err);
    fflush(stdout);
}

// A13, src/diagnose/initialize-variables.f90:44-67 and read-input.f90:55-76 -- host arithmetic
// (x86-64 baseline fp32, glibc powf/cosf/sinf), constants of xtt-lib-fortran/constants.f90:4-7.
int xee_profiles(int geometry, int density_mode, const float* domain, int nr, int nz, float* ra, float* za,
                 float* rcuva, float* exner, float* rho, float* sin_table, float* dr_out, float* dz_out)
{
    if (!domain || !ra || !za || !rcuva || !exner || !rho || !dr_out || !dz_out || nr < 2 || nz < 2)
        return fail(XEE_ERR_INVALID_ARGUMENT, "xee_profiles: bad argument");
    if (geometry != 0 && geometry != 1) return fail(XEE_ERR_INVALID_ARGUMENT, "geometry must be 0 or 1");
    const float g0 = 9.8f, theta0 = 298.0f, Rd = 287.0f, p0 = 101300.0f;
    const float Cv = 5.0f / 2.0f * Rd, Cp = Cv + Rd, kappa = Rd / Cp, h0 = Cp * theta0 / g0;
    float Lr1, Lr2, Lz1, Lz2, planet_radius = 0.0f;
    const float Lat1 = -90.0f, Lat2 = 90.0f;
    if (geometry == 0) {
        Lr1 = domain[0]; Lr2 = domain[1]; Lz1 = domain[2]; Lz2 = domain[3];
    } else {
        const float MATH_PI = acosf(-1.0f);                     // variables.f90:4
        const float DEG2RAD = MATH_PI / 180.0f;
        planet_radius = domain[0]; Lz1 = domain[1]; Lz2 = domain[2];
        Lr1 = Lat1 * DEG2RAD * planet_radius;                   // read-input.f90:69-70
        Lr2 = Lat2 * DEG2RAD * planet_radius;
    }
    const float dr = (Lr2 - Lr1) / (float)(nr - 1);
    const float dz = (Lz2 - Lz1) / (float)(nz - 1);
    for (int i = 0; i < nr; ++i) ra[i] = Lr1 + (float)i * dr;
    const float expo = 1.0f / kappa - 1.0f, pref = p0 / (theta0 * Rd);
    for (int j = 0; j < nz; ++j) {
        za[j] = Lz1 + (float)j * dz;
        if (density_mode == 0) { exner[j] = 1.0f - za[j] / h0; rho[j] = pref * powf(exner[j], expo); }
        else { exner[j] = 1.0f; rho[j] = 1.0f; }
    }
    if (geometry == 0) {
        for (int i = 0; i < nr; ++i) rcuva[i] = ra[i];
    } else {
        const float dlat = (Lat2 - Lat1) / (float)(nr - 1);
        for (int i = 0; i < nr; ++i) {
            const float lat = Lat1 + (float)i * dlat;
            rcuva[i] = planet_radius * cosf(lat);
            if (sin_table) sin_table[i] = sinf(lat);
        }
    }
    *dr_out = dr;
    *dz_out = dz;
    return XEE_OK;
}

// ---------------------------------------------------------------------------------------------
// Tier 2
// ---------------------------------------------------------------------------------------------

int xee_ensemble_create(xee_ensemble_t** out, int nr, int nz, int n_members, int device)
{
    if (!out) return fail(XEE_ERR_INVALID_ARGUMENT, "null out pointer");
    *out = nullptr;
    if (nr < 3 || nz < 3 || n_members < 1) return fail(XEE_ERR_INVALID_ARGUMENT, "need nr>=3, nz>=3, members>=1");
    XCHK(require_device());
    if (device < 0) HIPCHK(hipGetDevice(&device));
    HIPCHK(hipSetDevice(device));
    std::call_once(g_setup_once, [] {
        g_setup_status = sweep_kernel_setup();
        if (g_setup_status == 0) g_setup_status = exact_kernel_setup();
    });
    if (g_setup_status != 0) return fail(XEE_ERR_HIP, std::string("kernel setup: ") + hipGetErrorString((hipError_t)g_setup_status));
    xee_ensemble* e = new xee_ensemble();
    g_n_ensembles.fetch_add(1);
    e->device = device;
    e->g.nx = nr;
    e->g.ny = nz;
    e->g.pitch = ((nr + 63) / 64) * 64;
    e->g.members = n_members;
    e->g.plane = (size_t)e->g.pitch * nz;
    const Defaults dflt = read_defaults();
    e->opts = dflt.sweep;
    e->sync_checks = dflt.sync_checks != 0;
    e->exact_serial = dflt.exact_sum_serial != 0;
    int st = XEE_OK;
    auto guard = [&](int s) { if (st == XEE_OK) st = s; };
    if (create_stream(&e->stream, true) != hipSuccess) { delete e; return fail(XEE_ERR_HIP, "hipStreamCreate failed"); }
    e->own_stream = true;
    for (auto& ev : e->ev) if (hipEventCreate(&ev) != hipSuccess) guard(fail(XEE_ERR_HIP, "hipEventCreate failed"));
    // planes every solve needs
    guard(alloc_planes(e, &e->bc));
    guard(alloc_planes(e, &e->f));
    guard(alloc_planes(e, &e->psi[0]));
    guard(alloc_planes(e, &e->psi[1]));
    guard(alloc_planes(e, &e->coe, 9));
    guard(alloc_planes(e, &e->res));
    if (st == XEE_OK && hipMalloc(reinterpret_cast<void**>(&e->redo), 2 * sizeof(int)) != hipSuccess) guard(fail(XEE_ERR_ALLOC, "hipMalloc counters"));
    if (st == XEE_OK && hipMemsetAsync(e->redo, 0, 2 * sizeof(int), e->stream) != hipSuccess) guard(fail(XEE_ERR_HIP, "memset counters"));
    if (st == XEE_OK && hipMalloc(reinterpret_cast<void**>(&e->st), sizeof(MemberState) * n_members) != hipSuccess) guard(fail(XEE_ERR_ALLOC, "hipMalloc state"));
    if (st == XEE_OK && hipMalloc(reinterpret_cast<void**>(&e->sums), sizeof(float) * n_members) != hipSuccess) guard(fail(XEE_ERR_ALLOC, "hipMalloc sums"));
    if (st == XEE_OK && hipHostMalloc(reinterpret_cast<void**>(&e->st_host), sizeof(MemberState) * n_members) != hipSuccess) guard(fail(XEE_ERR_ALLOC, "hipHostMalloc state"));
    if (st == XEE_OK && hipHostMalloc(reinterpret_cast<void**>(&e->sums_host), sizeof(float) * n_members) != hipSuccess) guard(fail(XEE_ERR_ALLOC, "hipHostMalloc sums"));
    if (st == XEE_OK && hipMalloc(&e->ex_ws, exact_ws_bytes(e->g)) != hipSuccess) guard(fail(XEE_ERR_ALLOC, "hipMalloc exact-sum workspace"));
    if (st == XEE_OK && hipMalloc(reinterpret_cast<void**>(&e->ex_slow), sizeof(int) * n_members) != hipSuccess) guard(fail(XEE_ERR_ALLOC, "hipMalloc slow counters"));
    const size_t nv = (size_t)(nr > nz ? nr : nz) + 64;
    for (float** v : {&e->ra, &e->za, &e->rcuva, &e->rho, &e->exner})
        if (st == XEE_OK && hipMalloc(reinterpret_cast<void**>(v), nv * sizeof(float)) != hipSuccess) guard(fail(XEE_ERR_ALLOC, "hipMalloc profile"));
    if (st != XEE_OK) { xee_ensemble_destroy(e); return st; }
    launch_state_init(e->g, e->st, e->stream);
    e->rec.assign(n_members, xee_case_record_t{});
    for (int m = 0; m < n_members; ++m) e->rec[m].case_id = m;
    if (hipStreamSynchronize(e->stream) != hipSuccess) { xee_ensemble_destroy(e); return fail(XEE_ERR_HIP, "ensemble init failed"); }
    *out = e;
    return XEE_OK;
}

int xee_ensemble_destroy(xee_ensemble_t* e)
{
    if (!e) return XEE_OK;
    hipSetDevice(e->device);
    if (e->stream) hipStreamSynchronize(e->stream);
    for (auto& h : e->half) if (h) { xee_ensemble_destroy(h); h = nullptr; }
    if (e->is_view) {
        // what a view does not own: the planes of the ensemble it looks into, its member states, counters and host arrays
        e->A = e->B = e->C = e->bc = e->f = e->psi[0] = e->psi[1] = e->sA = e->sB = e->sC = e->coe = e->res = nullptr;
        e->eta = e->u = e->w = e->Q = e->terms = e->ra = e->za = e->rcuva = e->rho = e->exner = nullptr;
        e->base = nullptr; e->st = nullptr; e->st_host = nullptr; e->ex_slow = nullptr; e->redo = nullptr; e->uv_differs = nullptr;
        e->sums_host = nullptr; e->sx_ws = nullptr;
    }
    for (float* p : {e->A, e->B, e->C, e->bc, e->f, e->psi[0], e->psi[1], e->sA, e->sB, e->sC, e->coe, e->res,
                     e->eta, e->u, e->w, e->Q, e->terms, e->ra, e->za, e->rcuva, e->rho, e->exner, e->sums})
        if (p) hipFree(p);
    if (e->st) hipFree(e->st);
    if (e->ex_ws) hipFree(e->ex_ws);
    if (e->sx_ws) hipFree(e->sx_ws);
    if (e->ex_slow) hipFree(e->ex_slow);
    if (e->uv_differs) hipFree(e->uv_differs);
    if (e->base) hipFree(e->base);
    for (auto& c : e->cstream) if (c) { hipStreamSynchronize(c); hipStreamDestroy(c); }
    for (auto& ev : e->evE) if (ev) hipEventDestroy(ev);
    for (auto& ev : e->evF) if (ev) hipEventDestroy(ev);
    for (auto& pr : e->evB) for (auto& ev : pr) if (ev) hipEventDestroy(ev);
    for (auto& pr : e->evZ) for (auto& ev : pr) if (ev) hipEventDestroy(ev);
    if (e->res2) hipFree(e->res2);
    if (e->res3) hipFree(e->res3);
    if (e->ex_ws2) hipFree(e->ex_ws2);
    if (e->sums2) hipFree(e->sums2);
    if (e->ex_slow2) hipFree(e->ex_slow2);
    if (e->snap) hipFree(e->snap);
    if (e->st_host2) hipHostFree(e->st_host2);
    if (e->st_view) hipFree(e->st_view);
    if (e->redo) hipFree(e->redo);
    if (e->zstat) hipFree(e->zstat);
    if (e->shared_differs) hipFree(e->shared_differs);
    for (auto& kv : e->stash) if (kv.second) hipFree(kv.second);
    if (e->aos_stage) hipFree(e->aos_stage);
    if (e->gres) hipFree(e->gres);
    if (e->gsnap) hipFree(e->gsnap);
    if (e->gst_host) hipHostFree(e->gst_host);
    if (e->gst_view) hipFree(e->gst_view);
    for (auto& ev : e->gevE) if (ev) hipEventDestroy(ev);
    for (auto& ev : e->gevF) if (ev) hipEventDestroy(ev);
    if (e->st_host) hipHostFree(e->st_host);
    if (e->sums_host) hipHostFree(e->sums_host);
    for (auto& ev : e->ev) if (ev) hipEventDestroy(ev);
    if (e->own_stream && e->stream) hipStreamDestroy(e->stream);
    delete e;
    return XEE_OK;
}

int xee_ensemble_set_stream(xee_ensemble_t* e, void* hip_stream)
{
    XCHK(use(e));
    HIPCHK(hipStreamSynchronize(e->stream));
    if (e->own_stream && e->stream) { hipStreamDestroy(e->stream); e->own_stream = false; }
    if (hip_stream) {
        e->stream = reinterpret_cast<hipStream_t>(hip_stream);
    } else {
        HIPCHK(create_stream(&e->stream, true));
        e->own_stream = true;
    }
    return XEE_OK;
}

int xee_ensemble_set_case_ids(xee_ensemble_t* e, int first_case_id)
{
    if (!e) return fail(XEE_ERR_INVALID_ARGUMENT, "null ensemble");
    e->first_case_id = first_case_id;
    for (int m = 0; m < e->g.members; ++m) e->rec[m].case_id = first_case_id + m;
    return XEE_OK;
}

int xee_set_default_option(const char* key, int value)
{
    std::lock_guard<std::mutex> lk(g_defaults_mutex);
    return set_option_in(g_defaults.sweep, g_defaults.sync_checks, g_defaults.exact_sum_serial, key, value);
}

int xee_reset_default_options(void)
{
    std::lock_guard<std::mutex> lk(g_defaults_mutex);
    g_defaults = Defaults{};
    return XEE_OK;
}

int xee_debug_counters(int64_t* out /* 4 */)
{
    if (!out) return fail(XEE_ERR_INVALID_ARGUMENT, "null out");
    out[0] = g_n_ensembles.load(); out[1] = g_n_tier1_stage_allocs.load(); out[2] = g_n_thread_exit_releases.load(); out[3] = 0;
    return XEE_OK;
}

int xee_ensemble_set_option(xee_ensemble_t* e, const char* key, int value)
{
    if (!e) return fail(XEE_ERR_INVALID_ARGUMENT, "null ensemble");
    int sc = e->sync_checks ? 1 : 0, xs = e->exact_serial ? 1 : 0;
    XCHK(set_option_in(e->opts, sc, xs, key, value));
    e->base_shared_known = false;                       // ("share_planes" may have changed)
    e->sync_checks = sc != 0;
    e->exact_serial = xs != 0;
    return XEE_OK;
}

int xee_ensemble_sweep_info(xee_ensemble_t* e, char* name, int name_len, int* T, int* tile_w, int* tile_h)
{
    if (!e) return fail(XEE_ERR_INVALID_ARGUMENT, "null ensemble");
    if (e->opts.T == 1) e->last_sweep = SweepDesc{"k_sweep_stream", 1, e->g.nx, e->g.ny};   // one iteration per launch, no tiles
    if (name && name_len > 0) { strncpy(name, e->last_sweep.name, (size_t)name_len - 1); name[name_len - 1] = 0; }
    if (T) *T = e->z_pairs_now > 0 ? e->z_T_now : e->opts.T;   // the configured launch length (k_sweep_z: its own; the last launch of a block may be shorter)
    if (tile_w) *tile_w = e->last_sweep.tile_w;
    if (tile_h) *tile_h = e->last_sweep.tile_h;
    return XEE_OK;
}

int xee_ensemble_set_profiles(xee_ensemble_t* e, const float* ra, const float* za, const float* rcuva,
                              const float* rho, const float* exner, float dr, float dz)
{
    XCHK(use(e));
    if (!ra || !za || !rcuva || !rho || !exner) return fail(XEE_ERR_INVALID_ARGUMENT, "null profile");
    const size_t br = sizeof(float) * e->g.nx, bz = sizeof(float) * e->g.ny;
    HIPCHK(hipMemcpyAsync(e->ra, ra, br, hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipMemcpyAsync(e->rcuva, rcuva, br, hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipMemcpyAsync(e->za, za, bz, hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipMemcpyAsync(e->rho, rho, bz, hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipMemcpyAsync(e->exner, exner, bz, hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    e->dr = dr;
    e->dz = dz;
    e->have_profiles = true;
    return XEE_OK;
}

static int xfer(xee_ensemble_t* e, int field, int first, int count, float* host, bool up)
{
    XCHK(use(e));
    if (!host || first < 0 || count < 1 || first + count > e->g.members)
        return fail(XEE_ERR_INVALID_ARGUMENT, "bad member range or null host pointer");
    float* base = nullptr;
    int n1 = 0, n2 = 0;
    XCHK(field_ref(e, field, &base, &n1, &n2));
    if (up && field >= XEE_F_COE0) { e->uv.mask = 0; e->have_base = false; }   // planes changed behind assemble's back: assume nothing
    for (int k = 0; k < count; ++k) {
        float* d = base + (size_t)(first + k) * e->g.plane;
        float* h = host + (size_t)k * n1 * n2;
        if (up)
            HIPCHK(hipMemcpy2DAsync(d, sizeof(float) * e->g.pitch, h, sizeof(float) * n1, sizeof(float) * n1, n2, hipMemcpyHostToDevice, e->stream));
        else
            HIPCHK(hipMemcpy2DAsync(h, sizeof(float) * n1, d, sizeof(float) * e->g.pitch, sizeof(float) * n1, n2, hipMemcpyDeviceToHost, e->stream));
    }
    HIPCHK(hipStreamSynchronize(e->stream));
    return XEE_OK;
}

int xee_ensemble_upload(xee_ensemble_t* e, int field, int first, int count, const float* host)
{
    return xfer(e, field, first, count, const_cast<float*>(host), true);
}

int xee_ensemble_download(xee_ensemble_t* e, int field, int first, int count, float* host)
{
    return xfer(e, field, first, count, host, false);
}

int xee_ensemble_device_plane(xee_ensemble_t* e, int field, void** dptr, int* pitch, size_t* member_stride,
                              int* n1, int* n2)
{
    XCHK(use(e));
    float* base = nullptr;
    int a = 0, b = 0;
    XCHK(field_ref(e, field, &base, &a, &b));
    HIPCHK(hipStreamSynchronize(e->stream));   // the zero-fill of a fresh plane has landed
    if (dptr) *dptr = base;
    if (pitch) *pitch = e->g.pitch;
    if (member_stride) *member_stride = e->g.plane;
    if (n1) *n1 = a;
    if (n2) *n2 = b;
    return XEE_OK;
}

}  // extern "C"

namespace {
// the grid of the current member window and the float offset of its first member's plane
Grid window_grid(const xee_ensemble* e, size_t* off)
{
    Grid g = e->g;
    *off = 0;
    if (e->win_count > 0) { g.members = e->win_count; *off = (size_t)e->win_first * e->g.plane; }
    return g;
}
}  // namespace

extern "C" {

int xee_ensemble_member_window(xee_ensemble_t* e, int first, int count)
{
    if (!e) return fail(XEE_ERR_INVALID_ARGUMENT, "null ensemble");
    if (count == 0) { e->win_first = 0; e->win_count = 0; return XEE_OK; }
    if (first < 0 || count < 0 || first + count > e->g.members) return fail(XEE_ERR_INVALID_ARGUMENT, "member_window: range outside the ensemble");
    e->win_first = first; e->win_count = count;
    return XEE_OK;
}

int xee_ensemble_forcing_from_B(xee_ensemble_t* e)
{
    XCHK(use(e));
    if (!e->B) return fail(XEE_ERR_INVALID_ARGUMENT, "B has not been uploaded");
    size_t off;
    const Grid g = window_grid(e, &off);
    launch_negate(g, e->B + off, e->f + off, e->stream);
    return check_launch("negate");
}

int xee_ensemble_forcing_thermal(xee_ensemble_t* e)
{
    XCHK(use(e));
    if (!e->have_profiles) return fail(XEE_ERR_INVALID_ARGUMENT, "profiles not set");
    if (!e->Q) return fail(XEE_ERR_INVALID_ARGUMENT, "Q has not been uploaded");
    XCHK(alloc_planes(e, &e->terms));
    size_t off;
    const Grid g = window_grid(e, &off);
    launch_rhs_thermal(g, e->Q + off, e->ra, e->exner, e->terms + off, e->f + off, e->stream);
    return check_launch("forcing_thermal");
}

int xee_ensemble_forcing_zero(xee_ensemble_t* e)
{
    XCHK(use(e));
    size_t off;
    const Grid g = window_grid(e, &off);
    HIPCHK(hipMemsetAsync(e->f + off, 0, sizeof(float) * g.plane * g.members, e->stream));
    return XEE_OK;
}

int xee_ensemble_field_save(xee_ensemble_t* e, int field, int slot)
{
    XCHK(use(e));
    if (field < 0 || field >= XEE_F_COE0 || slot < 0 || slot > 63) return fail(XEE_ERR_INVALID_ARGUMENT, "field_save: field below XEE_F_COE0, slot 0..63");
    float* src = nullptr;
    int n1 = 0, n2 = 0;
    XCHK(field_ref(e, field, &src, &n1, &n2));
    float*& dst = e->stash[field * 64 + slot];
    const size_t bytes = sizeof(float) * e->g.plane * e->g.members;
    if (!dst) {
        if (hipMalloc(reinterpret_cast<void**>(&dst), bytes) != hipSuccess) { dst = nullptr; return fail(XEE_ERR_ALLOC, "hipMalloc stash slot"); }
        HIPCHK(hipMemsetAsync(dst, 0, bytes, e->stream));
    }
    size_t off;
    const Grid g = window_grid(e, &off);
    HIPCHK(hipMemcpyAsync(dst + off, src + off, sizeof(float) * g.plane * g.members, hipMemcpyDeviceToDevice, e->stream));
    return XEE_OK;
}

int xee_ensemble_field_restore(xee_ensemble_t* e, int field, int slot)
{
    XCHK(use(e));
    if (field < 0 || field >= XEE_F_COE0 || slot < 0 || slot > 63) return fail(XEE_ERR_INVALID_ARGUMENT, "field_restore: field below XEE_F_COE0, slot 0..63");
    auto it = e->stash.find(field * 64 + slot);
    if (it == e->stash.end() || !it->second) return fail(XEE_ERR_INVALID_ARGUMENT, "field_restore: nothing was saved in this slot");
    float* dst = nullptr;
    int n1 = 0, n2 = 0;
    XCHK(field_ref(e, field, &dst, &n1, &n2));
    size_t off;
    const Grid g = window_grid(e, &off);
    HIPCHK(hipMemcpyAsync(dst + off, it->second + off, sizeof(float) * g.plane * g.members, hipMemcpyDeviceToDevice, e->stream));
    return XEE_OK;
}

int xee_ensemble_assemble(xee_ensemble_t* e, int variant)
{
    XCHK(use(e));
    if (!e->have_profiles) return fail(XEE_ERR_INVALID_ARGUMENT, "profiles not set");
    if (!e->A || !e->B || !e->C) return fail(XEE_ERR_INVALID_ARGUMENT, "A, B, C must be uploaded first");
    if (variant != XEE_BAROTROPIC && variant != XEE_BAROCLINIC) return fail(XEE_ERR_INVALID_ARGUMENT, "bad variant");
    XCHK(alloc_planes(e, &e->sA));
    XCHK(alloc_planes(e, &e->sB));
    XCHK(alloc_planes(e, &e->sC));
    launch_solver_coef(e->g, e->A, e->B, e->C, e->rcuva, e->rho, e->sA, e->sB, e->sC, e->stream);
    launch_cal_coe(e->g, e->sA, variant == XEE_BAROCLINIC ? e->sB : nullptr, e->sC, e->coe, e->dr, e->dz, e->stream);
    XCHK(check_launch("assemble"));
    XCHK(detect_uniform_planes(e));
    XCHK(alloc_planes(e, &e->base, 5));
    e->g.two_pq4 = 2.0f * (4.0f * e->dr * e->dz);
    launch_base_planes(e->g, e->sA, variant == XEE_BAROCLINIC ? e->sB : nullptr, e->sC, e->base, e->dr, e->dz, e->stream);
    XCHK(check_launch("base planes"));
    e->have_base = true;
    e->base_shared_known = false; e->g.base_shared = 0;
    e->base_baro = (variant == XEE_BAROTROPIC);
    return XEE_OK;
}

int xee_ensemble_assemble_solver(xee_ensemble_t* e, int use_b)
{
    XCHK(use(e));
    if (!e->sA || !e->sC || (use_b && !e->sB)) return fail(XEE_ERR_INVALID_ARGUMENT, "solver coefficients have not been uploaded / assembled");
    launch_cal_coe(e->g, e->sA, use_b ? e->sB : nullptr, e->sC, e->coe, e->dr, e->dz, e->stream);
    XCHK(check_launch("assemble_solver"));
    XCHK(detect_uniform_planes(e));
    XCHK(alloc_planes(e, &e->base, 5));
    e->g.two_pq4 = 2.0f * (4.0f * e->dr * e->dz);
    launch_base_planes(e->g, e->sA, use_b ? e->sB : nullptr, e->sC, e->base, e->dr, e->dz, e->stream);
    XCHK(check_launch("base planes"));
    e->have_base = true;
    e->base_shared_known = false; e->g.base_shared = 0;
    e->base_baro = !use_b;
    return XEE_OK;
}

// solve_elliptic, xtt-lib-fortran/elliptic_tools.f90:93-265, for every member at once.
}  // extern "C"

namespace {

// solve_elliptic with the residual checks PIPELINED behind the sweeps.  The check of iteration k
// (exact norm + state machine) runs on a check stream while the next block(s) of sweeps already run on
// e->stream; a member found to have stopped at check k has, by then, been iterated further, so the
// iterate right after every check iteration is kept in a snapshot plane (one per slot)
// and its result is taken from there.  Sweeps skip a member as soon as they are shown a view of the member states
// in which its `active` flag is cleared.  Same values as the synchronous driver (tests run both).
//
// Depth.  A large batch keeps the host ONE check behind the sweeps (two slots, one check stream): its block of sweeps
// takes milliseconds and the check fits in its shadow.  For a small batch the check -- eight dependent, latency-bound
// kernels, ~165 us for one 1024x512 case, 80 of them the serial walk -- takes LONGER than the block of sweeps it hides
// behind (~125 us), and the sweep stream stood idle waiting for the host (profiles/r02, timeline of the `single`
// workload).  There the host stays TWO checks behind (three slots) and consecutive checks alternate between two
// check streams with their own workspaces, so that norm k overlaps norm k-1; only the state machine is chained
// (check k's update waits for check k-1's).
//
// What the sweep stream carries per check, besides sweeps, decides a small batch's speed (each event record or copy
// is a packet of ~5 us between 9 us kernels), so: the state machine itself publishes the member states (a device view
// per check for the sweeps, a pinned host copy for this loop -- no copy kernels); a slot is reused only after the HOST
// has seen its previous check finish (no stream wait needed); and the per-block begin/end events that time the sweeps
// are recorded for large batches only -- a small batch's `sweep_ms` is the span of the sweep stream from the first
// launch to the last (it carries nothing but sweeps).
int solve_pipelined(xee_ensemble* e, int max_iter, int cs, ConvParams cp, float alpha, int tb_T)
{
    const Grid& g = e->g;
    hipStream_t s = e->stream;
    // The norm's wide kernels (chunk sums, scans, summaries: ~75 us for 32 cases) stay on the sweep stream for a large
    // batch -- beside the sweeps they took compute units from every launch of the next block (+12 % per sweep launch,
    // profiles/r02) -- and only the serial walk (one wave per member) and the state machine run in the shadow of the
    // sweeps.  A small batch leaves the machine half empty anyway: there the whole norm runs beside the sweeps.
    // (the depth follows from the batch size alone; with option exact_sum_serial a large batch keeps depth 1 and its per-block
    // events, only the norm -- then the serial reference-order walk -- runs whole on the check stream)
    const bool large = (size_t)g.members * g.nx * g.ny >= ((size_t)4 << 20);
    const bool wide_on_sweep_stream = large && !e->exact_serial;
    const int D = large ? 1 : 2;                  // checks in flight behind the sweeps
    const int NS = D + 1;                         // slots (residual plane, snapshot, events, host copy of the states)
    constexpr int NV = 4;                         // views of the member states (see below)
    const bool block_events = large;
    if (!e->cstream[0]) HIPCHK(create_stream(&e->cstream[0], false));
    if (!e->evE[0]) {
        for (auto& ev : e->evE) HIPCHK(hipEventCreate(&ev));
        for (auto& ev : e->evF) HIPCHK(hipEventCreate(&ev));
        for (auto& pr : e->evB) for (auto& ev : pr) HIPCHK(hipEventCreate(&ev));
        for (auto& pr : e->evZ) for (auto& ev : pr) HIPCHK(hipEventCreate(&ev));
        HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&e->st_host2), sizeof(MemberState) * 3 * g.members));
        HIPCHK(hipMalloc(reinterpret_cast<void**>(&e->st_view), sizeof(MemberState) * NV * g.members));
    }
    if (D == 2 && !e->cstream[1]) {
        HIPCHK(create_stream(&e->cstream[1], false));
        HIPCHK(hipMalloc(&e->ex_ws2, exact_ws_bytes(g)));
        HIPCHK(hipMalloc(reinterpret_cast<void**>(&e->sums2), sizeof(float) * g.members));
        HIPCHK(hipMalloc(reinterpret_cast<void**>(&e->ex_slow2), sizeof(int) * g.members));
    }
    // Views.  Check j's state machine writes view j % NV.  The sweeps start on view NV-1 (a copy of the initial states)
    // and, at the head of the block that ends in check k, move to the view of check k-1-D: the newest check the host has
    // digested, so it is complete and no stream wait is needed.  Until the next move that view is only read: the state
    // machines that may run meanwhile are those of checks k-D .. k, which write other views (NV = 4 > D + 1).
    // (a stale view only costs work: a member found stopped keeps iterating until the sweeps see it and its result is
    // taken from the snapshot of its stopping check anyway)
    for (int v = 0; v < NV; ++v)
        HIPCHK(hipMemcpyAsync(e->st_view + (size_t)v * g.members, e->st, sizeof(MemberState) * g.members, hipMemcpyDeviceToDevice, s));
    const MemberState* view = e->st_view + (size_t)(NV - 1) * g.members;
    XCHK(alloc_planes(e, &e->res2));
    if (D == 2) XCHK(alloc_planes(e, &e->res3));
    XCHK(alloc_planes(e, &e->snap, NS));
    e->snap_planes = NS;
    hipStream_t cst[2] = {e->cstream[0], e->cstream[D - 1]};
    void* wsb[2] = {e->ex_ws, D == 2 ? e->ex_ws2 : e->ex_ws};
    float* sumb[2] = {e->sums, D == 2 ? e->sums2 : e->sums};
    int* slowb[2] = {e->ex_slow, D == 2 ? e->ex_slow2 : e->ex_slow};
    float* resbuf[3] = {e->res, e->res2, e->res3};
    float* snapbuf[3] = {e->snap, e->snap + (size_t)g.members * g.plane, e->snap + 2 * (size_t)g.members * g.plane};
    // every snapshot slot starts as a full copy of the first guess: the rim never changes, and a residual-writing sweep
    // launch that fills the snapshot itself only stores interior points
    launch_copy2(g, e->psi[e->cur], snapbuf[0], snapbuf[1], s);
    if (D == 2) launch_copy2(g, e->psi[e->cur], snapbuf[2], nullptr, s);

    int cur = e->cur;
    std::vector<int> stop_slot(g.members, -1);    // snapshot slot holding a stopped member's result
    std::vector<char> active(g.members, 1);
    e->timing = xee_timing_t{};
    e->timing.sweep_streams = 1;
    if (!e->is_view) HIPCHK(hipMemsetAsync(e->redo, 0, 2 * sizeof(int), s));   // the solve's diagnostic counters (a view: its parent did)
    HIPCHK(hipEventRecord(e->ev[0], s));
    int cnt = 0, k = 0;                           // k = number of checks enqueued
    int digested = 0;                             // ... and of checks whose outcome the host has looked at
    int final_check = -1;                         // index of a check that coincides with max_iter
    bool all_done = (max_iter == 0), block_open = false, sweeps_end_recorded = false;

    int z_in_slot[3] = {0, 0, 0};                         // k_sweep_z launches between a slot's evZ pair
    // digest the results of the checks up to j, in order (check j's pipeline has finished when its slot's evF has)
    auto digest_upto = [&](int j) -> int {
        for (; digested <= j; ++digested) {
            const int sl = digested % NS;
            HIPCHK(hipEventSynchronize(e->evF[sl]));
            if (block_events) {
                float ms = 0.f;
                HIPCHK(hipEventElapsedTime(&ms, e->evB[sl][0], e->evB[sl][1]));
                e->timing.sweep_ms += ms;
                if (z_in_slot[sl] > 0) {
                    HIPCHK(hipEventElapsedTime(&ms, e->evZ[sl][0], e->evZ[sl][1]));
                    e->timing.z_ms += ms;
                    e->timing.z_timed_launches += z_in_slot[sl];
                }
            }
            const MemberState* sh = e->st_host2 + (size_t)sl * g.members;
            all_done = true;
            for (int m = 0; m < g.members; ++m) {
                if (active[m] && !sh[m].active) { active[m] = 0; stop_slot[m] = sl; }
                if (sh[m].active) all_done = false;
            }
            memcpy(e->st_host, sh, sizeof(MemberState) * g.members);
        }
        return XEE_OK;
    };

    // Graphs.  A small batch is bound by how fast the host enqueues (one 1024x512 case: 13 launches of ~9 us per block
    // and eight more per check): a whole block -- the launches between two checks -- is captured once per distinct set
    // of arguments (ping-pong phase x slot x view: at most 2 x 3 x 4) and replayed as a graph.
    struct BlockGraph { hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr; bool snapped = false; int launches = 0, z_launches = 0; long long tiles = 0; };
    struct GraphCache {
        std::map<int, BlockGraph> m;
        ~GraphCache() { for (auto& kv : m) { if (kv.second.exec) hipGraphExecDestroy(kv.second.exec); if (kv.second.graph) hipGraphDestroy(kv.second.graph); } }
    } graphs;
    // (not on the legacy default stream, which cannot be captured; any capture failure falls back to plain launches)
    bool use_graphs = !large && tb_T > 1 && cs > tb_T && max_iter / cs >= 64 && s != nullptr;
    // the launches from iteration `from` up to `target` (a check iteration gets the residual plane and the snapshot)
    std::vector<int> plan;
    int z_enqueued = 0;                                   // k_sweep_z launches among those of the last enqueue_sweeps
    auto enqueue_sweeps = [&](int from, int target, bool to_check, int sl, int* cur_io, bool* snapped, int* launches, long long* tiles) {
        *snapped = false; *launches = 0; *tiles = 0; z_enqueued = 0;
        plan_block(target - from, to_check, tb_T, e->z_pairs_now > 0 ? e->z_T_now : 0, &plan);
        // (large batches: the block's run of k_sweep_z launches between its own pair of events -- xee_timing_t.z_ms)
        const bool time_z = block_events && to_check;
        bool z_open = false;
        for (size_t i = 0; i < plan.size(); ++i) {
            const int n = plan[i];
            const bool check = to_check && i + 1 == plan.size();
            const bool will_stream = e->z_pairs_now > 0 && alpha == 1.0f && sweep_z_can(n, check, e->base_baro);
            if (time_z && will_stream && !z_open) { hipEventRecord(e->evZ[sl][0], s); z_open = true; }
            if (z_open && !will_stream) { hipEventRecord(e->evZ[sl][1], s); z_open = false; }
            e->last_sweep.tiles = 0; e->last_sweep.streamed = false;
            if (n == 1)
                launch_sweep(g, e->psi[*cur_io], e->psi[*cur_io ^ 1], e->coe, e->f, check ? resbuf[sl] : nullptr, view, alpha, s);
            else if (launch_blocked(e, e->psi[*cur_io], e->psi[*cur_io ^ 1], check ? resbuf[sl] : nullptr, alpha, n, view,
                                    check ? snapbuf[sl] : nullptr))
                *snapped = true;
            *cur_io ^= 1;
            *launches += 1;
            *tiles += e->last_sweep.tiles;
            if (e->last_sweep.streamed) z_enqueued += 1;
        }
        if (z_open) hipEventRecord(e->evZ[sl][1], s);
        if (time_z) z_in_slot[sl] = z_enqueued;
        if (to_check && !*snapped) launch_copy_active(g, e->psi[*cur_io], snapbuf[sl], view, s);   // the iterate this check belongs to
    };

    while (cnt < max_iter && !all_done) {
        const int next_check = (cnt / cs + 1) * cs;
        const int target = next_check < max_iter ? next_check : max_iter;
        const bool check = target == next_check, last = target == max_iter;
        const int sl = k % NS, ci = k % D;
        if (block_events && !block_open) { HIPCHK(hipEventRecord(e->evB[sl][0], s)); block_open = true; }
        // slot sl last belonged to check k-NS = k-1-D, which the host digested at the end of the previous check's
        // iteration of this loop: it is free, and its check's view is the newest one that is safe to show the sweeps
        const int vi = k >= NS ? (k - NS) % NV : NV - 1;
        view = e->st_view + (size_t)vi * g.members;
        bool snapped = false;
        int launches = 0;
        long long tiles = 0;
        if (use_graphs && check && cnt % cs == 0) {
            const int key = cur | (sl << 1) | (vi << 3);
            BlockGraph& bg = graphs.m[key];
            if (!bg.exec) {
                int c2 = cur;
                if (hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed) == hipSuccess) {
                    enqueue_sweeps(cnt, target, true, sl, &c2, &bg.snapped, &bg.launches, &bg.tiles);
                    bg.z_launches = z_enqueued;
                    if (hipStreamEndCapture(s, &bg.graph) != hipSuccess || hipGraphInstantiate(&bg.exec, bg.graph, nullptr, nullptr, 0) != hipSuccess)
                        bg.exec = nullptr;
                }
                if (!bg.exec) { (void)hipGetLastError(); use_graphs = false; }
            }
        }
        if (use_graphs && check && cnt % cs == 0) {
            BlockGraph& bg = graphs.m[cur | (sl << 1) | (vi << 3)];
            HIPCHK(hipGraphLaunch(bg.exec, s));
            snapped = bg.snapped; launches = bg.launches; tiles = bg.tiles; z_enqueued = bg.z_launches;
            cur ^= (launches & 1);
        } else {
            enqueue_sweeps(cnt, target, check, sl, &cur, &snapped, &launches, &tiles);
        }
        e->timing.z_launches += z_enqueued;
        e->timing.sweep_launches += launches;
        e->timing.tiles += tiles;
        e->timing.iterations += target - cnt;
        cnt = target;
        if (check) {
            hipStream_t c = cst[ci];
            if (block_events) { HIPCHK(hipEventRecord(e->evB[sl][1], s)); block_open = false; }
            if (wide_on_sweep_stream) {
                // the workspace is shared between checks: check k-1 must have finished with it
                if (k >= 1) HIPCHK(hipStreamWaitEvent(s, e->evF[(k - 1) % NS], 0));
                launch_exact_sumsq_prepare(g, resbuf[sl], wsb[ci], view, s);
            }
            HIPCHK(hipEventRecord(e->evE[sl], s));
            HIPCHK(hipStreamWaitEvent(c, e->evE[sl], 0));
            if (e->exact_serial) launch_seqsum(g, resbuf[sl], 1, g.nx - 1, 1, g.ny - 1, true, sumb[ci], e->st, c);
            else if (wide_on_sweep_stream) launch_exact_sumsq_walk(g, resbuf[sl], wsb[ci], sumb[ci], slowb[ci], e->st, c);
            else launch_exact_sumsq(g, resbuf[sl], wsb[ci], sumb[ci], slowb[ci], e->st, c);
            // the state machine is sequential in the checks: at depth 2 check k-1 ran on the other stream
            if (D == 2 && k >= 1) HIPCHK(hipStreamWaitEvent(c, e->evF[(k - 1) % NS], 0));
            cp.cnt = cnt;
            cp.is_last = last ? 1 : 0;
            launch_conv_update(g, sumb[ci], e->st, cp, c, e->st_view + (size_t)(k % NV) * g.members,
                               e->st_host2 + (size_t)sl * g.members);
            HIPCHK(hipEventRecord(e->evF[sl], c));
            e->timing.check_launches += 1;
            XCHK(check_launch("sweep/check"));
            if (last) final_check = k;
            if (k >= D) XCHK(digest_upto(k - D));                            // D checks behind the sweeps
            ++k;
        } else if (last) {
            // max_iter is not a check iteration: every earlier check must be settled before :242-248 applies
            XCHK(digest_upto(k - 1));
            for (int i = 0; i < D; ++i) HIPCHK(hipStreamSynchronize(cst[i]));
            launch_finalize(g, e->st, max_iter, s);
            HIPCHK(hipMemcpyAsync(e->st_host, e->st, sizeof(MemberState) * g.members, hipMemcpyDeviceToHost, s));
            HIPCHK(hipEventRecord(e->ev[2], s));
            sweeps_end_recorded = true;
            HIPCHK(hipStreamSynchronize(s));
            XCHK(check_launch("finalize"));
            for (int m = 0; m < g.members; ++m) active[m] = 0;               // their result is psi[cur] itself
            all_done = true;
        }
    }
    if (!sweeps_end_recorded) HIPCHK(hipEventRecord(e->ev[2], s));           // the end of the sweeps on their stream
    // the checks still in flight: their outcome decides whose result is a snapshot (a member may have stopped at any
    // of them; whatever was enqueued after its stopping check is discarded below)
    XCHK(digest_upto(k - 1));
    for (int i = 0; i < D; ++i) HIPCHK(hipStreamSynchronize(cst[i]));
    HIPCHK(hipEventRecord(e->ev[1], s));
    if (max_iter == 0) HIPCHK(hipMemcpyAsync(e->st_host, e->st, sizeof(MemberState) * g.members, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, e->ev[0], e->ev[1]));
    e->timing.solve_ms = ms;
    if (!block_events) {
        HIPCHK(hipEventElapsedTime(&ms, e->ev[0], e->ev[2]));
        e->timing.sweep_ms = ms;
    }

    // :259-264  results into one plane.  A member that stopped at a check has been iterated further
    // while its check was in flight (unless that check was the very last thing enqueued): take its snapshot.
    const bool tail_clean = (final_check >= 0 && final_check == k - 1);       // nothing ran after the last check
    for (int m = 0; m < g.members; ++m) {
        if (stop_slot[m] < 0) continue;                                       // ended by finalize: psi[cur] is it
        if (tail_clean && stop_slot[m] == ((k - 1) % NS) && e->st_host[m].iters == max_iter) continue;
        HIPCHK(hipMemcpyAsync(e->psi[cur] + (size_t)m * g.plane, snapbuf[stop_slot[m]] + (size_t)m * g.plane,
                              sizeof(float) * g.plane, hipMemcpyDeviceToDevice, s));
    }
    {
        int n[2] = {0, 0};                        // tiles redone by the exact body: residual-writing launches, plain launches
        HIPCHK(hipMemcpyAsync(n, e->redo, sizeof(n), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        e->timing.redo_tiles = n[0]; e->timing.inline_redo_tiles = n[1];
    }
    // XEE_F_RESIDUAL = the residual plane of each member's LAST check (include/xee.h): checks rotated through the
    // slots' planes; a stopped member's is the one of its stopping check, everyone else's the one of the last check enqueued
    for (int m = 0; m < g.members && k >= 1; ++m) {
        const int slot = stop_slot[m] >= 0 ? stop_slot[m] : ((k - 1) % NS);
        if (slot != 0)
            HIPCHK(hipMemcpyAsync(e->res + (size_t)m * g.plane, resbuf[slot] + (size_t)m * g.plane, sizeof(float) * g.plane,
                                  hipMemcpyDeviceToDevice, s));
    }
    // the slow-chunk counters of the last check, where callers look for them
    if (D == 2 && k >= 1 && ((k - 1) % D) == 1 && !e->exact_serial)
        HIPCHK(hipMemcpyAsync(e->ex_slow, e->ex_slow2, sizeof(int) * g.members, hipMemcpyDeviceToDevice, s));
    HIPCHK(hipStreamSynchronize(s));
    e->cur = cur;
    return XEE_OK;
}

// solve_elliptic for a SMALL batch (one 1024x512 case: 242 workgroups per launch, a launch every 9 us): the sweeps of
// GB consecutive check blocks are ONE graph of 52 kernels, and the sweep stream carries nothing else -- no event record
// between blocks (each is a barrier packet that drains the queue: 6 us), one after the graph, at the boundary between two
// graph launches where the queue drains anyway (16 us, now paid once per GB blocks).  The GB checks of a graph start when
// the graph has ended and run, as in solve_pipelined, beside the sweeps of the NEXT graph (two check streams, only the
// state machines chained).  Slots: a check's residual plane, snapshot, event and host copy of the member states live in
// slot k % GS; the graph of checks [k, k + GB) is launched once every check up to k + GB - 1 - GS has been digested by the
// host -- with GS = 3 GB those belong to the graph before the previous one, so the host never waits in steady state.
// Views: check j's state machine writes view j % GS; the sweeps of the graph of checks [k, k + GB) are shown the view of
// check k - GB - 1 (digested by then; the checks in flight, k - GB .. k + GB - 1, write other views), so a graph depends on
// k % GS and the ping-pong phase only: three graphs per solve.  A member that stops is iterated on for up to GS more
// blocks before the sweeps see
// it -- harmless, its result is the snapshot of its stopping check (whose slot is not overwritten: by the time the slot
// comes round again the sweeps skip the member), but wasted work: the driver is used for long solves only.
int solve_graphed(xee_ensemble* e, int max_iter, int cs, ConvParams cp, float alpha, int tb_T, bool* ran)
{
    // blocks per graph: the option's value, but no more than four when the ring (3 GB slots of two planes) would exceed 128 MB
    const int GB = (e->opts.graph_blocks > 4 && (size_t)e->g.members * e->g.plane * sizeof(float) > ((size_t)2 << 20) + 65536) ? 4 : e->opts.graph_blocks;
    const int GS = 3 * GB, GV = GS + 1;                       // view GS = the initial states (never written)
    const bool block_events = e->opts.graph_events != 0;     // (experiment: an external event record per block inside the graph)
    const Grid& g = e->g;
    hipStream_t s = e->stream;
    *ran = false;
    if (!e->cstream[0]) HIPCHK(create_stream(&e->cstream[0], false));
    if (!e->cstream[1]) {
        HIPCHK(create_stream(&e->cstream[1], false));
        HIPCHK(hipMalloc(&e->ex_ws2, exact_ws_bytes(g)));
        HIPCHK(hipMalloc(reinterpret_cast<void**>(&e->sums2), sizeof(float) * g.members));
        HIPCHK(hipMalloc(reinterpret_cast<void**>(&e->ex_slow2), sizeof(int) * g.members));
    }
    if (e->gres && e->g_slots < GS) {
        HIPCHK(hipStreamSynchronize(s));
        hipFree(e->gres); hipFree(e->gsnap); hipHostFree(e->gst_host); hipFree(e->gst_view);
        e->gres = e->gsnap = nullptr; e->gst_host = e->gst_view = nullptr;
    }
    if (!e->gres) {
        e->g_slots = GS;
        XCHK(alloc_planes(e, &e->gres, GS));
        XCHK(alloc_planes(e, &e->gsnap, GS));
        HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&e->gst_host), sizeof(MemberState) * GS * g.members));
        HIPCHK(hipMalloc(reinterpret_cast<void**>(&e->gst_view), sizeof(MemberState) * GV * g.members));
        for (auto& ev : e->gevE) if (!ev) HIPCHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        for (auto& ev : e->gevF) if (!ev) HIPCHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    }
    const size_t mp = (size_t)g.members * g.plane;
    for (int v = 0; v < GV; ++v)
        HIPCHK(hipMemcpyAsync(e->gst_view + (size_t)v * g.members, e->st, sizeof(MemberState) * g.members, hipMemcpyDeviceToDevice, s));
    // every snapshot slot starts as a full copy of the first guess (the rim never changes; the sweeps store interior points)
    for (int sl = 0; sl < GS; sl += 2) launch_copy2(g, e->psi[e->cur], e->gsnap + sl * mp, e->gsnap + (sl + 1) * mp, s);
    hipStream_t cst[2] = {e->cstream[0], e->cstream[1]};
    void* wsb[2] = {e->ex_ws, e->ex_ws2};
    float* sumb[2] = {e->sums, e->sums2};
    int* slowb[2] = {e->ex_slow, e->ex_slow2};

    int cur = e->cur;
    std::vector<int> stop_slot(g.members, -1);
    std::vector<char> active(g.members, 1);
    e->timing = xee_timing_t{};
    e->timing.sweep_streams = 1;
    HIPCHK(hipMemsetAsync(e->redo, 0, 2 * sizeof(int), s));
    HIPCHK(hipEventRecord(e->ev[0], s));
    int cnt = 0, k = 0, digested = 0;
    bool all_done = false, converging = false;    // converging: an active member met the criteria at the newest digested check
    const int n_checks = max_iter / cs;

    auto digest_upto = [&](int j) -> int {
        for (; digested <= j; ++digested) {
            const int sl = digested % GS;
            HIPCHK(hipEventSynchronize(e->gevF[sl]));
            const MemberState* sh = e->gst_host + (size_t)sl * g.members;
            all_done = true;
            converging = false;
            for (int m = 0; m < g.members; ++m) {
                if (active[m] && !sh[m].active) { active[m] = 0; stop_slot[m] = sl; }
                if (sh[m].active) { all_done = false; converging = converging || sh[m].converge_cnt > 0; }
            }
            memcpy(e->st_host, sh, sizeof(MemberState) * g.members);
        }
        return XEE_OK;
    };
    // the launches of ONE check block on stream s, from iteration `from` to the check iteration `from + cs`
    auto enqueue_block = [&](int from, int sl, const MemberState* view, int* cur_io, int* launches, long long* tiles) {
        int c2 = from;
        const int target = from + cs;
        bool snapped = false;
        while (c2 < target) {
            const int rem = (target - c2) % tb_T;
            const int n = rem ? rem : tb_T;             // the short launch first: the residual-writing one is full-length
            c2 += n;
            const bool check = c2 == target;
            e->last_sweep.tiles = 0; e->last_sweep.streamed = false;
            if (n == 1)
                launch_sweep(g, e->psi[*cur_io], e->psi[*cur_io ^ 1], e->coe, e->f, check ? e->gres + sl * mp : nullptr, view, alpha, s);
            else if (launch_blocked(e, e->psi[*cur_io], e->psi[*cur_io ^ 1], check ? e->gres + sl * mp : nullptr, alpha, n, view,
                                    check ? e->gsnap + sl * mp : nullptr))
                snapped = true;
            *cur_io ^= 1;
            *launches += 1;
            *tiles += e->last_sweep.tiles;
        }
        if (!snapped) launch_copy_active(g, e->psi[*cur_io], e->gsnap + sl * mp, view, s);
    };
    struct BlockGraph { hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr; int launches = 0; long long tiles = 0; int flips = 0; };
    struct GraphCache {
        std::map<int, BlockGraph> m;
        ~GraphCache() { for (auto& kv : m) { if (kv.second.exec) hipGraphExecDestroy(kv.second.exec); if (kv.second.graph) hipGraphDestroy(kv.second.graph); } }
    } graphs;
    bool use_graphs = true;

    while (k < n_checks && !all_done) {
        // a whole graph -- or single blocks: at the tail, and while a member is counting down to its stop (whatever is
        // enqueued behind a stop is wasted work: up to GS blocks of it with whole graphs, three with single blocks)
        const int gb = (n_checks - k >= GB && use_graphs && !converging) ? GB : 1;
        // the slots about to be written are free (one more than needed: at most GS - 1 checks in flight, so that none of them
        // writes the view shown to the sweeps), and the view's check has been digested.  A graph is shown the states after
        // check k - GB - 1 -- a function of k alone, so that only GS / GB distinct graphs exist per ping-pong phase -- single
        // blocks the newest digested one.
        XCHK(digest_upto(k + gb - GS));
        if (gb == GB) XCHK(digest_upto(k - GB - 1));
        if (all_done) break;
        const int vi = gb == GB ? (k >= GB + 1 ? (k - GB - 1) % GS : GS) : (digested >= 1 ? (digested - 1) % GS : GS);
        const MemberState* view = e->gst_view + (size_t)vi * g.members;
        int launches = 0;
        long long tiles = 0;
        if (gb == GB) {
            const int key = cur | ((k % GS) << 1) | ((vi == GS ? 1 : 0) << 7);     // ping-pong phase x first slot (x the initial view)
            BlockGraph& bg = graphs.m[key];
            if (!bg.exec) {
                int c2 = cur;
                if (hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed) == hipSuccess) {
                    bool ok = true;
                    for (int i = 0; i < GB; ++i) {
                        enqueue_block(cnt + i * cs, (k + i) % GS, view, &c2, &bg.launches, &bg.tiles);
                        if (block_events && i + 1 < GB) ok = ok && hipEventRecordWithFlags(e->gevE[(k + i) % GS], s, hipEventRecordExternal) == hipSuccess;
                    }
                    if (hipStreamEndCapture(s, &bg.graph) != hipSuccess || !ok || hipGraphInstantiate(&bg.exec, bg.graph, nullptr, nullptr, 0) != hipSuccess)
                        bg.exec = nullptr;
                    bg.flips = c2 ^ cur;
                }
                if (!bg.exec) { (void)hipGetLastError(); use_graphs = false; graphs.m.erase(key); continue; }   // single blocks from here on
            }
            HIPCHK(hipGraphLaunch(bg.exec, s));
            launches = bg.launches; tiles = bg.tiles; cur ^= bg.flips;
        } else {
            enqueue_block(cnt, k % GS, view, &cur, &launches, &tiles);
        }
        const int ev_last = (k + gb - 1) % GS;
        HIPCHK(hipEventRecord(e->gevE[ev_last], s));                     // the residual planes of checks k .. k + gb - 1 are written
        e->timing.sweep_launches += launches;
        e->timing.tiles += tiles;
        e->timing.iterations += (int64_t)gb * cs;
        for (int i = 0; i < gb; ++i, ++k) {
            const int sl = k % GS, ci = k % 2;
            hipStream_t c = cst[ci];
            cnt += cs;
            HIPCHK(hipStreamWaitEvent(c, e->gevE[block_events ? sl : ev_last], 0));
            if (e->exact_serial) launch_seqsum(g, e->gres + sl * mp, 1, g.nx - 1, 1, g.ny - 1, true, sumb[ci], e->st, c);
            else launch_exact_sumsq(g, e->gres + sl * mp, wsb[ci], sumb[ci], slowb[ci], e->st, c);
            if (k >= 1) HIPCHK(hipStreamWaitEvent(c, e->gevF[(k - 1) % GS], 0));     // the state machine is sequential in the checks
            cp.cnt = cnt;
            cp.is_last = cnt == max_iter ? 1 : 0;
            launch_conv_update(g, sumb[ci], e->st, cp, c, e->gst_view + (size_t)sl * g.members, e->gst_host + (size_t)sl * g.members);
            HIPCHK(hipEventRecord(e->gevF[sl], c));
            e->timing.check_launches += 1;
        }
        XCHK(check_launch("sweep/check (graphed)"));
        XCHK(digest_upto(k - 1 - (gb == GB ? GB : 2)));                    // the host stays one graph (or two single blocks) behind the sweeps
    }
    // iterations past the last check (max_iter is not a multiple of check_step): every check settled first, then :242-248
    XCHK(digest_upto(k - 1));
    for (int i = 0; i < 2; ++i) HIPCHK(hipStreamSynchronize(cst[i]));
    const bool stopped_early = all_done;
    bool tail_ran = false;
    if (!stopped_early && cnt < max_iter) {
        // (the members still active take the remaining iterations from psi[cur]; they see the final states)
        int c2 = cnt;
        while (c2 < max_iter) {
            const int n = (max_iter - c2) < tb_T ? (max_iter - c2) : tb_T;
            e->last_sweep.tiles = 0; e->last_sweep.streamed = false;
            if (n == 1) launch_sweep(g, e->psi[cur], e->psi[cur ^ 1], e->coe, e->f, nullptr, e->st, alpha, s);
            else launch_blocked(e, e->psi[cur], e->psi[cur ^ 1], nullptr, alpha, n, e->st, nullptr);
            cur ^= 1; c2 += n;
            e->timing.sweep_launches += 1; e->timing.tiles += e->last_sweep.tiles;
        }
        e->timing.iterations += max_iter - cnt;
        launch_finalize(g, e->st, max_iter, s);
        HIPCHK(hipMemcpyAsync(e->st_host, e->st, sizeof(MemberState) * g.members, hipMemcpyDeviceToHost, s));
        tail_ran = true;
    }
    HIPCHK(hipEventRecord(e->ev[2], s));
    HIPCHK(hipEventRecord(e->ev[1], s));
    HIPCHK(hipStreamSynchronize(s));
    XCHK(check_launch("solve (graphed)"));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, e->ev[0], e->ev[1]));
    e->timing.solve_ms = ms;
    e->timing.sweep_ms = ms;                       // (the sweep stream carries nothing but sweeps)

    // :259-264  results into one plane.  A member that stopped at a check has been iterated further meanwhile: its result is
    // the snapshot of its stopping check -- unless that was the very last thing enqueued and it ran to max_iter
    const bool tail_clean = !tail_ran && k >= 1 && cnt == max_iter;
    for (int m = 0; m < g.members; ++m) {
        if (stop_slot[m] < 0) continue;                                       // ended by finalize: psi[cur] is it
        if (tail_clean && stop_slot[m] == ((k - 1) % GS) && e->st_host[m].iters == max_iter) continue;
        HIPCHK(hipMemcpyAsync(e->psi[cur] + (size_t)m * g.plane, e->gsnap + stop_slot[m] * mp + (size_t)m * g.plane,
                              sizeof(float) * g.plane, hipMemcpyDeviceToDevice, s));
    }
    {
        int n[2] = {0, 0};
        HIPCHK(hipMemcpyAsync(n, e->redo, sizeof(n), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        e->timing.redo_tiles = n[0]; e->timing.inline_redo_tiles = n[1];
    }
    // XEE_F_RESIDUAL = the residual plane of each member's LAST check
    for (int m = 0; m < g.members && k >= 1; ++m) {
        const int slot = stop_slot[m] >= 0 ? stop_slot[m] : ((k - 1) % GS);
        HIPCHK(hipMemcpyAsync(e->res + (size_t)m * g.plane, e->gres + slot * mp + (size_t)m * g.plane, sizeof(float) * g.plane,
                              hipMemcpyDeviceToDevice, s));
    }
    if (k >= 1 && ((k - 1) % 2) == 1 && !e->exact_serial)
        HIPCHK(hipMemcpyAsync(e->ex_slow, e->ex_slow2, sizeof(int) * g.members, hipMemcpyDeviceToDevice, s));
    HIPCHK(hipStreamSynchronize(s));
    e->cur = cur;
    *ran = true;
    return XEE_OK;
}


// The view of members [m0, m0 + n) of `e` for one of the two concurrent halves of a large batch's solve (created once, kept).
int make_half_view(xee_ensemble* e, int h, int m0, int n)
{
    if (e->half[h] && e->half[h]->g.members == n) return XEE_OK;
    if (e->half[h]) { xee_ensemble_destroy(e->half[h]); e->half[h] = nullptr; }
    xee_ensemble* s = new xee_ensemble();
    s->is_view = true;
    s->device = e->device;
    s->g = e->g;
    s->g.members = n;
    s->g.kstride = (size_t)e->g.members * e->g.plane;
    e->half[h] = s;
    if (create_stream(&s->stream, true) != hipSuccess) return fail(XEE_ERR_HIP, "hipStreamCreate (half) failed");
    s->own_stream = true;
    for (auto& ev : s->ev) HIPCHK(hipEventCreate(&ev));
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&s->sums), sizeof(float) * n));
    HIPCHK(hipMalloc(&s->ex_ws, exact_ws_bytes(s->g)));
    s->rec.assign(n, xee_case_record_t{});
    (void)m0;
    return XEE_OK;
}

// (re)point a half at its member range and copy the settings of this solve
void aim_half_view(xee_ensemble* e, xee_ensemble* s, int m0)
{
    const size_t off = (size_t)m0 * e->g.plane;
    s->f = e->f + off; s->psi[0] = e->psi[0] + off; s->psi[1] = e->psi[1] + off; s->res = e->res + off; s->bc = e->bc + off;
    s->coe = e->coe + off; s->base = e->base ? e->base + off : nullptr;
    s->st = e->st + m0; s->st_host = e->st_host + m0; s->ex_slow = e->ex_slow + m0; s->redo = e->redo;
    s->cur = e->cur; s->opts = e->opts; s->have_base = e->have_base; s->base_baro = e->base_baro; s->uv = e->uv;
    s->z_pairs_now = e->z_pairs_now; s->z_T_now = e->z_T_now; s->g.base_shared = e->g.base_shared; s->g.two_pq4 = e->g.two_pq4;
    s->exact_serial = e->exact_serial; s->sync_checks = e->sync_checks;
}

}  // namespace

extern "C" {

int xee_ensemble_solve(xee_ensemble_t* e, int max_iter, int check_step, int converge_time, int lost_rate,
                       float r1, float r2, float alpha, int flags)
{
    XCHK(use(e));
    // :112-129
    const bool chk_abs = r1 > 0.0f, chk_rel = r2 > 0.0f;
    if (!chk_abs && !chk_rel)
        return fail(XEE_ERR_INVALID_CRITERIA, "[check_abs_err] and [check_rel_err] cannot both be non-positive.");
    if (max_iter < 0) return fail(XEE_ERR_INVALID_ARGUMENT, "max_iter < 0");
    ConvParams cp{};
    cp.r1 = chk_abs ? r1 : FLT_MAX;
    cp.r2 = chk_rel ? r2 : FLT_MAX;
    const int cs = check_step > 0 ? check_step : 100;          // :131-144
    cp.converge_time = converge_time > 0 ? converge_time : 10;
    cp.lost_rate = lost_rate > 0 ? lost_rate : 5;
    cp.n_interior = (e->g.nx - 2) * (e->g.ny - 2);
    cp.debug = (flags & XEE_SOLVE_TRACE) ? 2 : 0;
    const Grid& g = e->g;
    hipStream_t s = e->stream;
    XCHK(base_shared_update(e));                        // (which generating planes all members share: Grid.base_shared)

    // :171-174  workspace = dat; the first iteration reads dat.  Both ping-pong planes carry the rim.
    if (flags & XEE_SOLVE_KEEP_PSI) launch_copy2(g, e->psi[e->cur], e->psi[e->cur ^ 1], nullptr, s);
    else { launch_copy2(g, e->bc, e->psi[0], e->psi[1], s); e->cur = 0; }
    launch_state_init(g, e->st, s);
    int cur = e->cur;                                   // plane holding the newest iterate
    std::vector<int> final_buf(g.members, cur);        // plane a member's result was written to
    std::vector<char> was_active(g.members, 1);
    const int tb_T = (flags & XEE_SOLVE_SINGLE_STEP) ? 1 : e->opts.T;

    // default: residual checks pipelined behind the sweeps; option "sync_checks" (or the per-check
    // trace, whose lines must appear in order) selects the simple synchronous driver below
    if (!e->sync_checks && cp.debug == 0) {
        // a small batch with a long solve ahead: several check blocks per graph (solve_graphed); else one block at a time
        const bool small = (size_t)g.members * g.nx * g.ny < ((size_t)4 << 20);
        bool ran = false;
        if (small && e->opts.graph_blocks && e->opts.kernel != 4 && tb_T > 1 && cs > tb_T && max_iter / cs >= 64 && s != nullptr)
            XCHK(solve_graphed(e, max_iter, cs, cp, alpha, tb_T, &ran));
        // a large batch that runs on k_sweep_q alone (no single-iteration launches): two halves of the member range, side by side
        const bool q_only = e->have_base && (e->opts.kernel == 0 || e->opts.kernel == 3 || e->opts.kernel == 4) && tb_T > 1 && cs % tb_T != 1 &&
                            (max_iter % cs) % tb_T != 1 && g.plane * sizeof(float) < ((size_t)1 << 31);
        // (measured on 1024x512 batches, tools/split_probe.py: +2 .. +4 % at 32 members, nothing at 64 and 256, a loss below 24 --
        // the halves must themselves still be large batches)
        const bool split = !ran && !small && q_only && e->opts.split_halves && g.members >= 32 && max_iter > 0 && s != nullptr && !e->is_view;
        // plain launches by the streaming k_sweep_z where the batch fills the device with it and every member qualifies
        int first_bad = -1;
        if (!ran && max_iter > 0) XCHK(z_decide(e, alpha, tb_T, split ? 2 : 1, split ? (g.members + 1) / 2 : g.members, &first_bad));
        // Only some members keep the batch off the stream (e.g. the zero-forcing kind of a source-term sweep, whose cases come
        // last): the members before the first of them stream, the others take the tiled sweep, side by side
        int z_split_pairs = 0;
        if (e->z_pairs_now == 0 && first_bad >= 8 && !ran && !small && q_only && e->opts.split_halves && max_iter > 0 && s != nullptr && !e->is_view) {
            Grid gv = g;
            gv.members = first_bad;
            // (chunked as if it had the device to itself: the tiled launches find room beside it, and once the shorter of the two streams
            // is through the other has the device -- tools/elig_split_probe.py, 16 + 8 cases: 39.2 ms with 1920 waves, 47.5 with 960; 32 + 16:
            // 1920 either way; the tiled sweep for all: 61.1 ms)
            z_split_pairs = sweep_z_pairs(gv, first_bad, 1, z_threshold(e, first_bad), e->z_T_now);
            // (too few of them to fill the device: as one of two streams -- the 8 + 16 cases of the bench's source-term sweep, whose thermal
            // kind has zeros in its forcing too)
            if (z_split_pairs <= 0) z_split_pairs = sweep_z_pairs(gv, first_bad, 2, z_threshold(e, first_bad), e->z_T_now);
            if (z_split_pairs > 0 && e->opts.z_pairs > 0) z_split_pairs = e->opts.z_pairs;
        }
        if (split || z_split_pairs > 0) {
            const int n0 = z_split_pairs > 0 ? first_bad : (g.members + 1) / 2, n1 = g.members - n0;
            XCHK(make_half_view(e, 0, 0, n0));
            XCHK(make_half_view(e, 1, n0, n1));
            aim_half_view(e, e->half[0], 0);
            aim_half_view(e, e->half[1], n0);
            if (z_split_pairs > 0) { e->half[0]->z_pairs_now = z_split_pairs; e->half[1]->z_pairs_now = 0; }
            HIPCHK(hipMemsetAsync(e->redo, 0, 2 * sizeof(int), s));
            HIPCHK(hipStreamSynchronize(s));                     // first guess, states and counters are in place
            int st2[2] = {XEE_OK, XEE_OK};
            std::string msg[2];
            auto work = [&](int h) {
                if (hipSetDevice(e->device) != hipSuccess) { st2[h] = XEE_ERR_HIP; msg[h] = "hipSetDevice failed"; return; }
                st2[h] = solve_pipelined(e->half[h], max_iter, cs, cp, alpha, tb_T);
                if (st2[h] != XEE_OK) msg[h] = g_err;
            };
            std::thread other(work, 1);
            work(0);
            other.join();
            for (int h = 0; h < 2; ++h) if (st2[h] != XEE_OK) return fail(st2[h], "solve (half " + std::to_string(h) + "): " + msg[h]);
            e->cur = e->half[0]->cur;
            if (e->half[1]->cur != e->cur) {
                // (a half whose members all stopped ends early, possibly in the other ping-pong plane: its results move over)
                HIPCHK(hipMemcpyAsync(e->psi[e->cur] + (size_t)n0 * g.plane, e->psi[e->cur ^ 1] + (size_t)n0 * g.plane,
                                      sizeof(float) * g.plane * n1, hipMemcpyDeviceToDevice, s));
                HIPCHK(hipStreamSynchronize(s));
            }
            e->last_sweep = e->half[0]->last_sweep;
            e->timing = e->half[0]->timing;
            const xee_timing_t& t1 = e->half[1]->timing;
            e->timing.solve_ms = std::max(e->timing.solve_ms, t1.solve_ms);
            e->timing.sweep_ms = std::max(e->timing.sweep_ms, t1.sweep_ms);   // the halves' sweeps run side by side
            e->timing.sweep_launches += t1.sweep_launches; e->timing.tiles += t1.tiles; e->timing.check_launches += t1.check_launches;
            e->timing.z_launches += t1.z_launches; e->timing.z_timed_launches += t1.z_timed_launches;
            e->timing.z_ms = std::max(e->timing.z_ms, t1.z_ms);
            e->timing.sweep_streams = 2;
            {
                int n[2] = {0, 0};
                HIPCHK(hipMemcpy(n, e->redo, sizeof(n), hipMemcpyDeviceToHost));
                e->timing.redo_tiles = n[0]; e->timing.inline_redo_tiles = n[1];
            }
            ran = true;
        }
        if (!ran) XCHK(solve_pipelined(e, max_iter, cs, cp, alpha, tb_T));
        for (int m = 0; m < g.members; ++m) {
            xee_case_record_t& r = e->rec[m];
            r.iters = e->st_host[m].iters;
            r.err_bits = e->st_host[m].err_bits;
            r.err_now = e->st_host[m].err_now;
            r.ratio = e->st_host[m].ratio;
        }
        return XEE_OK;
    }

    e->timing = xee_timing_t{};
    e->timing.sweep_streams = 1;
    HIPCHK(hipMemsetAsync(e->redo, 0, 2 * sizeof(int), s));
    HIPCHK(hipEventRecord(e->ev[0], s));
    bool block_open = false;
    int cnt = 0;
    bool all_done = (max_iter == 0);
    while (cnt < max_iter && !all_done) {
        // iterations until the next residual check (or the end): one launch does up to tb_T of them
        const int next_check = (cnt / cs + 1) * cs;
        const int target = next_check < max_iter ? next_check : max_iter;
        const int n = (target - cnt) < tb_T ? (target - cnt) : tb_T;
        cnt += n;
        const bool check = cnt == next_check;
        const bool last = cnt == max_iter;
        if (!block_open) { HIPCHK(hipEventRecord(e->ev[2], s)); block_open = true; }
        e->last_sweep.tiles = 0;
        if (n == 1)
            launch_sweep(g, e->psi[cur], e->psi[cur ^ 1], e->coe, e->f, check ? e->res : nullptr, e->st, alpha, s);
        else
            launch_blocked(e, e->psi[cur], e->psi[cur ^ 1], check ? e->res : nullptr, alpha, n);
        cur ^= 1;
        e->timing.sweep_launches += 1;
        e->timing.tiles += e->last_sweep.tiles;
        e->timing.iterations += n;
        if (check) {
            HIPCHK(hipEventRecord(e->ev[3], s));
            if (e->exact_serial) launch_seqsum(g, e->res, 1, g.nx - 1, 1, g.ny - 1, true, e->sums, e->st, s);
            else launch_exact_sumsq(g, e->res, e->ex_ws, e->sums, e->ex_slow, e->st, s);
            cp.cnt = cnt;
            cp.is_last = last ? 1 : 0;
            launch_conv_update(g, e->sums, e->st, cp, s);
            e->timing.check_launches += 1;
        } else if (last) {
            launch_finalize(g, e->st, max_iter, s);
        }
        if (check || last) {
            XCHK(check_launch("sweep/check"));
            HIPCHK(hipMemcpyAsync(e->st_host, e->st, sizeof(MemberState) * g.members, hipMemcpyDeviceToHost, s));
            HIPCHK(hipStreamSynchronize(s));
            if (check) {
                float ms = 0.f;
                HIPCHK(hipEventElapsedTime(&ms, e->ev[2], e->ev[3]));
                e->timing.sweep_ms += ms;
                block_open = false;
            }
            all_done = true;
            for (int m = 0; m < g.members; ++m) {
                if (was_active[m]) final_buf[m] = cur;          // it took part in the launch just done
                was_active[m] = (char)e->st_host[m].active;
                if (e->st_host[m].active) all_done = false;
            }
        }
    }
    HIPCHK(hipEventRecord(e->ev[1], s));
    if (max_iter == 0) {
        HIPCHK(hipMemcpyAsync(e->st_host, e->st, sizeof(MemberState) * g.members, hipMemcpyDeviceToHost, s));
    }
    HIPCHK(hipStreamSynchronize(s));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, e->ev[0], e->ev[1]));
    e->timing.solve_ms = ms;
    {
        int n[2] = {0, 0};
        HIPCHK(hipMemcpy(n, e->redo, sizeof(n), hipMemcpyDeviceToHost));
        e->timing.redo_tiles = n[0]; e->timing.inline_redo_tiles = n[1];
    }

    // :259-264  every member's result into one plane.  A stopped member's planes are never touched
    // again, so its result still sits in the plane its last launch wrote.
    for (int m = 0; m < g.members; ++m) {
        if (final_buf[m] != cur)
            HIPCHK(hipMemcpyAsync(e->psi[cur] + (size_t)m * g.plane, e->psi[final_buf[m]] + (size_t)m * g.plane,
                                  sizeof(float) * g.plane, hipMemcpyDeviceToDevice, s));
    }
    HIPCHK(hipStreamSynchronize(s));
    e->cur = cur;
    for (int m = 0; m < g.members; ++m) {
        xee_case_record_t& r = e->rec[m];
        r.iters = e->st_host[m].iters;
        r.err_bits = e->st_host[m].err_bits;
        r.err_now = e->st_host[m].err_now;
        r.ratio = e->st_host[m].ratio;
    }
    return XEE_OK;
}

// OPT-IN, NON-PARITY: 4-colour SOR with the reference's stopping rule.  The
// check at a multiple of check_step uses the residual of the CURRENT iterate.  Iterates differ from
// the reference's Jacobi; the fixed point is the same.
int xee_ensemble_solve_sor(xee_ensemble_t* e, int max_iter, int check_step, int converge_time, int lost_rate,
                           float r1, float r2, float omega, int flags)
{
    XCHK(use(e));
    const bool chk_abs = r1 > 0.0f, chk_rel = r2 > 0.0f;
    if (!chk_abs && !chk_rel)
        return fail(XEE_ERR_INVALID_CRITERIA, "[check_abs_err] and [check_rel_err] cannot both be non-positive.");
    if (max_iter < 0 || !(omega > 0.0f && omega < 2.0f)) return fail(XEE_ERR_INVALID_ARGUMENT, "need max_iter >= 0 and 0 < omega < 2");
    ConvParams cp{};
    cp.r1 = chk_abs ? r1 : FLT_MAX;
    cp.r2 = chk_rel ? r2 : FLT_MAX;
    const int cs = check_step > 0 ? check_step : 100;
    cp.converge_time = converge_time > 0 ? converge_time : 10;
    cp.lost_rate = lost_rate > 0 ? lost_rate : 5;
    cp.n_interior = (e->g.nx - 2) * (e->g.ny - 2);
    cp.debug = (flags & XEE_SOLVE_TRACE) ? 2 : 0;
    const Grid& g = e->g;
    hipStream_t s = e->stream;
    if (!(flags & XEE_SOLVE_KEEP_PSI)) { launch_copy2(g, e->bc, e->psi[0], e->psi[1], s); e->cur = 0; }
    launch_state_init(g, e->st, s);
    e->timing = xee_timing_t{};
    e->timing.sweep_streams = 1;
    e->last_sweep = SweepDesc{(e->have_base && e->opts.kernel != 1) ? (e->base_baro ? "k_sor_base<red/black>" : "k_sor_base<4 colours>") : "k_sor_color", 1, g.nx, g.ny};
    HIPCHK(hipEventRecord(e->ev[0], s));
    bool all_done = (max_iter == 0);
    for (int cnt = 1; cnt <= max_iter && !all_done; ++cnt) {
        const bool from_base = e->have_base && e->opts.kernel != 1;
        if (from_base) launch_sor_iteration_base(g, e->psi[e->cur], e->base, e->f, e->st, omega, e->base_baro, s);
        else launch_sor_iteration(g, e->psi[e->cur], e->coe, e->f, e->st, omega, s);
        e->timing.sweep_launches += (from_base && e->base_baro) ? 2 : 4;
        e->timing.iterations += 1;
        const bool check = (cnt % cs) == 0, last = cnt == max_iter;
        if (check) {
            if (from_base) launch_residual_base(g, e->psi[e->cur], e->base, e->f, e->res, e->st, e->base_baro, s);
            else launch_residual(g, e->psi[e->cur], e->coe, e->f, e->res, e->st, s);
            launch_exact_sumsq(g, e->res, e->ex_ws, e->sums, e->ex_slow, e->st, s);
            cp.cnt = cnt;
            cp.is_last = last ? 1 : 0;
            launch_conv_update(g, e->sums, e->st, cp, s);
            e->timing.check_launches += 1;
        } else if (last) {
            launch_finalize(g, e->st, max_iter, s);
        }
        if (check || last) {
            XCHK(check_launch("sor/check"));
            HIPCHK(hipMemcpyAsync(e->st_host, e->st, sizeof(MemberState) * g.members, hipMemcpyDeviceToHost, s));
            HIPCHK(hipStreamSynchronize(s));
            all_done = true;
            for (int m = 0; m < g.members; ++m) if (e->st_host[m].active) all_done = false;
        }
    }
    HIPCHK(hipEventRecord(e->ev[1], s));
    if (max_iter == 0) HIPCHK(hipMemcpyAsync(e->st_host, e->st, sizeof(MemberState) * g.members, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, e->ev[0], e->ev[1]));
    e->timing.solve_ms = ms;
    for (int m = 0; m < g.members; ++m) {
        xee_case_record_t& r = e->rec[m];
        r.iters = e->st_host[m].iters;
        r.err_bits = e->st_host[m].err_bits;
        r.err_now = e->st_host[m].err_now;
        r.ratio = e->st_host[m].ratio;
    }
    return XEE_OK;
}

int xee_ensemble_cal_eta(xee_ensemble_t* e)
{
    XCHK(use(e));
    if (!e->have_profiles) return fail(XEE_ERR_INVALID_ARGUMENT, "profiles not set");
    XCHK(alloc_planes(e, &e->eta));
    launch_cal_eta(e->g, e->psi[e->cur], e->eta, e->ra, e->rcuva, e->rho, e->exner, e->stream);
    return check_launch("cal_eta");
}

int xee_ensemble_cal_uw(xee_ensemble_t* e)
{
    XCHK(use(e));
    if (!e->have_profiles) return fail(XEE_ERR_INVALID_ARGUMENT, "profiles not set");
    XCHK(alloc_planes(e, &e->u));
    XCHK(alloc_planes(e, &e->w));
    launch_cal_uw(e->g, e->psi[e->cur], e->u, e->w, e->ra, e->za, e->rcuva, e->rho, e->stream);
    return check_launch("cal_uw");
}

namespace {
// e->sums[m] = sequential fp32 sum of e->terms over [0,nx-1) x [0,ny-1), reference order (old-diagnose/diagnose.f90:1029-1048)
int ee_sum(xee_ensemble* e, hipStream_t s)
{
    const Grid& g = e->g;
    if (e->exact_serial) {
        launch_seqsum(g, e->terms, 0, g.nx - 1, 0, g.ny - 1, false, e->sums, nullptr, s);
        return XEE_OK;
    }
    if (!e->sx_ws && hipMalloc(&e->sx_ws, exact_signed_ws_bytes(g)) != hipSuccess) {
        e->sx_ws = nullptr;
        return fail(XEE_ERR_ALLOC, "hipMalloc signed exact-sum workspace");
    }
    launch_exact_sum_signed(g, e->terms, 0, g.nx - 1, 0, g.ny - 1, e->sx_ws, e->sums, e->ex_slow, nullptr, s);
    return XEE_OK;
}
}  // namespace

int xee_ensemble_ee_scalars(xee_ensemble_t* e)
{
    XCHK(use(e));
    if (!e->have_profiles) return fail(XEE_ERR_INVALID_ARGUMENT, "profiles not set");
    if (!e->Q || !e->eta) return fail(XEE_ERR_INVALID_ARGUMENT, "Q must be uploaded and eta computed first");
    XCHK(alloc_planes(e, &e->terms));
    const Grid& g = e->g;
    hipStream_t s = e->stream;
    std::vector<float> sq(g.members);
    launch_terms_weight(g, e->Q, e->terms, e->ra, e->za, e->rcuva, e->rho, s);
    XCHK(ee_sum(e, s));
    HIPCHK(hipMemcpyAsync(e->sums_host, e->sums, sizeof(float) * g.members, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    for (int m = 0; m < g.members; ++m) sq[m] = e->sums_host[m];
    e->sx_slow.assign(2 * (size_t)g.members, 0);
    if (!e->exact_serial) HIPCHK(hipMemcpy(e->sx_slow.data(), e->ex_slow, sizeof(int) * g.members, hipMemcpyDeviceToHost));
    launch_terms_Qeta(g, e->Q, e->eta, e->terms, e->ra, e->za, e->rcuva, e->rho, s);
    XCHK(ee_sum(e, s));
    HIPCHK(hipMemcpyAsync(e->sums_host, e->sums, sizeof(float) * g.members, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    XCHK(check_launch("ee_scalars"));
    if (!e->exact_serial) HIPCHK(hipMemcpy(e->sx_slow.data() + g.members, e->ex_slow, sizeof(int) * g.members, hipMemcpyDeviceToHost));
    for (int m = 0; m < g.members; ++m) {
        e->rec[m].sum_Q = sq[m];
        e->rec[m].sum_Qeta = e->sums_host[m];
        e->rec[m].efficiency = e->sums_host[m] / sq[m];      // old-diagnose/diagnose.f90:792-793
    }
    return XEE_OK;
}

int xee_ensemble_ee_slow_chunks(xee_ensemble_t* e, int* sum_Q_slow, int* sum_Qeta_slow)
{
    if (!e || !sum_Q_slow || !sum_Qeta_slow) return fail(XEE_ERR_INVALID_ARGUMENT, "null argument");
    if (e->sx_slow.size() != 2 * (size_t)e->g.members) return fail(XEE_ERR_INVALID_ARGUMENT, "xee_ensemble_ee_scalars has not run");
    memcpy(sum_Q_slow, e->sx_slow.data(), sizeof(int) * e->g.members);
    memcpy(sum_Qeta_slow, e->sx_slow.data() + e->g.members, sizeof(int) * e->g.members);
    return XEE_OK;
}

int xee_ensemble_records(xee_ensemble_t* e, xee_case_record_t* out)
{
    if (!e || !out) return fail(XEE_ERR_INVALID_ARGUMENT, "null argument");
    memcpy(out, e->rec.data(), sizeof(xee_case_record_t) * e->g.members);
    return XEE_OK;
}

int xee_ensemble_timing(xee_ensemble_t* e, xee_timing_t* out)
{
    if (!e || !out) return fail(XEE_ERR_INVALID_ARGUMENT, "null argument");
    *out = e->timing;
    out->shared_planes = e->base_shared_known ? (e->g.base_shared & 15) : 0;
    return XEE_OK;
}

int xee_quotient_selfcheck(int device, uint64_t seed, int blocks, int per_thread, int d_exp_lo, int d_exp_hi,
                           uint64_t* mismatches, float* first_bad)
{
    if (!mismatches || blocks < 1 || per_thread < 1 || d_exp_lo > d_exp_hi || d_exp_lo < -120 || d_exp_hi > 119)
        return fail(XEE_ERR_INVALID_ARGUMENT, "quotient_selfcheck: bad arguments");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n < 1) return fail(XEE_ERR_NO_DEVICE, "no HIP device");
    HIPCHK(hipSetDevice(device < 0 ? 0 : device));
    unsigned long long* d_cnt = nullptr;
    float* d_bad = nullptr;
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&d_cnt), 2 * sizeof(unsigned long long)));
    hipError_t he = hipMalloc(reinterpret_cast<void**>(&d_bad), 2 * sizeof(float));
    unsigned long long h_cnt[2] = {0, 0};
    float h_bad[2] = {0.f, 0.f};
    if (he == hipSuccess) he = hipMemset(d_cnt, 0, 2 * sizeof(unsigned long long));
    if (he == hipSuccess) he = hipMemset(d_bad, 0, 2 * sizeof(float));
    if (he == hipSuccess) {
        launch_quotient_check(seed, blocks, per_thread, d_exp_lo, d_exp_hi, d_cnt, d_bad, nullptr);
        he = hipMemcpy(h_cnt, d_cnt, sizeof(h_cnt), hipMemcpyDeviceToHost);
    }
    if (he == hipSuccess) he = hipMemcpy(h_bad, d_bad, sizeof(h_bad), hipMemcpyDeviceToHost);
    if (he == hipSuccess) he = hipGetLastError();
    hipFree(d_cnt);
    if (d_bad) hipFree(d_bad);
    if (he != hipSuccess) return fail(XEE_ERR_HIP, std::string("quotient_selfcheck: ") + hipGetErrorString(he));
    *mismatches = h_cnt[0];
    if (first_bad) { first_bad[0] = h_bad[0]; first_bad[1] = h_bad[1]; }
    return XEE_OK;
}

int xee_ensemble_time_sweeps(xee_ensemble_t* e, int iters, int variant, float alpha, double* elapsed_ms,
                             int64_t* launches)
{
    XCHK(use(e));
    if (iters < 1 || (variant != 0 && variant != 1)) return fail(XEE_ERR_INVALID_ARGUMENT, "iters >= 1, variant 0 (streaming) or 1 (temporally blocked)");
    const Grid& g = e->g;
    hipStream_t s = e->stream;
    launch_state_init(g, e->st, s);
    int64_t nl = 0;
    int cur = e->cur;
    // (option "kernel" = 4: launches of sweep_z_T() iterations by k_sweep_z where the members qualify)
    e->z_pairs_now = 0;
    XCHK(base_shared_update(e));
    if (variant == 1 && e->opts.kernel == 4) XCHK(z_decide(e, alpha, e->opts.T, 1, g.members));
    const int len = e->z_pairs_now > 0 ? e->z_T_now : e->opts.T;
    HIPCHK(hipEventRecord(e->ev[0], s));
    for (int k = 0; k < iters;) {
        const int n = variant == 0 ? 1 : ((iters - k) < len ? (iters - k) : len);
        if (n == 1) launch_sweep(g, e->psi[cur], e->psi[cur ^ 1], e->coe, e->f, nullptr, e->st, alpha, s);
        else launch_blocked(e, e->psi[cur], e->psi[cur ^ 1], nullptr, alpha, n);
        cur ^= 1;
        k += n;
        ++nl;
    }
    HIPCHK(hipEventRecord(e->ev[1], s));
    HIPCHK(hipStreamSynchronize(s));
    XCHK(check_launch("time_sweeps"));
    e->cur = cur;
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, e->ev[0], e->ev[1]));
    if (elapsed_ms) *elapsed_ms = ms;
    if (launches) *launches = nl;
    return XEE_OK;
}

// ---------------------------------------------------------------------------------------------
// Tier 1: host arrays in, host arrays out, through a one-member ensemble
// ---------------------------------------------------------------------------------------------

namespace {
// The one-member ensemble behind the Tier-1 entry points is KEPT per thread and reused while the grid stays the same:
// creating and destroying one costs ~15 ms (thirty allocations, pinned host memory, streams, events, their release)
// against 1-3 ms of work for a small solve -- the reference program and the legacy program make dozens of such calls.
// A reused ensemble is put back into the state of a fresh one: every plane it holds zero-filled, flags and options reset
// (process defaults are read again).  A call that reports an error drops the cached ensemble; xee_tier1_release() frees
// it explicitly (it holds device memory: about forty planes of the last grid).
struct Tier1Cache { xee_ensemble_t* e = nullptr; int nx = 0, ny = 0; };
thread_local Tier1Cache g_t1;
void tier1_release_all();
// A thread that ends (a pool worker, an OpenMP team member) releases what its Tier-1 calls kept on the device: the
// reference's procedures are stateless, so such a caller has no reason to call xee_tier1_release itself.  The main
// thread's thread_locals die before any static object does, i.e. while the HIP runtime is still up; a thread that ends
// later than that (during process exit) leaks instead.
struct Tier1ThreadGuard {
    bool armed = false;
    ~Tier1ThreadGuard()
    {
        if (armed && !g_process_exiting.load()) { tier1_release_all(); g_n_thread_exit_releases.fetch_add(1); }
    }
};
thread_local Tier1ThreadGuard g_t1_guard;
// caches above this many bytes of planes are not kept between calls: a solve on such a grid takes seconds, the 15 ms of
// a fresh ensemble do not matter, and ~50 planes of a 4096x2048 grid are 1.7 GB
constexpr size_t kTier1KeepBytes = (size_t)1 << 30;

void tier1_drop()
{
    if (g_t1.e) xee_ensemble_destroy(g_t1.e);
    g_t1 = Tier1Cache{};
}

int tier1_reset(xee_ensemble* e)
{
    HIPCHK(hipSetDevice(e->device));
    hipStream_t s = e->stream;
    const size_t pb = e->g.plane * e->g.members * sizeof(float);
    for (float* p : {e->A, e->B, e->C, e->bc, e->f, e->psi[0], e->psi[1], e->sA, e->sB, e->sC, e->res, e->eta, e->u, e->w,
                     e->Q, e->terms, e->res2, e->res3})
        if (p) HIPCHK(hipMemsetAsync(p, 0, pb, s));
    if (e->coe) HIPCHK(hipMemsetAsync(e->coe, 0, 9 * pb, s));
    if (e->base) HIPCHK(hipMemsetAsync(e->base, 0, 4 * pb, s));
    if (e->snap) HIPCHK(hipMemsetAsync(e->snap, 0, e->snap_planes * pb, s));
    if (e->gres) { HIPCHK(hipMemsetAsync(e->gres, 0, e->g_slots * pb, s)); HIPCHK(hipMemsetAsync(e->gsnap, 0, e->g_slots * pb, s)); }
    const size_t nv = (size_t)(e->g.nx > e->g.ny ? e->g.nx : e->g.ny) + 64;
    for (float* v : {e->ra, e->za, e->rcuva, e->rho, e->exner})
        if (v) HIPCHK(hipMemsetAsync(v, 0, nv * sizeof(float), s));
    const Defaults dflt = read_defaults();
    e->opts = dflt.sweep;
    e->sync_checks = dflt.sync_checks != 0;
    e->exact_serial = dflt.exact_sum_serial != 0;
    e->have_profiles = false;
    e->have_base = false;
    e->base_baro = false;
    e->uv = UniformVals{};
    e->dr = e->dz = 0.f;
    e->cur = 0;
    e->win_first = e->win_count = 0;
    e->first_case_id = 0;
    e->timing = xee_timing_t{};
    e->timing.sweep_streams = 1;
    e->last_sweep = SweepDesc{};
    e->sx_slow.clear();
    e->rec.assign(e->g.members, xee_case_record_t{});
    for (int m = 0; m < e->g.members; ++m) e->rec[m].case_id = m;
    launch_state_init(e->g, e->st, s);
    return XEE_OK;
}

struct Tier1 {
    xee_ensemble_t* e = nullptr;
    unsigned long long failures0 = 0;
    int acquire(int nx, int ny)
    {
        failures0 = g_failures;
        g_t1_guard.armed = true;
        int dev = 0;
        if (g_t1.e && (g_t1.nx != nx || g_t1.ny != ny || hipGetDevice(&dev) != hipSuccess || dev != g_t1.e->device)) tier1_drop();
        if (g_t1.e) {
            const int st = tier1_reset(g_t1.e);
            if (st != XEE_OK) { tier1_drop(); return st; }
        } else {
            XCHK(xee_ensemble_create(&g_t1.e, nx, ny, 1, -1));
            g_t1.nx = nx; g_t1.ny = ny;
        }
        e = g_t1.e;
        return XEE_OK;
    }
    ~Tier1()
    {
        if (!e) return;
        // whatever went wrong, the next call starts afresh; and a large grid's planes are not held between calls
        if (g_failures != failures0 || e->g.plane * sizeof(float) * 50 > kTier1KeepBytes) tier1_drop();
    }
};
}  // namespace


namespace {
// What the last xee_cal_coe of this thread produced, kept on the device: the solver coefficients it was given and the
// nine coefficient planes it returned.  xee_solve_elliptic receives an arbitrary host `coe`; when the one it is handed
// holds exactly these bits (compared on the device after the upload it needs anyway), the operator's generating planes
// can be rebuilt from a, b, c and the solve runs k_sweep_q instead of the nine-plane kernel -- the drop-in user
// (the reference's main.f90: cal_coe, then solve_elliptic with that coe, diagnose.f90:7,15) gets the fast path.
struct CoeCache {
    int device = -1, nx = 0, ny = 0;
    float dx = 0.f, dy = 0.f;
    size_t plane = 0;
    float *abc = nullptr, *coe = nullptr;      // 3 + 9 planes
    bool b_zero = false, valid = false;
    ~CoeCache() { /* process exit: the runtime may already be gone; leak the two buffers */ }
};
thread_local CoeCache g_cc;
thread_local SweepDesc g_last_tier1_sweep{};

void tier1_release_all()
{
    tier1_drop();
    if (g_cc.abc) hipFree(g_cc.abc);
    if (g_cc.coe) hipFree(g_cc.coe);
    g_cc.abc = g_cc.coe = nullptr; g_cc.plane = 0; g_cc.device = -1; g_cc.valid = false;
}

int cache_store(xee_ensemble* e, const float* b_host, float dx, float dy)
{
    CoeCache& cc = g_cc;
    cc.valid = false;
    if (sizeof(float) * 12 * e->g.plane > kTier1KeepBytes) return XEE_OK;   // too large to keep: the solve takes the nine-plane kernel
    if (cc.device != e->device || cc.plane != e->g.plane) {
        if (cc.abc) hipFree(cc.abc);
        if (cc.coe) hipFree(cc.coe);
        cc.abc = cc.coe = nullptr;
        if (hipMalloc(reinterpret_cast<void**>(&cc.abc), sizeof(float) * 3 * e->g.plane) != hipSuccess ||
            hipMalloc(reinterpret_cast<void**>(&cc.coe), sizeof(float) * 9 * e->g.plane) != hipSuccess) {
            if (cc.abc) hipFree(cc.abc);
            cc.abc = cc.coe = nullptr; cc.plane = 0; cc.device = -1;
            (void)hipGetLastError();
            return XEE_OK;                       // no cache, no harm: the solve takes the nine-plane kernel
        }
        cc.device = e->device; cc.plane = e->g.plane;
    }
    const size_t pb = sizeof(float) * e->g.plane;
    HIPCHK(hipMemcpyAsync(cc.abc, e->sA, pb, hipMemcpyDeviceToDevice, e->stream));
    HIPCHK(hipMemcpyAsync(cc.abc + e->g.plane, e->sB, pb, hipMemcpyDeviceToDevice, e->stream));
    HIPCHK(hipMemcpyAsync(cc.abc + 2 * e->g.plane, e->sC, pb, hipMemcpyDeviceToDevice, e->stream));
    HIPCHK(hipMemcpyAsync(cc.coe, e->coe, 9 * pb, hipMemcpyDeviceToDevice, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    cc.nx = e->g.nx; cc.ny = e->g.ny; cc.dx = dx; cc.dy = dy;
    cc.b_zero = true;                            // b = +0 everywhere: the BAROTROPIC operator (diagnose.f90:6)
    const size_t nb = (size_t)(cc.nx - 1) * (cc.ny - 1);
    for (size_t q = 0; q < nb; ++q) {
        unsigned bits;
        memcpy(&bits, b_host + q, sizeof bits);
        if (bits != 0u) { cc.b_zero = false; break; }
    }
    cc.valid = true;
    return XEE_OK;
}

// after the caller's coe is on the device: is it the cached one?  then build the generating planes
int cache_try(xee_ensemble* e)
{
    const CoeCache& cc = g_cc;
    if (!cc.valid || cc.device != e->device || cc.nx != e->g.nx || cc.ny != e->g.ny || cc.plane != e->g.plane) return XEE_OK;
    if (!e->uv_differs) HIPCHK(hipMalloc(reinterpret_cast<void**>(&e->uv_differs), sizeof(unsigned)));
    launch_coe_compare(e->g, e->coe, cc.coe, e->uv_differs, e->stream);
    unsigned differs = 1u;
    HIPCHK(hipMemcpyAsync(&differs, e->uv_differs, sizeof(unsigned), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    if (differs) return XEE_OK;
    XCHK(alloc_planes(e, &e->base, 5));
    e->g.two_pq4 = 2.0f * (4.0f * cc.dx * cc.dy);
    launch_base_planes(e->g, cc.abc, cc.b_zero ? nullptr : cc.abc + e->g.plane, cc.abc + 2 * e->g.plane, e->base, cc.dx, cc.dy, e->stream);
    XCHK(check_launch("base planes (Tier 1)"));
    e->have_base = true;
    e->base_shared_known = false; e->g.base_shared = 0;
    e->base_baro = cc.b_zero;
    return XEE_OK;
}
}  // namespace

int xee_tier1_release(void)
{
    tier1_release_all();
    return XEE_OK;
}

int xee_last_sweep_info(char* name, int name_len, int* T, int* tile_w, int* tile_h)
{
    if (name && name_len > 0) { strncpy(name, g_last_tier1_sweep.name, (size_t)name_len - 1); name[name_len - 1] = 0; }
    if (T) *T = g_last_tier1_sweep.T;
    if (tile_w) *tile_w = g_last_tier1_sweep.tile_w;
    if (tile_h) *tile_h = g_last_tier1_sweep.tile_h;
    return XEE_OK;
}

namespace {
// the device staging buffer for the reference's AoS coe(9,nx,ny), allocated once per kept ensemble
int aos_stage(xee_ensemble* e, float** out)
{
    if (!e->aos_stage) {
        const size_t bytes = sizeof(float) * 9 * (size_t)e->g.nx * e->g.ny;
        if (hipMalloc(reinterpret_cast<void**>(&e->aos_stage), bytes) != hipSuccess) { e->aos_stage = nullptr; return fail(XEE_ERR_ALLOC, "hipMalloc coe staging"); }
        g_n_tier1_stage_allocs.fetch_add(1);
    }
    *out = e->aos_stage;
    return XEE_OK;
}
}  // namespace

int xee_cal_coe(const float* a, const float* b, const float* c, float* coe, float dx, float dy, int nx, int ny, int* err)
{
    if (!a || !b || !c || !coe || !err) return fail(XEE_ERR_INVALID_ARGUMENT, "null argument");
    *err = 1;                                                   // elliptic_tools.f90:33
    Tier1 sc;
    XCHK(sc.acquire(nx, ny));
    xee_ensemble_t* e = sc.e;
    XCHK(xee_ensemble_upload(e, XEE_F_SOLVER_A, 0, 1, a));
    XCHK(xee_ensemble_upload(e, XEE_F_SOLVER_B, 0, 1, b));
    XCHK(xee_ensemble_upload(e, XEE_F_SOLVER_C, 0, 1, c));
    launch_cal_coe(e->g, e->sA, e->sB, e->sC, e->coe, dx, dy, e->stream);
    float* aos = nullptr;
    const size_t bytes = sizeof(float) * 9 * (size_t)nx * ny;
    XCHK(aos_stage(e, &aos));
    // only interior entries are written (the reference leaves the rim untouched): start from the caller's bytes
    hipError_t he = hipMemcpyAsync(aos, coe, bytes, hipMemcpyHostToDevice, e->stream);
    if (he == hipSuccess) { launch_coe_soa_to_aos(e->g, e->coe, aos, e->stream); he = hipMemcpyAsync(coe, aos, bytes, hipMemcpyDeviceToHost, e->stream); }
    if (he == hipSuccess) he = hipStreamSynchronize(e->stream);
    if (he == hipSuccess) he = hipGetLastError();
    if (he != hipSuccess) return fail(XEE_ERR_HIP, std::string("xee_cal_coe: ") + hipGetErrorString(he));
    XCHK(cache_store(e, b, dx, dy));
    *err = 0;                                                   // :58
    return XEE_OK;
}

static int upload_coe_aos(xee_ensemble_t* e, const float* coe)
{
    float* aos = nullptr;
    const size_t bytes = sizeof(float) * 9 * (size_t)e->g.nx * e->g.ny;
    XCHK(aos_stage(e, &aos));
    hipError_t he = hipMemcpyAsync(aos, coe, bytes, hipMemcpyHostToDevice, e->stream);
    if (he == hipSuccess) { launch_coe_aos_to_soa(e->g, aos, e->coe, e->stream); he = hipStreamSynchronize(e->stream); }
    if (he == hipSuccess) he = hipGetLastError();
    if (he != hipSuccess) return fail(XEE_ERR_HIP, std::string("coe upload: ") + hipGetErrorString(he));
    return detect_uniform_planes(e);
}

int xee_do_elliptic(const float* psi, const float* coe, float* outdat, int nx, int ny, int* err)
{
    if (!psi || !coe || !outdat || !err) return fail(XEE_ERR_INVALID_ARGUMENT, "null argument");
    Tier1 sc;
    XCHK(sc.acquire(nx, ny));
    xee_ensemble_t* e = sc.e;
    XCHK(upload_coe_aos(e, coe));
    XCHK(xee_ensemble_upload(e, XEE_F_BC_INIT, 0, 1, psi));
    // interior only: seed the output plane with the caller's outdat so its rim survives
    XCHK(xee_ensemble_upload(e, XEE_F_RESIDUAL, 0, 1, outdat));
    launch_apply(e->g, e->bc, e->coe, e->res, e->stream);
    XCHK(check_launch("apply"));
    XCHK(xee_ensemble_download(e, XEE_F_RESIDUAL, 0, 1, outdat));
    *err = 1;                                                   // elliptic_tools.f90:73 (never cleared)
    return XEE_OK;
}

int xee_solve_elliptic(int* max_iter, int check_step, int converge_time, int lost_rate, float* strategy_r1,
                       float* strategy_r2, float alpha, float* dat, const float* coe, const float* f,
                       float* workspace, int nx, int ny, int* err, int debug)
{
    if (!max_iter || !strategy_r1 || !strategy_r2 || !dat || !coe || !f || !err)
        return fail(XEE_ERR_INVALID_ARGUMENT, "null argument");
    (void)workspace;   // scratch in the reference; the device keeps its own ping-pong plane
    const bool quiet_judge = (debug & XEE_DEBUG_NO_JUDGE) != 0;   // the Fortran shim prints it itself
    debug &= 0xff;
    if (!(*strategy_r1 > 0.0f) && !(*strategy_r2 > 0.0f)) {
        printf(" ERROR: [check_abs_err] and [check_rel_err] cannot both be non-positive.\n");
        return fail(XEE_ERR_INVALID_CRITERIA, "both criteria non-positive");
    }
    if (debug == 1 || debug == 2) {                             // :146-158
        printf(" ----- Solve Elliptic Inputs -----\n");
        printf("   max_iter       :  %d\n", *max_iter);
        printf("   strategy_r1    :  %.8G\n", *strategy_r1 > 0.0f ? *strategy_r1 : FLT_MAX);
        printf("   strategy_r2    :  %.8G\n", *strategy_r2 > 0.0f ? *strategy_r2 : FLT_MAX);
        printf("   alpha          :  %.8G\n", alpha);
        printf("   (nx, ny)       : ( %d ,  %d )\n", nx, ny);
        printf("   debug          :  %d\n", debug);
        printf("   check step     :  %d\n", check_step > 0 ? check_step : 100);
        printf("   converge time :  %d\n", converge_time > 0 ? converge_time : 10);
        printf(" ---------------------------------\n");
    }
    Tier1 sc;
    XCHK(sc.acquire(nx, ny));
    xee_ensemble_t* e = sc.e;
    XCHK(upload_coe_aos(e, coe));
    XCHK(cache_try(e));                                         // this thread's last cal_coe made exactly this coe?
    XCHK(xee_ensemble_upload(e, XEE_F_BC_INIT, 0, 1, dat));
    XCHK(xee_ensemble_upload(e, XEE_F_FORCING, 0, 1, f));
    XCHK(xee_ensemble_solve(e, *max_iter, check_step, converge_time, lost_rate, *strategy_r1, *strategy_r2, alpha,
                            debug == 2 ? XEE_SOLVE_TRACE : XEE_SOLVE_DEFAULT));
    g_last_tier1_sweep = e->last_sweep;
    XCHK(xee_ensemble_download(e, XEE_F_PSI, 0, 1, dat));
    const xee_case_record_t& r = e->rec[0];
    *err = r.err_bits;
    if (*max_iter > 0) {                                        // :253 (only reached when the loop ran)
        *max_iter = r.iters;
        *strategy_r1 = r.err_now;
        *strategy_r2 = r.ratio;
        if (!quiet_judge) xee_judge_error(*err);                // :254
    }
    return XEE_OK;
}

int xee_residual_norm(const float* res, int nx, int ny, float* sum_sq, float* rms, int* slow_chunks)
{
    if (!res || !sum_sq || !rms) return fail(XEE_ERR_INVALID_ARGUMENT, "null argument");
    Tier1 sc;
    XCHK(sc.acquire(nx, ny));
    xee_ensemble_t* e = sc.e;
    XCHK(xee_ensemble_upload(e, XEE_F_RESIDUAL, 0, 1, res));
    if (e->exact_serial) launch_seqsum(e->g, e->res, 1, nx - 1, 1, ny - 1, true, e->sums, nullptr, e->stream);
    else launch_exact_sumsq(e->g, e->res, e->ex_ws, e->sums, e->ex_slow, nullptr, e->stream);
    XCHK(check_launch("residual_norm"));
    int slow = 0;
    HIPCHK(hipMemcpyAsync(e->sums_host, e->sums, sizeof(float), hipMemcpyDeviceToHost, e->stream));
    if (!e->exact_serial) HIPCHK(hipMemcpyAsync(&slow, e->ex_slow, sizeof(int), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    *sum_sq = e->sums_host[0];
    *rms = sqrtf(*sum_sq / (float)((nx - 2) * (ny - 2)));      // :199, correctly rounded on the host as on the device
    if (slow_chunks) *slow_chunks = slow;
    return XEE_OK;
}

int xee_solver_coefficients(const float* rhoA, const float* rhoB, const float* rhoC, const float* rcuva,
                            const float* rho, float* solverA, float* solverB, float* solverC, int nr, int nz)
{
    if (!rhoA || !rhoB || !rhoC || !rcuva || !rho || !solverA || !solverB || !solverC)
        return fail(XEE_ERR_INVALID_ARGUMENT, "null argument");
    Tier1 sc;
    XCHK(sc.acquire(nr, nz));
    xee_ensemble_t* e = sc.e;
    std::vector<float> zr(nr, 0.f), zz(nz, 0.f);
    XCHK(xee_ensemble_set_profiles(e, zr.data(), zz.data(), rcuva, rho, zz.data(), 0.f, 0.f));
    XCHK(xee_ensemble_upload(e, XEE_F_A, 0, 1, rhoA));
    XCHK(xee_ensemble_upload(e, XEE_F_B, 0, 1, rhoB));
    XCHK(xee_ensemble_upload(e, XEE_F_C, 0, 1, rhoC));
    XCHK(alloc_planes(e, &e->sA));
    XCHK(alloc_planes(e, &e->sB));
    XCHK(alloc_planes(e, &e->sC));
    launch_solver_coef(e->g, e->A, e->B, e->C, e->rcuva, e->rho, e->sA, e->sB, e->sC, e->stream);
    XCHK(check_launch("solver_coef"));
    XCHK(xee_ensemble_download(e, XEE_F_SOLVER_A, 0, 1, solverA));
    XCHK(xee_ensemble_download(e, XEE_F_SOLVER_B, 0, 1, solverB));
    XCHK(xee_ensemble_download(e, XEE_F_SOLVER_C, 0, 1, solverC));
    return XEE_OK;
}

int xee_cal_eta(const float* rchi, float* eta, const float* ra, const float* rcuva, const float* rho,
                const float* exner, int nr, int nz)
{
    if (!rchi || !eta || !ra || !rcuva || !rho || !exner) return fail(XEE_ERR_INVALID_ARGUMENT, "null argument");
    Tier1 sc;
    XCHK(sc.acquire(nr, nz));
    xee_ensemble_t* e = sc.e;
    std::vector<float> zz(nz, 0.f);
    XCHK(xee_ensemble_set_profiles(e, ra, zz.data(), rcuva, rho, exner, 0.f, 0.f));
    XCHK(xee_ensemble_upload(e, XEE_F_PSI, 0, 1, rchi));
    XCHK(xee_ensemble_cal_eta(e));
    return xee_ensemble_download(e, XEE_F_ETA, 0, 1, eta);
}

int xee_cal_uw(const float* rpsi, float* u, float* w, const float* ra, const float* za, const float* rcuva,
               const float* rho, int nr, int nz)
{
    if (!rpsi || !u || !w || !ra || !za || !rcuva || !rho) return fail(XEE_ERR_INVALID_ARGUMENT, "null argument");
    Tier1 sc;
    XCHK(sc.acquire(nr, nz));
    xee_ensemble_t* e = sc.e;
    XCHK(xee_ensemble_set_profiles(e, ra, za, rcuva, rho, rho, 0.f, 0.f));
    XCHK(xee_ensemble_upload(e, XEE_F_PSI, 0, 1, rpsi));
    XCHK(xee_ensemble_cal_uw(e));
    XCHK(xee_ensemble_download(e, XEE_F_U, 0, 1, u));
    return xee_ensemble_download(e, XEE_F_W, 0, 1, w);
}

static int integrate_common(const float* wgt, const float* eta, const float* ra, const float* za,
                            const float* rcuva, const float* rho, int nr, int nz, float* sum)
{
    if (!wgt || !ra || !za || !rcuva || !rho || !sum) return fail(XEE_ERR_INVALID_ARGUMENT, "null argument");
    Tier1 sc;
    XCHK(sc.acquire(nr, nz));
    xee_ensemble_t* e = sc.e;
    XCHK(xee_ensemble_set_profiles(e, ra, za, rcuva, rho, rho, 0.f, 0.f));
    XCHK(xee_ensemble_upload(e, XEE_F_Q, 0, 1, wgt));
    XCHK(alloc_planes(e, &e->terms));
    if (eta) {
        XCHK(xee_ensemble_upload(e, XEE_F_ETA, 0, 1, eta));
        launch_terms_Qeta(e->g, e->Q, e->eta, e->terms, e->ra, e->za, e->rcuva, e->rho, e->stream);
    } else {
        launch_terms_weight(e->g, e->Q, e->terms, e->ra, e->za, e->rcuva, e->rho, e->stream);
    }
    XCHK(ee_sum(e, e->stream));
    XCHK(check_launch("integrate"));
    HIPCHK(hipMemcpyAsync(e->sums_host, e->sums, sizeof(float), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    *sum = e->sums_host[0];
    return XEE_OK;
}

int xee_integrate_weight_B(const float* weight, const float* ra, const float* za, const float* rcuva,
                           const float* rho, int nr, int nz, float* sum)
{
    return integrate_common(weight, nullptr, ra, za, rcuva, rho, nr, nz, sum);
}

int xee_cal_sum_Qeta(const float* Q, const float* eta, const float* ra, const float* za, const float* rcuva,
                     const float* rho, int nr, int nz, float* sum)
{
    if (!eta) return fail(XEE_ERR_INVALID_ARGUMENT, "null eta");
    return integrate_common(Q, eta, ra, za, rcuva, rho, nr, nz, sum);
}

int xee_cal_wtheta(const float* w_A, const float* theta_B, float* wtheta_B, int nr, int nz)
{
    if (!w_A || !theta_B || !wtheta_B) return fail(XEE_ERR_INVALID_ARGUMENT, "null argument");
    Tier1 sc;
    XCHK(sc.acquire(nr, nz));
    xee_ensemble_t* e = sc.e;
    XCHK(xee_ensemble_upload(e, XEE_F_W, 0, 1, w_A));
    XCHK(xee_ensemble_upload(e, XEE_F_Q, 0, 1, theta_B));
    XCHK(alloc_planes(e, &e->terms));
    launch_wtheta(e->g, e->w, e->Q, e->terms, e->stream);
    XCHK(check_launch("wtheta"));
    float* keep = e->Q;           // download the (nr-1,nz-1) result through the Q field's shape
    e->Q = e->terms;
    int st = xee_ensemble_download(e, XEE_F_Q, 0, 1, wtheta_B);
    e->Q = keep;
    return st;
}

int xee_rhs_thermal(const float* Q_B, const float* ra, const float* exner, float* rhs_O, int nr, int nz)
{
    if (!Q_B || !ra || !exner || !rhs_O) return fail(XEE_ERR_INVALID_ARGUMENT, "null argument");
    Tier1 sc;
    XCHK(sc.acquire(nr, nz));
    xee_ensemble_t* e = sc.e;
    std::vector<float> zz(nz, 0.f);
    XCHK(xee_ensemble_set_profiles(e, ra, zz.data(), ra, zz.data(), exner, 0.f, 0.f));
    XCHK(xee_ensemble_upload(e, XEE_F_Q, 0, 1, Q_B));
    XCHK(alloc_planes(e, &e->terms));
    launch_rhs_thermal(e->g, e->Q, e->ra, e->exner, e->terms, e->f, e->stream);
    XCHK(check_launch("rhs_thermal"));
    return xee_ensemble_download(e, XEE_F_FORCING, 0, 1, rhs_O);
}

int xee_rhs_from_B(const float* b_B, float* f_O, int nr, int nz)
{
    if (!b_B || !f_O) return fail(XEE_ERR_INVALID_ARGUMENT, "null argument");
    Tier1 sc;
    XCHK(sc.acquire(nr, nz));
    xee_ensemble_t* e = sc.e;
    XCHK(xee_ensemble_upload(e, XEE_F_Q, 0, 1, b_B));
    launch_rhs_from_B(e->g, e->Q, e->f, e->stream);
    XCHK(check_launch("rhs_from_B"));
    return xee_ensemble_download(e, XEE_F_FORCING, 0, 1, f_O);
}

}  // extern "C"
