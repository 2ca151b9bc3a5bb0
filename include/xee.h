/*
 * xee.h -- C-ABI of libxee, the MI355X (gfx950) implementation of the elliptic streamfunction
 * solve and energy-efficiency diagnostic of meteorologytoday/XLab-EE-fortran.
 *
 * This header is the drop-in boundary.  Every entry point names the reference interface it
 * replaces (path:line under the reference tree).  Plain C types only; no HIP / torch types.
 *
 * Conventions shared by all entry points
 *   - float = IEEE binary32, int = int32; all arrays are caller-owned.
 *   - 2-D arrays use the reference's Fortran layout f(n1,n2), column-major: C index j*n1 + i with
 *     i the r index (fastest) and j the z index -- identical to the reference's .bin files
 *     (xtt-lib-fortran/field_tools.f90:30-76) and to a numpy (n2,n1) C-order array.
 *   - `coe` on the Tier-1 calls keeps the reference's AoS layout coe(9,nx,ny).
 *   - Every function returns a status code (XEE_OK == 0).  Nothing calls exit(): where the
 *     reference STOPs (elliptic_tools.f90:126-129) the status is XEE_ERR_INVALID_CRITERIA and the
 *     Fortran shim (xlab-ee-fortran_amd/fortran/elliptic_tools.f90) performs the STOP.
 *   - Results are bit-identical to the reference's fp32 arithmetic (no FMA contraction,
 *     correctly-rounded division/sqrt, sequential fp32 sums in the reference's loop order).
 *   - There is NO CPU fallback: without a usable gfx950 device every compute entry point returns
 *     XEE_ERR_NO_DEVICE.
 */
#ifndef XEE_H
#define XEE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define XEE_VERSION 100

/* status codes */
#define XEE_OK                    0
#define XEE_ERR_INVALID_CRITERIA  1   /* strategy_r1 <= 0 and strategy_r2 <= 0 (elliptic_tools.f90:126-129) */
#define XEE_ERR_INVALID_ARGUMENT  2
#define XEE_ERR_NO_DEVICE         3
#define XEE_ERR_HIP               4
#define XEE_ERR_ALLOC             5
#define XEE_ERR_TIMEOUT           6   /* a collective of the multi-GPU path did not complete in time (a rank is missing) */

/* solver error bitmask, xtt-lib-fortran/elliptic_tools.f90:3-4 */
#define XEE_ERRBIT_OVER_MAX_ITERATION 1
#define XEE_ERRBIT_EXPLODE            2

/* operator variants, src/diagnose/variables.f90:13-15 and diagnose.f90:6,33 */
#define XEE_BAROTROPIC 0   /* solverB_B := 0 */
#define XEE_BAROCLINIC 1   /* solverB_B := saved_solverB_B */

int         xee_version(void);
const char *xee_last_error(void);          /* text of the last failure on this thread */
int         xee_device_count(int *count);  /* number of visible HIP devices (0 is not an error) */

/* Run-time options (integers, by name).  The library reads NO environment variables.  Defaults are process-wide
 * and are copied into every ensemble at creation (the Tier-1 entry points create theirs internally);
 * xee_ensemble_set_option changes one ensemble.  Every combination gives bit-identical results -- the options
 * choose between implementations of the same arithmetic (tests run them against each other and the oracle):
 *   "tb_T"             iterations per temporally blocked sweep launch, 1..16 (8); 1 = the streaming kernel
 *   "kernel"           0 auto | 1 nine coefficient planes (k_sweep_tb) | 2 generating planes with the IEEE
 *                      division (k_sweep_tb4) | 3 generating planes, hoisted reciprocal (k_sweep_q) | 4 as 3, with the plain
 *                      launches streamed along z by single waves (k_sweep_z) wherever that kernel applies.  Auto is 3
 *                      whenever xee_ensemble_assemble / xee_cal_coe built the planes, and 4 for a batch that fills the
 *                      device with k_sweep_z (e.g. 32 cases of 1024x512): BAROTROPIC operator -- or BAROCLINIC, for members
 *                      that all share A and C (generating planes AP, CQ; each may have its own B) --, alpha = 1, every member's
 *                      |coe5| within [2^-120, 2^120] and its smallest interior |forcing| at least 2^26 times
 *                      max(2^-100, max |coe5| 2^-120) -- tested per solve (one pass over three planes).  Launches of 6 iterations
 *                      (BAROCLINIC: 4); the residual-writing launch of a check block too where the block is whole launches
 *                      (100 = 16 x 6 + 4 = 25 x 4), else it stays k_sweep_q's
 *   "z_pairs"          chunks of rows per strip of k_sweep_z, in pairs: 1..64, 0 = chosen from the batch (0)
 *   "share_planes"     0: every member reads its own copy of the operator's generating planes even where all members hold the
 *                      same ones (1: such planes are read from one copy -- tested once per assembled operator)
 *   "q_exact_only"     1: k_sweep_q / k_sweep_z run their exact (fallback) body on every tile / wave
 *   "elision"          0: k_sweep_tb loads coefficient planes even when they are uniform over the batch (1)
 *   "tile"             NW*100+R: waves x rows per lane of a blocked tile, 0 = the kernel's default
 *   "sync_checks"      1: residual checks synchronous instead of pipelined behind the sweeps (0)
 *   "exact_sum_serial" 1: order-exact sums by a single-lane walk instead of the parallel algorithm (0)
 * Unknown keys and out-of-range values return XEE_ERR_INVALID_ARGUMENT. */
int xee_set_default_option(const char *key, int value);
int xee_reset_default_options(void);

/* ---------------------------------------------------------------------------------------------
 * Tier 1 -- drop-in replacements, host arrays in / host arrays out.
 * ------------------------------------------------------------------------------------------- */

/* replaces cal_coe, xtt-lib-fortran/elliptic_tools.f90:8-60.
 * a(nx-1,ny-2), b(nx-1,ny-1), c(nx-2,ny-1) -> coe(9,nx,ny); only interior entries are written.
 * *err: 1 on entry to the computation, 0 on success (as the reference). */
int xee_cal_coe(const float *a, const float *b, const float *c, float *coe,
                float dx, float dy, int nx, int ny, int *err);

/* replaces do_elliptic, elliptic_tools.f90:64-90.  Interior of outdat only; *err := 1 (sic). */
int xee_do_elliptic(const float *psi, const float *coe, float *outdat, int nx, int ny, int *err);

/* replaces solve_elliptic, elliptic_tools.f90:93-265.  *max_iter, *strategy_r1, *strategy_r2 and
 * *err are overwritten with iterations used, final RMS residual, final |ratio| and the error
 * bitmask (:253).  `workspace` is scratch (unspecified content on return, as for the reference's
 * callers).  debug: 1/2 echo the inputs, 2 also prints the per-check trace (:146-158,:202-204).
 * OR-ing XEE_DEBUG_NO_JUDGE into debug suppresses the judge_error text of :254 for hosts that print
 * it through their own I/O unit (the Fortran shim does, to keep its stdout ordering). */
#define XEE_DEBUG_NO_JUDGE 0x100
int xee_solve_elliptic(int *max_iter, int check_step, int converge_time, int lost_rate,
                       float *strategy_r1, float *strategy_r2, float alpha,
                       float *dat, const float *coe, const float *f, float *workspace,
                       int nx, int ny, int *err, int debug);

/* which sweep kernel this thread's last xee_solve_elliptic launched (name, iterations per launch, tile extent): when the
 * coe it was handed is bit for bit what this thread's last xee_cal_coe returned, the operator's generating planes are
 * rebuilt from that call's a, b, c and the solve takes the fast kernel (k_sweep_q); any other coe takes the nine-plane
 * kernel (k_sweep_tb).  Same results either way. */
int xee_last_sweep_info(char *name, int name_len, int *T, int *tile_w, int *tile_h);

/* The Tier-1 entry points work through a one-member device ensemble that is KEPT per calling thread and reused while the
 * grid (nx, ny) and the device stay the same -- a reused one is put back into the state of a fresh one first (planes
 * zero-filled, options from the process defaults).  It holds device memory for about forty planes of the last grid, plus
 * the copy of this thread's last xee_cal_coe; this call frees both (the next Tier-1 call simply builds them again).
 * A call that fails drops the cached ensemble by itself. */
int xee_tier1_release(void);
/* test aid: out[0] device ensembles created so far in this process, out[1] device staging buffers the Tier-1 entry points
 * allocated, out[2] Tier-1 caches released because their thread ended, out[3] reserved */
int xee_debug_counters(int64_t *out /* 4 */);

/* replaces the residual norm inside solve_elliptic, elliptic_tools.f90:193-199, as a stand-alone call:
 * *sum_sq = the sequential fp32 sum (i outer, j inner) of res(i,j)**2 over the interior, bit for bit;
 * *rms = sqrt(sum_sq / ((nx-2)*(ny-2))).  slow_chunks (may be NULL) reports how many 32-term chunks the
 * parallel algorithm had to add term by term (xee_exactsum.hip). */
int xee_residual_norm(const float *res, int nx, int ny, float *sum_sq, float *rms, int *slow_chunks);

/* replaces judge_error, elliptic_tools.f90:333-358 (prints the same texts to stdout). */
void xee_judge_error(int err);

/* replaces the 1-D profile set-up, src/diagnose/initialize-variables.f90:44-67 (+ the domain lines of
 * read-input.f90:55-76).  HOST-side work by design: rho needs glibc powf and the spherical geometry
 * glibc cosf/sinf, which no device libm reproduces bit for bit (hard part 5).
 * geometry 0: domain = {Lr1, Lr2, Lz1, Lz2}; geometry 1 (spherical): domain = {planet_radius, Lz1, Lz2}
 * with latitudes in DEGREES fed to cos/sin exactly as the reference does (quirk 9).
 * density_mode 0: DENSITY_NORMAL, 1: DENSITY_BOUSSINESQ.  sin_table may be NULL. */
int xee_profiles(int geometry, int density_mode, const float *domain, int nr, int nz,
                 float *ra, float *za, float *rcuva, float *exner, float *rho, float *sin_table,
                 float *dr, float *dz);

/* replaces the solver-coefficient assembly loops, src/diagnose/initialize-variables.f90:72-95.
 * rhoA/rhoB/rhoC (nr,nz), rcuva[nr], rho[nz] -> solverA (nr-1,nz-2), solverB (nr-1,nz-1),
 * solverC (nr-2,nz-1). */
int xee_solver_coefficients(const float *rhoA, const float *rhoB, const float *rhoC,
                            const float *rcuva, const float *rho,
                            float *solverA, float *solverB, float *solverC, int nr, int nz);

/* replaces cal_eta, src/diagnose/quick-tools1.f90:1-13 (internal procedure of program main that
 * reads host-associated ra, rcuva, rho, exner -- passed explicitly here).  eta (nr-1,nz). */
int xee_cal_eta(const float *rchi, float *eta, const float *ra, const float *rcuva,
                const float *rho, const float *exner, int nr, int nz);

/* replaces cal_uw, src/diagnose/quick-tools1.f90:15-41.  u (nr,nz-1), w (nr-1,nz). */
int xee_cal_uw(const float *rpsi, float *u, float *w, const float *ra, const float *za,
               const float *rcuva, const float *rho, int nr, int nz);

/* replaces integrate_weight_B / cal_sum_Q / cal_sum_wtheta,
 * src/old-diagnose/diagnose.f90:1029-1048,1050-1071,1094-1113: weight (nr-1,nz-1) -> *sum,
 * sequential fp32 accumulation in the reference's order (i outer, j inner). */
int xee_integrate_weight_B(const float *weight, const float *ra, const float *za,
                           const float *rcuva, const float *rho, int nr, int nz, float *sum);

/* replaces cal_sum_Qeta, src/old-diagnose/diagnose.f90:1073-1092.  Q (nr-1,nz-1), eta (nr-1,nz). */
int xee_cal_sum_Qeta(const float *Q, const float *eta, const float *ra, const float *za,
                     const float *rcuva, const float *rho, int nr, int nz, float *sum);

/* replaces cal_wtheta, src/old-diagnose/diagnose.f90:1117-1127. */
int xee_cal_wtheta(const float *w_A, const float *theta_B, float *wtheta_B, int nr, int nz);

/* replaces the RHS source-term assembly of the legacy sweep,
 * src/old-diagnose/diagnose.f90:383-387 + 395-406 (thermal: g0/theta0 d/dr(Q/(Cp exner)) averaged
 * to O points) and :523-538 (f = -1/4 sum of the 4 neighbouring B cells).  Outputs are (nr,nz). */
int xee_rhs_thermal(const float *Q_B, const float *ra, const float *exner, float *rhs_O, int nr, int nz);
int xee_rhs_from_B(const float *b_B, float *f_O, int nr, int nz);
/* replaces the momentum source term RHS_rpsi_mom, src/old-diagnose/diagnose.f90:412-436:
 * -(T(i,j)+T(i-1,j))/rcuva(i)**2 with T = d_dz_B2A(sqrt(m2)*F); m2, F on B cells (nr-1,nz-1) -> (nr,nz).
 * Rows the reference reads from an uninitialised workspace (j = nz-1) are defined as 0 (quirk 16). */
int xee_rhs_momentum(const float *m2_B, const float *F_B, const float *za, const float *rcuva,
                     float *rhs_O, int nr, int nz);

/* replaces the six finite-difference helpers of src/diagnose/quick-tools2.f90:1-85 (internal procedures
 * of program main reading host-associated ra, za, rcuva -- passed explicitly here).  Shapes (n1,n2):
 *   XEE_D_DZ_B2A     (nr-1,nz-1) -> (nr-1,nz)   rows J=2..nz-2 written, others keep the caller's bytes   :1-13
 *   XEE_D_DZ_O2C     (nr,nz)     -> (nr,nz-1)                                                         :16-25
 *   XEE_D_DR_B2B     (nr-1,nz-1) -> (nr-1,nz-1) one-sided at the ends                                  :27-43
 *   XEE_D_DR_B2C     (nr-1,nz-1) -> (nr,nz-1)   columns I=2..nr-1 written                               :45-57
 *   XEE_D_DR_O2A     (nr,nz)     -> (nr-1,nz)                                                         :59-68
 *   XEE_D_RCUVDR_O2A (nr,nz)     -> (nr-1,nz)   d_dr_O2A / ((rcuva(I)+rcuva(I+1))/2)                   :71-85 */
enum { XEE_D_DZ_B2A = 0, XEE_D_DZ_O2C = 1, XEE_D_DR_B2B = 2, XEE_D_DR_B2C = 3, XEE_D_DR_O2A = 4, XEE_D_RCUVDR_O2A = 5 };
int xee_fd_stencil(int which, const float *from, float *to, const float *ra, const float *za,
                   const float *rcuva, int nr, int nz);

/* replaces cal_exchange_conversion, src/old-diagnose/diagnose.f90:1143-1174 (K9), with REAL r, dr, dz: the
 * reference declares them `integer` (:1146), which truncates them and makes its number meaningless, so
 * this entry is a documented fix, not a parity target.  bndconv (nr-1,2); *sum accumulated sequentially. */
int xee_exchange_conversion(const float *rpsi, const float *rchi, const float *rhoC, const float *ra,
                            const float *za, const float *rho, float *bndconv, float *sum, int nr, int nz);
/* ... and exactly as the legacy program computes it, truncation included (r = int((ra(i)+ra(i+1))/2), dr = int(ra(2)-ra(1)),
 * dz = int(za(2)-za(1)), :1146-1152): bit-identical to the reference's bndconv files and sums (pinned against the legacy
 * binary, tests/test_oracle_legacy_chain.py); only meaningful on domains measured in units much smaller than dr, dz. */
int xee_exchange_conversion_legacy(const float *rpsi, const float *rchi, const float *rhoC, const float *ra,
                                   const float *za, const float *rho, float *bndconv, float *sum, int nr, int nz);

/* Stage II of the legacy TENDENCY mode (src/old-diagnose/diagnose.f90:462-518): forecast of the thermal field
 * between the two inversions.  Grids: A (nr-1,nz), B (nr-1,nz-1), C (nr,nz-1), packed, i fastest.
 * replaces the loop at :470-479: dtheta/dt = J - theta0/g0 (rhoA w)-bar-z + theta0/g0 (rhoB u)-bar-r on B cells */
int xee_theta_tendency(const float *JJ_B, const float *rhoA_A, const float *w_A, const float *rhoB_C, const float *u_C,
                       float *dtheta_dt_B, int nr, int nz);
/* replaces relativeTheta, :893-912: theta0 at (1,1), integrated along r on the first row, then up every column */
int xee_relative_theta(const float *dtheta_dz_A, const float *dtheta_dr_C, const float *ra, const float *za,
                       float *theta_B, int nr, int nz);
/* replaces :489-518.  theta_B in: dtheta/dt, out: the forecast theta; rhoB_B, rhoA_A, rhoB_C updated in place;
 * b_anomaly_B and solver_b_anomaly_B written.  One deliberate difference: the legacy code adds to row nz-1 of rhoA_A a
 * workspace row d_dz_B2A never wrote (uninitialised); that row is left alone here. */
int xee_theta_anomaly_update(float *theta_B, float testing_dt, float *rhoB_B, float *rhoA_A, float *rhoB_C,
                             float *b_anomaly_B, float *solver_b_anomaly_B, const float *ra, const float *za,
                             const float *rcuva, const float *rho, int nr, int nz);

/* replaces the staggered averages of the legacy program's input planes, src/old-diagnose/diagnose.f90:329-354:
 * rhoA_A (nr-1,nz), rhoB_C (nr,nz-1), rhoB_B (nr-1,nz-1, = b_basic_B :348), rhoC_C (nr,nz-1). */
int xee_legacy_averages(const float *rhoA_in, const float *rhoB_in, const float *rhoC_in, float *rhoA_A, float *rhoB_C,
                        float *rhoB_B, float *rhoC_C, int nr, int nz);
/* replaces :383-387, JJ_B = Q / (Cp * exner(j)) on B cells (the field the program writes as J-B.bin) */
int xee_JJ_B(const float *Q_B, const float *exner, float *JJ_B, int nr, int nz);
/* elementwise fp32 sum of two packed fields (f = RHS_rpsi_thm + RHS_rpsi_mom :456, solver_b_basic + solver_b_anomaly :572,
 * the chi sums of :736-737) */
int xee_add_fields(const float *a, const float *b, float *out, size_t n);

/* replaces the m2 integration, src/old-diagnose/diagnose.f90:359-380, in its INTENDED form (the legacy lines read
 * m2(0,j) and use stale loop indices in the first column): m2(1,j) from its own row, then
 * m2(i,j) = m2(i-1,j) + rcuva(i)**3.0 * rhoC_C(i,j) * (ra(i+1)-ra(i-1)) / 2.0 [/ sin_table(i)], a sequential fp32 sum along r.
 * rhoC_C (nr,nz-1) -> m2_B (nr-1,nz-1); sin_table NULL for the cylindrical geometry.  The powers (glibc powf on the
 * 1-D profile, as the Fortran runtime does) are taken on the host. */
int xee_m2_from_C(const float *rhoC_C, const float *ra, const float *rcuva, const float *sin_table, float *m2_B,
                  int nr, int nz);

/* ---------------------------------------------------------------------------------------------
 * Tier 2 -- device-resident ensemble of independent cases on ONE GPU.
 * All members share nr, nz and the 1-D profiles; each has its own A, B, C, forcing, bc_init, Q.
 * ------------------------------------------------------------------------------------------- */

typedef struct xee_ensemble xee_ensemble_t;

/* per-member field identifiers for upload / download / device_plane */
enum {
    XEE_F_A = 0,        /* rhoA_in (nr,nz)            initialize-variables.f90:33 */
    XEE_F_B = 1,        /* rhoB_in (nr,nz)            :34 */
    XEE_F_C = 2,        /* rhoC_in (nr,nz)            :35 */
    XEE_F_BC_INIT = 3,  /* bc_init_in (nr,nz)         :36 */
    XEE_F_FORCING = 4,  /* f (nr,nz)                  :38-42, diagnose.f90:2 */
    XEE_F_PSI = 5,      /* strf: solution (nr,nz)     diagnose.f90:13-15 */
    XEE_F_SOLVER_A = 6, /* solverA_A (nr-1,nz-2)      initialize-variables.f90:72-77 */
    XEE_F_SOLVER_B = 7, /* solverB_B (nr-1,nz-1)      :80-86 */
    XEE_F_SOLVER_C = 8, /* solverC_C (nr-2,nz-1)      :90-95 */
    XEE_F_ETA = 9,      /* eta (nr-1,nz)              quick-tools1.f90:1-13 */
    XEE_F_U = 10,       /* u_C (nr,nz-1)              quick-tools1.f90:15-41 */
    XEE_F_W = 11,       /* w_A (nr-1,nz) */
    XEE_F_Q = 12,       /* heating Q on B cells (nr-1,nz-1), EE scalars (old-diagnose :1050-1092) */
    XEE_F_RESIDUAL = 13,/* residual plane of the last check iteration (nr,nz), interior valid */
    XEE_F_COE0 = 16     /* XEE_F_COE0+k, k=0..8: coefficient plane k (nr,nz), interior valid */
};

/* One fixed-size record per case; this is what the multi-GPU gather moves. */
typedef struct {
    int32_t case_id;     /* global case index (caller-assigned, see xee_ensemble_set_case_ids) */
    int32_t iters;       /* iterations used           (max_iter on return, elliptic_tools.f90:253) */
    int32_t err_bits;    /* XEE_ERRBIT_*              */
    int32_t reserved;
    float   err_now;     /* final RMS residual        (strategy_r1 on return) */
    float   ratio;       /* final |ratio|             (strategy_r2 on return) */
    float   sum_Q;       /* integrate_weight_B(Q)     old-diagnose :1069 */
    float   sum_Qeta;    /* cal_sum_Qeta(Q, eta)      old-diagnose :1073-1092 */
    float   efficiency;  /* sum_Qeta / sum_Q          old-diagnose :792-793 */
    float   scalars[23]; /* reserved for further EE scalars */
} xee_case_record_t;     /* 128 bytes */

typedef struct {
    double solve_ms;        /* device time of the last xee_ensemble_solve (HIP events on its stream) */
    double sweep_ms;        /* of which: sweep kernels (events around the sweep launches only) */
    int64_t sweep_launches; /* number of sweep kernel launches in the last solve */
    int64_t iterations;     /* Jacobi iterations enqueued in the last solve (per member) */
    int64_t check_launches; /* residual-norm / state-machine launches */
    int64_t redo_tiles;        /* tiles of the last solve that the blocked sweep's optimistic body rejected (non-finite iterates,
                                  values outside the range of its strength-reduced division, -0 iterates, exotic coefficients)
                                  and the exact body recomputed in the same workgroup: in residual-writing launches ... */
    int64_t inline_redo_tiles; /* ... and in the plain launches; 0 on the common path */
    int64_t tiles;             /* tile executions of the blocked sweep in the last solve (launches x tiles per launch):
                                  (redo_tiles + inline_redo_tiles) / tiles = the fraction that took the exact body */
    int64_t sweep_streams;     /* 1, or 2 when a large batch was solved as two halves of its member range side by side: then
                                  sweep_launches counts both halves' launches (each over half the members) and sweep_ms is the
                                  longer of the two halves' sweep times -- the average duration of one launch, as a profiler
                                  sees it, is sweep_ms / (sweep_launches / sweep_streams) */
    int64_t z_launches;        /* of sweep_launches: launches of the streaming k_sweep_z (large batches; the others of
                                  such a solve -- the residual-writing launch of every check block and a remainder -- are
                                  k_sweep_q's) */
    double  z_ms;              /* device time between the first and the last k_sweep_z launch of every check block (HIP events on the
                                  launches' stream; large batches only; two halves: the longer of the two) and the number of launches */
    int64_t z_timed_launches;  /* between those events: one k_sweep_z launch lasts z_ms / (z_timed_launches / sweep_streams) */
    int64_t shared_planes;     /* bit k: generating plane k of the operator (0 AP, 1 CQ, 2 BX, 3 BY) held the same bits for every
                                  member in the last solve, and k_sweep_q / k_sweep_z read ONE copy of it for all of them (members
                                  that share their operator: an ensemble that varies the forcing only; option "share_planes") */
} xee_timing_t;

/* flags of xee_ensemble_solve */
#define XEE_SOLVE_DEFAULT        0
#define XEE_SOLVE_SINGLE_STEP    1   /* one iteration per launch (streaming kernel only)          */
#define XEE_SOLVE_KEEP_PSI       2   /* start from the current PSI instead of copying BC_INIT      */
#define XEE_SOLVE_TRACE          4   /* per-check "Iter/err_now/ratio" trace (debug==2, :202-204)    */

int xee_ensemble_create(xee_ensemble_t **ens, int nr, int nz, int n_members, int device);
int xee_ensemble_destroy(xee_ensemble_t *ens);
/* run on a caller-provided hipStream_t (e.g. torch's current stream); NULL = the ensemble's own */
int xee_ensemble_set_stream(xee_ensemble_t *ens, void *hip_stream);
int xee_ensemble_set_case_ids(xee_ensemble_t *ens, int first_case_id);
int xee_ensemble_set_option(xee_ensemble_t *ens, const char *key, int value);
/* which sweep kernel the last solve launched (name, iterations per launch, tile extent in points) */
int xee_ensemble_sweep_info(xee_ensemble_t *ens, char *name, int name_len, int *T, int *tile_w, int *tile_h);
/* A13 profiles are host-side work (glibc powf/cosf, initialize-variables.f90:44-67); hand them in */
int xee_ensemble_set_profiles(xee_ensemble_t *ens, const float *ra, const float *za,
                              const float *rcuva, const float *rho, const float *exner,
                              float dr, float dz);
/* host <-> device for `count` consecutive members starting at `first`; host holds count packed
 * (n1,n2) arrays of the field's logical shape */
int xee_ensemble_upload(xee_ensemble_t *ens, int field, int first, int count, const float *host);
int xee_ensemble_download(xee_ensemble_t *ens, int field, int first, int count, float *host);
/* raw device view of a field so that a producer already on the GPU can fill it in place:
 * element (i,j) of member m lives at dptr[m*member_stride + j*pitch + i] */
int xee_ensemble_device_plane(xee_ensemble_t *ens, int field, void **dptr, int *pitch,
                              size_t *member_stride, int *n1, int *n2);
/* f := -rhoB_in (DYNAMIC_EFFICIENCY, initialize-variables.f90:40-42) */
int xee_ensemble_forcing_from_B(xee_ensemble_t *ens);
/* f := g0/theta0 * d/dr( Q/(Cp*exner) ) averaged to O points, from the uploaded Q (B cells): the thermal
 * source term of the legacy sweep, old-diagnose/diagnose.f90:383-387,395-406;  f := 0 (:553) */
int xee_ensemble_forcing_thermal(xee_ensemble_t *ens);
int xee_ensemble_forcing_zero(xee_ensemble_t *ens);
/* Device-resident copies of a field (all members): save the current content of `field` (any XEE_F_* below XEE_F_COE0) into
 * stash slot `slot` (0..63, allocated on first use), or copy a slot back into the field.  A host that cycles through several
 * right-hand sides or boundary fields per operator -- the source-term sweep, old-diagnose/diagnose.f90:548-673: rchi_bc,
 * f = 0, f = -dB, f = -B0 in turn -- keeps them in HBM and switches without touching PCIe. */
/* Member window: the forcing_* calls above and field_save / field_restore below act on members [first, first + count)
 * only until the window is changed (count = 0: all members again).  Lets one batch hold cases that share an operator
 * but differ in their right-hand side -- the sweep's {-B, thermal, zero} kinds of one timestep. */
int xee_ensemble_member_window(xee_ensemble_t *ens, int first, int count);
int xee_ensemble_field_save(xee_ensemble_t *ens, int field, int slot);
int xee_ensemble_field_restore(xee_ensemble_t *ens, int field, int slot);
/* solver coefficients (K1) then the 9-point stencil (K2) for the chosen operator variant */
int xee_ensemble_assemble(xee_ensemble_t *ens, int variant);
/* the 9-point stencil (K2) from the solver coefficients CURRENTLY held in XEE_F_SOLVER_A / _B / _C (uploaded, or left by
 * a previous assemble): what the legacy sweep needs for L(A, B0+dB, C) (old-diagnose/diagnose.f90:572) and L(A, 0, C)
 * (:553).  use_b == 0 takes solverB_B = 0. */
int xee_ensemble_assemble_solver(xee_ensemble_t *ens, int use_b);
/* strf := bc_init_in, then solve_elliptic on every member; per-member results in the records */
int xee_ensemble_solve(xee_ensemble_t *ens, int max_iter, int check_step, int converge_time,
                       int lost_rate, float strategy_r1, float strategy_r2, float alpha, int flags);
/* OPT-IN, NON-PARITY smoother (the "red/black SOR" the north star words): 4-colour
 * successive over-relaxation, 0 < omega < 2, same stopping rule and records as xee_ensemble_solve but the
 * check uses the residual of the current iterate.  Converges to the same fixed point in far fewer
 * iterations; its iterates are NOT the reference's, so parity holds only to solver tolerance. */
int xee_ensemble_solve_sor(xee_ensemble_t *ens, int max_iter, int check_step, int converge_time,
                           int lost_rate, float strategy_r1, float strategy_r2, float omega, int flags);
int xee_ensemble_cal_eta(xee_ensemble_t *ens);
int xee_ensemble_cal_uw(xee_ensemble_t *ens);
/* sum_Q, sum_Qeta, efficiency from Q and the current eta (order-exact fp32 sums) */
int xee_ensemble_ee_scalars(xee_ensemble_t *ens);
/* diagnostic of the order-exact signed sums of the last xee_ensemble_ee_scalars: per member, how many 32-term chunks of
 * each integral the parallel algorithm had to add term by term (of (nr-1)*ceil((nz-1)/32)): small on the fast path */
int xee_ensemble_ee_slow_chunks(xee_ensemble_t *ens, int *sum_Q_slow /* n_members */, int *sum_Qeta_slow /* n_members */);
int xee_ensemble_records(xee_ensemble_t *ens, xee_case_record_t *out /* n_members */);
int xee_ensemble_timing(xee_ensemble_t *ens, xee_timing_t *out);
/* measurement aid: `iters` bare sweep iterations (no checks) of kernel `variant`
 * (0 = streaming, 1 = temporally blocked) between HIP events; does not touch the records */
int xee_ensemble_time_sweeps(xee_ensemble_t *ens, int iters, int variant, float alpha,
                             double *elapsed_ms, int64_t *launches);
/* validation aid for the blocked sweep's strength-reduced update (csrc/xee_sweepq.hip): draws
 * blocks*256*per_thread operand pairs (a, d), |d| in [2^d_exp_lo, 2^(d_exp_hi+1)), a over the range the
 * kernel admits for that d, and counts the pairs whose hoisted-reciprocal quotient differs from the IEEE
 * division a/d (the quotient of elliptic_tools.f90:238).  Must report 0. */
int xee_quotient_selfcheck(int device, uint64_t seed, int blocks, int per_thread, int d_exp_lo, int d_exp_hi,
                           uint64_t *mismatches, float *first_bad /* 2: a, d of one failing pair */);

/* ---------------------------------------------------------------------------------------------
 * The LEGACY program as ONE call: src/old-diagnose/diagnose.f90:283-841, TENDENCY mode -- source
 * terms, Stage I -> II -> III (up to six chi solves), integral check, exchange conversion, efficiency.txt -- for a batch of
 * cases that share grid, mode and solver lines.  The stage order lives in csrc/xee_legacy.hip and nowhere else: the Fortran
 * host bin/xee-old-diagnose and the Python driver legacy_diagnose.py both call this.  Solver lines: strategy 1 (stop when
 * err_now < r at a check) or 2 (|ratio| < r, ten passes); max_iter a multiple of 100.  Defined where the legacy program has
 * undefined behaviour: Q, F are (nr-1,nz-1) B-cell fields as given; m2 in its intended form; the
 * "w/ boundary" sums are 0 without a chi boundary field.
 * ------------------------------------------------------------------------------------------- */
typedef struct xee_legacy xee_legacy_t;
typedef struct {
    int   geometry;        /* 0 CYLINDRICAL, 1 SPHERICAL */
    int   density_mode;    /* 0 DENSITY_NORMAL, 1 DENSITY_BOUSSINESQ */
    float domain[4];       /* as xee_profiles */
    float testing_dt;      /* :489 */
    int   op_mode;         /* 0 BAROTROPIC, 1 BAROCLINIC, 2 BARO_ALL: which of L(B=0), L(B0+dB) Stage III and the checks use */
    int   psi_strategy;  float psi_r;  int psi_max_iter;  float psi_alpha;     /* the rpsi solver line (:147-151) */
    int   chi_strategy;  float chi_r;  int chi_max_iter;  float chi_alpha;     /* the rchi solver line */
    int   legacy_int;      /* exchange conversion with the program's INTEGER r, dr, dz (:1146); 0: real metric factors */
    int   chained;         /* first guesses carried from solve to solve as the program does (:545); 0: every solve from its own start */
    int   device;          /* -1: the current device */
    int   first_case_id;
} xee_legacy_config_t;
/* A, B, C: n_cases planes (nr,nz); Q, F: n_cases planes (nr-1,nz-1); psi_bc, chi_bc: n_cases planes (nr,nz) or NULL */
int xee_legacy_run(xee_legacy_t **result, const xee_legacy_config_t *cfg, int nr, int nz, int n_cases,
                   const float *A, const float *B, const float *C, const float *Q, const float *F,
                   const float *psi_bc, const float *chi_bc);
int xee_legacy_destroy(xee_legacy_t *result);
/* the files the legacy program writes (same names, shapes (n1,n2) with n1 contiguous), in the order they were produced */
int xee_legacy_field_count(const xee_legacy_t *result, int *n);
int xee_legacy_field_info(const xee_legacy_t *result, int index, char *name, int name_len, int *n1, int *n2);
int xee_legacy_field(const xee_legacy_t *result, int icase, const char *name, float *out);
/* the scalars behind efficiency.txt: sum_Q, sum_dtheta_dt, sum_Qeta_<op>_<rhs>, sum_wtheta_<op>_JF, sum_bndconv[2]_<op> */
int xee_legacy_scalar_count(const xee_legacy_t *result, int *n);
int xee_legacy_scalar_info(const xee_legacy_t *result, int index, char *key, int key_len);
int xee_legacy_scalar(const xee_legacy_t *result, int icase, const char *key, float *value);
/* (iterations, final residue) of every solve of the case, in the program's order ("Relaxation uses N steps ...") */
int xee_legacy_relax(const xee_legacy_t *result, int icase, int *n_solves, int *iters, float *err_now, int capacity);
int xee_legacy_record(const xee_legacy_t *result, int icase, xee_case_record_t *rec);
/* efficiency.txt (:779-841, every label; numbers as ES16.9 so that an fp32 value survives the round trip) */
int xee_legacy_efficiency_text(const xee_legacy_t *result, int icase, double elapsed_s, char *text, int text_len);
/* every file of the case under its legacy name, and efficiency.txt, into `directory` */
int xee_legacy_write_outputs(const xee_legacy_t *result, int icase, const char *directory, double elapsed_s);

/* ---------------------------------------------------------------------------------------------
 * Multi-GPU: one process per GPU, cases partitioned in consecutive blocks, ONE collective -- an all-gather of the
 * 128-byte case records over RCCL/xGMI (SURVEY.md 8b "xee_gather_scalars", 8e "ncclAllGather").  The reference has no
 * counterpart (it is a single-threaded CPU program); this is how an ensemble host collects the per-case scalars.
 * ------------------------------------------------------------------------------------------- */
typedef struct xee_comm xee_comm_t;
#define XEE_COMM_ID_BYTES 128
/* rank 0 makes the rendezvous id and hands its XEE_COMM_ID_BYTES bytes to the other ranks (file, broadcast, ...) */
int xee_comm_unique_id(char *id);
/* NOT collective: can this process take part (librccl loadable, a HIP device visible)?  Hosts agree on this over their own
 * channel BEFORE anybody enters the collective xee_comm_create -- a rank that failed earlier would leave the others
 * blocked inside it.  0 = yes, else a status with xee_last_error() set. */
int xee_comm_available(void);
/* the file the RCCL entry points were resolved from (dladdr on ncclAllGather), for the record */
int xee_comm_library(char *path, int path_len);
/* collective over all ranks: joins the communicator on `device` (-1: the current device).  The rendezvous runs on a helper
 * thread and is given up after xee_comm_set_timeout's seconds (default 120): XEE_ERR_TIMEOUT instead of a hang when a rank
 * never arrives (the process should then exit: the helper thread stays blocked inside RCCL). */
int xee_comm_create(xee_comm_t **comm, const char *id, int rank, int world, int device);
/* seconds after which xee_comm_create / xee_gather_scalars give up (process-wide; <= 0: wait for ever) */
int xee_comm_set_timeout(double seconds);
int xee_comm_rank(const xee_comm_t *comm, int *rank, int *world);
int xee_comm_destroy(xee_comm_t *comm);
/* collective: every rank contributes per_rank records (pad a short block with case_id = -1) and receives the
 * world * per_rank records of all ranks, ordered by rank, in `all` (host memory on both sides) */
int xee_gather_scalars(xee_comm_t *comm, const xee_case_record_t *local, int per_rank, xee_case_record_t *all);

#ifdef __cplusplus
}
#endif
#endif
