/*
 * xee_oracle.h -- CPU restatement ("the oracle") of the elliptic-solve + energy-efficiency
 * diagnostic hot path of meteorologytoday/XLab-EE-fortran.
 *
 * THIS IS TEST INFRASTRUCTURE. It is the checker the HIP path is compared against; nothing in the
 * product (xlab-ee-fortran_amd/, include/) may link, import or call it. Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg use it.
 *
 * Parity status: PINNED. Every function below is checked bit-for-bit against the reference itself,
 * compiled unmodified with `amdflang -O2` into oracle/_ref/ (see oracle/Makefile), and against the
 * md5 anchors in BASELINE.md section 4 (tests/test_oracle_golden.py, tests/golden/).
 *
 * Arithmetic contract: IEEE binary32, round-to-nearest-even, NO fused multiply-add, NO
 * reassociation, denormals kept. Build with `-O2 -ffp-contract=off` and never `-ffast-math`.
 * Evaluation order inside every expression follows the Fortran source left to right.
 *
 * Array convention: every 2-D array is the reference's Fortran array f(n1,n2) stored column-major,
 * i.e. C index [(j)*n1 + (i)] with 0-based i (r, fastest) and j (z).  This equals a numpy (n2,n1)
 * C-order array, the layout of the reference's .bin files (xtt-lib-fortran/field_tools.f90:30-76).
 * `coe` keeps the reference's AoS layout coe(9,nx,ny): C index [((j)*nx + (i))*9 + k].
 */
#ifndef XEE_ORACLE_H
#define XEE_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* error bits, xtt-lib-fortran/elliptic_tools.f90:3-4 */
#define ORC_ERR_OVER_MAX_ITERATION 1
#define ORC_ERR_EXPLODE            2

/* physical constants, xtt-lib-fortran/constants.f90:4-7 (computed in fp32 exactly as written) */
typedef struct {
    float g0, theta0, Rd, Cv, Cp, kappa, h0, p0;
} orc_constants_t;
void orc_constants(orc_constants_t *c);

/* A13: 1-D profiles, src/diagnose/initialize-variables.f90:44-67 (+ read-input.f90:55-76).
 * geometry: 0 cylindrical (domain = Lr1 Lr2 Lz1 Lz2), 1 spherical (domain = planet_radius Lz1 Lz2,
 * latitudes in DEGREES fed to cos/sin exactly as the reference does).
 * density_mode: 0 DENSITY_NORMAL, 1 DENSITY_BOUSSINESQ.
 * Outputs: ra[nr], za[nz], rcuva[nr], exner[nz], rho[nz], sin_table[nr] (spherical only, else
 * untouched), *dr, *dz. */
void orc_profiles(int geometry, int density_mode, const float *domain, int nr, int nz,
                  float *ra, float *za, float *rcuva, float *exner, float *rho, float *sin_table,
                  float *dr, float *dz);

/* A6: solver coefficients, src/diagnose/initialize-variables.f90:72-95.
 * in : rhoA, rhoB, rhoC (nr,nz); rcuva[nr]; rho[nz]
 * out: sA (nr-1,nz-2), sB (nr-1,nz-1), sC (nr-2,nz-1) */
void orc_solver_coefficients(const float *rhoA, const float *rhoB, const float *rhoC,
                             const float *rcuva, const float *rho,
                             float *sA, float *sB, float *sC, int nr, int nz);

/* A1: xtt-lib-fortran/elliptic_tools.f90:8-60.  Only interior coe entries are written. */
void orc_cal_coe(const float *a, const float *b, const float *c, float *coe,
                 float dx, float dy, int nx, int ny, int *err);

/* A2: xtt-lib-fortran/elliptic_tools.f90:64-90.  Interior of outdat only; err := 1 (sic). */
void orc_do_elliptic(const float *psi, const float *coe, float *outdat, int nx, int ny, int *err);

/* A3+A4: xtt-lib-fortran/elliptic_tools.f90:93-265.  Returns 0, or 1 when both criteria are
 * non-positive (the reference prints and STOPs there, :126-129; nothing is touched then).
 * In/out scalars are overwritten with results exactly like the reference (:253). When no check
 * ever ran (max_iter < check_step) the reference reports uninitialised err_now/ratio; the oracle
 * defines them as 0.  debug==2 prints the per-check trace to stdout (:202-204). */
int orc_solve_elliptic(int *max_iter, int check_step, int converge_time, int lost_rate,
                       float *strategy_r1, float *strategy_r2, float alpha,
                       float *dat, const float *coe, const float *f, float *workspace,
                       int nx, int ny, int *err, int debug);

/* Same algorithm and bit-identical results, but walking memory the way the reference does
 * (i outer / j inner over column-major arrays, AoS coe): the like-for-like CPU timing mode. */
int orc_solve_elliptic_reforder(int *max_iter, int check_step, int converge_time, int lost_rate,
                                float *strategy_r1, float *strategy_r2, float alpha,
                                float *dat, const float *coe, const float *f, float *workspace,
                                int nx, int ny, int *err, int debug);

/* A4 alone: sequential fp32 RMS of the interior of a residual plane, i outer / j inner
 * (elliptic_tools.f90:193-199). */
float orc_residual_rms(const float *res, int nx, int ny);
float orc_residual_sumsq(const float *res, int nx, int ny);   /* the raw sequential sum of squares */

/* finite differences, src/diagnose/quick-tools2.f90 (and the identical copies in
 * src/old-diagnose/diagnose.f90:943-1027).  Untouched output entries stay untouched. */
void orc_d_dz_B2A(const float *from, float *to, const float *za, int nr, int nz);     /* :1-13  */
void orc_d_dz_O2C(const float *from, float *to, const float *za, int nr, int nz);     /* :16-25 */
void orc_d_dr_B2B(const float *from, float *to, const float *ra, int nr, int nz);     /* :27-43 */
void orc_d_dr_B2C(const float *from, float *to, const float *ra, int nr, int nz);     /* :45-57 */
void orc_d_dr_O2A(const float *from, float *to, const float *ra, int nr, int nz);     /* :59-68 */
void orc_d_rcuvdr_O2A(const float *from, float *to, const float *ra, const float *rcuva,
                      int nr, int nz);                                                /* :71-85 */

/* A8: src/diagnose/quick-tools1.f90:1-13 */
void orc_cal_eta(const float *rchi, float *eta, const float *ra, const float *rcuva,
                 const float *rho, const float *exner, int nr, int nz);
/* A9: src/diagnose/quick-tools1.f90:15-41;  u (nr,nz-1), w (nr-1,nz) */
void orc_cal_uw(const float *rpsi, float *u, float *w, const float *ra, const float *za,
                const float *rcuva, const float *rho, int nr, int nz);

/* A10: EE domain integrals, src/old-diagnose/diagnose.f90:1029-1127.  Sequential fp32, i outer. */
float orc_integrate_weight_B(const float *weight, const float *ra, const float *za,
                             const float *rcuva, const float *rho, int nr, int nz);   /* :1029-1048 */
float orc_cal_sum_Qeta(const float *Q, const float *eta, const float *ra, const float *za,
                       const float *rcuva, const float *rho, int nr, int nz);         /* :1073-1092 */
void  orc_cal_wtheta(const float *w_A, const float *theta_B, float *wtheta_B, int nr, int nz); /* :1117-1127 */

/* A11: RHS source-term assembly, src/old-diagnose/diagnose.f90 */
void orc_JJ_B(const float *Q, const float *exner, float *JJ, int nr, int nz);          /* :383-387 */
void orc_rhs_thm(const float *JJ_B, float *rhs, float *wksp_C, const float *ra,
                 int nr, int nz);                                                      /* :395-406 */
void orc_rhs_mom(const float *m2, const float *F, float *rhs, float *wksp_B, float *wksp_A,
                 const float *za, const float *rcuva, int nr, int nz);                 /* :412-436 */
void orc_cell_average_B(const float *in_O, float *out_B, int nr, int nz);              /* :341-346 */
void orc_f_from_B(const float *b_B, float *f_O, int nr, int nz);                       /* :523-538 */

/* K9: boundary exchange conversion with the reference's `integer :: r, dr, dz` quirk made
 * explicit and FIXED (real r, dr, dz); src/old-diagnose/diagnose.f90:1143-1174.  parity unpinned
 * (the legacy code truncates r, dr, dz to integers, which makes its value meaningless). */
float orc_exchange_conversion(const float *rpsi, const float *rchi, const float *rhoC,
                              float *bndconv, const float *ra, const float *za,
                              const float *rho, int nr, int nz);

/* ... and with the truncation the legacy program really performs (pinned against the legacy binary) */
float orc_exchange_conversion_legacy(const float *rpsi, const float *rchi, const float *rhoC,
                                     float *bndconv, const float *ra, const float *za,
                                     const float *rho, int nr, int nz);

/* ---- legacy TENDENCY mode, Stage II (src/old-diagnose/diagnose.f90:462-518): forecast of the thermal field.
 * parity unpinned against the legacy binary (it has undefined behaviour elsewhere); restated from
 * the source lines cited at each function.  Grids: A (nr-1,nz), B (nr-1,nz-1), C (nr,nz-1). */
void orc_theta_tendency(const float *JJ_B, const float *rhoA_A, const float *w_A, const float *rhoB_C,
                        const float *u_C, float *dtheta_dt_B, int nr, int nz);                      /* :470-479 */
void orc_relative_theta(float *theta_B, const float *dtheta_dz_A, const float *dtheta_dr_C,
                        const float *ra, const float *za, int nr, int nz);                          /* :893-912 */
/* :489-518: theta := dtheta_dt * testing_dt; b anomaly = -g0/theta0 * d_dr_B2B(theta), added to rhoB_B; rhoA_A rows
 * 2..nz-2 += g0/theta0 * d_dz_B2A(theta) (the legacy code also adds row nz-1 of a workspace d_dz_B2A never wrote:
 * that row is left alone here); rhoB_C(2:nr-1,:) re-averaged from rhoB_B; theta := relativeTheta(...);
 * solver_b_anomaly_B = b anomaly / rcuv-bar / rho-bar. */
void orc_theta_anomaly_update(float *theta_B, float testing_dt, float *rhoB_B, float *rhoA_A, float *rhoB_C,
                              float *b_anomaly_B, float *solver_b_anomaly_B, const float *ra, const float *za,
                              const float *rcuva, const float *rho, int nr, int nz);

/* src/old-diagnose/diagnose.f90:359-380: angular momentum squared on B cells by integrating rcuv^3 * C along r.
 * The legacy lines read out of bounds (m2(i-1,j) with i = 1 in the cylindrical loop) and use stale loop indices in
 * the first column (rhoC_C(i,j) after the previous loops); restated as INTENDED: the first column from its own row,
 * the recurrence from the second column on.  sin_table may be NULL (cylindrical).  parity unpinned. */
void orc_m2_from_C(const float *rhoC_C, float *m2_B, const float *ra, const float *rcuva, const float *sin_table,
                   int nr, int nz);

#ifdef __cplusplus
}
#endif
#endif
