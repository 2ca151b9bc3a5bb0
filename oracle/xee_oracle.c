/*
 * xee_oracle.c -- CPU restatement of the reference hot path.  TEST INFRASTRUCTURE ONLY
 * (see xee_oracle.h for the contract, the array conventions and the parity status).
 *
 * Build: gcc -O3 -ffp-contract=off -fno-fast-math -fPIC -shared (oracle/Makefile).
 * Every function cites the reference lines it restates.  Indices in comments are the
 * reference's 1-based Fortran indices; code uses 0-based i = I-1, j = J-1.
 */
#include "xee_oracle.h"

#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* f(i,j) of a Fortran array with leading dimension n1, 0-based */
#define IDX(i, j, n1) ((size_t)(j) * (size_t)(n1) + (size_t)(i))

/* ------------------------------------------------------------------------------------------ */
/* xtt-lib-fortran/constants.f90:4-5: g0=9.8, theta0=298.0, Rd=287.0, Cv=5.0/2.0*Rd,
 * Cp=Cv+Rd, kappa=Rd/Cp, h0=Cp*theta0/g0, p0=101300.0 -- all real(4), folded left to right. */
void orc_constants(orc_constants_t *c)
{
    volatile float g0 = 9.8f, theta0 = 298.0f, Rd = 287.0f, p0 = 101300.0f;
    float Cv = 5.0f / 2.0f * Rd;
    float Cp = Cv + Rd;
    c->g0 = g0;
    c->theta0 = theta0;
    c->Rd = Rd;
    c->Cv = Cv;
    c->Cp = Cp;
    c->kappa = Rd / Cp;
    c->h0 = Cp * theta0 / g0;
    c->p0 = p0;
}

/* src/diagnose/initialize-variables.f90:44-67, read-input.f90:55-76, variables.f90:4 */
void orc_profiles(int geometry, int density_mode, const float *domain, int nr, int nz,
                  float *ra, float *za, float *rcuva, float *exner, float *rho, float *sin_table,
                  float *dr_out, float *dz_out)
{
    orc_constants_t k;
    orc_constants(&k);
    float Lr1, Lr2, Lz1, Lz2, planet_radius = 0.0f, Lat1 = -90.0f, Lat2 = 90.0f;
    if (geometry == 0) {
        Lr1 = domain[0]; Lr2 = domain[1]; Lz1 = domain[2]; Lz2 = domain[3];
    } else {
        /* read-input.f90:67-70: Lr = Lat * DEG2RAD * planet_radius, MATH_PI = acos(-1.0) */
        const float MATH_PI = 3.14159274f; /* acosf(-1.0f) */
        const float DEG2RAD = MATH_PI / 180.0f;
        planet_radius = domain[0]; Lz1 = domain[1]; Lz2 = domain[2];
        Lr1 = Lat1 * DEG2RAD * planet_radius;
        Lr2 = Lat2 * DEG2RAD * planet_radius;
    }
    float dr = (Lr2 - Lr1) / (float)(nr - 1);   /* :45 */
    float dz = (Lz2 - Lz1) / (float)(nz - 1);
    for (int i = 0; i < nr; ++i) ra[i] = Lr1 + (float)i * dr;   /* :47-49 */
    float expo = 1.0f / k.kappa - 1.0f;
    float pref = k.p0 / (k.theta0 * k.Rd);
    for (int j = 0; j < nz; ++j) {              /* :52-57 */
        za[j] = Lz1 + (float)j * dz;
        if (density_mode == 0) {
            exner[j] = 1.0f - za[j] / k.h0;
            rho[j] = pref * powf(exner[j], expo);
        } else {
            exner[j] = 1.0f;
            rho[j] = 1.0f;
        }
    }
    if (geometry == 0) {                        /* :59-60 */
        for (int i = 0; i < nr; ++i) rcuva[i] = ra[i];
    } else {                                    /* :61-67, degrees fed to cos/sin (sic) */
        float dlat = (Lat2 - Lat1) / (float)(nr - 1);
        for (int i = 0; i < nr; ++i) {
            float lat = Lat1 + (float)i * dlat;
            rcuva[i] = planet_radius * cosf(lat);
            if (sin_table) sin_table[i] = sinf(lat);
        }
    }
    *dr_out = dr;
    *dz_out = dz;
}

/* src/diagnose/initialize-variables.f90:72-95 */
void orc_solver_coefficients(const float *rhoA, const float *rhoB, const float *rhoC,
                             const float *rcuva, const float *rho,
                             float *sA, float *sB, float *sC, int nr, int nz)
{
    /* solverA_A(I,J) = (A(I,J+1)+A(I+1,J+1)) / (rcuva(I)+rcuva(I+1)) / rho(J+1),  I<=nr-1, J<=nz-2 */
    for (int j = 0; j < nz - 2; ++j)
        for (int i = 0; i < nr - 1; ++i)
            sA[IDX(i, j, nr - 1)] = (rhoA[IDX(i, j + 1, nr)] + rhoA[IDX(i + 1, j + 1, nr)])
                                    / (rcuva[i] + rcuva[i + 1]) / rho[j + 1];
    /* solverB_B(I,J) = (B(I,J)+B(I+1,J)+B(I,J+1)+B(I+1,J+1)) / (rcuva(I)+rcuva(I+1)) / (rho(J)+rho(J+1)) */
    for (int j = 0; j < nz - 1; ++j)
        for (int i = 0; i < nr - 1; ++i)
            sB[IDX(i, j, nr - 1)] = (rhoB[IDX(i, j, nr)] + rhoB[IDX(i + 1, j, nr)]
                                     + rhoB[IDX(i, j + 1, nr)] + rhoB[IDX(i + 1, j + 1, nr)])
                                    / (rcuva[i] + rcuva[i + 1]) / (rho[j] + rho[j + 1]);
    /* solverC_C(I,J) = (C(I+1,J)+C(I+1,J+1)) / rcuva(I+1) / (rho(J)+rho(J+1)),  I<=nr-2, J<=nz-1 */
    for (int j = 0; j < nz - 1; ++j)
        for (int i = 0; i < nr - 2; ++i)
            sC[IDX(i, j, nr - 2)] = (rhoC[IDX(i + 1, j, nr)] + rhoC[IDX(i + 1, j + 1, nr)])
                                    / rcuva[i + 1] / (rho[j] + rho[j + 1]);
}

/* xtt-lib-fortran/elliptic_tools.f90:8-60 */
void orc_cal_coe(const float *a, const float *b, const float *c, float *coe,
                 float dx, float dy, int nx, int ny, int *err)
{
    float PP = dx * dx;          /* dx**2 */
    float QQ = dy * dy;
    float PQ4 = 4.0f * dx * dy;  /* 4*dx*dy, left to right */
    float twoPQ4 = 2.0f * PQ4;
    *err = 1;
    const int na = nx - 1, nb = nx - 1, nc = nx - 2;
    for (int j = 1; j <= ny - 2; ++j) {
        for (int i = 1; i <= nx - 2; ++i) {
            /* Fortran (I,J) = (i+1,j+1);  a(I,J-1) -> a0[i][j-1], a(I-1,J-1) -> a0[i-1][j-1] ... */
            float Ap = a[IDX(i, j - 1, na)] / PP;
            float Am = a[IDX(i - 1, j - 1, na)] / PP;
            float Cp = c[IDX(i - 1, j, nc)] / QQ;
            float Cm = c[IDX(i - 1, j - 1, nc)] / QQ;
            float BXp = (b[IDX(i, j, nb)] + b[IDX(i, j - 1, nb)]) / twoPQ4;
            float BXm = (b[IDX(i - 1, j, nb)] + b[IDX(i - 1, j - 1, nb)]) / twoPQ4;
            float BYp = (b[IDX(i - 1, j, nb)] + b[IDX(i, j, nb)]) / twoPQ4;
            float BYm = (b[IDX(i - 1, j - 1, nb)] + b[IDX(i, j - 1, nb)]) / twoPQ4;
            float *w = coe + IDX(i, j, nx) * 9;
            w[0] = -(BXm + BYp);
            w[1] = Cp + (BXp - BXm);
            w[2] = BXp + BYp;
            w[3] = Am - (BYp - BYm);
            w[4] = -(Am + Ap + Cm + Cp);
            w[5] = Ap + (BYp - BYm);
            w[6] = BXm + BYm;
            w[7] = Cm - (BXp - BXm);
            w[8] = -(BXp + BYm);
        }
    }
    *err = 0;
}

/* xtt-lib-fortran/elliptic_tools.f90:64-90; k=1..3 row J+1, 4..6 row J, 7..9 row J-1 */
static inline float stencil9(const float *w, const float *psi, int i, int j, int nx)
{
    float r = w[0] * psi[IDX(i - 1, j + 1, nx)];
    r = r + w[1] * psi[IDX(i, j + 1, nx)];
    r = r + w[2] * psi[IDX(i + 1, j + 1, nx)];
    r = r + w[3] * psi[IDX(i - 1, j, nx)];
    r = r + w[4] * psi[IDX(i, j, nx)];
    r = r + w[5] * psi[IDX(i + 1, j, nx)];
    r = r + w[6] * psi[IDX(i - 1, j - 1, nx)];
    r = r + w[7] * psi[IDX(i, j - 1, nx)];
    r = r + w[8] * psi[IDX(i + 1, j - 1, nx)];
    return r;
}

void orc_do_elliptic(const float *psi, const float *coe, float *outdat, int nx, int ny, int *err)
{
    *err = 1; /* :73, never cleared (sic) */
    for (int j = 1; j <= ny - 2; ++j)
        for (int i = 1; i <= nx - 2; ++i)
            outdat[IDX(i, j, nx)] = stencil9(coe + IDX(i, j, nx) * 9, psi, i, j, nx);
}

/* xtt-lib-fortran/elliptic_tools.f90:193-199: err_now = sqrt(sum/((nx-2)*(ny-2))), sequential
 * fp32, I outer / J inner, to_dat**2.0 == x*x. */
float orc_residual_sumsq(const float *res, int nx, int ny)
{
    float s = 0.0f;
    for (int i = 1; i <= nx - 2; ++i)
        for (int j = 1; j <= ny - 2; ++j) {
            float x = res[IDX(i, j, nx)];
            s = s + x * x;
        }
    return s;
}

float orc_residual_rms(const float *res, int nx, int ny)
{
    return sqrtf(orc_residual_sumsq(res, nx, ny) / (float)((nx - 2) * (ny - 2)));
}

/* xtt-lib-fortran/elliptic_tools.f90:333-358 */
static void judge_error_print(int err)
{
    int known = 0;
    if (err == 0) { printf(" Elliptic Tools: Iteration success.\n"); known = 1; }
    if (err & ORC_ERR_OVER_MAX_ITERATION) { printf(" Elliptic Tools: [Error] Max iteration reached.\n"); known = 1; }
    if (err & ORC_ERR_EXPLODE) { printf(" Elliptic Tools: [Error] Iteration explodes.\n"); known = 1; }
    if (!known) printf(" Elliptic Tools: Unknown error code %d\n", err);
}

typedef struct {
    int check_abs, check_rel;
    int check_step, converge_time, lost_rate;
} solve_setup_t;

/* :112-144.  Returns 1 when both criteria are off. */
static int solve_setup(solve_setup_t *s, int check_step, int converge_time, int lost_rate,
                       float *r1, float *r2)
{
    if (*r1 > 0.0f) s->check_abs = 1; else { s->check_abs = 0; *r1 = FLT_MAX; }
    if (*r2 > 0.0f) s->check_rel = 1; else { s->check_rel = 0; *r2 = FLT_MAX; }
    s->check_step = check_step > 0 ? check_step : 100;
    s->converge_time = converge_time > 0 ? converge_time : 10;
    s->lost_rate = lost_rate > 0 ? lost_rate : 5;
    return (!s->check_abs && !s->check_rel) ? 1 : 0;
}

/* The convergence state machine, :192-234.  Returns 1 to stop. */
typedef struct { int converge_cnt, lose_cnt; float err_before, err_now, ratio; } conv_state_t;

static int conv_check(conv_state_t *st, const solve_setup_t *s, float err_now, float r1, float r2,
                      int cnt, int debug)
{
    int stop = 0;
    st->err_now = err_now;
    float ratio = (st->err_before - err_now) / st->err_before;
    if (debug == 2) printf("Iter: %8d, err_now: %12.3E, ratio: %12.3E\n", cnt, err_now, ratio);
    ratio = fabsf(ratio);
    st->ratio = ratio;
    if (st->err_before == 0.0f) {
        stop = 1;
    } else if (err_now < r1 && ratio < r2) {
        st->converge_cnt += 1;
        st->lose_cnt = 0;
        if (st->converge_cnt >= s->converge_time) stop = 1;
    } else if (st->converge_cnt > 0) {
        st->lose_cnt += 1;
        if (st->lose_cnt >= s->lost_rate) {
            st->converge_cnt -= 1;
            st->lose_cnt = 0;
        }
    }
    st->err_before = err_now;
    return stop;
}

/* Fast-walking variant: SoA copy of coe, i-inner loops.  Same values in the same per-point
 * order, so bit-identical to the reference (Jacobi is order-independent across points). */
int orc_solve_elliptic(int *max_iter, int check_step, int converge_time, int lost_rate,
                       float *strategy_r1, float *strategy_r2, float alpha,
                       float *dat, const float *coe, const float *f, float *workspace,
                       int nx, int ny, int *err, int debug)
{
    solve_setup_t s;
    if (solve_setup(&s, check_step, converge_time, lost_rate, strategy_r1, strategy_r2)) {
        printf(" ERROR: [check_abs_err] and [check_rel_err] cannot both be non-positive.\n");
        return 1;
    }
    const size_t n = (size_t)nx * ny;
    float *w = (float *)malloc(sizeof(float) * n * 9);
    float *res = (float *)malloc(sizeof(float) * n);
    if (!w || !res) { free(w); free(res); return 2; }
    memset(w, 0, sizeof(float) * n * 9);
    for (int j = 1; j <= ny - 2; ++j)
        for (int i = 1; i <= nx - 2; ++i)
            for (int k = 0; k < 9; ++k) w[(size_t)k * n + IDX(i, j, nx)] = coe[IDX(i, j, nx) * 9 + k];

    conv_state_t st = {0, 0, FLT_MAX, 0.0f, 0.0f};
    *err = 0;
    memcpy(workspace, dat, sizeof(float) * n);   /* :171 */
    float *fr = workspace, *to = dat;            /* :173-174 */
    const float r1 = *strategy_r1, r2 = *strategy_r2;
    const int maxit = *max_iter;

    for (int cnt = 1; cnt <= maxit; ++cnt) {
        int flag = (cnt % s.check_step) == 0;
        float *tmp = fr; fr = to; to = tmp;      /* :185-187 */
        int stop = 0;
        for (int j = 1; j <= ny - 2; ++j) {
            const float *p0 = fr + IDX(0, j - 1, nx), *p1 = fr + IDX(0, j, nx), *p2 = fr + IDX(0, j + 1, nx);
            const size_t o = IDX(0, j, nx);
            for (int i = 1; i <= nx - 2; ++i) {
                float r = w[0 * n + o + i] * p2[i - 1];
                r = r + w[1 * n + o + i] * p2[i];
                r = r + w[2 * n + o + i] * p2[i + 1];
                r = r + w[3 * n + o + i] * p1[i - 1];
                r = r + w[4 * n + o + i] * p1[i];
                r = r + w[5 * n + o + i] * p1[i + 1];
                r = r + w[6 * n + o + i] * p0[i - 1];
                r = r + w[7 * n + o + i] * p0[i];
                r = r + w[8 * n + o + i] * p0[i + 1];
                res[o + i] = r - f[o + i];       /* :190 */
            }
        }
        if (flag) stop = conv_check(&st, &s, orc_residual_rms(res, nx, ny), r1, r2, cnt, debug);
        for (int j = 1; j <= ny - 2; ++j) {      /* :236-240 */
            const size_t o = IDX(0, j, nx);
            for (int i = 1; i <= nx - 2; ++i)
                to[o + i] = fr[o + i] + alpha * res[o + i] / (-w[4 * n + o + i]);
        }
        if (cnt == maxit) { stop = 1; *err |= ORC_ERR_OVER_MAX_ITERATION; }   /* :242-248 */
        if (stop) {
            *max_iter = cnt; *strategy_r1 = st.err_now; *strategy_r2 = st.ratio;   /* :253 */
            if (debug >= 0) judge_error_print(*err);
            break;
        }
    }
    if (to != dat) memcpy(dat, workspace, sizeof(float) * n);   /* :259-264 */
    free(w); free(res);
    return 0;
}

/* Reference-order variant: the memory walk of elliptic_tools.f90:75-88,190,236-240 as written
 * (I outer / J inner, AoS coe).  Used as the like-for-like CPU timing "port". */
int orc_solve_elliptic_reforder(int *max_iter, int check_step, int converge_time, int lost_rate,
                                float *strategy_r1, float *strategy_r2, float alpha,
                                float *dat, const float *coe, const float *f, float *workspace,
                                int nx, int ny, int *err, int debug)
{
    solve_setup_t s;
    if (solve_setup(&s, check_step, converge_time, lost_rate, strategy_r1, strategy_r2)) {
        printf(" ERROR: [check_abs_err] and [check_rel_err] cannot both be non-positive.\n");
        return 1;
    }
    const size_t n = (size_t)nx * ny;
    conv_state_t st = {0, 0, FLT_MAX, 0.0f, 0.0f};
    *err = 0;
    memcpy(workspace, dat, sizeof(float) * n);
    float *fr = workspace, *to = dat;
    const float r1 = *strategy_r1, r2 = *strategy_r2;
    const int maxit = *max_iter;
    for (int cnt = 1; cnt <= maxit; ++cnt) {
        int flag = (cnt % s.check_step) == 0;
        float *tmp = fr; fr = to; to = tmp;
        int stop = 0;
        for (int i = 1; i <= nx - 2; ++i)
            for (int j = 1; j <= ny - 2; ++j)
                to[IDX(i, j, nx)] = stencil9(coe + IDX(i, j, nx) * 9, fr, i, j, nx);
        for (int j = 1; j <= ny - 2; ++j)       /* array-section statement :190 */
            for (int i = 1; i <= nx - 2; ++i)
                to[IDX(i, j, nx)] = to[IDX(i, j, nx)] - f[IDX(i, j, nx)];
        if (flag) stop = conv_check(&st, &s, orc_residual_rms(to, nx, ny), r1, r2, cnt, debug);
        for (int i = 1; i <= nx - 2; ++i)
            for (int j = 1; j <= ny - 2; ++j)
                to[IDX(i, j, nx)] = fr[IDX(i, j, nx)] + alpha * to[IDX(i, j, nx)] / (-coe[IDX(i, j, nx) * 9 + 4]);
        if (cnt == maxit) { stop = 1; *err |= ORC_ERR_OVER_MAX_ITERATION; }
        if (stop) {
            *max_iter = cnt; *strategy_r1 = st.err_now; *strategy_r2 = st.ratio;
            if (debug >= 0) judge_error_print(*err);
            break;
        }
    }
    if (to != dat) memcpy(dat, workspace, sizeof(float) * n);
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* src/diagnose/quick-tools2.f90 */

void orc_d_dz_B2A(const float *from, float *to, const float *za, int nr, int nz)
{   /* :7-12  I=1..nr-1, J=2..nz-2: (from(I,J)-from(I,J-1)) / ((za(J+1)-za(J-1))/2.0) */
    for (int j = 1; j <= nz - 3; ++j)
        for (int i = 0; i < nr - 1; ++i)
            to[IDX(i, j, nr - 1)] = (from[IDX(i, j, nr - 1)] - from[IDX(i, j - 1, nr - 1)])
                                    / ((za[j + 1] - za[j - 1]) / 2.0f);
}

void orc_d_dz_O2C(const float *from, float *to, const float *za, int nr, int nz)
{   /* :20-24 */
    for (int j = 0; j < nz - 1; ++j)
        for (int i = 0; i < nr; ++i)
            to[IDX(i, j, nr)] = (from[IDX(i, j + 1, nr)] - from[IDX(i, j, nr)]) / (za[j + 1] - za[j]);
}

void orc_d_dr_B2B(const float *from, float *to, const float *ra, int nr, int nz)
{   /* :31-42 one-sided at the ends, centred inside; note ra is indexed with the B-grid index */
    for (int j = 0; j < nz - 1; ++j)
        for (int i = 0; i < nr - 1; ++i) {
            int m, n;
            if (i == 0) { m = 0; n = 1; } else if (i == nr - 2) { m = -1; n = 0; } else { m = -1; n = 1; }
            to[IDX(i, j, nr - 1)] = (from[IDX(i + m, j, nr - 1)] - from[IDX(i + n, j, nr - 1)])
                                    / (ra[i + m] - ra[i + n]);
        }
}

void orc_d_dr_B2C(const float *from, float *to, const float *ra, int nr, int nz)
{   /* :51-56  I=2..nr-1: (from(I,J)-from(I-1,J)) / ((ra(I+1)-ra(I-1))/2.0) */
    for (int j = 0; j < nz - 1; ++j)
        for (int i = 1; i <= nr - 2; ++i)
            to[IDX(i, j, nr)] = (from[IDX(i, j, nr - 1)] - from[IDX(i - 1, j, nr - 1)])
                                / ((ra[i + 1] - ra[i - 1]) / 2.0f);
}

void orc_d_dr_O2A(const float *from, float *to, const float *ra, int nr, int nz)
{   /* :63-67 */
    for (int j = 0; j < nz; ++j)
        for (int i = 0; i < nr - 1; ++i)
            to[IDX(i, j, nr - 1)] = (from[IDX(i + 1, j, nr)] - from[IDX(i, j, nr)]) / (ra[i + 1] - ra[i]);
}

void orc_d_rcuvdr_O2A(const float *from, float *to, const float *ra, const float *rcuva, int nr, int nz)
{   /* :77-83 */
    orc_d_dr_O2A(from, to, ra, nr, nz);
    for (int j = 0; j < nz; ++j)
        for (int i = 0; i < nr - 1; ++i)
            to[IDX(i, j, nr - 1)] = to[IDX(i, j, nr - 1)] / ((rcuva[i] + rcuva[i + 1]) / 2.0f);
}

/* src/diagnose/quick-tools1.f90:1-13: eta = eta * g0 / (rho(j) * Cp * exner(j) * theta0) */
void orc_cal_eta(const float *rchi, float *eta, const float *ra, const float *rcuva,
                 const float *rho, const float *exner, int nr, int nz)
{
    orc_constants_t k;
    orc_constants(&k);
    orc_d_rcuvdr_O2A(rchi, eta, ra, rcuva, nr, nz);
    for (int j = 0; j < nz; ++j) {
        float den = rho[j] * k.Cp * exner[j] * k.theta0;
        for (int i = 0; i < nr - 1; ++i)
            eta[IDX(i, j, nr - 1)] = eta[IDX(i, j, nr - 1)] * k.g0 / den;
    }
}

/* src/diagnose/quick-tools1.f90:15-41 */
void orc_cal_uw(const float *rpsi, float *u, float *w, const float *ra, const float *za,
                const float *rcuva, const float *rho, int nr, int nz)
{
    orc_d_rcuvdr_O2A(rpsi, w, ra, rcuva, nr, nz);
    orc_d_dz_O2C(rpsi, u, za, nr, nz);
    for (size_t k = 0; k < (size_t)nr * (nz - 1); ++k) u[k] = -u[k];
    for (int j = 0; j < nz; ++j)
        for (int i = 0; i < nr - 1; ++i)
            w[IDX(i, j, nr - 1)] = w[IDX(i, j, nr - 1)] / rho[j];
    for (int j = 0; j < nz - 1; ++j)
        for (int i = 0; i < nr; ++i) {
            if (ra[i] != 0.0f)
                u[IDX(i, j, nr)] = u[IDX(i, j, nr)] / (rcuva[i] * (rho[j] + rho[j + 1]) / 2.0f);
            else
                u[IDX(i, j, nr)] = 0.0f;
        }
}

/* src/old-diagnose/diagnose.f90:1029-1048: sequential fp32, I outer / J inner,
 * weight*rho_*rcuv*dr*dz left to right. */
float orc_integrate_weight_B(const float *weight, const float *ra, const float *za,
                             const float *rcuva, const float *rho, int nr, int nz)
{
    float s = 0.0f;
    for (int i = 0; i < nr - 1; ++i)
        for (int j = 0; j < nz - 1; ++j) {
            float rcuv = (rcuva[i] + rcuva[i + 1]) / 2.0f;
            float dr = ra[i + 1] - ra[i];
            float dz = za[j + 1] - za[j];
            float rho_ = (rho[j + 1] + rho[j]) / 2.0f;
            s = s + weight[IDX(i, j, nr - 1)] * rho_ * rcuv * dr * dz;
        }
    return s;
}

/* src/old-diagnose/diagnose.f90:1073-1092 */
float orc_cal_sum_Qeta(const float *Q, const float *eta, const float *ra, const float *za,
                       const float *rcuva, const float *rho, int nr, int nz)
{
    float s = 0.0f;
    for (int i = 0; i < nr - 1; ++i)
        for (int j = 0; j < nz - 1; ++j) {
            float rcuv = (rcuva[i] + rcuva[i + 1]) / 2.0f;
            float dr = ra[i + 1] - ra[i];
            float dz = za[j + 1] - za[j];
            float rho_ = (rho[j + 1] + rho[j]) / 2.0f;
            s = s + ((eta[IDX(i, j, nr - 1)] + eta[IDX(i, j + 1, nr - 1)]) / 2.0f)
                        * Q[IDX(i, j, nr - 1)] * rho_ * rcuv * dr * dz;
        }
    return s;
}

/* src/old-diagnose/diagnose.f90:1117-1127 */
void orc_cal_wtheta(const float *w_A, const float *theta_B, float *wtheta_B, int nr, int nz)
{
    for (int j = 0; j < nz - 1; ++j)
        for (int i = 0; i < nr - 1; ++i)
            wtheta_B[IDX(i, j, nr - 1)] = ((w_A[IDX(i, j, nr - 1)] + w_A[IDX(i, j + 1, nr - 1)]) / 2.0f)
                                          * theta_B[IDX(i, j, nr - 1)];
}

/* src/old-diagnose/diagnose.f90:383-387 */
void orc_JJ_B(const float *Q, const float *exner, float *JJ, int nr, int nz)
{
    orc_constants_t k;
    orc_constants(&k);
    for (int j = 0; j < nz - 1; ++j)
        for (int i = 0; i < nr - 1; ++i)
            JJ[IDX(i, j, nr - 1)] = Q[IDX(i, j, nr - 1)] / (k.Cp * exner[j]);
}

/* src/old-diagnose/diagnose.f90:395-406.  wksp_C (nr,nz-1) must be zero-initialised by the
 * caller where the reference leaves it untouched (columns 1 and nr). */
void orc_rhs_thm(const float *JJ_B, float *rhs, float *wksp_C, const float *ra, int nr, int nz)
{
    orc_constants_t k;
    orc_constants(&k);
    memset(rhs, 0, sizeof(float) * (size_t)nr * nz);
    orc_d_dr_B2C(JJ_B, wksp_C, ra, nr, nz);
    for (int j = 1; j <= nz - 2; ++j)
        for (int i = 1; i <= nr - 2; ++i)
            rhs[IDX(i, j, nr)] = (wksp_C[IDX(i, j, nr)] + wksp_C[IDX(i, j - 1, nr)]) / 2.0f;
    for (size_t q = 0; q < (size_t)nr * nz; ++q) rhs[q] = rhs[q] * k.g0 / k.theta0;
}

/* src/old-diagnose/diagnose.f90:412-436.  wksp_A (nr-1,nz) zero-initialised by the caller. */
void orc_rhs_mom(const float *m2, const float *F, float *rhs, float *wksp_B, float *wksp_A,
                 const float *za, const float *rcuva, int nr, int nz)
{
    memset(rhs, 0, sizeof(float) * (size_t)nr * nz);
    for (int j = 0; j < nz - 1; ++j)
        for (int i = 0; i < nr - 1; ++i)
            wksp_B[IDX(i, j, nr - 1)] = sqrtf(m2[IDX(i, j, nr - 1)]) * F[IDX(i, j, nr - 1)];
    orc_d_dz_B2A(wksp_B, wksp_A, za, nr, nz);
    for (int j = 1; j <= nz - 2; ++j)
        for (int i = 1; i <= nr - 2; ++i)
            rhs[IDX(i, j, nr)] = -(wksp_A[IDX(i, j, nr - 1)] + wksp_A[IDX(i - 1, j, nr - 1)])
                                 / (rcuva[i] * rcuva[i]);   /* rcuva(i)**2.0 */
}

/* src/old-diagnose/diagnose.f90:341-346: 4-point average of an O-grid plane onto B cells */
void orc_cell_average_B(const float *in_O, float *out_B, int nr, int nz)
{
    for (int j = 0; j < nz - 1; ++j)
        for (int i = 0; i < nr - 1; ++i)
            out_B[IDX(i, j, nr - 1)] = (in_O[IDX(i, j, nr)] + in_O[IDX(i + 1, j, nr)]
                                        + in_O[IDX(i, j + 1, nr)] + in_O[IDX(i + 1, j + 1, nr)]) / 4.0f;
}

/* src/old-diagnose/diagnose.f90:523-538: f(I,J) = -(b(I-1,J-1)+b(I-1,J)+b(I,J)+b(I,J-1))/4.0 */
void orc_f_from_B(const float *b_B, float *f_O, int nr, int nz)
{
    memset(f_O, 0, sizeof(float) * (size_t)nr * nz);
    for (int j = 1; j <= nz - 2; ++j)
        for (int i = 1; i <= nr - 2; ++i)
            f_O[IDX(i, j, nr)] = -(b_B[IDX(i - 1, j - 1, nr - 1)] + b_B[IDX(i - 1, j, nr - 1)]
                                   + b_B[IDX(i, j, nr - 1)] + b_B[IDX(i, j - 1, nr - 1)]) / 4.0f;
}

/* src/old-diagnose/diagnose.f90:1143-1174 with real (not integer) r, dr, dz.  parity unpinned. */
static float exchange_conversion_impl(const float *rpsi, const float *rchi, const float *rhoC,
                                      float *bndconv, const float *ra, const float *za,
                                      const float *rho, int nr, int nz, int legacy_int)
{
    float sum = 0.0f;
    /* legacy_int: `integer :: i, r, dr, dz` (:1146) -- the assignments truncate toward zero, the mixed-mode
     * expressions convert back to real */
    float dz = za[1] - za[0];
    float dr = ra[1] - ra[0];
    if (legacy_int) { dz = (float)(int)dz; dr = (float)(int)dr; }
    const int t = nz - 1;
    for (int i = 0; i < nr - 1; ++i) {
        float r = (ra[i] + ra[i + 1]) / 2.0f;
        if (legacy_int) r = (float)(int)r;
        float bot = ((rhoC[IDX(i, 0, nr)] + rhoC[IDX(i + 1, 0, nr)]) / (2.0f * rho[0]))
                    * (((rpsi[IDX(i, 0, nr)] + rpsi[IDX(i + 1, 0, nr)]) / 2.0f)
                           * ((rchi[IDX(i, 1, nr)] + rchi[IDX(i + 1, 1, nr)] - rchi[IDX(i, 0, nr)] - rchi[IDX(i + 1, 0, nr)]) / (2.0f * dz))
                       - ((rchi[IDX(i, 0, nr)] + rchi[IDX(i + 1, 0, nr)]) / 2.0f)
                             * ((rpsi[IDX(i, 1, nr)] + rpsi[IDX(i + 1, 1, nr)] - rpsi[IDX(i, 0, nr)] - rpsi[IDX(i + 1, 0, nr)]) / (2.0f * dz)))
                    / (r * r);
        float top = ((rhoC[IDX(i, t, nr)] + rhoC[IDX(i + 1, t, nr)]) / (2.0f * rho[t]))
                    * (((rpsi[IDX(i, t, nr)] + rpsi[IDX(i + 1, t, nr)]) / 2.0f)
                           * ((rchi[IDX(i, t, nr)] + rchi[IDX(i + 1, t, nr)] - rchi[IDX(i, t - 1, nr)] - rchi[IDX(i + 1, t - 1, nr)]) / (2.0f * dz))
                       - ((rchi[IDX(i, t, nr)] + rchi[IDX(i + 1, t, nr)]) / 2.0f)
                             * ((rpsi[IDX(i, t, nr)] + rpsi[IDX(i + 1, t, nr)] - rpsi[IDX(i, t - 1, nr)] - rpsi[IDX(i + 1, t - 1, nr)]) / (2.0f * dz)))
                    / (r * r);
        bndconv[IDX(i, 0, nr - 1)] = bot;
        bndconv[IDX(i, 1, nr - 1)] = top;
        sum = sum - (top - bot) * r * dr;
    }
    return sum;
}

float orc_exchange_conversion(const float *rpsi, const float *rchi, const float *rhoC,
                              float *bndconv, const float *ra, const float *za,
                              const float *rho, int nr, int nz)
{
    return exchange_conversion_impl(rpsi, rchi, rhoC, bndconv, ra, za, rho, nr, nz, 0);
}

/* the same as the legacy program really computes it: r, dr, dz declared INTEGER (:1146).  Pinned against the legacy
 * binary on a metre-sized domain (tests/test_oracle_legacy_chain.py); on a unit domain r = 0 and the value is Inf/NaN. */
float orc_exchange_conversion_legacy(const float *rpsi, const float *rchi, const float *rhoC,
                                     float *bndconv, const float *ra, const float *za,
                                     const float *rho, int nr, int nz)
{
    return exchange_conversion_impl(rpsi, rchi, rhoC, bndconv, ra, za, rho, nr, nz, 1);
}

/* src/old-diagnose/diagnose.f90:470-479.  `theta0/g0 * (x) / 2.0` is ((theta0/g0) * x) / 2.0, the sum left to right. */
void orc_theta_tendency(const float *JJ_B, const float *rhoA_A, const float *w_A, const float *rhoB_C,
                        const float *u_C, float *dtheta_dt_B, int nr, int nz)
{
    orc_constants_t k;
    orc_constants(&k);
    const float c = k.theta0 / k.g0;
    for (int j = 0; j < nz - 1; ++j)
        for (int i = 0; i < nr - 1; ++i) {
            float a = c * (rhoA_A[IDX(i, j, nr - 1)] * w_A[IDX(i, j, nr - 1)]
                           + rhoA_A[IDX(i, j + 1, nr - 1)] * w_A[IDX(i, j + 1, nr - 1)]) / 2.0f;
            float b = c * (rhoB_C[IDX(i, j, nr)] * u_C[IDX(i, j, nr)]
                           + rhoB_C[IDX(i + 1, j, nr)] * u_C[IDX(i + 1, j, nr)]) / 2.0f;
            dtheta_dt_B[IDX(i, j, nr - 1)] = JJ_B[IDX(i, j, nr - 1)] - a + b;
        }
}

/* src/old-diagnose/diagnose.f90:893-912: integrate the gradients from theta0 at (1,1): along r on the first
 * row, then up every column; sequential fp32 additions. */
void orc_relative_theta(float *theta_B, const float *dtheta_dz_A, const float *dtheta_dr_C,
                        const float *ra, const float *za, int nr, int nz)
{
    orc_constants_t k;
    orc_constants(&k);
    for (size_t q = 0; q < (size_t)(nr - 1) * (nz - 1); ++q) theta_B[q] = k.theta0;
    for (int i = 1; i <= nr - 2; ++i) {
        float dist = (ra[i + 1] - ra[i - 1]) / 2.0f;
        theta_B[IDX(i, 0, nr - 1)] = theta_B[IDX(i - 1, 0, nr - 1)] + dist * dtheta_dr_C[IDX(i, 0, nr)];
    }
    for (int i = 0; i < nr - 1; ++i)
        for (int j = 1; j <= nz - 2; ++j) {
            float dist = (za[j + 1] - za[j - 1]) / 2.0f;
            theta_B[IDX(i, j, nr - 1)] = theta_B[IDX(i, j - 1, nr - 1)] + dist * dtheta_dz_A[IDX(i, j, nr - 1)];
        }
}

/* src/old-diagnose/diagnose.f90:489-518 (see the header for the one deliberate difference) */
void orc_theta_anomaly_update(float *theta_B, float testing_dt, float *rhoB_B, float *rhoA_A, float *rhoB_C,
                              float *b_anomaly_B, float *solver_b_anomaly_B, const float *ra, const float *za,
                              const float *rcuva, const float *rho, int nr, int nz)
{
    orc_constants_t k;
    orc_constants(&k);
    const size_t nB = (size_t)(nr - 1) * (nz - 1), nA = (size_t)(nr - 1) * nz, nC = (size_t)nr * (nz - 1);
    float *wB = (float *)calloc(nB, sizeof(float)), *wA = (float *)calloc(nA, sizeof(float));
    float *gA = (float *)malloc(nA * sizeof(float)), *gC = (float *)malloc(nC * sizeof(float));
    for (size_t q = 0; q < nB; ++q) theta_B[q] = theta_B[q] * testing_dt;                          /* :489 */
    orc_d_dr_B2B(theta_B, wB, ra, nr, nz);                                                         /* :492 */
    for (size_t q = 0; q < nB; ++q) b_anomaly_B[q] = -(k.g0 / k.theta0 * wB[q]);                   /* :493 */
    for (size_t q = 0; q < nB; ++q) rhoB_B[q] = rhoB_B[q] + b_anomaly_B[q];                        /* :494 */
    orc_d_dz_B2A(theta_B, wA, za, nr, nz);                                                         /* :499 */
    for (int j = 1; j <= nz - 3; ++j)                                                              /* :500 */
        for (int i = 0; i < nr - 1; ++i)
            rhoA_A[IDX(i, j, nr - 1)] = rhoA_A[IDX(i, j, nr - 1)] + k.g0 / k.theta0 * wA[IDX(i, j, nr - 1)];
    for (int j = 0; j < nz - 1; ++j)                                                               /* :503-508 */
        for (int i = 1; i <= nr - 2; ++i)
            rhoB_C[IDX(i, j, nr)] = (rhoB_B[IDX(i - 1, j, nr - 1)] + rhoB_B[IDX(i, j, nr - 1)]) / 2.0f;
    for (size_t q = 0; q < nA; ++q) gA[q] = rhoA_A[q] * (k.theta0 / k.g0);                         /* :509 */
    for (size_t q = 0; q < nC; ++q) gC[q] = rhoB_C[q] * (-k.theta0 / k.g0);
    orc_relative_theta(theta_B, gA, gC, ra, za, nr, nz);
    for (int j = 0; j < nz - 1; ++j)                                                               /* :513-518 */
        for (int i = 0; i < nr - 1; ++i)
            solver_b_anomaly_B[IDX(i, j, nr - 1)] = b_anomaly_B[IDX(i, j, nr - 1)]
                                                    / ((rcuva[i] + rcuva[i + 1]) / 2.0f) / ((rho[j] + rho[j + 1]) / 2.0f);
    free(wB); free(wA); free(gA); free(gC);
}

/* src/old-diagnose/diagnose.f90:359-380, intended form (see the header).  `x**3.0` is powf, as flang lowers it. */
void orc_m2_from_C(const float *rhoC_C, float *m2_B, const float *ra, const float *rcuva, const float *sin_table,
                   int nr, int nz)
{
    for (int j = 0; j < nz - 1; ++j) {
        float first = powf((rcuva[1] - rcuva[0]) / 4.0f, 3.0f) * rhoC_C[IDX(0, j, nr)] * (ra[1] - ra[0]) / 2.0f;     /* :361,:372 */
        if (sin_table) first = first / ((sin_table[1] + 3.0f * sin_table[0]) / 4.0f);                                /* :373 */
        m2_B[IDX(0, j, nr - 1)] = first;
        for (int i = 1; i <= nr - 2; ++i) {
            float t = powf(rcuva[i], 3.0f) * rhoC_C[IDX(i, j, nr)] * (ra[i + 1] - ra[i - 1]) / 2.0f;                 /* :365,:377 */
            if (sin_table) t = t / sin_table[i];                                                                     /* :378 */
            m2_B[IDX(i, j, nr - 1)] = m2_B[IDX(i - 1, j, nr - 1)] + t;
        }
    }
}
