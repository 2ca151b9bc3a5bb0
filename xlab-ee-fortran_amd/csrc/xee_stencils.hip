// xee_stencils.hip -- the remaining finite-difference / integral stencils of the path as Tier-1 entry
// points (host arrays in the reference's layouts, computed on the GPU, no CPU fallback):
//   xee_fd_stencil          the six finite differences of src/diagnose/quick-tools2.f90:1-85
//                           (identical copies in src/old-diagnose/diagnose.f90:943-1027)
//   xee_rhs_momentum        RHS_rpsi_mom, src/old-diagnose/diagnose.f90:412-436
//   xee_exchange_conversion boundary exchange-conversion line integral, old-diagnose :1143-1174 (K9)
//   xee_theta_tendency, xee_relative_theta, xee_theta_anomaly_update   Stage II of the legacy TENDENCY mode,
//                           old-diagnose :462-518, :893-912 (forecast of the thermal field)
// Same bit-exactness discipline as xee_kernels.hip (no contraction, left-to-right, IEEE division).
#include "../../include/xee.h"
#include "xee_device.h"

#include <cmath>
#include <vector>

#pragma clang fp contract(off)

namespace {

using xee::record_failure;

// a packed device copy of a host array, freed on scope exit
struct DevBuf {
    float* p = nullptr;
    size_t n = 0;
    ~DevBuf() { if (p) (void)hipFree(p); }
    int put(const float* host, size_t count)
    {
        n = count;
        if (hipMalloc(reinterpret_cast<void**>(&p), sizeof(float) * (count ? count : 1)) != hipSuccess)
            return record_failure(XEE_ERR_ALLOC, "hipMalloc failed");
        if (host && count && hipMemcpy(p, host, sizeof(float) * count, hipMemcpyHostToDevice) != hipSuccess)
            return record_failure(XEE_ERR_HIP, "hipMemcpy H2D failed");
        return XEE_OK;
    }
    int get(float* host) const
    {
        if (hipMemcpy(host, p, sizeof(float) * n, hipMemcpyDeviceToHost) != hipSuccess)
            return record_failure(XEE_ERR_HIP, "hipMemcpy D2H failed");
        return XEE_OK;
    }
};

int need_device()
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return record_failure(XEE_ERR_NO_DEVICE, "no HIP device visible (libxee has no CPU fallback)");
    return XEE_OK;
}

int finish(const char* what)
{
    hipError_t e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipGetLastError();
    if (e != hipSuccess) return record_failure(XEE_ERR_HIP, what);
    return XEE_OK;
}

#define TRY(x) do { int s_ = (x); if (s_ != XEE_OK) return s_; } while (0)

// which: see include/xee.h.  Arrays are packed Fortran (n1,n2), i fastest.  1-based (I,J) in comments.
__global__ __launch_bounds__(256) void k_fd_stencil(int which, const float* __restrict__ from, float* __restrict__ to,
        const float* __restrict__ ra, const float* __restrict__ za, const float* __restrict__ rcuva, int nr, int nz)
{
    const int i = blockIdx.x * 64 + threadIdx.x;
    const int j = blockIdx.y * 4 + threadIdx.y;
    switch (which) {
        case XEE_D_DZ_B2A:       // quick-tools2.f90:7-12: I=1..nr-1, J=2..nz-2; from (nr-1,nz-1) -> to (nr-1,nz)
            if (i < nr - 1 && j >= 1 && j <= nz - 3)
                to[(size_t)j * (nr - 1) + i] = (from[(size_t)j * (nr - 1) + i] - from[(size_t)(j - 1) * (nr - 1) + i])
                                               / ((za[j + 1] - za[j - 1]) / 2.0f);
            break;
        case XEE_D_DZ_O2C:       // :20-24: from (nr,nz) -> to (nr,nz-1)
            if (i < nr && j < nz - 1)
                to[(size_t)j * nr + i] = (from[(size_t)(j + 1) * nr + i] - from[(size_t)j * nr + i]) / (za[j + 1] - za[j]);
            break;
        case XEE_D_DR_B2B:       // :31-42: one-sided at the ends; (nr-1,nz-1) -> (nr-1,nz-1)
            if (i < nr - 1 && j < nz - 1) {
                const int m = (i == 0) ? 0 : -1, n = (i == nr - 2) ? 0 : 1;
                to[(size_t)j * (nr - 1) + i] = (from[(size_t)j * (nr - 1) + i + m] - from[(size_t)j * (nr - 1) + i + n])
                                               / (ra[i + m] - ra[i + n]);
            }
            break;
        case XEE_D_DR_B2C:       // :51-56: I=2..nr-1; from (nr-1,nz-1) -> to (nr,nz-1)
            if (i >= 1 && i <= nr - 2 && j < nz - 1)
                to[(size_t)j * nr + i] = (from[(size_t)j * (nr - 1) + i] - from[(size_t)j * (nr - 1) + i - 1])
                                         / ((ra[i + 1] - ra[i - 1]) / 2.0f);
            break;
        case XEE_D_DR_O2A:       // :63-67: from (nr,nz) -> to (nr-1,nz)
        case XEE_D_RCUVDR_O2A:   // :77-83: ... then / ((rcuva(I)+rcuva(I+1))/2)
            if (i < nr - 1 && j < nz) {
                float t = (from[(size_t)j * nr + i + 1] - from[(size_t)j * nr + i]) / (ra[i + 1] - ra[i]);
                if (which == XEE_D_RCUVDR_O2A) t = t / ((rcuva[i] + rcuva[i + 1]) / 2.0f);
                to[(size_t)j * (nr - 1) + i] = t;
            }
            break;
        default: break;
    }
}

// wksp_B = sqrt(m2)*F on B cells, then d_dz_B2A, then RHS(I,J) = -(wA(I,J)+wA(I-1,J))/rcuva(I)**2
__global__ __launch_bounds__(256) void k_rhs_momentum(const float* __restrict__ m2, const float* __restrict__ F,
        const float* __restrict__ za, const float* __restrict__ rcuva, float* __restrict__ rhs, int nr, int nz)
{
    const int i = blockIdx.x * 64 + threadIdx.x;
    const int j = blockIdx.y * 4 + threadIdx.y;
    if (i >= nr || j >= nz) return;
    float v = 0.0f;
    if (i >= 1 && i <= nr - 2 && j >= 1 && j <= nz - 2) {
        // d_dz_B2A writes wksp_A(I,J) for J=2..nz-2 only (quick-tools2.f90:8); row nz-1, which the
        // reference reads from an uninitialised workspace, is defined as 0 here
        auto wA = [&](int ii) -> float {
            if (j > nz - 3) return 0.0f;
            const float b1 = sqrtf(m2[(size_t)j * (nr - 1) + ii]) * F[(size_t)j * (nr - 1) + ii];
            const float b0 = sqrtf(m2[(size_t)(j - 1) * (nr - 1) + ii]) * F[(size_t)(j - 1) * (nr - 1) + ii];
            return (b1 - b0) / ((za[j + 1] - za[j - 1]) / 2.0f);
        };
        v = -(wA(i) + wA(i - 1)) / (rcuva[i] * rcuva[i]);
    }
    rhs[(size_t)j * nr + i] = v;
}

// K9 with REAL r, dr, dz (the reference declares them integer, old-diagnose/diagnose.f90:1146, which
// truncates them and makes its value meaningless -- fixed and documented).
__global__ void k_exchange_terms(const float* __restrict__ rpsi, const float* __restrict__ rchi,
        const float* __restrict__ rhoC, const float* __restrict__ ra, const float* __restrict__ za,
        const float* __restrict__ rho, float* __restrict__ bndconv, int nr, int nz, int legacy_int)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nr - 1) return;
    // legacy_int: the reference's `integer :: r, dr, dz` -- assignments truncate toward zero
    float dz = za[1] - za[0];
    float r = (ra[i] + ra[i + 1]) / 2.0f;
    if (legacy_int) { dz = (float)(int)dz; r = (float)(int)r; }
    const int t = nz - 1;
    auto at = [&](const float* f, int ii, int jj) { return f[(size_t)jj * nr + ii]; };
    const float bot = ((at(rhoC, i, 0) + at(rhoC, i + 1, 0)) / (2.0f * rho[0]))
        * (((at(rpsi, i, 0) + at(rpsi, i + 1, 0)) / 2.0f)
               * ((at(rchi, i, 1) + at(rchi, i + 1, 1) - at(rchi, i, 0) - at(rchi, i + 1, 0)) / (2.0f * dz))
           - ((at(rchi, i, 0) + at(rchi, i + 1, 0)) / 2.0f)
                 * ((at(rpsi, i, 1) + at(rpsi, i + 1, 1) - at(rpsi, i, 0) - at(rpsi, i + 1, 0)) / (2.0f * dz)))
        / (r * r);
    const float top = ((at(rhoC, i, t) + at(rhoC, i + 1, t)) / (2.0f * rho[t]))
        * (((at(rpsi, i, t) + at(rpsi, i + 1, t)) / 2.0f)
               * ((at(rchi, i, t) + at(rchi, i + 1, t) - at(rchi, i, t - 1) - at(rchi, i + 1, t - 1)) / (2.0f * dz))
           - ((at(rchi, i, t) + at(rchi, i + 1, t)) / 2.0f)
                 * ((at(rpsi, i, t) + at(rpsi, i + 1, t) - at(rpsi, i, t - 1) - at(rpsi, i + 1, t - 1)) / (2.0f * dz)))
        / (r * r);
    bndconv[i] = bot;
    bndconv[(nr - 1) + i] = top;
}

// sum = sum - (top - bot) * r * dr, sequentially over i (order-exact, one lane; nr-1 terms)
__global__ void k_exchange_sum(const float* __restrict__ bndconv, const float* __restrict__ ra, float* __restrict__ out, int nr, int legacy_int)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float dr = ra[1] - ra[0];
    if (legacy_int) dr = (float)(int)dr;
    float s = 0.0f;
    for (int i = 0; i < nr - 1; ++i) {
        float r = (ra[i] + ra[i + 1]) / 2.0f;
        if (legacy_int) r = (float)(int)r;
        s = s - (bndconv[(nr - 1) + i] - bndconv[i]) * r * dr;
    }
    out[0] = s;
}

// ---- legacy TENDENCY mode, Stage II (src/old-diagnose/diagnose.f90:462-518) -----------------------------------
__device__ __forceinline__ float s_g0() { return 9.8f; }        // xtt-lib-fortran/constants.f90:4
__device__ __forceinline__ float s_theta0() { return 298.0f; }

// :470-479  dtheta/dt on B cells; `theta0/g0 * (x) / 2.0` = ((theta0/g0) * x) / 2.0
__global__ __launch_bounds__(256) void k_theta_tendency(const float* __restrict__ JJ, const float* __restrict__ rhoA,
        const float* __restrict__ wA, const float* __restrict__ rhoBC, const float* __restrict__ uC,
        float* __restrict__ out, int nr, int nz)
{
    const int i = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y;
    if (i >= nr - 1 || j >= nz - 1) return;
    const float c = s_theta0() / s_g0();
    const size_t a0 = (size_t)j * (nr - 1) + i, a1 = (size_t)(j + 1) * (nr - 1) + i, c0 = (size_t)j * nr + i;
    const float a = c * (rhoA[a0] * wA[a0] + rhoA[a1] * wA[a1]) / 2.0f;
    const float b = c * (rhoBC[c0] * uC[c0] + rhoBC[c0 + 1] * uC[c0 + 1]) / 2.0f;
    out[a0] = JJ[a0] - a + b;
}

__global__ __launch_bounds__(256) void k_scale(float* __restrict__ x, float s, size_t n)
{
    const size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (q < n) x[q] = x[q] * s;
}

// :492-494  b anomaly = -(g0/theta0 * d_dr_B2B(theta)); rhoB_B += b anomaly;  :513-518 solver_b_anomaly
__global__ __launch_bounds__(256) void k_b_anomaly(const float* __restrict__ theta, float* __restrict__ banom,
        float* __restrict__ rhoBB, float* __restrict__ sban, const float* __restrict__ ra,
        const float* __restrict__ rcuva, const float* __restrict__ rho, int nr, int nz)
{
    const int i = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y;
    if (i >= nr - 1 || j >= nz - 1) return;
    const int m = (i == 0) ? 0 : -1, n = (i == nr - 2) ? 0 : 1;                         // d_dr_B2B, quick-tools2.f90:31-42
    const size_t o = (size_t)j * (nr - 1) + i;
    const float d = (theta[o + m] - theta[o + n]) / (ra[i + m] - ra[i + n]);
    const float b = -(s_g0() / s_theta0() * d);
    banom[o] = b;
    rhoBB[o] = rhoBB[o] + b;
    sban[o] = b / ((rcuva[i] + rcuva[i + 1]) / 2.0f) / ((rho[j] + rho[j + 1]) / 2.0f);
}

// :499-500  rhoA_A rows 2..nz-2 += g0/theta0 * d_dz_B2A(theta)
__global__ __launch_bounds__(256) void k_rhoA_update(const float* __restrict__ theta, float* __restrict__ rhoA,
        const float* __restrict__ za, int nr, int nz)
{
    const int i = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y;
    if (i >= nr - 1 || j < 1 || j > nz - 3) return;
    const size_t o = (size_t)j * (nr - 1) + i;
    const float d = (theta[o] - theta[o - (nr - 1)]) / ((za[j + 1] - za[j - 1]) / 2.0f);  // d_dz_B2A, quick-tools2.f90:7-12
    rhoA[o] = rhoA[o] + s_g0() / s_theta0() * d;
}

// :503-508  rhoB_C(I,J) = (rhoB_B(I-1,J) + rhoB_B(I,J)) / 2, I = 2..nr-1
__global__ __launch_bounds__(256) void k_rhoBC_update(const float* __restrict__ rhoBB, float* __restrict__ rhoBC, int nr, int nz)
{
    const int i = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y;
    if (i < 1 || i > nr - 2 || j >= nz - 1) return;
    rhoBC[(size_t)j * nr + i] = (rhoBB[(size_t)j * (nr - 1) + i - 1] + rhoBB[(size_t)j * (nr - 1) + i]) / 2.0f;
}

// :893-912 relativeTheta: one lane integrates along the first row, then one lane per column integrates upwards;
// the gradients are scaled on the fly (gz = dz_A * sz, gr = dr_C * sr) as the caller's array expressions do
__global__ void k_relative_theta_row(float* __restrict__ theta, const float* __restrict__ drC, float sr,
        const float* __restrict__ ra, int nr)
{
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    float t = s_theta0();
    theta[0] = t;
    for (int i = 1; i <= nr - 2; ++i) {
        const float dist = (ra[i + 1] - ra[i - 1]) / 2.0f;
        t = t + dist * (drC[i] * sr);
        theta[i] = t;
    }
}
__global__ __launch_bounds__(64) void k_relative_theta_cols(float* __restrict__ theta, const float* __restrict__ dzA, float sz,
        const float* __restrict__ za, int nr, int nz)
{
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= nr - 1) return;
    float t = theta[i];
    for (int j = 1; j <= nz - 2; ++j) {
        const float dist = (za[j + 1] - za[j - 1]) / 2.0f;
        t = t + dist * (dzA[(size_t)j * (nr - 1) + i] * sz);
        theta[(size_t)j * (nr - 1) + i] = t;
    }
}

// old-diagnose/diagnose.f90:359-380 (intended form): per row j a sequential sum along r; cube[i] = rcuva(i)**3.0 (host powf),
// cube[0] = ((rcuva(2)-rcuva(1))/4)**3.0
__global__ __launch_bounds__(64) void k_m2_from_C(const float* __restrict__ rhoC, float* __restrict__ m2, const float* __restrict__ ra,
        const float* __restrict__ cube, const float* __restrict__ sin_table /* or nullptr */, int nr, int nz)
{
    const int j = blockIdx.x * 64 + threadIdx.x;
    if (j >= nz - 1) return;
    float acc = cube[0] * rhoC[(size_t)j * nr] * (ra[1] - ra[0]) / 2.0f;
    if (sin_table) acc = acc / ((sin_table[1] + 3.0f * sin_table[0]) / 4.0f);
    m2[(size_t)j * (nr - 1)] = acc;
    for (int i = 1; i <= nr - 2; ++i) {
        float t = cube[i] * rhoC[(size_t)j * nr + i] * (ra[i + 1] - ra[i - 1]) / 2.0f;
        if (sin_table) t = t / sin_table[i];
        acc = acc + t;
        m2[(size_t)j * (nr - 1) + i] = acc;
    }
}

}  // namespace

// ---- legacy program, staggered averages of the input planes (src/old-diagnose/diagnose.f90:329-354) -----------------
// rhoA_A(i,j) = (A(i,j)+A(i+1,j))/2 on A (nr-1,nz);  rhoB_C(i,j) = (B(i,j)+B(i,j+1))/2 and rhoC_C likewise on C (nr,nz-1);
// rhoB_B = the four-cell average /4 on B (nr-1,nz-1)
__global__ __launch_bounds__(256) void k_legacy_averages(const float* __restrict__ A, const float* __restrict__ B, const float* __restrict__ Cc,
        float* __restrict__ rhoA_A, float* __restrict__ rhoB_C, float* __restrict__ rhoB_B, float* __restrict__ rhoC_C, int nr, int nz)
{
    const int i = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y;
    if (i >= nr || j >= nz) return;
    auto at = [&](const float* f, int ii, int jj) { return f[(size_t)jj * nr + ii]; };
    if (i < nr - 1) rhoA_A[(size_t)j * (nr - 1) + i] = (at(A, i, j) + at(A, i + 1, j)) / 2.0f;
    if (j < nz - 1) {
        rhoB_C[(size_t)j * nr + i] = (at(B, i, j) + at(B, i, j + 1)) / 2.0f;
        rhoC_C[(size_t)j * nr + i] = (at(Cc, i, j) + at(Cc, i, j + 1)) / 2.0f;
        if (i < nr - 1)
            rhoB_B[(size_t)j * (nr - 1) + i] = (at(B, i, j) + at(B, i + 1, j) + at(B, i, j + 1) + at(B, i + 1, j + 1)) / 4.0f;
    }
}

// JJ_B(i,j) = Q(i,j) / (Cp * exner(j))   (:383-387)
__global__ __launch_bounds__(256) void k_JJ_B(const float* __restrict__ Q, const float* __restrict__ exner, float* __restrict__ JJ, int nr, int nz)
{
    const int i = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y;
    if (i >= nr - 1 || j >= nz - 1) return;
    const float Rd = 287.0f, Cv = 5.0f / 2.0f * Rd, Cp = Cv + Rd;      // xtt-lib-fortran/constants.f90:4-7
    JJ[(size_t)j * (nr - 1) + i] = Q[(size_t)j * (nr - 1) + i] / (Cp * exner[j]);
}

__global__ __launch_bounds__(256) void k_add_fields(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, size_t n)
{
    const size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (q < n) out[q] = a[q] + b[q];
}

extern "C" {

int xee_m2_from_C(const float* rhoC_C, const float* ra, const float* rcuva, const float* sin_table, float* m2_B, int nr, int nz)
{
    if (!rhoC_C || !ra || !rcuva || !m2_B || nr < 3 || nz < 4) return record_failure(XEE_ERR_INVALID_ARGUMENT, "xee_m2_from_C: bad argument");
    TRY(need_device());
    std::vector<float> cube(nr);
    cube[0] = powf((rcuva[1] - rcuva[0]) / 4.0f, 3.0f);                    // :361 (host libm, like the Fortran runtime)
    for (int i = 1; i < nr; ++i) cube[i] = powf(rcuva[i], 3.0f);           // :365
    DevBuf dC, dra, dcube, dsin, dM;
    TRY(dC.put(rhoC_C, (size_t)nr * (nz - 1))); TRY(dra.put(ra, nr)); TRY(dcube.put(cube.data(), nr));
    if (sin_table) TRY(dsin.put(sin_table, nr));
    TRY(dM.put(nullptr, (size_t)(nr - 1) * (nz - 1)));
    hipLaunchKernelGGL(k_m2_from_C, dim3((nz + 63) / 64), dim3(64), 0, 0, dC.p, dM.p, dra.p, dcube.p, sin_table ? dsin.p : nullptr, nr, nz);
    TRY(finish("xee_m2_from_C"));
    return dM.get(m2_B);
}

int xee_theta_tendency(const float* JJ_B, const float* rhoA_A, const float* w_A, const float* rhoB_C, const float* u_C,
                       float* dtheta_dt_B, int nr, int nz)
{
    if (!JJ_B || !rhoA_A || !w_A || !rhoB_C || !u_C || !dtheta_dt_B || nr < 3 || nz < 4)
        return record_failure(XEE_ERR_INVALID_ARGUMENT, "xee_theta_tendency: bad argument");
    TRY(need_device());
    const size_t nB = (size_t)(nr - 1) * (nz - 1), nA = (size_t)(nr - 1) * nz, nC = (size_t)nr * (nz - 1);
    DevBuf dJ, dRA, dW, dRC, dU, dO;
    TRY(dJ.put(JJ_B, nB)); TRY(dRA.put(rhoA_A, nA)); TRY(dW.put(w_A, nA)); TRY(dRC.put(rhoB_C, nC)); TRY(dU.put(u_C, nC));
    TRY(dO.put(nullptr, nB));
    hipLaunchKernelGGL(k_theta_tendency, dim3((nr + 63) / 64, (nz + 3) / 4), dim3(64, 4), 0, 0, dJ.p, dRA.p, dW.p, dRC.p, dU.p, dO.p, nr, nz);
    TRY(finish("xee_theta_tendency"));
    return dO.get(dtheta_dt_B);
}

int xee_relative_theta(const float* dtheta_dz_A, const float* dtheta_dr_C, const float* ra, const float* za,
                       float* theta_B, int nr, int nz)
{
    if (!dtheta_dz_A || !dtheta_dr_C || !ra || !za || !theta_B || nr < 3 || nz < 4)
        return record_failure(XEE_ERR_INVALID_ARGUMENT, "xee_relative_theta: bad argument");
    TRY(need_device());
    DevBuf dA, dC, dra, dza, dT;
    TRY(dA.put(dtheta_dz_A, (size_t)(nr - 1) * nz)); TRY(dC.put(dtheta_dr_C, (size_t)nr * (nz - 1)));
    TRY(dra.put(ra, nr)); TRY(dza.put(za, nz)); TRY(dT.put(nullptr, (size_t)(nr - 1) * (nz - 1)));
    hipLaunchKernelGGL(k_relative_theta_row, dim3(1), dim3(64), 0, 0, dT.p, dC.p, 1.0f, dra.p, nr);
    hipLaunchKernelGGL(k_relative_theta_cols, dim3((nr + 63) / 64), dim3(64), 0, 0, dT.p, dA.p, 1.0f, dza.p, nr, nz);
    TRY(finish("xee_relative_theta"));
    return dT.get(theta_B);
}

int xee_theta_anomaly_update(float* theta_B, float testing_dt, float* rhoB_B, float* rhoA_A, float* rhoB_C,
                             float* b_anomaly_B, float* solver_b_anomaly_B, const float* ra, const float* za,
                             const float* rcuva, const float* rho, int nr, int nz)
{
    if (!theta_B || !rhoB_B || !rhoA_A || !rhoB_C || !b_anomaly_B || !solver_b_anomaly_B || !ra || !za || !rcuva || !rho ||
        nr < 3 || nz < 4)
        return record_failure(XEE_ERR_INVALID_ARGUMENT, "xee_theta_anomaly_update: bad argument");
    TRY(need_device());
    const size_t nB = (size_t)(nr - 1) * (nz - 1), nA = (size_t)(nr - 1) * nz, nC = (size_t)nr * (nz - 1);
    DevBuf dT, dBB, dRA, dRC, dBa, dSb, dra, dza, drc, drho;
    TRY(dT.put(theta_B, nB)); TRY(dBB.put(rhoB_B, nB)); TRY(dRA.put(rhoA_A, nA)); TRY(dRC.put(rhoB_C, nC));
    TRY(dBa.put(nullptr, nB)); TRY(dSb.put(nullptr, nB));
    TRY(dra.put(ra, nr)); TRY(dza.put(za, nz)); TRY(drc.put(rcuva, nr)); TRY(drho.put(rho, nz));
    const dim3 grid((nr + 63) / 64, (nz + 3) / 4), block(64, 4);
    hipLaunchKernelGGL(k_scale, dim3((unsigned)((nB + 255) / 256)), dim3(256), 0, 0, dT.p, testing_dt, nB);                  // :489
    hipLaunchKernelGGL(k_b_anomaly, grid, block, 0, 0, dT.p, dBa.p, dBB.p, dSb.p, dra.p, drc.p, drho.p, nr, nz);          // :492-494, :513-518
    hipLaunchKernelGGL(k_rhoA_update, grid, block, 0, 0, dT.p, dRA.p, dza.p, nr, nz);                                      // :499-500
    hipLaunchKernelGGL(k_rhoBC_update, grid, block, 0, 0, dBB.p, dRC.p, nr, nz);                                           // :503-508
    // :509 relativeTheta(theta, rhoA_A * (theta0/g0), rhoB_C * (-theta0/g0))
    hipLaunchKernelGGL(k_relative_theta_row, dim3(1), dim3(64), 0, 0, dT.p, dRC.p, -298.0f / 9.8f, dra.p, nr);
    hipLaunchKernelGGL(k_relative_theta_cols, dim3((nr + 63) / 64), dim3(64), 0, 0, dT.p, dRA.p, 298.0f / 9.8f, dza.p, nr, nz);
    TRY(finish("xee_theta_anomaly_update"));
    TRY(dT.get(theta_B)); TRY(dBB.get(rhoB_B)); TRY(dRA.get(rhoA_A)); TRY(dRC.get(rhoB_C)); TRY(dBa.get(b_anomaly_B));
    return dSb.get(solver_b_anomaly_B);
}

int xee_fd_stencil(int which, const float* from, float* to, const float* ra, const float* za, const float* rcuva,
                   int nr, int nz)
{
    if (!from || !to || !ra || !za || !rcuva || nr < 3 || nz < 4) return record_failure(XEE_ERR_INVALID_ARGUMENT, "xee_fd_stencil: bad argument");
    size_t nfrom, nto;
    switch (which) {
        case XEE_D_DZ_B2A: nfrom = (size_t)(nr - 1) * (nz - 1); nto = (size_t)(nr - 1) * nz; break;
        case XEE_D_DZ_O2C: nfrom = (size_t)nr * nz; nto = (size_t)nr * (nz - 1); break;
        case XEE_D_DR_B2B: nfrom = nto = (size_t)(nr - 1) * (nz - 1); break;
        case XEE_D_DR_B2C: nfrom = (size_t)(nr - 1) * (nz - 1); nto = (size_t)nr * (nz - 1); break;
        case XEE_D_DR_O2A:
        case XEE_D_RCUVDR_O2A: nfrom = (size_t)nr * nz; nto = (size_t)(nr - 1) * nz; break;
        default: return record_failure(XEE_ERR_INVALID_ARGUMENT, "xee_fd_stencil: unknown stencil id");
    }
    TRY(need_device());
    DevBuf dfrom, dto, dra, dza, drc;
    TRY(dfrom.put(from, nfrom));
    TRY(dto.put(to, nto));          // entries the reference leaves untouched keep the caller's bytes
    TRY(dra.put(ra, nr));
    TRY(dza.put(za, nz));
    TRY(drc.put(rcuva, nr));
    hipLaunchKernelGGL(k_fd_stencil, dim3((nr + 63) / 64, (nz + 3) / 4), dim3(64, 4), 0, 0, which, dfrom.p, dto.p, dra.p, dza.p, drc.p, nr, nz);
    TRY(finish("xee_fd_stencil"));
    return dto.get(to);
}

int xee_rhs_momentum(const float* m2_B, const float* F_B, const float* za, const float* rcuva, float* rhs_O, int nr, int nz)
{
    if (!m2_B || !F_B || !za || !rcuva || !rhs_O || nr < 3 || nz < 4) return record_failure(XEE_ERR_INVALID_ARGUMENT, "xee_rhs_momentum: bad argument");
    TRY(need_device());
    DevBuf dm, dF, dza, drc, dout;
    TRY(dm.put(m2_B, (size_t)(nr - 1) * (nz - 1)));
    TRY(dF.put(F_B, (size_t)(nr - 1) * (nz - 1)));
    TRY(dza.put(za, nz));
    TRY(drc.put(rcuva, nr));
    TRY(dout.put(nullptr, (size_t)nr * nz));
    hipLaunchKernelGGL(k_rhs_momentum, dim3((nr + 63) / 64, (nz + 3) / 4), dim3(64, 4), 0, 0, dm.p, dF.p, dza.p, drc.p, dout.p, nr, nz);
    TRY(finish("xee_rhs_momentum"));
    return dout.get(rhs_O);
}

static int exchange_conversion_impl(const float* rpsi, const float* rchi, const float* rhoC, const float* ra, const float* za,
                                    const float* rho, float* bndconv, float* sum, int nr, int nz, int legacy_int)
{
    if (!rpsi || !rchi || !rhoC || !ra || !za || !rho || !bndconv || !sum || nr < 3 || nz < 3)
        return record_failure(XEE_ERR_INVALID_ARGUMENT, "xee_exchange_conversion: bad argument");
    TRY(need_device());
    DevBuf dpsi, dchi, drc, dra, dza, drho, dbnd, dsum;
    const size_t n = (size_t)nr * nz;
    TRY(dpsi.put(rpsi, n)); TRY(dchi.put(rchi, n)); TRY(drc.put(rhoC, n));
    TRY(dra.put(ra, nr)); TRY(dza.put(za, nz)); TRY(drho.put(rho, nz));
    TRY(dbnd.put(nullptr, (size_t)(nr - 1) * 2)); TRY(dsum.put(nullptr, 1));
    hipLaunchKernelGGL(k_exchange_terms, dim3((nr - 1 + 255) / 256), dim3(256), 0, 0, dpsi.p, dchi.p, drc.p, dra.p, dza.p, drho.p, dbnd.p, nr, nz, legacy_int);
    hipLaunchKernelGGL(k_exchange_sum, dim3(1), dim3(64), 0, 0, dbnd.p, dra.p, dsum.p, nr, legacy_int);
    TRY(finish("xee_exchange_conversion"));
    TRY(dbnd.get(bndconv));
    return dsum.get(sum);
}

int xee_exchange_conversion(const float* rpsi, const float* rchi, const float* rhoC, const float* ra, const float* za,
                            const float* rho, float* bndconv, float* sum, int nr, int nz)
{
    return exchange_conversion_impl(rpsi, rchi, rhoC, ra, za, rho, bndconv, sum, nr, nz, 0);
}

int xee_exchange_conversion_legacy(const float* rpsi, const float* rchi, const float* rhoC, const float* ra, const float* za,
                                   const float* rho, float* bndconv, float* sum, int nr, int nz)
{
    return exchange_conversion_impl(rpsi, rchi, rhoC, ra, za, rho, bndconv, sum, nr, nz, 1);
}


int xee_legacy_averages(const float* rhoA_in, const float* rhoB_in, const float* rhoC_in, float* rhoA_A, float* rhoB_C,
                        float* rhoB_B, float* rhoC_C, int nr, int nz)
{
    if (!rhoA_in || !rhoB_in || !rhoC_in || !rhoA_A || !rhoB_C || !rhoB_B || !rhoC_C || nr < 3 || nz < 3)
        return record_failure(XEE_ERR_INVALID_ARGUMENT, "xee_legacy_averages: bad argument");
    TRY(need_device());
    DevBuf dA, dB, dC, oA, oBC, oBB, oCC;
    const size_t n = (size_t)nr * nz;
    TRY(dA.put(rhoA_in, n)); TRY(dB.put(rhoB_in, n)); TRY(dC.put(rhoC_in, n));
    TRY(oA.put(nullptr, (size_t)(nr - 1) * nz)); TRY(oBC.put(nullptr, (size_t)nr * (nz - 1)));
    TRY(oBB.put(nullptr, (size_t)(nr - 1) * (nz - 1))); TRY(oCC.put(nullptr, (size_t)nr * (nz - 1)));
    hipLaunchKernelGGL(k_legacy_averages, dim3((nr + 63) / 64, (nz + 3) / 4), dim3(64, 4), 0, 0, dA.p, dB.p, dC.p, oA.p, oBC.p, oBB.p, oCC.p, nr, nz);
    TRY(finish("xee_legacy_averages"));
    TRY(oA.get(rhoA_A)); TRY(oBC.get(rhoB_C)); TRY(oBB.get(rhoB_B));
    return oCC.get(rhoC_C);
}

int xee_JJ_B(const float* Q_B, const float* exner, float* JJ_B, int nr, int nz)
{
    if (!Q_B || !exner || !JJ_B || nr < 3 || nz < 3) return record_failure(XEE_ERR_INVALID_ARGUMENT, "xee_JJ_B: bad argument");
    TRY(need_device());
    DevBuf dQ, dex, dJ;
    TRY(dQ.put(Q_B, (size_t)(nr - 1) * (nz - 1))); TRY(dex.put(exner, nz)); TRY(dJ.put(nullptr, (size_t)(nr - 1) * (nz - 1)));
    hipLaunchKernelGGL(k_JJ_B, dim3((nr + 62) / 64, (nz + 2) / 4), dim3(64, 4), 0, 0, dQ.p, dex.p, dJ.p, nr, nz);
    TRY(finish("xee_JJ_B"));
    return dJ.get(JJ_B);
}

int xee_add_fields(const float* a, const float* b, float* out, size_t n)
{
    if (!a || !b || !out || n == 0) return record_failure(XEE_ERR_INVALID_ARGUMENT, "xee_add_fields: bad argument");
    TRY(need_device());
    DevBuf da, db, dout;
    TRY(da.put(a, n)); TRY(db.put(b, n)); TRY(dout.put(nullptr, n));
    hipLaunchKernelGGL(k_add_fields, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, da.p, db.p, dout.p, n);
    TRY(finish("xee_add_fields"));
    return dout.get(out);
}

}  // extern "C"
