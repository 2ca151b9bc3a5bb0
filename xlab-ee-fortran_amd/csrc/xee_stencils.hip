// xee_stencils.hip -- the remaining finite-difference / integral stencils of the path as Tier-1 entry
// points (host arrays in the reference's layouts, computed on the GPU, no CPU fallback):
//   xee_fd_stencil          the six finite differences of src/diagnose/quick-tools2.f90:1-85
//                           (identical copies in src/old-diagnose/diagnose.f90:943-1027)
//   xee_rhs_momentum        RHS_rpsi_mom, src/old-diagnose/diagnose.f90:412-436
//   xee_exchange_conversion boundary exchange-conversion line integral, old-diagnose :1143-1174 (K9)
// Same bit-exactness discipline as xee_kernels.hip (no contraction, left-to-right, IEEE division).
#include "../../include/xee.h"
#include "xee_device.h"

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
        const float* __restrict__ rho, float* __restrict__ bndconv, int nr, int nz)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nr - 1) return;
    const float dz = za[1] - za[0];
    const float r = (ra[i] + ra[i + 1]) / 2.0f;
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
__global__ void k_exchange_sum(const float* __restrict__ bndconv, const float* __restrict__ ra, float* __restrict__ out, int nr)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const float dr = ra[1] - ra[0];
    float s = 0.0f;
    for (int i = 0; i < nr - 1; ++i) {
        const float r = (ra[i] + ra[i + 1]) / 2.0f;
        s = s - (bndconv[(nr - 1) + i] - bndconv[i]) * r * dr;
    }
    out[0] = s;
}

}  // namespace

extern "C" {

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

int xee_exchange_conversion(const float* rpsi, const float* rchi, const float* rhoC, const float* ra, const float* za,
                            const float* rho, float* bndconv, float* sum, int nr, int nz)
{
    if (!rpsi || !rchi || !rhoC || !ra || !za || !rho || !bndconv || !sum || nr < 3 || nz < 3)
        return record_failure(XEE_ERR_INVALID_ARGUMENT, "xee_exchange_conversion: bad argument");
    TRY(need_device());
    DevBuf dpsi, dchi, drc, dra, dza, drho, dbnd, dsum;
    const size_t n = (size_t)nr * nz;
    TRY(dpsi.put(rpsi, n)); TRY(dchi.put(rchi, n)); TRY(drc.put(rhoC, n));
    TRY(dra.put(ra, nr)); TRY(dza.put(za, nz)); TRY(drho.put(rho, nz));
    TRY(dbnd.put(nullptr, (size_t)(nr - 1) * 2)); TRY(dsum.put(nullptr, 1));
    hipLaunchKernelGGL(k_exchange_terms, dim3((nr - 1 + 255) / 256), dim3(256), 0, 0, dpsi.p, dchi.p, drc.p, dra.p, dza.p, drho.p, dbnd.p, nr, nz);
    hipLaunchKernelGGL(k_exchange_sum, dim3(1), dim3(64), 0, 0, dbnd.p, dra.p, dsum.p, nr);
    TRY(finish("xee_exchange_conversion"));
    TRY(dbnd.get(bndconv));
    return dsum.get(sum);
}

}  // extern "C"
