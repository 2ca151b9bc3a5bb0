// xee_sweep4.hip -- the temporally blocked Jacobi sweep fed by the operator's FOUR GENERATING PLANES
// instead of its nine derived coefficient planes.
//
// cal_coe (xtt-lib-fortran/elliptic_tools.f90:37-54) builds the 9 stencil weights of a point from 8
// pre-divided staggered quantities, each of which it shares with a neighbour:
//     Ap = a(i,j-1)/PP      = AP(i,j)        Am = AP(i-1,j)
//     Cp = c(i-1,j)/QQ      = CQ(i,j)        Cm = CQ(i,j-1)
//     BXp = (b(i,j)+b(i,j-1))/(2*PQ4) = BX(i,j)     BXm = BX(i-1,j)
//     BYp = (b(i-1,j)+b(i,j))/(2*PQ4) = BY(i,j)     BYm = BY(i,j-1)
// The blocked kernel is bound by the bytes it loads per launch, so it loads AP, CQ, BX, BY (16 B per
// point; only AP, CQ = 8 B for the BAROTROPIC operator whose b is zero) and rebuilds w1..w9 in its
// prologue with cal_coe's own expressions -- i-1 from the neighbouring lane (DPP), j-1 from the row
// below in the same registers.  Same operands, same operations, same order => the same bits as the
// stored coefficients (tests compare the two paths bitwise).  44 -> 24 B per loaded point
// (28 -> 16 B BAROTROPIC).  The iteration loop is identical to k_sweep_tb (xee_kernels.hip).
#include "xee_device.h"

#include <cstdio>
#include <cstdlib>

#pragma clang fp contract(off)

namespace xee {

namespace {

// bound_ctrl: the lane without a source reads 0, so no `old` register has to be prepared
__device__ __forceinline__ float shr1(float src)   // lane l <- lane l-1 (lane 0 gets 0)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, src), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float shl1(float src)   // lane l <- lane l+1 (lane 63 gets 0)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, src), 0x130, 0xf, 0xf, true));
}

}  // namespace

// the four generating planes from the solver coefficients (the divisions of elliptic_tools.f90:37-44)
__global__ __launch_bounds__(256) void k_base_planes(Grid g, const float* __restrict__ a, const float* __restrict__ b,
        const float* __restrict__ c, float* __restrict__ base, float dx, float dy)
{
    const int i = blockIdx.x * 64 + threadIdx.x;
    const int j = blockIdx.y * 4 + threadIdx.y;
    if (i >= g.nx || j >= g.ny) return;
    const size_t o = (size_t)blockIdx.z * g.plane + (size_t)j * g.pitch + i;
    const size_t ks = (size_t)g.members * g.plane;
    const float PP = dx * dx, QQ = dy * dy, PQ4 = 4.0f * dx * dy;
    const float two_PQ4 = 2.0f * PQ4;
    float ap = 0.f, cq = 0.f, bx = 0.f, by = 0.f;
    if (i <= g.nx - 2 && j >= 1 && j <= g.ny - 2) {
        ap = a[o - g.pitch] / PP;                                          // a(i, j-1)
        bx = ((b ? b[o] : 0.f) + (b ? b[o - g.pitch] : 0.f)) / two_PQ4;    // b(i,j) + b(i,j-1)
    }
    if (i >= 1 && i <= g.nx - 2 && j <= g.ny - 2) {
        cq = c[o - 1] / QQ;                                                // c(i-1, j)
        by = ((b ? b[o - 1] : 0.f) + (b ? b[o] : 0.f)) / two_PQ4;          // b(i-1,j) + b(i,j)
    }
    base[o] = ap;
    base[o + ks] = cq;
    base[o + 2 * ks] = bx;
    base[o + 3 * ks] = by;
}

template <int NW, int R, bool WRITE_R, bool BARO>
__global__ __launch_bounds__(NW * 64) void k_sweep_tb4(Grid g, const float* __restrict__ psi_in,
        float* __restrict__ psi_out, const float* __restrict__ base, const float* __restrict__ f,
        float* __restrict__ res, const MemberState* __restrict__ st, float alpha, int T)
{
    __shared__ float edge[2][2][NW][64];
    const int m = blockIdx.z;
    if (st && !st[m].active) return;                      // workgroup-uniform
    const int lane = threadIdx.x & 63;
    const int w = threadIdx.x >> 6;
    constexpr int H = NW * R;
    const int x0 = (int)blockIdx.x * (64 - 2 * T) - T;
    const int y0 = (int)blockIdx.y * (H - 2 * T) - T;
    const int gx = x0 + lane;
    const int gy0 = y0 + w * R;
    const bool col_in = gx >= 0 && gx < g.nx;
    const bool col_int = gx >= 1 && gx <= g.nx - 2;
    const size_t mo = (size_t)m * g.plane;
    const size_t ks = (size_t)g.members * g.plane;

    // ---- prologue: load AP, CQ (, BX, BY), f, psi and rebuild the nine weights ----------------
    float c[9][R], ff[R], p[R];
    {
        // row gy0-1 of CQ and BY (the j-1 operands of this wave's first row)
        const int gyb = gy0 - 1;
        const bool inb = col_in && gyb >= 0 && gyb < g.ny;
        const size_t ob = mo + (size_t)(inb ? gyb : 0) * g.pitch + (inb ? gx : 0);
        float cq_m = inb ? base[ks + ob] : 0.0f;
        float by_m = (!BARO && inb) ? base[3 * ks + ob] : 0.0f;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int gy = gy0 + r;
            const bool in = col_in && gy >= 0 && gy < g.ny;
            const bool interior = col_int && gy >= 1 && gy <= g.ny - 2;
            const size_t o = mo + (size_t)(in ? gy : 0) * g.pitch + (in ? gx : 0);
            p[r] = in ? psi_in[o] : 0.0f;
            ff[r] = interior ? f[o] : 0.0f;
            const float Ap = in ? base[o] : 0.0f;
            const float Cp = in ? base[ks + o] : 0.0f;
            const float BXp = (!BARO && in) ? base[2 * ks + o] : 0.0f;
            const float BYp = (!BARO && in) ? base[3 * ks + o] : 0.0f;
            const float Am = shr1(Ap);                    // AP(i-1,j)
            const float BXm = BARO ? 0.0f : shr1(BXp);    // BX(i-1,j)
            const float Cm = cq_m, BYm = by_m;            // CQ(i,j-1), BY(i,j-1)
            // elliptic_tools.f90:46-54
            const float w1 = -(BXm + BYp);
            const float w2 = Cp + (BXp - BXm);
            const float w3 = BXp + BYp;
            const float w4 = Am - (BYp - BYm);
            const float w5 = -(Am + Ap + Cm + Cp);
            const float w6 = Ap + (BYp - BYm);
            const float w7 = BXm + BYm;
            const float w8 = Cm - (BXp - BXm);
            const float w9 = -(BXp + BYm);
            // no masking of the weights: a non-interior point's update is discarded below (`interior ?`),
            // and unmasked the four corner weights of the BAROTROPIC operator are compile-time +-0
            c[0][r] = w1; c[1][r] = w2; c[2][r] = w3; c[3][r] = w4; c[4][r] = w5;
            c[5][r] = w6; c[6][r] = w7; c[7][r] = w8; c[8][r] = w9;
            (void)interior;
            cq_m = Cp;
            by_m = BYp;
        }
    }

    // ---- T iterations (same as k_sweep_tb) ----------------------------------------------------
    const bool vx = (lane >= T || x0 <= 0) && (lane < 64 - T || x0 + 63 >= g.nx - 1);
    // one Jacobi step src -> dst (two register sets, so no value has to be moved between iterations)
    float q[R];
    auto step = [&](const float (&src)[R], float (&dst)[R], int t) {
        const int par = t & 1;
        edge[par][0][w][lane] = src[R - 1];
        edge[par][1][w][lane] = src[0];
        __syncthreads();
        const float below = (w > 0) ? edge[par][0][w - 1][lane] : 0.0f;
        const float above = (w < NW - 1) ? edge[par][1][w + 1][lane] : 0.0f;
        // the i-1 / i+1 views of rows -1..R of the old iterate, one DPP move each
        float lft[R + 2], rgt[R + 2];
        lft[0] = shr1(below); rgt[0] = shl1(below);
#pragma unroll
        for (int r = 0; r < R; ++r) { lft[r + 1] = shr1(src[r]); rgt[r + 1] = shl1(src[r]); }
        lft[R + 1] = shr1(above); rgt[R + 1] = shl1(above);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            // (skipping the rows that can no longer reach the stored region -- the valid band shrinks by
            // one row per step and side -- was measured: the extra branches cost more than they save)
            const float om = (r > 0) ? src[r - 1] : below;
            const float oc = src[r];
            const float on = (r + 1 < R) ? src[r + 1] : above;
            float Rv = c[0][r] * lft[r + 2];              // elliptic_tools.f90:77-85, left to right
            Rv = Rv + c[1][r] * on;
            Rv = Rv + c[2][r] * rgt[r + 2];
            Rv = Rv + c[3][r] * lft[r + 1];
            Rv = Rv + c[4][r] * oc;
            Rv = Rv + c[5][r] * rgt[r + 1];
            Rv = Rv + c[6][r] * lft[r];
            Rv = Rv + c[7][r] * om;
            Rv = Rv + c[8][r] * rgt[r];
            Rv = Rv - ff[r];                              // :190
            const float upd = oc + alpha * Rv / (-c[4][r]);   // :238
            const int gy = gy0 + r;
            const bool interior = col_int && gy >= 1 && gy <= g.ny - 2;
            dst[r] = interior ? upd : oc;
            if (WRITE_R && t == T - 1) {
                const int ly = w * R + r;
                const bool vy = (ly >= T || y0 <= 0) && (ly < H - T || y0 + H - 1 >= g.ny - 1);
                if (vx && vy && interior) res[mo + (size_t)gy * g.pitch + gx] = Rv;
            }
        }
    };
    for (int t = 0; t < T; t += 2) {
        step(p, q, t);
        if (t + 1 < T) step(q, p, t + 1);
    }
    const bool in_q = (T & 1) != 0;                       // odd T: the last step wrote q

#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int ly = w * R + r;
        const int gy = gy0 + r;
        const bool vy = (ly >= T || y0 <= 0) && (ly < H - T || y0 + H - 1 >= g.ny - 1);
        const bool interior = col_int && gy >= 1 && gy <= g.ny - 2;
        if (vx && vy && interior) psi_out[mo + (size_t)gy * g.pitch + gx] = in_q ? q[r] : p[r];
    }
}

// =============================================================================================
// Persistent form for the BAROTROPIC operator: a workgroup walks over tiles and, while it iterates
// on tile k, the planes of tile k+1 (psi, f, AP, CQ: 4R+1 rows of 64 floats per wave) are already
// streaming into LDS by LDS-DMA (global_load_lds: no registers, survives the loop's barriers because
// those are raw s_barrier + lgkmcnt waits, never a vmcnt drain).  This takes the load latency and
// the workgroup dispatch out of the per-tile cost; two workgroups stay resident per CU.
// Same arithmetic as k_sweep_tb4<.,.,.,true>.
// =============================================================================================
template <int NW, int R, bool WRITE_R>
__global__ __launch_bounds__(NW * 64, 4) void k_sweep_tb4p(Grid g, const float* __restrict__ psi_in,
        float* __restrict__ psi_out, const float* __restrict__ base, const float* __restrict__ f,
        float* __restrict__ res, const MemberState* __restrict__ st, float alpha, int T, int ntx, int nty)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int H = NW * R;
    constexpr int NROW = 4 * R + 1;                       // psi[R], f[R], AP[R], CQ[R+1] (row gy0-1 first)
    const int lane = threadIdx.x & 63;
    const int w = threadIdx.x >> 6;
    float* mine = smem + (size_t)w * NROW * 64;           // this wave's landing zone
    float* edge = smem + (size_t)NW * NROW * 64;          // [2][2][NW][64]
    const size_t ks = (size_t)g.members * g.plane;
    const int tiles_per_member = ntx * nty;
    const int ntiles = tiles_per_member * g.members;

    auto lds_row = [&](int row) {
        return (__attribute__((address_space(3))) void*)(mine + row * 64);
    };
    // start the DMA of one tile's rows; lanes outside the domain load nothing (masked when read)
    auto issue = [&](int tile) {
        const int m = tile / tiles_per_member, rem = tile % tiles_per_member;
        const int x0 = (rem % ntx) * (64 - 2 * T) - T, y0 = (rem / ntx) * (H - 2 * T) - T;
        const int gx = x0 + lane, gy0 = y0 + w * R;
        const bool col_in = gx >= 0 && gx < g.nx;
        const size_t mo = (size_t)m * g.plane;
        {
            const int gyb = gy0 - 1;
            if (col_in && gyb >= 0 && gyb < g.ny)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + ks + mo + (size_t)gyb * g.pitch + gx), lds_row(3 * R), 4, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int gy = gy0 + r;
            if (col_in && gy >= 0 && gy < g.ny) {
                const size_t o = mo + (size_t)gy * g.pitch + gx;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(psi_in + o), lds_row(r), 4, 0, 0);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(f + o), lds_row(R + r), 4, 0, 0);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + o), lds_row(2 * R + r), 4, 0, 0);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + ks + o), lds_row(3 * R + 1 + r), 4, 0, 0);
            }
        }
    };

    int tile = blockIdx.x;
    if (tile < ntiles) issue(tile);
    for (; tile < ntiles; tile += gridDim.x) {
        const int m = tile / tiles_per_member, rem = tile % tiles_per_member;
        const int x0 = (rem % ntx) * (64 - 2 * T) - T, y0 = (rem / ntx) * (H - 2 * T) - T;
        const int gx = x0 + lane, gy0 = y0 + w * R;
        const bool col_in = gx >= 0 && gx < g.nx;
        const bool col_int = gx >= 1 && gx <= g.nx - 2;
        const size_t mo = (size_t)m * g.plane;
        const bool act = !st || st[m].active;             // workgroup-uniform

        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's rows have landed
        float c[9][R], ff[R], p[R], q[R];
        {
            const int gyb = gy0 - 1;
            float cq_m = (col_in && gyb >= 0 && gyb < g.ny) ? mine[(3 * R) * 64 + lane] : 0.0f;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int gy = gy0 + r;
                const bool in = col_in && gy >= 0 && gy < g.ny;
                const bool interior = col_int && gy >= 1 && gy <= g.ny - 2;
                p[r] = in ? mine[r * 64 + lane] : 0.0f;
                ff[r] = interior ? mine[(R + r) * 64 + lane] : 0.0f;
                const float Ap = in ? mine[(2 * R + r) * 64 + lane] : 0.0f;
                const float Cp = in ? mine[(3 * R + 1 + r) * 64 + lane] : 0.0f;
                const float BXp = 0.0f, BYp = 0.0f, BXm = 0.0f, BYm = 0.0f;
                const float Am = shr1(Ap);
                const float Cm = cq_m;
                c[0][r] = -(BXm + BYp);                   // elliptic_tools.f90:46-54
                c[1][r] = Cp + (BXp - BXm);
                c[2][r] = BXp + BYp;
                c[3][r] = Am - (BYp - BYm);
                c[4][r] = -(Am + Ap + Cm + Cp);
                c[5][r] = Ap + (BYp - BYm);
                c[6][r] = BXm + BYm;
                c[7][r] = Cm - (BXp - BXm);
                c[8][r] = -(BXp + BYm);
                cq_m = Cp;
            }
        }
        // every value is in registers: the landing zone may be overwritten by the next tile's DMA
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (tile + (int)gridDim.x < ntiles) issue(tile + gridDim.x);
        if (!act) continue;                               // (uniform) stopped member: nothing to compute

        const bool vx = (lane >= T || x0 <= 0) && (lane < 64 - T || x0 + 63 >= g.nx - 1);
        auto step = [&](const float (&src)[R], float (&dst)[R], int t) {
            float* eb = edge + (size_t)(t & 1) * 2 * NW * 64;
            eb[w * 64 + lane] = src[R - 1];               // top row, for the wave above
            eb[(NW + w) * 64 + lane] = src[0];            // bottom row, for the wave below
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // raw barrier: no vmcnt drain
            const float below = (w > 0) ? eb[(w - 1) * 64 + lane] : 0.0f;
            const float above = (w < NW - 1) ? eb[(NW + w + 1) * 64 + lane] : 0.0f;
            float lft[R + 2], rgt[R + 2];
            lft[0] = shr1(below); rgt[0] = shl1(below);
#pragma unroll
            for (int r = 0; r < R; ++r) { lft[r + 1] = shr1(src[r]); rgt[r + 1] = shl1(src[r]); }
            lft[R + 1] = shr1(above); rgt[R + 1] = shl1(above);
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const float om = (r > 0) ? src[r - 1] : below;
                const float oc = src[r];
                const float on = (r + 1 < R) ? src[r + 1] : above;
                float Rv = c[0][r] * lft[r + 2];          // elliptic_tools.f90:77-85, left to right
                Rv = Rv + c[1][r] * on;
                Rv = Rv + c[2][r] * rgt[r + 2];
                Rv = Rv + c[3][r] * lft[r + 1];
                Rv = Rv + c[4][r] * oc;
                Rv = Rv + c[5][r] * rgt[r + 1];
                Rv = Rv + c[6][r] * lft[r];
                Rv = Rv + c[7][r] * om;
                Rv = Rv + c[8][r] * rgt[r];
                Rv = Rv - ff[r];                          // :190
                const float upd = oc + alpha * Rv / (-c[4][r]);   // :238
                const int gy = gy0 + r;
                const bool interior = col_int && gy >= 1 && gy <= g.ny - 2;
                dst[r] = interior ? upd : oc;
                if (WRITE_R && t == T - 1) {
                    const int ly = w * R + r;
                    const bool vy = (ly >= T || y0 <= 0) && (ly < H - T || y0 + H - 1 >= g.ny - 1);
                    if (vx && vy && interior) res[mo + (size_t)gy * g.pitch + gx] = Rv;
                }
            }
        };
        for (int t = 0; t < T; t += 2) {
            step(p, q, t);
            if (t + 1 < T) step(q, p, t + 1);
        }
        const bool in_q = (T & 1) != 0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int ly = w * R + r;
            const int gy = gy0 + r;
            const bool vy = (ly >= T || y0 <= 0) && (ly < H - T || y0 + H - 1 >= g.ny - 1);
            const bool interior = col_int && gy >= 1 && gy <= g.ny - 2;
            if (vx && vy && interior) psi_out[mo + (size_t)gy * g.pitch + gx] = in_q ? q[r] : p[r];
        }
        // the edge slots of this tile's last steps may still be read by slower waves: one barrier
        // before the next tile's first step writes them again
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
}

void launch_base_planes(const Grid& g, const float* sA, const float* sB, const float* sC, float* base,
                        float dx, float dy, hipStream_t s)
{
    hipLaunchKernelGGL(k_base_planes, dim3((g.nx + 63) / 64, (g.ny + 3) / 4, g.members), dim3(64, 4, 1), 0, s,
                       g, sA, sB, sC, base, dx, dy);
}

template <int NW, int R, bool BARO>
static void launch_tb4_cfg(const Grid& g, const float* psi_in, float* psi_out, const float* base, const float* f,
                           float* res, const MemberState* st, float alpha, int T, hipStream_t s)
{
    constexpr int H = NW * R;
    dim3 grid((g.nx - 1 + (64 - 2 * T) - 1) / (64 - 2 * T), (g.ny - 1 + (H - 2 * T) - 1) / (H - 2 * T), g.members);
    if (res)
        hipLaunchKernelGGL((k_sweep_tb4<NW, R, true, BARO>), grid, dim3(NW * 64), 0, s, g, psi_in, psi_out, base, f, res, st, alpha, T);
    else
        hipLaunchKernelGGL((k_sweep_tb4<NW, R, false, BARO>), grid, dim3(NW * 64), 0, s, g, psi_in, psi_out, base, f, res, st, alpha, T);
}

template <int NW, int R>
static bool launch_tb4p_cfg(const Grid& g, const float* psi_in, float* psi_out, const float* base, const float* f,
                            float* res, const MemberState* st, float alpha, int T, hipStream_t s)
{
    constexpr int H = NW * R;
    constexpr size_t lds = ((size_t)NW * (4 * R + 1) * 64 + 2 * 2 * NW * 64) * sizeof(float);
    static bool ready = false, ok = false;
    if (!ready) {
        ready = true;
        ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_sweep_tb4p<NW, R, true>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess &&
             hipFuncSetAttribute(reinterpret_cast<const void*>(&k_sweep_tb4p<NW, R, false>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess;
    }
    if (!ok) return false;
    const int ntx = (g.nx - 1 + (64 - 2 * T) - 1) / (64 - 2 * T), nty = (g.ny - 1 + (H - 2 * T) - 1) / (H - 2 * T);
    const long ntiles = (long)ntx * nty * g.members;
    const int grid = (int)(ntiles < 512 ? ntiles : 512);      // two resident workgroups per CU, 256 CUs
    if (res)
        hipLaunchKernelGGL((k_sweep_tb4p<NW, R, true>), dim3(grid), dim3(NW * 64), lds, s, g, psi_in, psi_out, base, f, res, st, alpha, T, ntx, nty);
    else
        hipLaunchKernelGGL((k_sweep_tb4p<NW, R, false>), dim3(grid), dim3(NW * 64), lds, s, g, psi_in, psi_out, base, f, res, st, alpha, T, ntx, nty);
    return true;
}

void launch_sweep_tb4(const Grid& g, const float* psi_in, float* psi_out, const float* base, const float* f,
                      float* res, const MemberState* st, float alpha, int T, bool barotropic, hipStream_t s)
{
    if (barotropic && getenv("XEE_TB_PERSIST") && !getenv("XEE_TB_CFG") &&
        launch_tb4p_cfg<8, 8>(g, psi_in, psi_out, base, f, res, st, alpha, T, s))
        return;
    static const int cfg = [] {
        const char* e = getenv("XEE_TB_CFG");
        int nw = 0, r = 0;
        if (e && sscanf(e, "%dx%d", &nw, &r) == 2) return nw * 100 + r;
        return 0;
    }();
#define TB4_CASE(NW, R) case NW * 100 + R: \
        if (barotropic) launch_tb4_cfg<NW, R, true>(g, psi_in, psi_out, base, f, res, st, alpha, T, s); \
        else launch_tb4_cfg<NW, R, false>(g, psi_in, psi_out, base, f, res, st, alpha, T, s); \
        break;
    switch (cfg) {
        TB4_CASE(8, 16)
        TB4_CASE(8, 12)
        TB4_CASE(8, 8)
        TB4_CASE(4, 16)
        TB4_CASE(4, 12)
        TB4_CASE(16, 8)
        default:
            if (barotropic) launch_tb4_cfg<8, 8, true>(g, psi_in, psi_out, base, f, res, st, alpha, T, s);
            else launch_tb4_cfg<8, 8, false>(g, psi_in, psi_out, base, f, res, st, alpha, T, s);
    }
#undef TB4_CASE
}

}  // namespace xee
