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

// the four generating planes from the solver coefficients (the divisions of elliptic_tools.f90:37-44), and a copy of b as a fifth
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
    base[o + 4 * ks] = b ? b[o] : 0.f;                    // (k_sweep_z<nine> reads b and divides itself: one plane instead of BX and BY)
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

void launch_sweep_tb4(const Grid& g, const float* psi_in, float* psi_out, const float* base, const float* f,
                      float* res, const MemberState* st, float alpha, int T, bool barotropic, const SweepOpts& o,
                      SweepDesc* desc, hipStream_t s)
{
    int cfg = o.tile;
    if (cfg != 816 && cfg != 808 && cfg != 416) cfg = 808;
    if (desc) { desc->name = "k_sweep_tb4"; desc->T = T; desc->tile_w = 64; desc->tile_h = (cfg / 100) * (cfg % 100); }
#define TB4_CASE(NW, R) case NW * 100 + R: \
        if (barotropic) launch_tb4_cfg<NW, R, true>(g, psi_in, psi_out, base, f, res, st, alpha, T, s); \
        else launch_tb4_cfg<NW, R, false>(g, psi_in, psi_out, base, f, res, st, alpha, T, s); \
        break;
    switch (cfg) {
        TB4_CASE(8, 16)
        TB4_CASE(8, 8)
        TB4_CASE(4, 16)
    }
#undef TB4_CASE
}

}  // namespace xee
