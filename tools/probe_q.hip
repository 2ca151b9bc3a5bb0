// probe_q.hip -- developer timeline of the blocked sweep (k_sweep_q): per-workgroup s_memtime stamps at the
// phase boundaries of a tile, for the bench workload (32 cases of 1024x512, BAROTROPIC, T = 6).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt \
//         -Iinclude tools/probe_q.hip -o tools/bin/probe_q
#include "../xlab-ee-fortran_amd/csrc/xee_sweepq.hip"

#include <algorithm>
#include <cmath>
#include <map>
#include <vector>

// experiment: the same tile body in a loop (grid = resident workgroups), with an optional start delay for half of them
template <int NW, int R, bool STAMP>
__global__ __launch_bounds__(NW * 64, 4) void k_probe_loop(xee::Grid g, const float* __restrict__ psi_in, float* __restrict__ psi_out,
        const float* __restrict__ base, const float* __restrict__ f, float alpha, int T, int ntx, int nty, int stagger, int swizzle,
        unsigned long long* __restrict__ stamps)
{
    using namespace xee;
    __shared__ float edge[2][2][NW][64];
    __shared__ int vote[2][NW];
    constexpr int H = NW * R;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tiles_per_member = ntx * nty;
    const int ntiles = tiles_per_member * g.members;
    const int nb = gridDim.x;
    const int b = (swizzle && nb % 8 == 0) ? (int)(blockIdx.x % 8) * (nb / 8) + (int)(blockIdx.x / 8) : (int)blockIdx.x;
    if (stagger > 0 && ((int)blockIdx.x / 8) % 2 == 1)       // every other workgroup of an XCD
        for (int i = 0; i < stagger; ++i) __builtin_amdgcn_s_sleep(127);
    for (int tile = b, k = 0; tile < ntiles; tile += nb, ++k) {
        const int m = tile / tiles_per_member, rem = tile % tiles_per_member;
        TileGeom tg;
        tg.lane = lane; tg.w = w;
        tg.x0 = (rem % ntx) * (64 - 2 * T) - T;
        tg.y0 = (rem / ntx) * (H - 2 * T) - T;
        tg.gx = tg.x0 + lane; tg.gy0 = tg.y0 + w * R;
        tg.col_in = tg.gx >= 0 && tg.gx < g.nx;
        tg.col_int = tg.gx >= 1 && tg.gx <= g.nx - 2;
        tg.vx = (lane >= T || tg.x0 <= 0) && (lane < 64 - T || tg.x0 + 63 >= g.nx - 1);
        tg.mo = (size_t)m * g.plane;
        tg.ks = (size_t)g.members * g.plane;
        tg.stamps = STAMP ? stamps + 32 * (size_t)tile : nullptr;
        if constexpr (STAMP) {
            stamp<STAMP>(tg, 0);
            if (threadIdx.x == 0)
                tg.stamps[7] = (unsigned long long)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) |
                               ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) << 32);
        }
        if (!sweep_tile<NW, R, false, true, true, STAMP>(g, tg, edge, vote[k & 1], psi_in, psi_out, base, f, nullptr, alpha, T)) {
            sweep_tile<NW, R, false, true, false, STAMP>(g, tg, edge, vote[k & 1], psi_in, psi_out, base, f, nullptr, alpha, T);
            __syncthreads();
        }
        stamp<STAMP>(tg, 6);
    }
}

// experiment: the slim star body (103 VGPRs) in a tile loop
template <int NW, int R>
__global__ __launch_bounds__(NW * 64, 4) void k_probe_sloop(xee::Grid g, const float* __restrict__ psi_in, float* __restrict__ psi_out,
        const float* __restrict__ base, const float* __restrict__ f, float alpha, int T, int ntx, int nty, int swizzle)
{
    using namespace xee;
    __shared__ xee_f2 fy[NW][R][64];
    __shared__ float edge[2][2][NW][64];
    __shared__ int vote[2][NW];
    constexpr int H = NW * R;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tiles_per_member = ntx * nty;
    const int ntiles = tiles_per_member * g.members;
    const int nb = gridDim.x;
    const int b = (swizzle && nb % 8 == 0) ? (int)(blockIdx.x % 8) * (nb / 8) + (int)(blockIdx.x / 8) : (int)blockIdx.x;
    for (int tile = b, k = 0; tile < ntiles; tile += nb, ++k) {
        const int m = tile / tiles_per_member, rem = tile % tiles_per_member;
        TileGeom tg;
        tg.lane = lane; tg.w = w;
        tg.x0 = (rem % ntx) * (64 - 2 * T) - T;
        tg.y0 = (rem / ntx) * (H - 2 * T) - T;
        tg.gx = tg.x0 + lane; tg.gy0 = tg.y0 + w * R;
        tg.col_in = tg.gx >= 0 && tg.gx < g.nx;
        tg.col_int = tg.gx >= 1 && tg.gx <= g.nx - 2;
        tg.vx = (lane >= T || tg.x0 <= 0) && (lane < 64 - T || tg.x0 + 63 >= g.nx - 1);
        tg.mo = (size_t)m * g.plane;
        tg.ks = (size_t)g.members * g.plane;
        tg.stamps = nullptr;
        slim_tile<NW, R, false>(g, tg, fy, edge, vote[k & 1], psi_in, psi_out, base, f, nullptr, alpha, T, nullptr);
    }
}

template <int NW, int R>
__global__ __launch_bounds__(NW * 64, 4) void k_probe_s1(xee::Grid g, const float* __restrict__ psi_in, float* __restrict__ psi_out,
        const float* __restrict__ base, const float* __restrict__ f, float alpha, int T)
{
    using namespace xee;
    __shared__ xee_f2 fy[NW][R][64];
    __shared__ float edge[2][2][NW][64];
    __shared__ int vote[NW];
    constexpr int H = NW * R;
    TileGeom tg;
    tg.lane = threadIdx.x & 63; tg.w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    tg.x0 = (int)blockIdx.x * (64 - 2 * T) - T;
    tg.y0 = (int)blockIdx.y * (H - 2 * T) - T;
    tg.gx = tg.x0 + tg.lane; tg.gy0 = tg.y0 + tg.w * R;
    tg.col_in = tg.gx >= 0 && tg.gx < g.nx;
    tg.col_int = tg.gx >= 1 && tg.gx <= g.nx - 2;
    tg.vx = (tg.lane >= T || tg.x0 <= 0) && (tg.lane < 64 - T || tg.x0 + 63 >= g.nx - 1);
    tg.mo = (size_t)blockIdx.z * g.plane;
    tg.ks = (size_t)g.members * g.plane;
    tg.stamps = nullptr;
    slim_tile<NW, R, false>(g, tg, fy, edge, vote, psi_in, psi_out, base, f, nullptr, alpha, T, nullptr);
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

int main(int argc, char** argv)
{
    using namespace xee;
    const int T = argc > 1 ? atoi(argv[1]) : 6;
    const int members = argc > 2 ? atoi(argv[2]) : 32;
    const int loop_mode = argc > 3 ? atoi(argv[3]) : 0;      // 1: tile-loop kernel
    const int stagger = argc > 4 ? atoi(argv[4]) : 0;
    const int swizzle = argc > 5 ? atoi(argv[5]) : 1;
    const int per_cu = argc > 6 ? atoi(argv[6]) : 2;
    Grid g{};
    g.nx = 1024; g.ny = 512; g.pitch = 1024; g.plane = (size_t)g.pitch * g.ny; g.members = members;
    const size_t n = g.plane * members;
    std::vector<float> h(n);
    float *psi0, *psi1, *f, *base;
    CK(hipMalloc(&psi0, n * 4)); CK(hipMalloc(&psi1, n * 4)); CK(hipMalloc(&f, n * 4)); CK(hipMalloc(&base, 4 * n * 4));
    unsigned s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)(s >> 8) * (1.0f / 16777216.0f); };
    for (auto& v : h) v = 1e-3f * (rnd() - 0.5f);
    CK(hipMemcpy(psi0, h.data(), n * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(psi1, h.data(), n * 4, hipMemcpyHostToDevice));
    for (auto& v : h) v = 1e-6f * (rnd() - 0.5f);
    CK(hipMemcpy(f, h.data(), n * 4, hipMemcpyHostToDevice));
    for (int k = 0; k < 4; ++k) {
        for (auto& v : h) v = k < 2 ? (1.0f + 0.2f * rnd()) : 0.0f;
        CK(hipMemcpy(base + (size_t)k * n, h.data(), n * 4, hipMemcpyHostToDevice));
    }
    #ifndef PROBE_NW
#define PROBE_NW 8
#define PROBE_R 8
#endif
    constexpr int NW = PROBE_NW, R = PROBE_R, H = NW * R;
    dim3 grid((g.nx - 1 + (64 - 2 * T) - 1) / (64 - 2 * T), (g.ny - 1 + (H - 2 * T) - 1) / (H - 2 * T), g.members);
    const size_t nwg = (size_t)grid.x * grid.y * grid.z;
    unsigned long long* stamps;
    CK(hipMalloc(&stamps, nwg * 32 * sizeof(unsigned long long)));
    CK(hipMemset(stamps, 0, nwg * 32 * sizeof(unsigned long long)));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms_plain = 0, ms_stamp = 0;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        for (int k = 0; k < 10; ++k)
            if (loop_mode == 2)
                hipLaunchKernelGGL((k_probe_sloop<NW, R>), dim3(256 * per_cu), dim3(NW * 64), 0, 0, g, (k & 1) ? psi1 : psi0, (k & 1) ? psi0 : psi1,
                                   base, f, 1.0f, T, (int)grid.x, (int)grid.y, swizzle);
            else if (loop_mode == 3)
                hipLaunchKernelGGL((k_probe_s1<NW, R>), grid, dim3(NW * 64), 0, 0, g, (k & 1) ? psi1 : psi0, (k & 1) ? psi0 : psi1, base, f, 1.0f, T);
            else if (loop_mode)
                hipLaunchKernelGGL((k_probe_loop<NW, R, false>), dim3(256 * per_cu), dim3(NW * 64), 0, 0, g, (k & 1) ? psi1 : psi0, (k & 1) ? psi0 : psi1,
                                   base, f, 1.0f, T, (int)grid.x, (int)grid.y, stagger, swizzle, (unsigned long long*)nullptr);
            else
            hipLaunchKernelGGL((k_sweep_q<NW, R, false, true, false>), grid, dim3(NW * 64), 0, 0, g, (k & 1) ? psi1 : psi0, (k & 1) ? psi0 : psi1,
                               base, f, (float*)nullptr, (const MemberState*)nullptr, 1.0f, T, 0, (unsigned long long*)nullptr);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms_plain, e0, e1));
    }
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        if (loop_mode)
            hipLaunchKernelGGL((k_probe_loop<NW, R, true>), dim3(256 * per_cu), dim3(NW * 64), 0, 0, g, psi0, psi1, base, f, 1.0f, T,
                               (int)grid.x, (int)grid.y, stagger, swizzle, stamps);
        else
        hipLaunchKernelGGL((k_sweep_q<NW, R, false, true, true>), grid, dim3(NW * 64), 0, 0, g, psi0, psi1, base, f, (float*)nullptr,
                           (const MemberState*)nullptr, 1.0f, T, 0, stamps);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms_stamp, e0, e1));
    }
    std::vector<unsigned long long> st(nwg * 32);
    CK(hipMemcpy(st.data(), stamps, nwg * 32 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    // (s_memtime is per XCD: stamps are only compared within one CU)
    std::map<unsigned long long, std::vector<size_t>> cu;
    for (size_t i = 0; i < nwg; ++i) cu[((st[i * 32 + 7] >> 8) & 0xffull) | ((st[i * 32 + 7] >> 32) << 8)].push_back(i);
    std::vector<double> spans;
    for (auto& kv : cu) {
        unsigned long long a0 = ~0ull, a1 = 0;
        for (size_t i : kv.second) { a0 = std::min(a0, st[i * 32]); a1 = std::max(a1, st[i * 32 + 6]); }
        spans.push_back((double)(a1 - a0));
    }
    std::sort(spans.begin(), spans.end());
    const double tick_us = (ms_plain * 100.0) / spans[spans.size() / 2];      // plain launch time over the median CU's stamped span
    printf("T=%d members=%d: %zu workgroups; plain %.1f us/launch, stamped %.1f us/launch; median CU span %.0f ticks -> %.1f MHz\n",
           T, members, nwg, ms_plain * 100.0, ms_stamp * 1e3, spans[spans.size() / 2], 1.0 / tick_us);
    const char* names[6] = {"start->loads issued", "loads issued->weights built", "weights->T steps done", "steps->validated", "validated->stores issued", "stores->end"};
    double sum[6] = {0};
    for (size_t i = 0; i < nwg; ++i)
        for (int k = 0; k < 6; ++k) sum[k] += (double)(st[i * 32 + k + 1] - st[i * 32 + k]);
    for (int k = 0; k < 6; ++k) printf("  %-30s %8.0f ticks  %7.3f us\n", names[k], sum[k] / nwg, sum[k] / nwg * tick_us);
    // per-CU timeline: how much of a CU's time has 0 / 1 / 2 workgroups inside their T-step loop
    double inloop[4] = {0, 0, 0, 0}, resident[4] = {0, 0, 0, 0};
    for (auto& kv : cu) {
        std::vector<std::pair<unsigned long long, int>> ev, evr;
        for (size_t i : kv.second) {
            ev.push_back({st[i * 32 + 2], +1}); ev.push_back({st[i * 32 + 3], -1});
            evr.push_back({st[i * 32 + 0], +1}); evr.push_back({st[i * 32 + 6], -1});
        }
        auto acc = [&](std::vector<std::pair<unsigned long long, int>>& e, double* out) {
            std::sort(e.begin(), e.end());
            int n = 0; unsigned long long prev = e.front().first;
            for (auto& x : e) { out[std::min(n, 3)] += (double)(x.first - prev); prev = x.first; n += x.second; }
        };
        acc(evr, resident); { auto e2 = ev; e2.push_back({evr.empty() ? 0 : std::min_element(evr.begin(), evr.end())->first, 0}); acc(e2, inloop); }
    }
    double tot = 0; for (double x : spans) tot += x;
    printf("  %zu CUs seen, %.1f workgroups each\n", cu.size(), (double)nwg / cu.size());
    printf("  CU time with 0/1/2/3+ workgroups resident: %.3f %.3f %.3f %.3f\n", resident[0] / tot, resident[1] / tot, resident[2] / tot, resident[3] / tot);
    printf("  CU time with 0/1/2/3+ workgroups in the T-step loop: %.3f %.3f %.3f %.3f\n", inloop[0] / tot, inloop[1] / tot, inloop[2] / tot, inloop[3] / tot);
    double gaps = 0; size_t ng = 0, fallbacks = 0;
    {   // inside the loop (wave 0 of each workgroup): barrier wait, inner rows, edge rows + LDS write, per step
        double bw[6] = {0}, inner[6] = {0}, tail[6] = {0}, gap[6] = {0};
        for (size_t i = 0; i < nwg; ++i)
            for (int t = 0; t < std::min(T, 6); ++t) {
                const unsigned long long* x = &st[i * 32 + 8 + 4 * t];
                bw[t] += (double)(x[1] - x[0]); inner[t] += (double)(x[2] - x[1]); tail[t] += (double)(x[3] - x[2]);
                if (t + 1 < std::min(T, 6)) gap[t] += (double)(x[4] - x[3]);
            }
        for (int t = 0; t < std::min(T, 6); ++t)
            printf("  step %d: barrier wait %6.0f  inner rows %6.0f  edge rows+write %6.0f  to next %5.0f ticks\n", t, bw[t] / nwg, inner[t] / nwg, tail[t] / nwg, gap[t] / nwg);
    }
    (void)gaps; (void)ng; (void)fallbacks;
    return 0;
}
