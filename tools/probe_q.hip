// probe_q.hip -- developer timeline of the PRODUCTION blocked sweep (k_sweep_q<8,8,false,true,true> with its STAMP template
// flag on: the same code, plus an s_memtime at each phase boundary of a tile) on the bench workload shape
// (32 cases of 1024x512, BAROTROPIC, alpha = 1).  Prints, per phase, the median over workgroups and how many workgroups a
// CU has in its step loop on average.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize \
//         -Iinclude tools/probe_q.hip -o tools/bin/probe_q && tools/bin/probe_q [T] [members]
#include "../xlab-ee-fortran_amd/csrc/xee_sweepq.hip"

#include <algorithm>
#include <cmath>
#include <map>
#include <vector>

namespace xee {
int record_failure(int code, const char*) { return code; }
// (referenced by launch_sweep_q's fallback for planes of 2 GiB and more; never called here)
void launch_sweep_tb4(const Grid&, const float*, float*, const float*, const float*, float*, const MemberState*, float, int, bool,
                      const SweepOpts&, SweepDesc*, hipStream_t) {}
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv)
{
    using namespace xee;
    const int T = argc > 1 ? atoi(argv[1]) : 8, M = argc > 2 ? atoi(argv[2]) : 32;
    constexpr int NW = 8, R = 8, H = NW * R;
    Grid g; g.nx = 1024; g.ny = 512; g.pitch = 1024; g.members = M; g.plane = (size_t)g.pitch * g.ny;
    const size_t n = g.plane * M;
    float *psi0, *psi1, *base, *f;
    CK(hipMalloc((void**)&psi0, n * 4)); CK(hipMalloc((void**)&psi1, n * 4)); CK(hipMalloc((void**)&base, 4 * n * 4)); CK(hipMalloc((void**)&f, n * 4));
    {   // A = C = 1 on a unit square: AP = CQ = 1/dx^2-ish, f = smooth + noise, psi = 0
        std::vector<float> h(n), z(n, 0.0f);
        for (size_t q = 0; q < n; ++q) h[q] = 1.0e6f * (1.0f + 1e-3f * (float)((q * 2654435761u) % 1000) / 1000.0f);
        CK(hipMemcpy(base, h.data(), n * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(base + n, h.data(), n * 4, hipMemcpyHostToDevice));
        CK(hipMemset(base + 2 * n, 0, 2 * n * 4));
        for (size_t q = 0; q < n; ++q) h[q] = 1e-2f * sinf(0.01f * (float)(q % 1024)) + 1e-4f * (float)((q * 40503u) % 997) / 997.0f + 1e-5f;
        CK(hipMemcpy(f, h.data(), n * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(psi0, z.data(), n * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(psi1, z.data(), n * 4, hipMemcpyHostToDevice));
    }
    const int ntx = (g.nx - 1 + (64 - 2 * T) - 1) / (64 - 2 * T), nty = (g.ny - 1 + (H - 2 * T) - 1) / (H - 2 * T);
    const size_t nwg = (size_t)ntx * nty * M;
    const dim3 grid((unsigned)(8 * ((nwg + 7) / 8)));
    unsigned long long* stamps;
    CK(hipMalloc((void**)&stamps, nwg * 32 * 8)); CK(hipMemset(stamps, 0, nwg * 32 * 8));
    int* redo; CK(hipMalloc((void**)&redo, 4096)); CK(hipMemset(redo, 0, 4096));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int k = 0; k < 24; ++k) {
        if (k == 4) CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((k_sweep_q<NW, R, false, true, true, false>), grid, dim3(NW * 64), 0, 0, g, (k & 1) ? psi1 : psi0, (k & 1) ? psi0 : psi1,
                           base, f, (float*)nullptr, (float*)nullptr, (const MemberState*)nullptr, 1.0f, T, 0, redo, (unsigned long long*)nullptr, ntx, nty);
    }
    CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize());
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    const double plain_us = ms * 1e3 / 20.0;
    hipLaunchKernelGGL((k_sweep_q<NW, R, false, true, true, true>), grid, dim3(NW * 64), 0, 0, g, psi0, psi1, base, f, (float*)nullptr, (float*)nullptr,
                       (const MemberState*)nullptr, 1.0f, T, 0, redo, stamps, ntx, nty);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> st(nwg * 32);
    CK(hipMemcpy(st.data(), stamps, nwg * 32 * 8, hipMemcpyDeviceToHost));
    // phases: 0 start, 1 loads issued, 2 weights built, 3 steps done, 4 validated, 5 stores issued, 6 end; [7] = HW_ID | XCC_ID << 32
    const char* names[6] = {"start -> loads issued", "-> weights built (memory latency)", "-> T steps done", "-> validated (vote)", "-> stores issued", "-> end"};
    std::vector<std::vector<double>> d(6);
    struct Cu { unsigned long long t0 = ~0ull, t1 = 0; double in_loop = 0, resident = 0; };
    std::map<unsigned long long, Cu> per_cu;          // (the counters of different XCDs are not aligned: spans are taken per CU)
    for (size_t wgi = 0; wgi < nwg; ++wgi) {
        const unsigned long long* s = &st[wgi * 32];
        if (!s[0] || !s[6] || s[6] < s[0]) continue;
        for (int k = 0; k < 6; ++k) d[k].push_back((double)(s[k + 1] - s[k]));
        Cu& c = per_cu[s[7] & 0xffffffff0000ff00ull];
        c.t0 = std::min(c.t0, s[0]); c.t1 = std::max(c.t1, s[6]);
        c.in_loop += (double)(s[3] - s[2]); c.resident += (double)(s[6] - s[0]);
    }
    std::vector<double> spans;
    for (auto& kv : per_cu) spans.push_back((double)(kv.second.t1 - kv.second.t0));
    std::sort(spans.begin(), spans.end());
    const double span = spans[spans.size() / 2];
    printf("T=%d members=%d: %zu workgroups; plain launch %.1f us; stamped launch: median CU busy span %.0f ticks (s_memtime counts shader cycles: %.1f us at 2.4 GHz)\n",
           T, M, nwg, plain_us, span, span / 2400.0);
    double total = 0;
    for (int k = 0; k < 6; ++k) {
        std::sort(d[k].begin(), d[k].end());
        const double med = d[k][d[k].size() / 2];
        total += med;
        printf("  %-36s median %8.0f ticks   (p10 %.0f, p90 %.0f)\n", names[k], med, d[k][d[k].size() / 10], d[k][d[k].size() * 9 / 10]);
    }
    printf("  workgroup lifetime (sum of medians) %.0f ticks; step loop = %.0f %% of it\n", total, 100.0 * d[2][d[2].size() / 2] / total);
    // average number of workgroups of a CU inside the step loop, over the launch
    double in_loop = 0, resident = 0, tot = 0;
    for (auto& kv : per_cu) { in_loop += kv.second.in_loop; resident += kv.second.resident; tot += (double)(kv.second.t1 - kv.second.t0); }
    printf("  %zu CUs seen; per CU on average %.2f workgroups resident and %.2f inside the step loop\n", per_cu.size(), resident / tot, in_loop / tot);
    return 0;
}
