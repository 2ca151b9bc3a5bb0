// ubench_valu.hip -- issue-rate microbenchmarks for the instructions the Jacobi sweep is made of (gfx950).
// Each kernel runs ITERS x 16 independent instances of one instruction per wave; the host reports cycles
// per wave-instruction per SIMD for 1, 2 and 4 waves per SIMD (grid = 256 CUs x that many workgroups).
//   hipcc --offload-arch=gfx950 -O2 tools/ubench_valu.hip -o /tmp/ubench_valu && /tmp/ubench_valu
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float float2v __attribute__((ext_vector_type(2)));

constexpr int ITERS = 4096;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

enum Kind { K_FMA, K_MUL, K_ADD, K_PKMUL, K_PKADD, K_PKFMA, K_RCP, K_DIVSCALE, K_DIVFMAS, K_DIVFIXUP, K_DPP, K_CNDMASK,
            K_CMP, K_DSREAD32, K_DSREAD64, K_DSWRITE32, K_MIX_DS_VALU, K_DIVFULL, K_FASTDIV, K_N };
static const char* kind_name[K_N] = {"v_fma_f32", "v_mul_f32", "v_add_f32", "v_pk_mul_f32", "v_pk_add_f32", "v_pk_fma_f32",
    "v_rcp_f32", "v_div_scale_f32", "v_div_fmas_f32", "v_div_fixup_f32", "v_mov_b32_dpp(wave_shr)", "v_cndmask_b32",
    "v_cmp_gt_f32(vop3->sgpr)", "ds_read_b32", "ds_read_b64", "ds_write_b32", "8 valu + 1 ds_read_b32 (per valu)",
    "IEEE a/b (compiler sequence)", "a*y, fma, fma (3-op quotient)"};

template <int KIND>
__global__ __launch_bounds__(256) void k_bench(float* out, float seed, unsigned long long* ticks)
{
    unsigned long long tk0;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tk0) : : "memory");
    __shared__ float lds[4096];
    float a[16];
    float2v p[16];
    const float b = seed + 1.0f, c = seed + 0.5f;
    float2v pb = {b, b}, pc = {c, c};
#pragma unroll
    for (int i = 0; i < 16; ++i) { a[i] = seed + i; p[i] = float2v{seed + i, seed - i}; }
    lds[threadIdx.x] = seed;
    lds[threadIdx.x + 256] = seed;
    __syncthreads();
    const unsigned laddr = (threadIdx.x & 255) * 8;
    for (int it = 0; it < ITERS; ++it) {
        if (KIND == K_FMA) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            REP16(X)
#undef X
        } else if (KIND == K_MUL) {
#define X(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            REP16(X)
#undef X
        } else if (KIND == K_ADD) {
#define X(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            REP16(X)
#undef X
        } else if (KIND == K_PKMUL) {
#define X(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pb));
            REP16(X)
#undef X
        } else if (KIND == K_PKADD) {
#define X(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pb));
            REP16(X)
#undef X
        } else if (KIND == K_PKFMA) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pb), "v"(pc));
            REP16(X)
#undef X
        } else if (KIND == K_RCP) {
#define X(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
            REP16(X)
#undef X
        } else if (KIND == K_DIVSCALE) {
#define X(i) asm volatile("v_div_scale_f32 %0, vcc, %0, %1, %0" : "+v"(a[i]) : "v"(b) : "vcc");
            REP16(X)
#undef X
        } else if (KIND == K_DIVFMAS) {
#define X(i) asm volatile("v_div_fmas_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c) : "vcc");
            REP16(X)
#undef X
        } else if (KIND == K_DIVFIXUP) {
#define X(i) asm volatile("v_div_fixup_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            REP16(X)
#undef X
        } else if (KIND == K_DPP) {
#define X(i) asm volatile("v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n s_nop 0" : "+v"(a[i]));
            // (s_nop 0 keeps the assembler's DPP read-after-write hazard rule satisfied; it is counted in the result)
            REP16(X)
#undef X
        } else if (KIND == K_CNDMASK) {
#define X(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : "vcc");
            REP16(X)
#undef X
        } else if (KIND == K_CMP) {
            unsigned long long s;
#define X(i) asm volatile("v_cmp_gt_f32_e64 %0, |%1|, %2" : "=s"(s) : "v"(a[i]), "v"(b));
            REP16(X)
#undef X
            if (s == 0x12345) a[0] += 1.0f;
        } else if (KIND == K_DSREAD32) {
#define X(i) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(a[i]) : "v"(laddr), "n"(i * 4));
            REP16(X)
#undef X
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if (KIND == K_DSREAD64) {
#define X(i) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(p[i]) : "v"(laddr), "n"(i * 8));
            REP16(X)
#undef X
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if (KIND == K_DSWRITE32) {
#define X(i) asm volatile("ds_write_b32 %0, %1 offset:%2" : : "v"(laddr), "v"(a[i]), "n"(i * 4) : "memory");
            REP16(X)
#undef X
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if (KIND == K_MIX_DS_VALU) {
            // 16 x (8 independent v_mul + 1 ds_read_b32): does the LDS issue hide behind the VALU stream?
            float t0, t1;
#define X(i) asm volatile("ds_read_b32 %2, %4 offset:%5\n v_mul_f32 %0, %0, %3\n v_mul_f32 %1, %1, %3\n v_mul_f32 %0, %0, %3\n v_mul_f32 %1, %1, %3\n" \
                          "v_mul_f32 %0, %0, %3\n v_mul_f32 %1, %1, %3\n v_mul_f32 %0, %0, %3\n v_mul_f32 %1, %1, %3" \
                          : "+v"(a[i]), "+v"(a[(i + 1) & 15]), "=v"(t0) : "v"(b), "v"(laddr), "n"(i * 4));
            REP16(X)
#undef X
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            (void)t1;
        } else if (KIND == K_DIVFULL) {
#pragma unroll
            for (int i = 0; i < 16; ++i) a[i] = a[i] / b;
        } else if (KIND == K_FASTDIV) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float q0 = a[i] * c;
                const float r = __builtin_fmaf(-b, q0, a[i]);
                a[i] = __builtin_fmaf(r, c, q0);
            }
        }
    }
    unsigned long long tk1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tk1) : : "memory");
    if (threadIdx.x == 0) ticks[blockIdx.x] = tk1 - tk0;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i] + p[i].x + p[i].y;
    if (s == 12345.678f) out[threadIdx.x] = s + lds[threadIdx.x];
}

template <int KIND>
static void run(float* d_out, int n_cu, double ghz)
{
    static unsigned long long* d_ticks = nullptr;
    if (!d_ticks) CK(hipMalloc(&d_ticks, 4096 * sizeof(unsigned long long)));
    std::vector<unsigned long long> h_ticks(4096);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    printf("%-36s", kind_name[KIND]);
    for (int wps : {1, 2, 4}) {
        const int grid = n_cu * wps;              // 256 threads = 4 waves = one wave per SIMD per workgroup
        hipLaunchKernelGGL(k_bench<KIND>, dim3(grid), dim3(256), 0, 0, d_out, 1.0f, d_ticks);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_bench<KIND>, dim3(grid), dim3(256), 0, 0, d_out, 1.0f, d_ticks);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double cyc = ms * 1e-3 * ghz * 1e9 / ((double)ITERS * 16 * wps);
        CK(hipMemcpy(h_ticks.data(), d_ticks, grid * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        double tsum = 0;
        for (int i = 0; i < grid; ++i) tsum += (double)h_ticks[i];
        // ticks per wave-instruction per SIMD: a workgroup's lifetime / (its instructions x co-resident waves per SIMD)
        printf("  %dw/SIMD: %6.2f cyc %6.2f tick", wps, cyc, tsum / grid / ((double)ITERS * 16 * wps));
    }
    printf("\n");
    fflush(stdout);
}

int main()
{
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const double ghz = prop.clockRate * 1e-6;
    printf("%s, %d CUs, clockRate %.3f GHz (cycles below assume that clock)\n", prop.gcnArchName, prop.multiProcessorCount, ghz);
    float* d_out;
    CK(hipMalloc(&d_out, 4096));
    const int n_cu = prop.multiProcessorCount;
    run<K_FMA>(d_out, n_cu, ghz);
    run<K_MUL>(d_out, n_cu, ghz);
    run<K_ADD>(d_out, n_cu, ghz);
    run<K_PKMUL>(d_out, n_cu, ghz);
    run<K_PKADD>(d_out, n_cu, ghz);
    run<K_PKFMA>(d_out, n_cu, ghz);
    run<K_RCP>(d_out, n_cu, ghz);
    run<K_DIVSCALE>(d_out, n_cu, ghz);
    run<K_DIVFMAS>(d_out, n_cu, ghz);
    run<K_DIVFIXUP>(d_out, n_cu, ghz);
    run<K_DPP>(d_out, n_cu, ghz);
    run<K_CNDMASK>(d_out, n_cu, ghz);
    run<K_CMP>(d_out, n_cu, ghz);
    run<K_DSREAD32>(d_out, n_cu, ghz);
    run<K_DSREAD64>(d_out, n_cu, ghz);
    run<K_DSWRITE32>(d_out, n_cu, ghz);
    run<K_MIX_DS_VALU>(d_out, n_cu, ghz);
    run<K_DIVFULL>(d_out, n_cu, ghz);
    run<K_FASTDIV>(d_out, n_cu, ghz);
    return 0;
}
