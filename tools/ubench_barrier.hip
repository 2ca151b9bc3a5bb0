// ubench_barrier.hip -- what one workgroup barrier per N VALU instructions costs on gfx950 (ticks of s_memtime).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
constexpr int ITERS = 2000;
template <int NV, int MODE>   // MODE 0: no barrier, 1: s_barrier, 2: LDS write + waitcnt + s_barrier + LDS read (the sweep's exchange)
__global__ void k(float* out, unsigned long long* ticks, float seed)
{
    __shared__ float lds[2][2048];
    float a[8];
    for (int i = 0; i < 8; ++i) a[i] = seed + i;
    const float b = seed + 1.0f;
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int k2 = 0; k2 < NV / 8; ++k2) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
        }
        if (MODE == 1) asm volatile("s_barrier" ::: "memory");
        if (MODE == 2) {
            lds[it & 1][threadIdx.x] = a[0];
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            a[1] += lds[it & 1][(threadIdx.x + 64) % blockDim.x];
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
    float s = 0; for (int i = 0; i < 8; ++i) s += a[i];
    if (s == 1234.5f) out[0] = s;
}
template <int NV, int MODE>
void run(int threads, int wgs_per_cu, float* d_out, unsigned long long* d_t)
{
    const int grid = 256 * wgs_per_cu;
    hipLaunchKernelGGL((k<NV, MODE>), dim3(grid), dim3(threads), 0, 0, d_out, d_t, 1.0f);
    hipLaunchKernelGGL((k<NV, MODE>), dim3(grid), dim3(threads), 0, 0, d_out, d_t, 1.0f);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(grid);
    CK(hipMemcpy(h.data(), d_t, grid * 8, hipMemcpyDeviceToHost));
    double s = 0; for (auto x : h) s += x;
    printf("  NV=%3d mode=%d threads=%4d wg/cu=%d: %7.1f ticks per iteration\n", NV, MODE, threads, wgs_per_cu, s / grid / ITERS);
}
int main()
{
    float* d_out; unsigned long long* d_t;
    CK(hipMalloc(&d_out, 64)); CK(hipMalloc(&d_t, 8 * 4096));
    for (int wg : {1, 2}) {
        run<184, 0>(512, wg, d_out, d_t); run<184, 1>(512, wg, d_out, d_t); run<184, 2>(512, wg, d_out, d_t);
        run<96, 0>(512, wg, d_out, d_t); run<96, 1>(512, wg, d_out, d_t); run<96, 2>(512, wg, d_out, d_t);
        run<8, 0>(512, wg, d_out, d_t); run<8, 1>(512, wg, d_out, d_t); run<8, 2>(512, wg, d_out, d_t);
    }
    run<184, 1>(256, 1, d_out, d_t); run<184, 1>(256, 4, d_out, d_t); run<184, 1>(1024, 1, d_out, d_t); run<184, 0>(1024, 1, d_out, d_t);
    return 0;
}
