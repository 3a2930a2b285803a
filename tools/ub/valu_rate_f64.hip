// micro-benchmark: issue cost (s_memtime ticks per instruction per wave) of the double-precision vector instructions the -p 1
// sharpen kernel is made of, with 1 / 2 / 4 waves per SIMD.  hipcc --offload-arch=gfx950 -O2 valu_rate_f64.hip -o valu_rate_f64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define REP8(x) x x x x x x x x
#define OP3(op) op " %0, %0, %8, %9\n" op " %1, %1, %8, %9\n" op " %2, %2, %8, %9\n" op " %3, %3, %8, %9\n" op " %4, %4, %8, %9\n" op " %5, %5, %8, %9\n" op " %6, %6, %8, %9\n" op " %7, %7, %8, %9"
#define OP2(op) op " %0, %0, %8\n" op " %1, %1, %8\n" op " %2, %2, %8\n" op " %3, %3, %8\n" op " %4, %4, %8\n" op " %5, %5, %8\n" op " %6, %6, %8\n" op " %7, %7, %8"
#define OP1(op) op " %0, %0\n" op " %1, %1\n" op " %2, %2\n" op " %3, %3\n" op " %4, %4\n" op " %5, %5\n" op " %6, %6\n" op " %7, %7"
#define REGS "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
template <int KIND> __global__ void k(unsigned long long* out, double* sink, int iters)
{
    double a0 = 1.0 + threadIdx.x, a1 = 1.5, a2 = 0.5, a3 = 3., a4 = 5., a5 = 7., a6 = 9., a7 = 2.;
    const double m = 1.0000001, c = 1e-9;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
        if constexpr (KIND == 0) { REP8(asm volatile(OP3("v_fma_f64") : REGS : "v"(m), "v"(c));) }
        if constexpr (KIND == 1) { REP8(asm volatile(OP2("v_add_f64") : REGS : "v"(c));) }
        if constexpr (KIND == 2) { REP8(asm volatile(OP2("v_mul_f64") : REGS : "v"(m));) }
        if constexpr (KIND == 3) { REP8(asm volatile(OP2("v_min_f64") : REGS : "v"(m));) }
        if constexpr (KIND == 4) { REP8(asm volatile(OP1("v_rcp_f64") : REGS);) }
        if constexpr (KIND == 5) { REP8(asm volatile(OP1("v_rsq_f64") : REGS);) }
        if constexpr (KIND == 6) { REP8(asm volatile(OP1("v_sqrt_f64") : REGS);) }
        if constexpr (KIND == 7) { REP8(asm volatile("v_mul_f64 %0, |%0|, |%8| clamp\n v_mul_f64 %1, |%1|, |%8| clamp\n v_mul_f64 %2, |%2|, |%8| clamp\n v_mul_f64 %3, |%3|, |%8| clamp\n v_mul_f64 %4, |%4|, |%8| clamp\n v_mul_f64 %5, |%5|, |%8| clamp\n v_mul_f64 %6, |%6|, |%8| clamp\n v_mul_f64 %7, |%7|, |%8| clamp" : REGS : "v"(m));) }
        if constexpr (KIND == 8) { REP8(asm volatile(OP1("v_mov_b64") : REGS);) }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
    sink[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
template <int KIND> void run(const char* name, unsigned long long* d, double* sink)
{
    const int iters = 64;
    for (int threads : {256, 512, 1024}) {
        hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(threads), 0, 0, d, sink, iters);
        hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(threads), 0, 0, d, sink, iters);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(256 * threads / 64);
        hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
        double s = 0;
        for (auto x : h) s += (double)x;
        printf("%-28s %d waves/SIMD: %.2f ticks per instruction per wave\n", name, threads / 256, s / h.size() / (iters * 64.0));
    }
}
int main()
{
    unsigned long long* d; double* sink;
    hipMalloc(&d, 256 * 16 * 8); hipMalloc(&sink, 256 * 1024 * 8);
    run<0>("v_fma_f64", d, sink); run<1>("v_add_f64", d, sink); run<2>("v_mul_f64", d, sink); run<7>("v_mul_f64 |a| |b| clamp", d, sink);
    run<3>("v_min_f64", d, sink); run<4>("v_rcp_f64", d, sink); run<5>("v_rsq_f64", d, sink); run<6>("v_sqrt_f64", d, sink);
    run<8>("v_mov_b64", d, sink);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 0, 0, d, sink, 4096); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h0; hipMemcpy(&h0, d, 8, hipMemcpyDeviceToHost);
    printf("s_memtime: %.1f ticks per microsecond (kernel %.3f ms, %llu ticks)\n", h0 / (ms * 1e3), ms, h0);
    return 0;
}
