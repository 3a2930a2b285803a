// micro-benchmark: do LDS traffic and vector arithmetic overlap on one SIMD of gfx950?
//   mode 0: every wave runs V vector instructions            mode 1: every wave runs L LDS writes (+ the reads back)
//   mode 2: even waves LDS, odd waves VALU (two waves per SIMD: one of each kind)   mode 3: every wave does both, LDS first then VALU
//   mode 4: every wave does both, interleaved (1 LDS op, 8 VALU)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
#define VALU8(a) asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(m), "v"(c));
template <int MODE, bool READ> __global__ void __launch_bounds__(512) k(unsigned long long* out, float* sink, int iters)
{
    extern __shared__ __attribute__((aligned(128))) char smem[];
    f2 a[8];
    for (int i = 0; i < 8; i++) a[i] = f2{1.0f + threadIdx.x + i, 2.f};
    const f2 m = {1.0000001f, 0.9999999f}, c = {1e-9f, -1e-9f};
    const int wave = threadIdx.x >> 6;
    f2* buf = (f2*)smem + wave * 4096;                     // a private 32 KB / 8 region per wave: 4096 float2 = 32 KB ... (8 waves x 8 KB)
    buf = (f2*)smem + wave * 1024;
    const int lane = threadIdx.x & 63;
    const bool do_lds = MODE == 1 || MODE == 3 || MODE == 4 || (MODE == 2 && (wave < 4));          // waves 0-3: first slot of each SIMD
    const bool do_valu = MODE == 0 || MODE == 3 || MODE == 4 || (MODE == 2 && (wave >= 4));
    f2 w[8];
    for (int i = 0; i < 8; i++) w[i] = f2{(float)lane, (float)i};
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
        if constexpr (MODE == 4) {
#pragma unroll
            for (int q = 0; q < 8; q++) {
                buf[lane + 64 * q] = w[q];
                __builtin_amdgcn_sched_barrier(0);
                VALU8(a)
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (READ) {
#pragma unroll
                for (int q = 0; q < 8; q++) w[q] = buf[lane + 64 * q];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
        } else {
            if (do_lds) {
#pragma unroll
                for (int q = 0; q < 8; q++) buf[lane + 64 * q] = w[q];
                if constexpr (READ) {
#pragma unroll
                    for (int q = 0; q < 8; q++) w[q] = buf[lane + 64 * q];
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            if (do_valu) { VALU8(a) VALU8(a) VALU8(a) VALU8(a) VALU8(a) VALU8(a) VALU8(a) VALU8(a) }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
    float s = 0;
    for (int i = 0; i < 8; i++) s += a[i].x + a[i].y + w[i].x + w[i].y;
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE, bool READ> void run(const char* name, unsigned long long* d, float* sink)
{
    const int iters = 200;
    for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL((k<MODE, READ>), dim3(256), dim3(512), 65536, 0, d, sink, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(256 * 8);
    hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    double lo = 0, hi = 0;
    for (int b = 0; b < 256; b++) for (int w = 0; w < 8; w++) (w < 4 ? lo : hi) += (double)h[b * 8 + w];
    printf("%-64s waves 0-3: %7.0f  waves 4-7: %7.0f cycles per iteration\n", name, lo / (256 * 4) / iters, hi / (256 * 4) / iters);
}
int main()
{
    unsigned long long* d; float* sink;
    hipMalloc(&d, 256 * 8 * 8); hipMalloc(&sink, 256 * 512 * 4);
    hipFuncSetAttribute((const void*)k<0, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    printf("per iteration and wave: 64 v_pk_fma_f32 and/or 8 ds_write_b64 (+ 8 ds_read_b64); 512 threads = 2 waves per SIMD, 256 workgroups\n");
    run<0, false>("VALU only (all waves)", d, sink);
    run<1, false>("LDS writes only (all waves)", d, sink);
    run<1, true>("LDS writes + reads only (all waves)", d, sink);
    run<2, false>("waves 0-3 LDS writes, waves 4-7 VALU", d, sink);
    run<2, true>("waves 0-3 LDS writes + reads, waves 4-7 VALU", d, sink);
    run<3, false>("every wave: LDS writes, then VALU", d, sink);
    run<3, true>("every wave: LDS writes + reads, then VALU", d, sink);
    run<4, false>("every wave: interleaved (1 write, 8 VALU) x 8", d, sink);
    run<4, true>("every wave: interleaved (1 write, 8 VALU) x 8, then reads", d, sink);
    return 0;
}
