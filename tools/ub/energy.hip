// micro-benchmark: what does an operation cost in ENERGY on this board?  Every kernel below keeps all 256 compute units busy with
// one kind of work for ~2 s while a host thread samples the device's own hwmon node (socket power, shader clock); the rate of the
// operation comes from the kernel's duration.  (power - power of the idle-spinning kernel) / rate = joules per operation.
//   hipcc --offload-arch=gfx950 -O2 -pthread tools/ub/energy.hip -o tools/scratch/energy ;  tools/scratch/energy
// Why: the overlapped frame of the upscaler runs into the 1400 W power limit -- its rate is the limit divided
// by the energy of a frame, so instruction counts have to be priced in joules.
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dirent.h>
#include <string>
#include <thread>
#include <unistd.h>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
#define REP8(x) x x x x x x x x

enum { K_SLEEP, K_VALU, K_PK, K_MIN3, K_RCP, K_PKF16, K_LDSW, K_LDSR, K_SWAP, K_DPP };

template <int KIND> __global__ void __launch_bounds__(512) k_alu(float* sink, long iters)
{
    __shared__ __attribute__((aligned(16))) float lds[16384];
    f2 a0 = {1.0f + threadIdx.x, 2.f}, a1 = {1.5f, 2.5f}, a2 = {0.5f, 0.25f}, a3 = {3.f, 4.f}, a4 = {5.f, 6.f}, a5 = {7.f, 8.f}, a6 = {9.f, 1.f}, a7 = {2.f, 3.f};
    const f2 m = {1.0000001f, 0.9999999f}, c = {1e-9f, -1e-9f};
    f4 q0 = {1, 2, 3, 4}, q1 = {5, 6, 7, 8};
    const unsigned la = threadIdx.x * 16u;
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = (float)i;
    __syncthreads();
    for (long it = 0; it < iters; it++) {
        if constexpr (KIND == K_SLEEP) { REP8(asm volatile("s_sleep 8");) }
        if constexpr (KIND == K_VALU) { REP8(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9" : "+v"(a0.x), "+v"(a1.x), "+v"(a2.x), "+v"(a3.x), "+v"(a4.x), "+v"(a5.x), "+v"(a6.x), "+v"(a7.x) : "v"(m.x), "v"(c.x));) }
        if constexpr (KIND == K_PK) { REP8(asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));) }
        if constexpr (KIND == K_MIN3) { REP8(asm volatile("v_min3_f32 %0, %0, %8, %9\n v_min3_f32 %1, %1, %8, %9\n v_min3_f32 %2, %2, %8, %9\n v_min3_f32 %3, %3, %8, %9\n v_min3_f32 %4, %4, %8, %9\n v_min3_f32 %5, %5, %8, %9\n v_min3_f32 %6, %6, %8, %9\n v_min3_f32 %7, %7, %8, %9" : "+v"(a0.x), "+v"(a1.x), "+v"(a2.x), "+v"(a3.x), "+v"(a4.x), "+v"(a5.x), "+v"(a6.x), "+v"(a7.x) : "v"(m.x), "v"(c.x));) }
        if constexpr (KIND == K_RCP) { REP8(asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7" : "+v"(a0.x), "+v"(a1.x), "+v"(a2.x), "+v"(a3.x), "+v"(a4.x), "+v"(a5.x), "+v"(a6.x), "+v"(a7.x));) }
        if constexpr (KIND == K_PKF16) { REP8(asm volatile("v_pk_fma_f16 %0, %0, %8, %9\n v_pk_fma_f16 %1, %1, %8, %9\n v_pk_fma_f16 %2, %2, %8, %9\n v_pk_fma_f16 %3, %3, %8, %9\n v_pk_fma_f16 %4, %4, %8, %9\n v_pk_fma_f16 %5, %5, %8, %9\n v_pk_fma_f16 %6, %6, %8, %9\n v_pk_fma_f16 %7, %7, %8, %9" : "+v"(a0.x), "+v"(a1.x), "+v"(a2.x), "+v"(a3.x), "+v"(a4.x), "+v"(a5.x), "+v"(a6.x), "+v"(a7.x) : "v"(m.x), "v"(c.x));) }
        if constexpr (KIND == K_LDSW) { REP8(asm volatile("ds_write_b64 %0, %1\n ds_write_b64 %0, %2 offset:8192\n ds_write_b64 %0, %3 offset:16384\n ds_write_b64 %0, %4 offset:24576\n s_waitcnt lgkmcnt(0)" :: "v"(la / 2), "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "memory");) }
        if constexpr (KIND == K_LDSR) { REP8(asm volatile("ds_read_b128 %0, %2\n ds_read_b128 %1, %2 offset:8192\n s_waitcnt lgkmcnt(0)" : "=&v"(q0), "=&v"(q1) : "v"(la) : "memory"); asm volatile("" :: "v"(q0), "v"(q1));) }
        if constexpr (KIND == K_SWAP) { REP8(asm volatile("v_permlane32_swap_b32 %0, %1\n v_permlane16_swap_b32 %2, %3\n v_permlane32_swap_b32 %4, %5\n v_permlane16_swap_b32 %6, %7\n s_nop 1" : "+v"(a0.x), "+v"(a1.x), "+v"(a2.x), "+v"(a3.x), "+v"(a4.x), "+v"(a5.x), "+v"(a6.x), "+v"(a7.x));) }
        if constexpr (KIND == K_DPP) { REP8(asm volatile("v_mov_b32_dpp %0, %1 row_ror:8 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %3 row_ror:8 row_mask:0xf bank_mask:0xc\n v_mov_b32_dpp %4, %5 row_shr:4 row_mask:0xf bank_mask:0xa\n v_mov_b32_dpp %6, %7 row_shl:4 row_mask:0xf bank_mask:0x5\n s_nop 1" : "+v"(a0.x), "+v"(a1.x), "+v"(a2.x), "+v"(a3.x), "+v"(a4.x), "+v"(a5.x), "+v"(a6.x), "+v"(a7.x));) }
    }
    sink[blockIdx.x * blockDim.x + threadIdx.x] = a0.x + a1.x + a2.x + a3.x + a4.x + a5.x + a6.x + a7.x + a0.y + a1.y + a2.y + a3.y + a4.y + a5.y + a6.y + a7.y + q0.x + q1.y + lds[threadIdx.x];
}
// streaming reads / writes / copy of a buffer far larger than the caches (16-byte accesses, whole kilobytes per wave instruction)
template <int MODE> __global__ void __launch_bounds__(512) k_mem(const f4* __restrict__ src, f4* __restrict__ dst, long n, int passes)
{
    f4 acc = {0, 0, 0, 0};
    for (int p = 0; p < passes; p++)
        for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
            if constexpr (MODE == 0) acc += __builtin_nontemporal_load(src + i);
            if constexpr (MODE == 1) __builtin_nontemporal_store(f4{(float)i, 1.f, 2.f, (float)p}, dst + i);
            if constexpr (MODE == 2) __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
        }
    if (MODE == 0 && acc.x == 12345.678f) dst[0] = acc;
}

struct Sampler {
    std::string dir;
    std::atomic<bool> stop{false};
    std::vector<double> pw, fr;
    std::thread th;
    static double rd(const std::string& p) { FILE* f = fopen(p.c_str(), "r"); if (!f) return -1; double v = -1; if (fscanf(f, "%lf", &v) != 1) v = -1; fclose(f); return v; }
    explicit Sampler(const char* pci)
    {
        for (int c = 0; c < 128 && dir.empty(); c++) {
            char link[256], real[512];
            snprintf(link, sizeof link, "/sys/class/drm/card%d/device", c);
            const ssize_t n = readlink(link, real, sizeof real - 1);
            if (n <= 0) continue;
            real[n] = 0;
            const char* b = strrchr(real, '/');
            if (!b || strcasecmp(b + 1, pci)) continue;
            const std::string hw = std::string(link) + "/hwmon";
            if (DIR* d = opendir(hw.c_str())) { while (dirent* e = readdir(d)) if (!strncmp(e->d_name, "hwmon", 5)) dir = hw + "/" + e->d_name; closedir(d); }
        }
    }
    void start() { stop = false; pw.clear(); fr.clear(); th = std::thread([this] { while (!stop) { double p = rd(dir + "/power1_average"); if (p < 0) p = rd(dir + "/power1_input"); if (p > 0) pw.push_back(p * 1e-6); const double f = rd(dir + "/freq1_input"); if (f > 0) fr.push_back(f * 1e-6); usleep(50000); } }); }
    void finish(double& p, double& f)
    {
        stop = true; th.join();
        auto med = [](std::vector<double> v) { if (v.size() < 5) return -1.0; v.erase(v.begin(), v.begin() + v.size() / 4); std::sort(v.begin(), v.end()); return v[v.size() / 2]; };   // (first quarter: ramp-up)
        p = med(pw); f = med(fr);
    }
};
#include <algorithm>

int main(int argc, char** argv)
{
    const bool only_mem = argc > 1 && !strcmp(argv[1], "mem");      // (LDS and HBM rows only)
    char pci[64] = "";
    hipDeviceGetPCIBusId(pci, sizeof pci, 0);
    Sampler S(pci);
    printf("# device %s, hwmon %s\n", pci, S.dir.empty() ? "(not found: power columns empty)" : S.dir.c_str());
    float* sink; hipMalloc(&sink, 256 * 512 * 4);
    const long NB = 1l << 32;                         // 4 GiB per buffer
    f4 *src, *dst; hipMalloc(&src, NB); hipMalloc(&dst, NB); hipMemset(src, 1, NB); hipMemset(dst, 0, NB);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    double p_idle = 0;
    printf("# %-34s %9s %9s %9s %14s %s\n", "kernel (256 x 512 threads)", "seconds", "watts", "sclk MHz", "rate", "energy per operation (above the sleeping kernel)");
    auto alu = [&](const char* name, auto kern, long iters, double ops_per_iter_per_wave, const char* unit, double bytes_per_op) {
        hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, sink, 16l);            // warm up
        hipDeviceSynchronize();
        S.start(); hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, sink, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        double p, f; S.finish(p, f);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double waves = 256.0 * 8, ops = waves * iters * ops_per_iter_per_wave, rate = ops / (ms * 1e-3);
        if (!strcmp(name, "s_sleep (idle spinning)")) p_idle = p;
        printf("  %-34s %9.2f %9.0f %9.0f %11.3g /s", name, ms * 1e-3, p, f, rate);
        if (ops > 0 && p > 0 && p_idle > 0 && strcmp(name, "s_sleep (idle spinning)")) {
            const double e = (p - p_idle) / rate;
            printf("   %.2f nJ per wave %s = %.1f pJ per lane", e * 1e9, unit, e * 1e12 / 64);
            if (bytes_per_op > 0) printf(" = %.2f pJ per byte", e * 1e12 / bytes_per_op);
        }
        printf("\n"); fflush(stdout);
    };
    const long IT = 12000000;
    alu("s_sleep (idle spinning)", k_alu<K_SLEEP>, IT / 16, 0, "", 0);
    if (!only_mem) {
    alu("v_fma_f32", k_alu<K_VALU>, IT, 64, "instruction", 0);
    alu("v_pk_fma_f32", k_alu<K_PK>, IT, 64, "instruction", 0);
    alu("v_min3_f32", k_alu<K_MIN3>, IT, 64, "instruction", 0);
    alu("v_rcp_f32", k_alu<K_RCP>, IT / 2, 64, "instruction", 0);
    alu("v_pk_fma_f16", k_alu<K_PKF16>, IT, 64, "instruction", 0);
    alu("v_permlane32/16_swap_b32", k_alu<K_SWAP>, IT, 32, "instruction", 0);
    alu("v_mov_b32_dpp (row_ror / shr / shl)", k_alu<K_DPP>, IT, 32, "instruction", 0);
    }
    alu("ds_write_b64", k_alu<K_LDSW>, IT / 4, 32, "instruction", 512);
    alu("ds_read_b128", k_alu<K_LDSR>, IT / 4, 16, "instruction", 1024);
    auto mem = [&](const char* name, auto kern, int passes, double bytes_per_pass) {
        hipLaunchKernelGGL(kern, dim3(256 * 4), dim3(512), 0, 0, src, dst, NB / 16 / 64, 1);
        hipDeviceSynchronize();
        S.start(); hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(256 * 4), dim3(512), 0, 0, src, dst, NB / 16, passes);
        hipEventRecord(e1); hipEventSynchronize(e1);
        double p, f; S.finish(p, f);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double rate = bytes_per_pass * passes / (ms * 1e-3);
        printf("  %-34s %9.2f %9.0f %9.0f %9.2f TB/s", name, ms * 1e-3, p, f, rate * 1e-12);
        if (p > 0 && p_idle > 0) printf("   %.1f pJ per byte (above the sleeping kernel; includes the loop's own instructions)", (p - p_idle) / rate * 1e12);
        printf("\n"); fflush(stdout);
    };
    mem("HBM read, 16 B per lane", k_mem<0>, 3000, (double)NB);
    mem("HBM write, 16 B per lane", k_mem<1>, 2000, (double)NB);
    mem("HBM copy (read + write)", k_mem<2>, 1200, 2.0 * NB);
    return 0;
}
