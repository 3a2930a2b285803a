// micro-test: can two kernels of ONE stream overlap, the second gated on the first by a device-side counter?
//   hipcc --offload-arch=gfx950 -O2 tools/ub/anyorder.hip -o tools/scratch/anyorder ; tools/scratch/anyorder
// Why: the reference records a frame's stages in ONE command buffer on ONE queue; the three colour planes of a stage are independent
// (SURVEY 2.1), only stage s+1 of plane c needs stage s of plane c.  A HIP stream orders whole kernels (the AQL barrier bit).
// hipExtLaunchKernel(.., flags = hipExtAnyOrderLaunch) clears that bit for ONE launch: it may start while its predecessors of the
// same stream still run.  Part 1 measures whether it does on this board / runtime (timestamps from the 100 MHz wall clock);
// part 2 hands 16 KB per producer workgroup to consumers of the NEXT kernel through a per-plane arrival counter and checks every
// word (producers: plain stores -> barrier -> one lane's agent-scope release -> counter; consumers: one lane polls, one agent-scope
// acquire, barrier, plain loads -- MI355X_MICROARCH.md, inter-workgroup visibility).
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void k_spin(unsigned long long* stamps, int slot, long ticks)
{
    const unsigned long long t0 = wall_clock64();
    while ((long)(wall_clock64() - t0) < ticks) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0 && blockIdx.x == 0) { stamps[2 * slot] = t0; stamps[2 * slot + 1] = wall_clock64(); }
}

// producer: workgroup b of plane c writes its 16 KB (4096 words of value f(epoch, c, b, i)), then arrives on cnt[c]
__global__ void __launch_bounds__(256) k_produce(unsigned* data, unsigned* cnt, int per_plane, unsigned epoch)
{
    const int b = blockIdx.x, c = blockIdx.y;
    uint4* dst = (uint4*)(data + ((size_t)c * per_plane + b) * 4096);
    for (int i = threadIdx.x; i < 1024; i += 256) {
        const unsigned v = epoch * 0x9e3779b9u + (unsigned)(c * per_plane + b) * 4096u + 4u * i;
        dst[i] = make_uint4(v, v + 1, v + 2, v + 3);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(&cnt[c], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// consumer: workgroup t of plane c reads column t (16 words) of EVERY producer block of plane c -- it needs the whole plane
__global__ void __launch_bounds__(256) k_consume(const unsigned* data, unsigned* cnt, int per_plane, unsigned epoch, unsigned target,
                                                 unsigned* errors, unsigned long long* waited, int* timeout_flag)
{
    const int t = blockIdx.x, c = blockIdx.y;
    __shared__ int ok;
    if (threadIdx.x == 0) {
        const unsigned long long t0 = wall_clock64();
        int good = 1;
        while (__hip_atomic_load(&cnt[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(16);
            if (wall_clock64() - t0 > 100000000ull) { good = 0; break; }      // 1 s: report, never hang the box
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        ok = good;
        if (t == 0) waited[c] = wall_clock64() - t0;
    }
    __syncthreads();
    if (!ok) { if (threadIdx.x == 0) *timeout_flag = 1; return; }
    unsigned bad = 0;
    for (int b = threadIdx.x; b < per_plane; b += 256) {
        const uint4* src = (const uint4*)(data + ((size_t)c * per_plane + b) * 4096) + 4 * t;
        for (int i = 0; i < 4; i++) {
            const uint4 q = src[i];
            const unsigned v = epoch * 0x9e3779b9u + (unsigned)(c * per_plane + b) * 4096u + 4u * (4 * t + i);
            bad += (q.x != v) + (q.y != v + 1) + (q.z != v + 2) + (q.w != v + 3);
        }
    }
    if (bad) atomicAdd(errors, bad);
}

int main()
{
    hipStream_t st;
    CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    unsigned long long* stamps;
    CHECK(hipMalloc(&stamps, 64 * sizeof(unsigned long long)));
    CHECK(hipMemset(stamps, 0, 64 * sizeof(unsigned long long)));
    // ---- part 1: A (50 us), B any-order (20 us), C ordered (1 us)
    for (int flag = 0; flag < 2; flag++) {
        unsigned long long* sp = stamps; long ta = 5000, tb = 2000, tc = 100; int s0 = 0, s1 = 1, s2 = 2;
        void* a0[] = {&sp, &s0, &ta}; void* a1[] = {&sp, &s1, &tb}; void* a2[] = {&sp, &s2, &tc};
        CHECK(hipExtLaunchKernel((const void*)k_spin, dim3(64), dim3(64), a0, 0, st, nullptr, nullptr, 0));
        CHECK(hipExtLaunchKernel((const void*)k_spin, dim3(64), dim3(64), a1, 0, st, nullptr, nullptr, flag ? hipExtAnyOrderLaunch : 0));
        CHECK(hipExtLaunchKernel((const void*)k_spin, dim3(64), dim3(64), a2, 0, st, nullptr, nullptr, 0));
        CHECK(hipStreamSynchronize(st));
        unsigned long long h[6];
        CHECK(hipMemcpy(h, stamps, sizeof(h), hipMemcpyDeviceToHost));
        printf("flags=%d: A [0, %.1f] us, B [%.1f, %.1f] us, C [%.1f, %.1f] us -> B %s A; C %s\n", flag, (h[1] - h[0]) * 0.01, (long)(h[2] - h[0]) * 0.01,
               (long)(h[3] - h[0]) * 0.01, (long)(h[4] - h[0]) * 0.01, (long)(h[5] - h[0]) * 0.01, h[2] < h[1] ? "OVERLAPS" : "follows",
               (h[4] >= h[1] && h[4] >= h[3]) ? "waits for both" : "DOES NOT WAIT");
    }
    // ---- part 2: producer (512 x 3 blocks of 16 KB) -> consumer (256 x 3 blocks), same stream, consumer any-order, counters monotonic
    const int per_plane = 512;
    unsigned *data, *cnt, *errors; unsigned long long* waited; int* tflag;
    CHECK(hipMalloc(&data, (size_t)3 * per_plane * 4096 * 4));
    CHECK(hipMalloc(&cnt, 64)); CHECK(hipMemset(cnt, 0, 64));
    CHECK(hipMalloc(&errors, 4)); CHECK(hipMemset(errors, 0, 4));
    CHECK(hipMalloc(&waited, 64)); CHECK(hipMalloc(&tflag, 4)); CHECK(hipMemset(tflag, 0, 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int mode = 0; mode < 2; mode++) {        // 0: ordered launches (the counter is already complete), 1: consumer any-order
        const int iters = 200;
        CHECK(hipMemset(cnt, 0, 64));
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0, st));
        for (int it = 0; it < iters; it++) {
            unsigned epoch = (unsigned)(mode * 1000 + it + 1), target = (unsigned)(it + 1) * per_plane;
            int pp = per_plane;
            void* pa[] = {&data, &cnt, &pp, &epoch};
            void* ca[] = {&data, &cnt, &pp, &epoch, &target, &errors, &waited, &tflag};
            CHECK(hipExtLaunchKernel((const void*)k_produce, dim3(per_plane, 3), dim3(256), pa, 0, st, nullptr, nullptr, 0));
            CHECK(hipExtLaunchKernel((const void*)k_consume, dim3(256, 3), dim3(256), ca, 0, st, nullptr, nullptr, mode ? hipExtAnyOrderLaunch : 0));
        }
        CHECK(hipEventRecord(e1, st));
        CHECK(hipEventSynchronize(e1));
        float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
        unsigned herr; int hflag; unsigned long long hw[3];
        CHECK(hipMemcpy(&herr, errors, 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(&hflag, tflag, 4, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(hw, waited, sizeof(hw), hipMemcpyDeviceToHost));
        printf("hand-off %s: %.2f us per producer+consumer pair, wrong words %u, timeouts %d, last waits %.1f %.1f %.1f us\n",
               mode ? "any-order consumer" : "ordered launches", ms * 1e3 / iters, herr, hflag, hw[0] * 0.01, hw[1] * 0.01, hw[2] * 0.01);
    }
    return 0;
}
