// What does v_cvt_pk_u8_f32 do with fractions, negatives and values beyond 255?  (hipcc --offload-arch=gfx950 -O2 -o /tmp/cvt tools/ub/cvt_pk_u8.hip && /tmp/cvt)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float* in, unsigned* out, int n)
{
    int i = threadIdx.x;
    if (i >= n) return;
    unsigned r = 0xAABBCCDDu, sel = 1u;
    asm volatile("v_cvt_pk_u8_f32 %0, %1, %2, %0" : "+v"(r) : "v"(in[i]), "v"(sel));
    out[i] = r;
    // the same with the fp32 rounding mode set to toward-zero (MODE bits 1:0 = 3), as cvt4_f_u8 does around its multiplies
    unsigned q = 0xAABBCCDDu;
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 3\n\ts_nop 1\n\tv_cvt_pk_u8_f32 %0, %1, %2, %0\n\ts_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0"
                 : "+v"(q) : "v"(in[i]), "v"(sel));
    out[64 + i] = q;
}
int main()
{
    const float h[] = {0.0f, 0.49f, 0.5f, 0.51f, 0.999f, 1.0f, 1.5f, 2.5f, 3.5f, 254.5f, 254.999f, 255.0f, 255.5f, 256.0f, 300.0f, 1e9f, -0.4f, -0.6f, -1.0f, -1e9f, __builtin_nanf(""), 127.5f, 128.5f};
    const int n = sizeof h / sizeof h[0];
    float* d; unsigned* o; unsigned ho[128];
    hipMalloc(&d, sizeof h); hipMalloc(&o, 4 * 128);
    hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, n);
    hipMemcpy(ho, o, 4 * 128, hipMemcpyDeviceToHost);
    for (int i = 0; i < n; i++) printf("%14.6g -> byte %3u   with round-toward-zero mode: %3u\n", h[i], (ho[i] >> 8) & 0xff, (ho[64 + i] >> 8) & 0xff);
    return 0;
}
