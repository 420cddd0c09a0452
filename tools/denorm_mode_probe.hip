// Does v_cvt_f32_f64 flush a subnormal RESULT when MODE.FP_DENORM selects flush for single precision?
// And do v_mul_f32 / v_max_f32 / v_cvt_f64_f32 flush subnormal INPUTS in that mode?
//   hipcc -O2 --offload-arch=gfx950 tools/denorm_mode_probe.hip -o /tmp/dprobe && /tmp/dprobe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>

__global__ void k(const double *in, unsigned *out, int flush)
{
    // hwreg(HW_REG_MODE = 1, offset 4, width 2): single-precision denormal control; 0 = flush in and out, 3 = keep
    if (flush) __builtin_amdgcn_s_setreg(1 | (4 << 6) | (1 << 11), 0);
    else       __builtin_amdgcn_s_setreg(1 | (4 << 6) | (1 << 11), 3);
    const int i = threadIdx.x;
    float f;
    asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f) : "v"(in[i]));
    out[4 * i + 0] = __float_as_uint(f);
    float tiny = __uint_as_float(0x00012345u | (i & 1 ? 0x80000000u : 0));
    float m, x; double d;
    asm volatile("v_mul_f32 %0, 1.0, %1" : "=v"(m) : "v"(tiny));
    asm volatile("v_max_f32 %0, %1, %1" : "=v"(x) : "v"(tiny));
    asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d) : "v"(tiny));
    out[4 * i + 1] = __float_as_uint(m);
    out[4 * i + 2] = __float_as_uint(x);
    out[4 * i + 3] = (unsigned)(__double_as_longlong(d) >> 32);
}

int main()
{
    double h[4] = {1e-40, -3e-39, 1e-37, -1e-45};
    double *d; unsigned *o; unsigned r[16];
    hipMalloc(&d, sizeof h); hipMalloc(&o, sizeof r);
    hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    for (int flush = 0; flush < 2; flush++) {
        hipLaunchKernelGGL(k, dim3(1), dim3(4), 0, 0, d, o, flush);
        hipMemcpy(r, o, sizeof r, hipMemcpyDeviceToHost);
        printf("single-precision denormals %s\n", flush ? "FLUSHED (mode 0)" : "kept (mode 3)");
        for (int i = 0; i < 4; i++)
            printf("  cvt_f32_f64(%g) = %08x   mul_f32(1.0, sub) = %08x   max_f32(sub, sub) = %08x   cvt_f64_f32(sub).hi = %08x\n",
                   h[i], r[4 * i], r[4 * i + 1], r[4 * i + 2], r[4 * i + 3]);
    }
    return 0;
}
