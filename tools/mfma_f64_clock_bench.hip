// What v_mfma_f64_16x16x4_f64 sustains on gfx950 when the operands are DATA: four accumulators per wave, operands
// cycling through eight random doubles per lane (registers, no memory traffic in the loop), for ~1 ms per launch, and
// the clock the chip holds meanwhile (s_memtime / s_memrealtime x 100 MHz: MI355X_MICROARCH.md, 'DVFS give-back').
// The round-1 bench (tools/mfma_f64_bench.hip) multiplied two near-constants and read 78 TFLOP/s = 64.5 cycles per MFMA
// at 2.4 GHz; the FIR kernels multiply audio samples by taps, and the chip lowers its clock under that load.
//   hipcc -O3 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form=1 tools/mfma_f64_clock_bench.hip -o /tmp/mfma_clock && /tmp/mfma_clock
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double v4f64 __attribute__((ext_vector_type(4)));

/* LDSR: ds_read_b64 per 4 MFMAs -- the operands then come from LDS as in the FIR kernels (fir_tile at four row tiles reads 2 per 4, at one row tile 8 per 4) */
template <int LDSR>
__global__ __launch_bounds__(256) void k(const double *in, double *out, unsigned long long *clk, int iters)
{
    __shared__ double sm[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) sm[i] = in[i];
    __syncthreads();
    v4f64 acc[4];
#pragma unroll
    for (int i = 0; i < 4; i++) acc[i] = {0, 0, 0, 0};
    double a[8], b[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        a[i] = in[(threadIdx.x * 16 + i) & 4095];
        b[i] = in[(threadIdx.x * 16 + 8 + i + blockIdx.x) & 4095];
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    const double *sp = sm + (threadIdx.x & 63);
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            /* LDSR = ds_read_b64 per 4 MFMAs, in pairs (a taps operand and a window operand); 1 = a pair every 8 MFMAs */
            if constexpr (LDSR > 0) {
                if (LDSR >= 2 || (u & 1) == 0) {
#pragma unroll
                    for (int r = 0; r < (LDSR + 1) / 2; r++) {
                        a[(u + r) & 7] = sp[((it * 8 + u) * 67 + r * 1031) & 4031];
                        b[(u + r) & 7] = sp[((it * 8 + u) * 131 + 64 + r * 517) & 4031];
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < 4; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[(u + i) & 7], b[u], acc[i], 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    double s = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) { clk[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = t1 - t0; clk[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = r1 - r0; }
}

int main()
{
    const int blocks = 256, iters = 1000;
    double *in, *out; unsigned long long *clk;
    (void)hipMalloc(&in, 4096 * 8); (void)hipMalloc(&out, blocks * 256 * 8); (void)hipMalloc(&clk, blocks * 4 * 16);
    for (int mode = 0; mode < 7; mode++) {
        std::vector<double> h(4096);
        unsigned long long x = 88172645463325252ull;
        for (auto &v : h) {
            x ^= x << 13; x ^= x >> 7; x ^= x << 17;
            const double r = (double)(x >> 11) / 9007199254740992.0 * 2.0 - 1.0;             /* uniform [-1, 1) */
            v = mode == 0 ? 1.0 : mode == 2 ? r : (double)(float)r;                          /* constants | floats widened (the FIR's operands) | full doubles */
        }
        hipMemcpy(in, h.data(), 4096 * 8, hipMemcpyHostToDevice);
        auto kern = mode == 3 ? k<1> : mode == 4 ? k<2> : mode == 5 ? k<4> : mode == 6 ? k<8> : k<0>;
        for (int w = 0; w < 200; w++) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, in, out, clk, iters);   /* ~0.2 s of load first */
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        for (int w = 0; w < 20; w++) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, in, out, clk, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> c(blocks * 4 * 2);
        hipMemcpy(c.data(), clk, c.size() * 8, hipMemcpyDeviceToHost);
        std::vector<double> ghz, cyc;
        for (int i = 0; i < blocks * 4; i++) { ghz.push_back((double)c[2 * i] / (double)c[2 * i + 1] * 0.1); cyc.push_back((double)c[2 * i] / (iters * 32.0)); }
        std::sort(ghz.begin(), ghz.end()); std::sort(cyc.begin(), cyc.end());
        const double flops = 20.0 * blocks * 4 * iters * 32 * 2048.0;
        printf("%-30s: %6.2f TFLOP/s wall (launch gaps included), %.2f cycles per MFMA in the loop, in-kernel clock %.3f GHz (p10 %.3f, p90 %.3f) -> %.2f TFLOP/s at that clock\n",
               mode == 0 ? "operands all 1.0" : mode == 1 ? "random floats, widened" : mode == 2 ? "random doubles" : mode == 3 ? "floats, 1 ds_read_b64 / 4 MFMA" :
               mode == 4 ? "floats, 2 ds_read_b64 / 4 MFMA" : mode == 5 ? "floats, 4 ds_read_b64 / 4 MFMA" : "floats, 8 ds_read_b64 / 4 MFMA", flops / (ms * 1e-3) / 1e12, cyc[cyc.size() / 2],
               ghz[ghz.size() / 2], ghz[ghz.size() / 10], ghz[ghz.size() * 9 / 10], 1024 * 2048.0 / cyc[cyc.size() / 2] * ghz[ghz.size() / 2] / 1e3);
    }
    return 0;
}
