// Can the hardware do the reference's truncating float product (dspMulFloatFloat, runtime/dsp_ieee754.h:335-375)?
//
// The reference assembles a * b from bit fields: +0.0 when either biased exponent is 0 or ea + eb - 127 < 1 (looked at BEFORE the
// mantissa product's carry), else sign | (ea + eb - 127 [+ 1 on carry]) << 23 | the top 24 bits of the 48-bit mantissa product,
// TRUNCATED; exponent 255 is read like any other and nothing stops the exponent field from overflowing.
// For operands with 1 <= ea, eb <= 254 and 128 <= ea + eb <= 380 that is IEEE round-toward-zero of the exact product.  This probe
//   1. runs v_mul_f32 under MODE.FP_ROUND (single) = toward zero, single-precision denormals flushed (the chain kernels' mode),
//      over pairs drawn to sit on every boundary, and compares with the reference's formula restated on the host;
//   2. the same through v_mul_f64 of the widened operands + v_cvt_f32_f64, with the DOUBLE round field toward zero and the single
//      one left at nearest -- does the narrowing conversion obey the double field?  (then adds need no mode switch at all);
//   3. checks that a v_add_f32 behind an s_setreg back to nearest rounds to nearest (the mode switch is taken at once);
//   4. times a loop of 16 products + 16 dependent adds with two s_setreg per 16 against the same loop without them.
//   hipcc -O2 --offload-arch=gfx950 tools/rtz_mul_probe.hip -o /tmp/rtzprobe && /tmp/rtzprobe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// HW_REG_MODE = 1.  FP_ROUND: bits [1:0] single, [3:2] double / half; 0 nearest even, 1 +inf, 2 -inf, 3 toward zero.
// FP_DENORM: bits [5:4] single, [7:6] double / half; 0 = flush in and out.
__global__ void k_mul(const unsigned *a, const unsigned *b, unsigned *o_sp, unsigned *o_dp, unsigned *o_add, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    __builtin_amdgcn_s_setreg(1 | (4 << 6) | (1 << 11), 0);          // single-precision denormals: flush
    float x = __uint_as_float(a[i]), y = __uint_as_float(b[i]);
    float p, q, s;
    double xd, yd, pd;
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 3\n\t"     // single: toward zero
                 "v_mul_f32 %0, %1, %2\n\t"
                 "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0\n\t"     // single: nearest
                 "v_add_f32 %3, %0, %4\n\t"
                 : "=&v"(p), "+v"(x), "+v"(y), "=&v"(s) : "v"(1.0f));
    o_sp[i] = __float_as_uint(p);
    o_add[i] = __float_as_uint(s);
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 3\n\t"     // double: toward zero, single stays at nearest
                 "v_cvt_f64_f32 %0, %3\n\t"
                 "v_cvt_f64_f32 %1, %4\n\t"
                 "s_nop 1\n\t"
                 "v_mul_f64 %2, %0, %1\n\t"
                 "s_nop 1\n\t"
                 "v_cvt_f32_f64 %5, %2\n\t"
                 "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 0\n\t"
                 : "=&v"(xd), "=&v"(yd), "=&v"(pd), "+v"(x), "+v"(y), "=&v"(q));
    o_dp[i] = __float_as_uint(q);
}

// 16 taps per group: products toward zero, then the dependent adds at nearest; `switches` = 0 leaves the mode alone (timing only)
template <int SWITCHES>
__global__ void k_loop(const float *x, const float *h, float *out, int groups)
{
    __builtin_amdgcn_s_setreg(1 | (4 << 6) | (1 << 11), 0);
    const int lane = blockIdx.x * blockDim.x + threadIdx.x;
    float acc = 0.0f;
    const float *xp = x + lane;
    for (int g = 0; g < groups; g++) {
        float xv[16], p[16];
#pragma unroll
        for (int j = 0; j < 16; j++) xv[j] = xp[16 * g + j];
        const float *hg = h + 16 * g;                    // wave-uniform: scalar loads
        if (SWITCHES) asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 3");
#pragma unroll
        for (int j = 0; j < 16; j++) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(p[j]) : "s"(hg[j]), "v"(xv[j]));
        if (SWITCHES) asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0");
#pragma unroll
        for (int j = 0; j < 16; j++) asm volatile("v_add_f32 %0, %0, %1" : "+v"(acc) : "v"(p[j]));
    }
    out[lane] = acc;
}

// what a pair of mode switches costs a wave: registers only, NPER products + NPER dependent sums per pair of s_setreg
template <int SWITCHES, int NPER>
__global__ void k_regs(float *out, int iters, float seed)
{
    float acc0 = seed, acc1 = seed * 0.5f, x = seed + threadIdx.x * 1e-3f, h = 0.99f;
    for (int it = 0; it < iters; it++) {
        float p[NPER];
        if (SWITCHES) asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 3");
#pragma unroll
        for (int j = 0; j < NPER; j++) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(p[j]) : "v"(h), "v"(x));
        if (SWITCHES) asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0");
#pragma unroll
        for (int j = 0; j < NPER; j += 2) {
            asm volatile("v_add_f32 %0, %0, %1" : "+v"(acc0) : "v"(p[j]));
            asm volatile("v_add_f32 %0, %0, %1" : "+v"(acc1) : "v"(p[j + 1]));
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc0 + acc1;
}

static unsigned ref_mul(unsigned A, unsigned B)        // dsp_ieee754.h:335-375, restated
{
    const int ea = (A >> 23) & 255, eb = (B >> 23) & 255;
    if (ea == 0 || eb == 0) return 0;
    int exp = ea + eb - 127;
    if (exp < 1) return 0;
    if ((A ^ B) & 0x80000000u) exp |= 1 << 8;
    const unsigned ma = ((A & 0x7FFFFF) | (1u << 23)) << 5, mb = ((B & 0x7FFFFF) | (1u << 23)) << 5;
    unsigned reshi = (unsigned)(((unsigned long long)ma * mb) >> 32);
    if (reshi & (1u << 25)) { exp++; reshi >>= 2; } else reshi >>= 1;
    reshi &= (1u << 23) - 1;
    return reshi | ((unsigned)exp << 23);
}

int main()
{
    std::mt19937 rng(20260104);
    std::vector<unsigned> a, b;
    auto mant = [&]() -> unsigned {
        switch (rng() % 6) {
        case 0: return 0;
        case 1: return 0x7FFFFF;
        case 2: return rng() & 0x7FFFFF;
        case 3: return (rng() & 0xFFF);                 // small mantissas: products without a carry, exact
        case 4: return 0x7FFFFF & ~(rng() & 0xFFF);
        default: return 0x3504F3 + (int)(rng() % 9) - 4;   // around sqrt(2): the product sits on the carry boundary
        }
    };
    auto add = [&](unsigned ea, unsigned eb) {
        a.push_back((rng() & 1u) << 31 | ea << 23 | mant());
        b.push_back((rng() & 1u) << 31 | eb << 23 | mant());
    };
    // (1) the bulk: exponents of audio (60 .. 140 each), (2) the underflow edge ea + eb - 127 in -2 .. 3, (3) the overflow edge
    // ea + eb - 127 in 252 .. 258, (4) zero / subnormal / exponent-255 operands
    for (int i = 0; i < 400000; i++) add(60 + rng() % 81, 60 + rng() % 81);
    for (int i = 0; i < 200000; i++) { const int ea = 1 + rng() % 130, s = 125 + rng() % 6; const int eb = s - ea; if (eb >= 1 && eb <= 254) add(ea, eb); }
    for (int i = 0; i < 100000; i++) { const int ea = 126 + rng() % 129, s = 379 + rng() % 7; const int eb = s - ea; if (eb >= 1 && eb <= 254) add(ea, eb); }
    for (int i = 0; i < 50000; i++) add(rng() % 3 == 0 ? 0 : rng() % 256, rng() % 3 == 0 ? 255 : rng() % 3 == 1 ? 0 : rng() % 256);
    const int n = (int)a.size();
    unsigned *da, *db, *dsp, *ddp, *dadd;
    CHECK(hipMalloc(&da, n * 4)); CHECK(hipMalloc(&db, n * 4)); CHECK(hipMalloc(&dsp, n * 4)); CHECK(hipMalloc(&ddp, n * 4)); CHECK(hipMalloc(&dadd, n * 4));
    CHECK(hipMemcpy(da, a.data(), n * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(db, b.data(), n * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_mul, dim3((n + 255) / 256), dim3(256), 0, 0, da, db, dsp, ddp, dadd, n);
    CHECK(hipDeviceSynchronize());
    std::vector<unsigned> sp(n), dp(n), ad(n);
    CHECK(hipMemcpy(sp.data(), dsp, n * 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(dp.data(), ddp, n * 4, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(ad.data(), dadd, n * 4, hipMemcpyDeviceToHost));
    // categories by the operands' exponents
    const char *names[6] = {"in range (1 <= ea, eb <= 254, 128 <= ea + eb <= 380)", "ea + eb == 127 (reference: +0 whatever the carry)", "ea + eb < 127 (reference: +0)",
                            "ea + eb >= 381 (exponent field overflows)", "an exponent of 0 (reference: +0)", "an exponent of 255, the other 1 .. 254"};
    long cnt[6] = {0}, bad_sp[6] = {0}, bad_dp[6] = {0}, bad_sp_mag[6] = {0};
    int shown[6] = {0};
    for (int i = 0; i < n; i++) {
        const int ea = (a[i] >> 23) & 255, eb = (b[i] >> 23) & 255;
        const int c = (ea == 0 || eb == 0) ? 4 : (ea == 255 || eb == 255) ? 5 : ea + eb < 127 ? 2 : ea + eb == 127 ? 1 : ea + eb >= 381 ? 3 : 0;
        const unsigned want = ref_mul(a[i], b[i]);
        cnt[c]++;
        if (sp[i] != want) {
            bad_sp[c]++;
            if ((sp[i] & 0x7FFFFFFF) != (want & 0x7FFFFFFF)) bad_sp_mag[c]++;
            if (shown[c] < 3) { shown[c]++; printf("    e.g. [%s] %08x * %08x: reference %08x, v_mul_f32 RTZ %08x, f64 path %08x\n", names[c], a[i], b[i], want, sp[i], dp[i]); }
        }
        if (dp[i] != want) bad_dp[c]++;
    }
    printf("v_mul_f32 under FP_ROUND(single) = toward zero, single denormals flushed  vs  dspMulFloatFloat   (%d pairs)\n", n);
    for (int c = 0; c < 6; c++)
        printf("  %-62s %8ld pairs: v_mul_f32 differs in %ld (%ld beyond the sign of a zero); v_mul_f64 + v_cvt_f32_f64 (double field RTZ) differs in %ld\n",
               names[c], cnt[c], bad_sp[c], bad_sp_mag[c], bad_dp[c]);
    // (3) the add behind the switch back: p + 1.0 must be the nearest float
    long bad_add = 0, n_add = 0;
    for (int i = 0; i < n; i++) {
        const int ea = (a[i] >> 23) & 255, eb = (b[i] >> 23) & 255;
        if (ea == 0 || eb == 0 || ea == 255 || eb == 255 || ea + eb < 128 || ea + eb > 380) continue;
        float p; memcpy(&p, &sp[i], 4);
        const float s = p + 1.0f;                          // host: nearest even
        unsigned sb; memcpy(&sb, &s, 4);
        n_add++;
        if (sb != ad[i]) bad_add++;
    }
    printf("v_add_f32 right behind s_setreg back to nearest: %ld of %ld sums differ from round-to-nearest\n", bad_add, n_add);

    // (4) timing: 1024 blocks x 256 lanes, 4096 taps
    const int lanes = 1024 * 256, taps = 4096;
    float *dx, *dh, *dout;
    CHECK(hipMalloc(&dx, (size_t)(lanes + taps) * 4)); CHECK(hipMalloc(&dh, taps * 4)); CHECK(hipMalloc(&dout, lanes * 4));
    std::vector<float> hx(lanes + taps), hh(taps);
    std::uniform_real_distribution<float> u(-1.0f, 1.0f);
    for (auto &v : hx) v = u(rng);
    for (auto &v : hh) v = u(rng) * 0.01f;
    CHECK(hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dh, hh.data(), hh.size() * 4, hipMemcpyHostToDevice));
    for (int sw = 0; sw < 2; sw++) {
        for (int rep = 0; rep < 3; rep++) {
            CHECK(hipDeviceSynchronize());
            auto t0 = std::chrono::steady_clock::now();
            if (sw) hipLaunchKernelGGL(k_loop<1>, dim3(1024), dim3(256), 0, 0, dx, dh, dout, taps / 16);
            else    hipLaunchKernelGGL(k_loop<0>, dim3(1024), dim3(256), 0, 0, dx, dh, dout, taps / 16);
            CHECK(hipDeviceSynchronize());
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            if (rep == 2) printf("loop of %d lanes x %d taps (4 waves per SIMD), 16 v_mul_f32 + 16 dependent v_add_f32 per group, %s: %.1f us = %.2f cycles per tap and wave at 2.4 GHz\n",
                                 lanes, taps, sw ? "two s_setreg per group" : "no mode switches", us, us * 2400.0 / taps / 4.0);
        }
    }
    // (5) the switches' own price, registers only: 1024 SIMDs x W waves, NPER products + NPER sums per pair of switches
    {
        float *dreg; CHECK(hipMalloc(&dreg, 4096 * 256 * 4));
        auto time_it = [&](auto kern, int blocks, const char *what) {
            for (int rep = 0; rep < 3; rep++) {
                (void)hipDeviceSynchronize();
                auto t0 = std::chrono::steady_clock::now();
                hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, dreg, 20000, 0.25f);
                (void)hipDeviceSynchronize();
                const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
                if (rep == 2) printf("  %-58s %8.1f us\n", what, us);
            }
        };
        printf("registers only, 20000 turns per wave (2.4 GHz: 1 us = 2400 cycles):\n");
        time_it(k_regs<0, 8>, 256, "1 wave/SIMD,  8 mul +  8 add per turn, no switches");
        time_it(k_regs<1, 8>, 256, "1 wave/SIMD,  8 mul +  8 add per turn, two s_setreg");
        time_it(k_regs<0, 16>, 256, "1 wave/SIMD, 16 mul + 16 add per turn, no switches");
        time_it(k_regs<1, 16>, 256, "1 wave/SIMD, 16 mul + 16 add per turn, two s_setreg");
        time_it(k_regs<0, 8>, 512, "2 waves/SIMD, 8 mul +  8 add per turn, no switches");
        time_it(k_regs<1, 8>, 512, "2 waves/SIMD, 8 mul +  8 add per turn, two s_setreg");
        time_it(k_regs<0, 8>, 1024, "4 waves/SIMD, 8 mul +  8 add per turn, no switches");
        time_it(k_regs<1, 8>, 1024, "4 waves/SIMD, 8 mul +  8 add per turn, two s_setreg");
        time_it(k_regs<1, 16>, 1024, "4 waves/SIMD, 16 mul + 16 add per turn, two s_setreg");
    }
    // the same sums on the host (toward-zero products by the reference's formula, nearest adds): lanes 0, 1, 77
    {
        std::vector<float> o(lanes);
        CHECK(hipMemcpy(o.data(), dout, lanes * 4, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int lane : {0, 1, 77, 4095, lanes - 1}) {
            float acc = 0.0f;
            for (int t = 0; t < taps; t++) {
                unsigned xa, hb; memcpy(&xa, &hx[lane + t], 4); memcpy(&hb, &hh[t], 4);
                const unsigned pr = ref_mul(hb, xa);
                float p; memcpy(&p, &pr, 4);
                acc = acc + p;
            }
            if (memcmp(&acc, &o[lane], 4)) { bad++; printf("  lane %d: host %a device %a\n", lane, acc, o[lane]); }
        }
        printf("loop with switches vs the reference's products summed on the host: %d of 5 lanes differ\n", bad);
    }
    return 0;
}
