// Which compute units does a stream made by hipExtStreamCreateWithCUMask run on?  Every workgroup records its XCC_ID and HW_ID
// (shader engine, CU); the host prints the set per mask.  Answers how mask bits map to (XCD, SE, CU) on this stack.
//   hipcc -O2 --offload-arch=gfx950 tools/cu_mask_probe.hip -o /tmp/cu_mask_probe && /tmp/cu_mask_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <set>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void where(unsigned *out, int spin)
{
    unsigned hw = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));       // HW_REG_HW_ID, 32 bits
    unsigned xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11));      // HW_REG_XCC_ID, 4 bits
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)spin) { }     // keep the CU busy so that workgroups spread
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
}

static void run(const char *name, hipStream_t st, unsigned *d, int nwg)
{
    CHECK(hipMemsetAsync(d, 0xFF, (size_t)nwg * 8, st));
    hipLaunchKernelGGL(where, dim3(nwg), dim3(256), 0, st, d, 2000);
    CHECK(hipStreamSynchronize(st));
    std::vector<unsigned> h((size_t)nwg * 2);
    CHECK(hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost));
    std::map<unsigned, std::set<unsigned>> per_xcc;       // xcc -> set of (se, cu)
    for (int i = 0; i < nwg; i++) {
        const unsigned hw = h[2 * i], xcc = h[2 * i + 1] & 0xF;
        const unsigned cu = (hw >> 8) & 0xF, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;      // gfx9 HW_ID: CU_ID [11:8], SH_ID [12], SE_ID [15:13]
        per_xcc[xcc].insert(se * 32 + sh * 16 + cu);
    }
    size_t total = 0;
    printf("%-44s", name);
    for (auto &kv : per_xcc) { printf(" xcc%u:%zu", kv.first, kv.second.size()); total += kv.second.size(); }
    printf("  = %zu CUs\n", total);
    if (total <= 40) { for (auto &kv : per_xcc) { printf("      xcc%u (se.cu):", kv.first); for (unsigned v : kv.second) printf(" %u.%u", v / 32, v % 16); printf("\n"); } }
}

int main()
{
    unsigned *d; const int nwg = 8192;
    CHECK(hipMalloc(&d, (size_t)nwg * 8));
    hipStream_t s0; CHECK(hipStreamCreate(&s0));
    run("unmasked stream", s0, d, nwg);
    struct { const char *name; uint32_t m[8]; } masks[] = {
        {"bits 0..31 (first word all ones)", {0xFFFFFFFFu, 0, 0, 0, 0, 0, 0, 0}},
        {"bits 0..7", {0xFFu, 0, 0, 0, 0, 0, 0, 0}},
        {"bits 0..15", {0xFFFFu, 0, 0, 0, 0, 0, 0, 0}},
        {"bit 0 of every word", {1, 1, 1, 1, 1, 1, 1, 1}},
        {"bits 0,1 of every word", {3, 3, 3, 3, 3, 3, 3, 3}},
        {"all but bits 0..15", {0xFFFF0000u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u}},
        {"every 16th bit (16 CUs)", {0x00010001u, 0x00010001u, 0x00010001u, 0x00010001u, 0x00010001u, 0x00010001u, 0x00010001u, 0x00010001u}},
        {"all but every 16th bit", {~0x00010001u, ~0x00010001u, ~0x00010001u, ~0x00010001u, ~0x00010001u, ~0x00010001u, ~0x00010001u, ~0x00010001u}},
    };
    for (auto &mk : masks) {
        hipStream_t st;
        hipError_t e = hipExtStreamCreateWithCUMask(&st, 8, mk.m);
        if (e != hipSuccess) { printf("%-44s hipExtStreamCreateWithCUMask: %s\n", mk.name, hipGetErrorString(e)); continue; }
        run(mk.name, st, d, nwg);
        CHECK(hipStreamDestroy(st));
    }
    return 0;
}
