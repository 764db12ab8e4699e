// Probe: what does a stream CU mask (hipExtStreamCreateWithCUMask) select on an MI355X?  For each candidate mask a grid of
// short blocks records its (XCC, SE, CU) from HW registers; prints how many distinct CUs per XCC ran work.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <set>
#include <vector>

__global__ void probe(unsigned* out) {
    if (threadIdx.x == 0) {
        unsigned xcc, hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        out[2 * blockIdx.x] = xcc & 0xf;
        out[2 * blockIdx.x + 1] = hw;
    }
    const long long s = wall_clock64();
    while (wall_clock64() - s < 500) { __builtin_amdgcn_s_sleep(8); }
}

static void run(const char* name, const std::vector<uint32_t>& mask) {
    hipStream_t st;
    hipError_t e = hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data());
    if (e != hipSuccess) { printf("%s: create failed: %s\n", name, hipGetErrorString(e)); return; }
    const int n = 4096;
    unsigned* d;
    if (hipMalloc(&d, n * 8) != hipSuccess) return;
    hipLaunchKernelGGL(probe, dim3(n), dim3(64), 0, st, d);
    if (hipStreamSynchronize(st) != hipSuccess) { printf("%s: sync failed\n", name); return; }
    std::vector<unsigned> h(2 * n);
    (void)hipMemcpy(h.data(), d, n * 8, hipMemcpyDeviceToHost);
    std::map<unsigned, std::set<unsigned>> per;
    for (int i = 0; i < n; ++i) per[h[2 * i]].insert(h[2 * i + 1] & 0xfff00);   // cu_id bits 11:8, sh 12, se 15:13 on gfx9
    printf("%-28s:", name);
    size_t tot = 0;
    for (auto& kv : per) { printf(" xcc%u=%zu", kv.first, kv.second.size()); tot += kv.second.size(); }
    printf("  total %zu\n", tot);
    (void)hipFree(d);
    (void)hipStreamDestroy(st);
}

int main() {
    std::vector<uint32_t> all(8, 0xffffffffu), lo(8, 0), hi(8, 0), even(8, 0), odd(8, 0), x4(8, 0), x4b(8, 0);
    for (int i = 0; i < 256; ++i) {
        if (i < 128) lo[i / 32] |= 1u << (i % 32); else hi[i / 32] |= 1u << (i % 32);
        if (i % 2 == 0) even[i / 32] |= 1u << (i % 32); else odd[i / 32] |= 1u << (i % 32);
        if (i % 8 < 4) x4[i / 32] |= 1u << (i % 32); else x4b[i / 32] |= 1u << (i % 32);
    }
    run("all 256 bits", all);
    run("bits 0..127", lo);
    run("bits 128..255", hi);
    run("even bits", even);
    run("odd bits", odd);
    run("bits with i%8 < 4", x4);
    run("bits with i%8 >= 4", x4b);
    return 0;
}
