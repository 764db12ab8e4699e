// Probe: which workgroups of a 512 x 256-thread, 72 KiB-LDS grid does the dispatcher co-schedule on one CU?
// Every block records its CU (__smid: xcc | se | cu) and its start time; all blocks spin ~30 us so the first 512 are
// co-resident.  Prints, per CU, the block ids it hosted, and the pairing rule that explains them.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>

__global__ __launch_bounds__(256, 2) void probe(unsigned* cu, long long* t0) {
    extern __shared__ char smem[];
    if (threadIdx.x == 0) {
        cu[blockIdx.x] = __smid();
        t0[blockIdx.x] = wall_clock64();
    }
    const long long s = wall_clock64();
    while (wall_clock64() - s < 3000) { __builtin_amdgcn_s_sleep(8); }  // 100 MHz ticks: 30 us
    if (threadIdx.x == 1000) smem[0] = 1;
}

int main() {
    const int n = 1536;
    unsigned* cu; long long* t0;
    hipMalloc(&cu, n * 4); hipMalloc(&t0, n * 8);
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 73728);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(probe, dim3(n), dim3(256), 73728, 0, cu, t0);
    hipDeviceSynchronize();
    std::vector<unsigned> hc(n); std::vector<long long> ht(n);
    hipMemcpy(hc.data(), cu, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(ht.data(), t0, n * 8, hipMemcpyDeviceToHost);
    long long tmin = ht[0];
    for (int i = 0; i < n; ++i) tmin = ht[i] < tmin ? ht[i] : tmin;
    std::map<unsigned, std::vector<int>> by;
    for (int i = 0; i < n; ++i) by[hc[i]].push_back(i);
    printf("distinct CUs: %zu\n", by.size());
    int shown = 0;
    int rule256 = 0, rule8 = 0, rule1 = 0, pairs = 0;
    for (auto& kv : by) {
        std::vector<int> first;
        for (int b : kv.second) if (ht[b] - tmin < 1000) first.push_back(b);  // started in the first 10 us
        if (first.size() == 2) {
            ++pairs;
            const int d = first[1] - first[0];
            rule256 += d == 256; rule8 += d == 8; rule1 += d == 1;
        }
        if (shown < 12) {
            printf("cu %5u:", kv.first);
            for (int b : kv.second) printf(" %d(%lld)", b, (ht[b] - tmin) / 100);
            printf("\n");
            ++shown;
        }
    }
    printf("CUs with exactly two first-round residents: %d; of those block-id difference 256: %d, 8: %d, 1: %d\n", pairs, rule256, rule8, rule1);
    return 0;
}
