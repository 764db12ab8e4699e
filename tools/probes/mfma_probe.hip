// Probe: how fast does the 192x192 kernel's inner pattern issue?  8 waves/WG, 1 WG/CU, 6x3 accumulators of
// v_mfma_f32_16x16x32_bf16 per wave, per "K-tile": 36 MFMAs [+ s_barrier] [+ 18 ds_read_b128 interleaved].
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int MODE>  // 0: MFMA only, 1: + barrier per tile, 2: + 18 LDS reads per tile (asm, prefetch style), 3: reads + barrier
__global__ __launch_bounds__(512, 2) void probe(float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 49152 / 4; i += 512) ((float*)smem)[i] = 0.001f * i;
    __syncthreads();
    f32x4 acc[6][3];
    for (int i = 0; i < 6; ++i) for (int j = 0; j < 3; ++j) acc[i][j] = (f32x4){0, 0, 0, 0};
    bf16x8 a0[6], b0[3], a1[6], b1[3];
    for (int i = 0; i < 6; ++i) { a0[i] = *(bf16x8*)(smem + i * 2048 + lane * 16); a1[i] = *(bf16x8*)(smem + 12288 + i * 2048 + lane * 16); }
    for (int j = 0; j < 3; ++j) { b0[j] = *(bf16x8*)(smem + 24576 + j * 2048 + lane * 16); b1[j] = *(bf16x8*)(smem + 32768 + j * 2048 + lane * 16); }
    const unsigned base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)smem + lane * 16;
#define DSR(dst, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(base), "n"(off))
#define ROW(r, bb, aa)                                                                    \
    acc[r][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bb[0], aa, acc[r][0], 0, 0, 0);   \
    acc[r][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bb[1], aa, acc[r][1], 0, 0, 0);   \
    acc[r][2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bb[2], aa, acc[r][2], 0, 0, 0);   \
    __builtin_amdgcn_sched_barrier(0)
    bf16x8 t[18];
    for (int it = 0; it < iters; ++it) {
        if (MODE & 1) __builtin_amdgcn_s_barrier();
        if (MODE & 2) { DSR(t[0], 0); DSR(t[1], 2048); }
        ROW(0, b0, a0[0]);
        if (MODE & 2) { DSR(t[2], 4096); DSR(t[3], 6144); }
        ROW(1, b0, a0[1]);
        if (MODE & 2) { DSR(t[4], 8192); DSR(t[5], 10240); }
        ROW(2, b0, a0[2]);
        if (MODE & 2) { DSR(t[6], 12288); DSR(t[7], 14336); }
        ROW(3, b0, a0[3]);
        if (MODE & 2) { DSR(t[8], 16384); DSR(t[9], 18432); }
        ROW(4, b0, a0[4]);
        if (MODE & 2) { DSR(t[10], 20480); DSR(t[11], 22528); }
        ROW(5, b0, a0[5]);
        if (MODE & 2) { DSR(t[12], 24576); }
        ROW(0, b1, a1[0]);
        if (MODE & 2) { DSR(t[13], 26624); }
        ROW(1, b1, a1[1]);
        if (MODE & 2) { DSR(t[14], 28672); }
        ROW(2, b1, a1[2]);
        if (MODE & 2) { DSR(t[15], 30720); }
        ROW(3, b1, a1[3]);
        if (MODE & 2) { DSR(t[16], 32768); }
        ROW(4, b1, a1[4]);
        if (MODE & 2) { DSR(t[17], 34816); }
        ROW(5, b1, a1[5]);
        if (MODE & 2) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            // keep the loaded values alive without using them as operands (would change the data pattern)
            for (int k = 0; k < 18; ++k) asm volatile("" ::"v"(t[k]));
        }
    }
    float s = 0;
    for (int i = 0; i < 6; ++i) for (int j = 0; j < 3; ++j) s += acc[i][j][0] + acc[i][j][3];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, int waves_per_wg) {
    float* out;
    hipMalloc(&out, 256 * 512 * 4);
    const int iters = 2000;
    hipFuncSetAttribute((const void*)probe<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 147456);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(64 * waves_per_wg), 147456, 0, out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
    }
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = 256.0 * waves_per_wg * iters * 36 * 16384.0;
    printf("%-34s waves/WG %d: %.3f ms  %.0f TFLOP/s  (%.0f ns per 36-MFMA tile per wave)\n", name, waves_per_wg, ms, flops / ms / 1e9, ms * 1e6 / iters);
    hipFree(out);
}

int main() {
    for (int w : {4, 8}) {
        run<0>("MFMA only", w);
        run<1>("MFMA + s_barrier per tile", w);
        run<2>("MFMA + 18 ds_read_b128", w);
        run<3>("MFMA + reads + barrier", w);
    }
    return 0;
}
