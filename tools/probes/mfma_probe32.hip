// Probe: would the 192x192 tile run faster as 4 waves x (3x3 accumulators of v_mfma_f32_32x32x16_bf16, 96x96 outputs per wave) than as the
// kernel's 8 waves x (6x3 of v_mfma_f32_16x16x32_bf16, 96x48 per wave)?  Same flops per workgroup and K-tile (36 MFMAs per wave either way),
// 24 instead of 18 ds_read_b128 per wave but 96 instead of 144 per workgroup.  MODE 0: MFMA only, 2: + LDS reads, 3: + reads + barrier.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int MODE>
__global__ __launch_bounds__(256, 1) void probe32(float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 49152 / 4; i += 256) ((float*)smem)[i] = 0.001f * i;
    __syncthreads();
    f32x16 acc[3][3];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    bf16x8 a[4][3], b[4][3];
    for (int k = 0; k < 4; ++k) for (int i = 0; i < 3; ++i) {
        a[k][i] = *(bf16x8*)(smem + (k * 3 + i) * 1024 + lane * 16);
        b[k][i] = *(bf16x8*)(smem + 16384 + (k * 3 + i) * 1024 + lane * 16);
    }
    const unsigned base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)smem + lane * 16;
#define DSR(dst, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(base), "n"(off))
    bf16x8 t[24];
    for (int it = 0; it < iters; ++it) {
        if (MODE & 1) __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                if (MODE & 2) { DSR(t[(k * 3 + i) * 2], (k * 3 + i) * 2048); DSR(t[(k * 3 + i) * 2 + 1], (k * 3 + i) * 2048 + 1024); }
#pragma unroll
                for (int j = 0; j < 3; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[k][j], a[k][i], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (MODE & 2) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            for (int k = 0; k < 24; ++k) asm volatile("" ::"v"(t[k]));
        }
    }
    float s = 0;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) s += acc[i][j][0] + acc[i][j][15];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name) {
    float* out;
    (void)hipMalloc(&out, 256 * 256 * 4);
    const int iters = 2000;
    (void)hipFuncSetAttribute((const void*)probe32<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 147456);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(probe32<MODE>, dim3(256), dim3(256), 147456, 0, out, iters);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
    }
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double flops = 256.0 * 4 * iters * 36 * 32768.0;
    printf("4 waves x 3x3 of 32x32x16, %-26s: %.3f ms  %.0f TFLOP/s  (%.0f ns per K-tile per wave)\n", name, ms, flops / ms / 1e9, ms * 1e6 / iters);
    (void)hipFree(out);
}

int main() {
    run<0>("MFMA only");
    run<2>("MFMA + 24 ds_read_b128");
    run<3>("MFMA + reads + barrier");
    return 0;
}
