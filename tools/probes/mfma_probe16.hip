// Probe: would 16 waves per workgroup (4 per SIMD, 48x48 outputs per wave = 3x3 accumulators, 12 ds_read_b128 per K-tile)
// hide the LDS/barrier latency of the 192x192 kernel better than its 8 waves of 6x3?  Same flops per K-tile per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int ROWS, int WAVES>   // ROWS x 3 accumulators per wave
__global__ __launch_bounds__(64 * WAVES) void probe(float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 49152 / 4; i += 64 * WAVES) ((float*)smem)[i] = 0.001f * i;
    __syncthreads();
    f32x4 acc[ROWS][3];
    for (int i = 0; i < ROWS; ++i) for (int j = 0; j < 3; ++j) acc[i][j] = (f32x4){0, 0, 0, 0};
    bf16x8 a0[ROWS], b0[3], a1[ROWS], b1[3], na0[ROWS], nb0[3], na1[ROWS], nb1[3];
    for (int i = 0; i < ROWS; ++i) { a0[i] = *(bf16x8*)(smem + i * 2048 + lane * 16); a1[i] = *(bf16x8*)(smem + 12288 + i * 2048 + lane * 16); }
    for (int j = 0; j < 3; ++j) { b0[j] = *(bf16x8*)(smem + 24576 + j * 2048 + lane * 16); b1[j] = *(bf16x8*)(smem + 32768 + j * 2048 + lane * 16); }
    const unsigned base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)smem + lane * 16;
#define DSR(dst, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(base), "n"(off))
    for (int it = 0; it < iters; ++it) {
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int i = 0; i < ROWS; ++i) {
            DSR(na0[i], 0 + 1024 * 0); DSR(na1[i], 12288);
            if (i < 3) { DSR(nb0[i], 24576); DSR(nb1[i], 32768); }
#pragma unroll
            for (int j = 0; j < 3; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0[j], a0[i], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int i = 0; i < ROWS; ++i) {
#pragma unroll
            for (int j = 0; j < 3; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1[j], a1[i], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        for (int i = 0; i < ROWS; ++i) { asm volatile("" ::"v"(na0[i]), "v"(na1[i])); }
        for (int j = 0; j < 3; ++j) { asm volatile("" ::"v"(nb0[j]), "v"(nb1[j])); }
    }
    float s = 0;
    for (int i = 0; i < ROWS; ++i) for (int j = 0; j < 3; ++j) s += acc[i][j][0] + acc[i][j][3];
    out[blockIdx.x * 64 * WAVES + threadIdx.x] = s;
}

template <int ROWS, int WAVES>
void run(const char* name) {
    float* out;
    (void)hipMalloc(&out, 256 * 1024 * 4);
    const int iters = 2000;
    (void)hipFuncSetAttribute((const void*)probe<ROWS, WAVES>, hipFuncAttributeMaxDynamicSharedMemorySize, 147456);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((probe<ROWS, WAVES>), dim3(256), dim3(64 * WAVES), 147456, 0, out, iters);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
    }
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double flops = 256.0 * WAVES * iters * (ROWS * 3 * 2) * 16384.0;
    printf("%-44s: %.3f ms  %.0f TFLOP/s\n", name, ms, flops / ms / 1e9);
    (void)hipFree(out);
}

int main() {
    run<6, 8>("8 waves x (6x3), reads + barrier");
    run<3, 16>("16 waves x (3x3), reads + barrier");
    run<6, 8>("8 waves x (6x3), reads + barrier (again)");
    run<3, 16>("16 waves x (3x3), reads + barrier (again)");
    return 0;
}
