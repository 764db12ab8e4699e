// Probe: vector-instruction issue cost on one SIMD as a function of resident waves, alone and beside MFMAs.
// Each wave runs `iters` trips of 32 instructions of one kind on 16 independent register chains (inline asm, so hipcc
// neither packs nor removes them); cycles come from s_memtime inside the kernel (the clock the chip holds under that load),
// reported as SIMD cycles per wave-instruction = elapsed / (instructions per wave * waves per SIMD).
//   kinds: 0 v_fma_f32, 1 v_pk_fma_f32, 2 v_exp_f32, 3 v_add_f32, 4 v_pk_add_f32, 5 v_cvt_pk_bf16_f32, 6 v_max3_f32,
//          7 = 4 x (1 v_mfma_32x32x16 + 6 v_fma_f32), 8 = 4 x (1 mfma + 3 v_exp_f32), 9 = mfma only (4 per trip)
// build: hipcc --offload-arch=gfx950 -O3 -o _bin/valu_probe valu_probe.hip ; run: _bin/valu_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ __launch_bounds__(256) void probe(unsigned long long* cyc, float* sink, int iters) {
    float r[16];
    f32x2 q[8];
    for (int i = 0; i < 16; ++i) r[i] = 0.001f * (threadIdx.x + i);
    for (int i = 0; i < 8; ++i) q[i] = (f32x2){r[2 * i], r[2 * i + 1]};
    f32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    bf16x8 fa, fb;
    for (int i = 0; i < 8; ++i) { fa[i] = (__bf16)(0.01f * i); fb[i] = (__bf16)(0.02f * i); }
    const float c = 1.0001f;
    __syncthreads();
    unsigned long long t0;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 2; ++rep) {
            if constexpr (KIND == 0) {
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(r[i]) : "v"(c));
            } else if constexpr (KIND == 1) {
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(q[i]) : "v"(q[(i + 1) & 7]));
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(q[i]) : "v"(q[(i + 1) & 7]));
            } else if constexpr (KIND == 2) {
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(r[i]));
            } else if constexpr (KIND == 3) {
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(r[i]) : "v"(c));
            } else if constexpr (KIND == 4) {
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(q[i]) : "v"(q[(i + 1) & 7]));
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(q[i]) : "v"(q[(i + 1) & 7]));
            } else if constexpr (KIND == 5) {
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(r[i]) : "v"(c));
            } else if constexpr (KIND == 6) {
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(c), "v"(r[(i + 1) & 15]));
            } else if constexpr (KIND == 7) {
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc, 0, 0, 0);
#pragma unroll
                    for (int i = 0; i < 6; ++i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(r[g * 6 + i]) : "v"(c));
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else if constexpr (KIND == 8) {
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc, 0, 0, 0);
#pragma unroll
                    for (int i = 0; i < 3; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(r[g * 3 + i]));
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    unsigned long long t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += r[i] + acc[i];
    for (int i = 0; i < 8; ++i) s += q[i][0] + q[i][1];
    if (s == 12345.678f) sink[0] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int KIND>
static void run(const char* name, int per_trip_valu, int per_trip_mfma) {
    const int iters = 4000;
    unsigned long long* cyc;
    float* sink;
    hipMalloc(&cyc, 256 * 8 * 4 * 8);
    hipMalloc(&sink, 4);
    for (int wgs = 1; wgs <= 8; wgs *= 2) {   // workgroups of 4 waves per CU = waves per SIMD
        const int grid = 256 * wgs;
        hipLaunchKernelGGL(probe<KIND>, dim3(grid), dim3(256), 0, 0, cyc, sink, iters);
        hipDeviceSynchronize();
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(probe<KIND>, dim3(grid), dim3(256), 0, 0, cyc, sink, iters);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        static unsigned long long h[256 * 8 * 4];
        hipMemcpy(h, cyc, grid * 4 * 8, hipMemcpyDeviceToHost);
        double avg = 0; for (int i = 0; i < grid * 4; ++i) avg += (double)h[i]; avg /= grid * 4;
        const double trips = (double)iters;
        printf("%-28s waves/SIMD %d: %7.1f cycles/trip/wave  -> %5.2f SIMD-cycles per VALU instr (%d/trip), %5.1f per MFMA (%d/trip); clock %.2f GHz\n", name, wgs,
               avg / trips, per_trip_valu ? avg / trips / per_trip_valu / wgs : 0.0, per_trip_valu, per_trip_mfma ? avg / trips / per_trip_mfma / wgs : 0.0,
               per_trip_mfma, avg / (ms * 1e6));
    }
}

int main() {
    run<0>("v_fma_f32", 32, 0);
    run<1>("v_pk_fma_f32", 32, 0);
    run<2>("v_exp_f32", 32, 0);
    run<3>("v_add_f32", 32, 0);
    run<4>("v_pk_add_f32", 32, 0);
    run<5>("v_cvt_pk_bf16_f32", 32, 0);
    run<6>("v_max3_f32", 32, 0);
    run<9>("mfma 32x32x16 only", 0, 4);
    run<7>("mfma + 6 v_fma each", 24, 4);
    run<8>("mfma + 3 v_exp each", 12, 4);
    return 0;
}
