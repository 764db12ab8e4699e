// Probe for a larger GEMM tile: 4 waves per workgroup = one per SIMD, 128 x 96 outputs per wave = 8 x 6
// accumulators of v_mfma_f32_16x16x32_bf16 (192 registers: AGPRs), per K-tile of 64: 96 MFMAs [+ 28 ds_read_b128 into the other fragment
// set, interleaved] [+ s_barrier].  Next to it the 192x192 kernel's pattern (8 waves, 6 x 3) from mfma_probe.hip for the same run.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_probe_big.hip -o tools/probes/_bin/mfma_probe_big
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define DSR(dst, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(base), "n"(off))

template <int MODE>  // bit 0: s_barrier per K-tile, bit 1: 28 LDS reads per K-tile (prefetch style into the other set)
__global__ __launch_bounds__(256, 1) void probe_big(float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 57344 / 4; i += 256) ((float*)smem)[i] = 0.001f * i;
    __syncthreads();
    f32x4 acc[8][6];
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 6; ++j) acc[i][j] = (f32x4){0, 0, 0, 0};
    bf16x8 xa[2][8], xb[2][6], ya[2][8], yb[2][6];      // two fragment sets (tile t computes from one while tile t+1 is read into the other)
    for (int h = 0; h < 2; ++h) {
        for (int i = 0; i < 8; ++i) { xa[h][i] = *(bf16x8*)(smem + (h * 8 + i) * 1024 + lane * 16); ya[h][i] = xa[h][i]; }
        for (int j = 0; j < 6; ++j) { xb[h][j] = *(bf16x8*)(smem + 16384 + (h * 6 + j) * 1024 + lane * 16); yb[h][j] = xb[h][j]; }
    }
    const unsigned base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)smem + lane * 16;
#define ROWB(r, bb, aa)                                                                                          \
    _Pragma("unroll") for (int j_ = 0; j_ < 6; ++j_) acc[r][j_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bb[j_], aa, acc[r][j_], 0, 0, 0); \
    __builtin_amdgcn_sched_barrier(0)
    // one K-tile: compute from (Ca, Cb), read the next tile's 28 fragments into (Na, Nb) -- two reads behind each of the first 14 MFMA rows
#define TILE(Ca, Cb, Na, Nb)                                                                                     \
    {                                                                                                            \
        if (MODE & 1) __builtin_amdgcn_s_barrier();                                                              \
        _Pragma("unroll") for (int h = 0; h < 2; ++h) {                                                          \
            _Pragma("unroll") for (int r = 0; r < 8; ++r) {                                                      \
                if (MODE & 2) {                                                                                  \
                    const int k = h * 8 + r;                                                                     \
                    if (k < 14) {                                                                                \
                        if (k < 8) { DSR(Na[0][k], k * 1024); DSR(Na[1][k], 8192 + k * 1024); }                  \
                        else { DSR(Nb[0][k - 8], 16384 + (k - 8) * 1024); DSR(Nb[1][k - 8], 16384 + 6144 + (k - 8) * 1024); } \
                    }                                                                                            \
                }                                                                                                \
                ROWB(r, Cb[h], Ca[h][r]);                                                                        \
            }                                                                                                    \
        }                                                                                                        \
        if (MODE & 2) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); }  \
    }
    for (int it = 0; it < iters; it += 2) {
        TILE(xa, xb, ya, yb)
        TILE(ya, yb, xa, xb)
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 6; ++j) s += acc[i][j][0] + acc[i][j][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>  // the 192x192 kernel's pattern: 8 waves, 6 x 3 accumulators, 18 reads per K-tile
__global__ __launch_bounds__(512, 2) void probe_192(float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 49152 / 4; i += 512) ((float*)smem)[i] = 0.001f * i;
    __syncthreads();
    f32x4 acc[6][3];
    for (int i = 0; i < 6; ++i) for (int j = 0; j < 3; ++j) acc[i][j] = (f32x4){0, 0, 0, 0};
    bf16x8 xa[2][6], xb[2][3], ya[2][6], yb[2][3];
    for (int h = 0; h < 2; ++h) {
        for (int i = 0; i < 6; ++i) { xa[h][i] = *(bf16x8*)(smem + (h * 6 + i) * 1024 + lane * 16); ya[h][i] = xa[h][i]; }
        for (int j = 0; j < 3; ++j) { xb[h][j] = *(bf16x8*)(smem + 16384 + (h * 3 + j) * 1024 + lane * 16); yb[h][j] = xb[h][j]; }
    }
    const unsigned base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)smem + lane * 16;
#define ROWS(r, bb, aa)                                                                                          \
    _Pragma("unroll") for (int j_ = 0; j_ < 3; ++j_) acc[r][j_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bb[j_], aa, acc[r][j_], 0, 0, 0); \
    __builtin_amdgcn_sched_barrier(0)
#define TILES(Ca, Cb, Na, Nb)                                                                                    \
    {                                                                                                            \
        if (MODE & 1) __builtin_amdgcn_s_barrier();                                                              \
        _Pragma("unroll") for (int h = 0; h < 2; ++h) {                                                          \
            _Pragma("unroll") for (int r = 0; r < 6; ++r) {                                                      \
                if (MODE & 2) {                                                                                  \
                    const int k = h * 6 + r;                                                                     \
                    if (k < 9) {                                                                                 \
                        if (k < 6) { DSR(Na[0][k], k * 1024); DSR(Na[1][k], 6144 + k * 1024); }                  \
                        else { DSR(Nb[0][k - 6], 16384 + (k - 6) * 1024); DSR(Nb[1][k - 6], 16384 + 3072 + (k - 6) * 1024); } \
                    }                                                                                            \
                }                                                                                                \
                ROWS(r, Cb[h], Ca[h][r]);                                                                        \
            }                                                                                                    \
        }                                                                                                        \
        if (MODE & 2) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); }  \
    }
    for (int it = 0; it < iters; it += 2) {
        TILES(xa, xb, ya, yb)
        TILES(ya, yb, xa, xb)
    }
    float s = 0;
    for (int i = 0; i < 6; ++i) for (int j = 0; j < 3; ++j) s += acc[i][j][0] + acc[i][j][3];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <typename K>
void run(const char* name, K kern, int threads, int lds, double mfma_per_tile_per_wave) {
    float* out;
    hipMalloc(&out, 256 * 512 * 4);
    const int iters = 2000;
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(256), dim3(threads), lds, 0, out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    const double flops = 256.0 * (threads / 64) * iters * mfma_per_tile_per_wave * 16384.0;
    printf("%-62s %.3f ms  %5.0f TFLOP/s\n", name, ms, flops / ms / 1e9);
    hipFree(out);
}

int main() {
    run("192x192 tile: 8 waves x 6x3, MFMA only", probe_192<0>, 512, 49152, 36);
    run("192x192 tile: 8 waves x 6x3, + 18 ds_read_b128 per K-tile", probe_192<2>, 512, 49152, 36);
    run("192x192 tile: 8 waves x 6x3, + reads + barrier", probe_192<3>, 512, 49152, 36);
    run("256x192 tile: 4 waves x 8x6 (AGPR accumulators), MFMA only", probe_big<0>, 256, 57344, 96);
    run("256x192 tile: 4 waves x 8x6, + 28 ds_read_b128 per K-tile", probe_big<2>, 256, 57344, 96);
    run("256x192 tile: 4 waves x 8x6, + reads + barrier", probe_big<3>, 256, 57344, 96);
    return 0;
}
