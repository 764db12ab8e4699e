// Probe of the operand path of the 192x192x64 NT GEMM WITHOUT the GEMM: every workgroup (512 threads, one per CU) walks the output tiles the
// persistent kernel would walk (same XCD-aware order, same A / B panels of qkv forward: A [12288, 768], B [2304, 768] bf16) and only STAGES their
// K-tiles into a ring in the LDS by 16-byte LDS-DMA -- no fragment reads, no MFMA.  Variants:
//   depth   = K-tiles in flight per wave before it waits (ring slots - 1); the GEMM runs depth 2 (3-slot ring of 48 KB)
//   sync    = 1: a workgroup barrier per K-tile (as the GEMM), 0: each wave free-runs behind its own counted vmcnt
//   ktile   = 64 (48 KB per stage: 6 pieces per wave) or 32 (24 KB per stage: 3 pieces per wave, twice the stages in the same LDS)
// Question: is ~55 GB/s per CU (what the GEMM's K loop pulls, and what its MFMA-less skeleton reaches) a limit of the L2 -> LDS path, or of how the
// GEMM drives it?   hipcc --offload-arch=gfx950 -O3 tools/probes/fill_probe.hip -o tools/probes/_bin/fill_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __bf16 bf16_t;
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
    const int base = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
    return base + (bid >> 3);
}
__device__ __forceinline__ void glds16_sv(const void* base_uniform, unsigned lane_byte_off, unsigned lds_byte_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(lane_byte_off), "s"(base_uniform), "s"(lds_byte_addr) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// KT = K-tile (64 or 32), DEPTH = K-tiles in flight, SYNC = barrier per K-tile
// W = 0: tiles in row-major order (the GEMM's); W > 0: column blocks of W tile columns, row-major inside a block, so that an XCD's contiguous
// chunk of the list is a (chunk / W)-row x W-column rectangle of tiles instead of whole tile rows
template <int KT, int DEPTH, int SYNC>
__global__ __launch_bounds__(512, 2) void fill_kernel(const bf16_t* A, const bf16_t* B, int M, int N, int K, int reps, unsigned* sink, int W) {
    extern __shared__ __attribute__((aligned(128))) char smem[];
    constexpr int TM = 192, ROWB = KT * 2;                 // bytes per tile row
    constexpr int OP = TM * ROWB, STAGE = 2 * OP, SLOTS = DEPTH + 1;
    constexpr int CH = ROWB / 16;                          // 16-byte chunks per row
    constexpr int PIECES = 2 * TM * CH / 512;              // per thread per K-tile (A rows then B rows)
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned sbase = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)smem);
    const int tiles_m = M / TM, tiles_n = N / TM, nwg = tiles_m * tiles_n, nt = K / KT;
    unsigned off[PIECES];
#pragma unroll
    for (int p = 0; p < PIECES; ++p) {
        const int i = p % (PIECES / 2);
        const int slot = i * 512 + tid;
        const int row = slot / CH, lc = (slot % CH) ^ ((row >> 1) & (CH - 1));
        off[p] = (unsigned)((row * K + lc * 8) * 2);
    }
    unsigned acc = 0;
    for (int rep = 0; rep < reps; ++rep) {
        int g = 0;                                          // running K-tile counter of this workgroup: ring slot = g % SLOTS
        for (int it = blockIdx.x; it < nwg; it += gridDim.x) {
            const int sid = xcd_remap(it, nwg);
            int tm = sid / tiles_n, tn = sid % tiles_n;
            if (W > 0) {
                const int per = tiles_m * W, b = sid / per, r = sid % per;
                const int w = (tiles_n - b * W) < W ? (tiles_n - b * W) : W;   // the last block may be narrower
                tm = r / w, tn = b * W + r % w;
            }
            const bf16_t* Ap = A + (long)tm * TM * K;
            const bf16_t* Bp = B + (long)tn * TM * K;
            for (int t = 0; t < nt; ++t, ++g) {
                const unsigned dst = sbase + (g % SLOTS) * STAGE;
#pragma unroll
                for (int p = 0; p < PIECES; ++p) {
                    const bool isA = p < PIECES / 2;
                    const int i = p % (PIECES / 2);
                    glds16_sv((isA ? Ap : Bp) + t * KT, off[p], dst + (isA ? 0 : OP) + (i * 512 + wave * 64) * 16);
                }
                wait_vm<PIECES * DEPTH>();                 // the K-tile issued DEPTH tiles ago has landed (this wave's pieces)
                if (SYNC) __builtin_amdgcn_s_barrier();
            }
        }
    }
    wait_vm<0>();
    __syncthreads();
    acc += ((unsigned*)smem)[tid];
    if (acc == 0x12345678u) sink[0] = acc;                  // keeps the staging alive
}

template <int KT, int DEPTH, int SYNC>
static void run(const bf16_t* A, const bf16_t* B, int M, int N, int K, unsigned* sink, const char* what, int W = 0) {
    constexpr int STAGE = 2 * 192 * KT * 2, SLOTS = DEPTH + 1;
    const size_t lds = (size_t)STAGE * SLOTS;
    if (lds > 160 * 1024) { printf("%-46s skipped (%zu KB of LDS)\n", what, lds / 1024); return; }
    CHECK(hipFuncSetAttribute((const void*)fill_kernel<KT, DEPTH, SYNC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int reps = 20;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((fill_kernel<KT, DEPTH, SYNC>), dim3(256), dim3(512), lds, 0, A, B, M, N, K, reps, sink, W);
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((fill_kernel<KT, DEPTH, SYNC>), dim3(256), dim3(512), lds, 0, A, B, M, N, K, reps, sink, W);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = (double)(M / 192) * (N / 192) * (K / KT) * STAGE * reps;
    printf("%-46s %7.1f us per pass   %6.2f TB/s   %5.1f GB/s per CU\n", what, ms * 1e3 / reps, bytes / ms / 1e9, bytes / ms / 1e6 / 256);
}

int main() {
    const int M = 12288, N = 2304, K = 768;
    bf16_t *A, *B; unsigned* sink;
    CHECK(hipMalloc(&A, (size_t)M * 3072 * 2)); CHECK(hipMalloc(&B, (size_t)3072 * 3072 * 2)); CHECK(hipMalloc(&sink, 64));
    std::vector<unsigned short> h((size_t)M * 3072);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned short)(0x3c00 + (rand() & 0x3ff));
    CHECK(hipMemcpy(A, h.data(), (size_t)M * 3072 * 2, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(B, h.data(), (size_t)3072 * 3072 * 2, hipMemcpyHostToDevice));
    printf("# operand staging only, qkv-forward panels (768 output tiles of 192 x 192, 12 K-tiles of 64 each = 453 MB per pass), 256 workgroups x 512 threads\n");
    run<64, 2, 1>(A, B, M, N, K, sink, "K-tile 64, 2 in flight, barrier (the GEMM's)");
    run<64, 2, 1>(A, B, M, N, K, sink, "  same, column blocks of 6 tiles", 6);
    run<64, 2, 1>(A, B, M, N, K, sink, "  same, column blocks of 4 tiles", 4);
    run<64, 2, 1>(A, B, M, N, K, sink, "  same, column blocks of 3 tiles", 3);
    run<64, 2, 1>(A, B, M, N, K, sink, "  same, column blocks of 2 tiles", 2);
    run<64, 2, 0>(A, B, M, N, K, sink, "K-tile 64, 2 in flight, no barrier");
    run<64, 2, 0>(A, B, M, N, K, sink, "  same, column blocks of 6 tiles", 6);
    run<64, 1, 1>(A, B, M, N, K, sink, "K-tile 64, 1 in flight, barrier");
    run<32, 2, 1>(A, B, M, N, K, sink, "K-tile 32, 2 in flight, barrier");
    run<32, 4, 1>(A, B, M, N, K, sink, "K-tile 32, 4 in flight, barrier");
    run<32, 5, 1>(A, B, M, N, K, sink, "K-tile 32, 5 in flight, barrier");
    run<32, 5, 0>(A, B, M, N, K, sink, "K-tile 32, 5 in flight, no barrier");
    printf("# the other shapes of the step (K-tile 64, 2 in flight, barrier); W = tile columns per column block, 0 = row-major\n");
    const int shapes[][2] = {{3072, 768}, {768, 3072}, {768, 2304}, {768, 768}};
    for (auto& sh : shapes)
        for (int W : {0, 2, 4, 6, 8}) {
            if (W >= sh[0] / 192 && W) continue;
            char what[96];
            snprintf(what, sizeof what, "N %4d K %4d W %d", sh[0], sh[1], W);
            run<64, 2, 1>(A, B, M, sh[0], sh[1], sink, what, W);
        }
    return 0;
}
