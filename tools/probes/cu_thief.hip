// A stand-in for a collective's kernel on a one-GPU box: `nwg` workgroups of 256 threads that hold 96 KiB of LDS each -- so none of them can
// share a CU with a 192x192 GEMM workgroup (144-156 of the CU's 160 KiB) or with another thief -- and spin on the 100 MHz real-time counter for `ms` milliseconds.
// Launched on a side stream next to the training step it shows what the step loses while RCCL's channels occupy that many CUs
// (tools/cu_thief_probe.py).  Every wave leaves when the time is up or after a bounded number of polls.
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/probes/cu_thief.hip -o tools/probes/_bin/libcu_thief.so
#include <hip/hip_runtime.h>

__global__ __launch_bounds__(256) void thief_kernel(unsigned long long ticks, unsigned* sink) {
    extern __shared__ unsigned hold[];       // 96 KiB (dynamic)
    hold[threadIdx.x] = threadIdx.x;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    unsigned acc = hold[(threadIdx.x * 7) & 8191];
    for (long i = 0; i < (1L << 26); ++i) {   // bounded: ~2^26 polls of >= 64 cycles each would be minutes; the clock ends it first
        if (__builtin_amdgcn_s_memrealtime() - t0 >= ticks) break;
        __builtin_amdgcn_s_sleep(8);
    }
    if (acc == 0xFFFFFFFFu) sink[0] = acc;   // keeps the LDS array alive
}

extern "C" int thief_launch(int nwg, double ms, unsigned* sink, hipStream_t stream) {
    static bool once = false;
    if (!once) {
        if (hipFuncSetAttribute((const void*)thief_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024) != hipSuccess) return -1;
        once = true;
    }
    hipLaunchKernelGGL(thief_kernel, dim3(nwg), dim3(256), 96 * 1024, stream, (unsigned long long)(ms * 1e5), sink);
    return (int)hipGetLastError();
}
