// HBM-bound row kernels for gfx950: LayerNorm forward/backward (fp32 statistics, bf16 output for the
// following MFMA GEMM), column sums (bias gradients), batch sums (query-embedding gradients), weight
// packing (fp32 master -> bf16 [N,K] and its transpose [K,N]), fp32->bf16 row casts and sequence
// assembly.  One wave (64 lanes) owns one row; each lane moves 16 B per access.
#include "vt_common.h"

namespace {

// per-lane vector width of the LayerNorm kernels: a row is 64 lanes x VEC floats x V pieces
template <int VEC> struct LnVec;
template <> struct LnVec<4> { typedef f32x4 F; typedef bf16x4 H; };
template <> struct LnVec<2> { typedef __attribute__((ext_vector_type(2))) float F; typedef bf16x2 H; };

// ------------------------------------------------------------------------------------------------
// LayerNorm forward: y = (x - mean) * rstd * gamma + beta, y in bf16, stats saved in fp32
// ------------------------------------------------------------------------------------------------
template <int V, int VEC>  // dim = 64 * VEC * V: 256..1024 with 16-B accesses, 128 / 384 (the discriminator's width) with 8-B
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, RowMap xmap, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, float eps, int64_t rows,
                                                      bf16_t* __restrict__ y, float* __restrict__ mean, float* __restrict__ rstd) {
    typedef typename LnVec<VEC>::F F;
    typedef typename LnVec<VEC>::H Hh;
    constexpr int DIM = 64 * VEC * V;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    F g[V], b[V];
#pragma unroll
    for (int i = 0; i < V; ++i) {
        g[i] = *(const F*)(gamma + (i * 64 + lane) * VEC);
        b[i] = *(const F*)(beta + (i * 64 + lane) * VEC);
    }
    for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < rows; row += (int64_t)gridDim.x * 4) {
        const float* xr = x + xmap(row) * DIM;
        F v[V];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < V; ++i) {
            v[i] = *(const F*)(xr + (i * 64 + lane) * VEC);
            if constexpr (VEC == 4) s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
            else s += v[i][0] + v[i][1];
        }
        const float mu = wave_sum(s) * (1.0f / DIM);
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < V; ++i)
#pragma unroll
            for (int r = 0; r < VEC; ++r) {
                const float d = v[i][r] - mu;
                q += d * d;
            }
        const float var = wave_sum(q) * (1.0f / DIM);
        const float rs = 1.0f / sqrtf(var + eps);
        bf16_t* yr = y + row * DIM;
#pragma unroll
        for (int i = 0; i < V; ++i) {
            Hh o;
#pragma unroll
            for (int r = 0; r < VEC; ++r) o[r] = f2bf((v[i][r] - mu) * rs * g[i][r] + b[i][r]);
            *(Hh*)(yr + (i * 64 + lane) * VEC) = o;
        }
        if (lane == 0) {
            mean[row] = mu;
            rstd[row] = rs;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// LayerNorm backward (+ residual-stream add).  With xh = (x-mean)*rstd, a = dy*gamma:
//   dx = rstd * (a - mean(a) - xh * mean(a*xh));   out = dres + dx  (fp32, and a bf16 copy)
// Column partials per workgroup: dgamma = sum dy*xh, dbeta = sum dy, dxsum = sum out (the bias
// gradient of the Linear whose output this residual stream is).  partial layout [grid][3][DIM].
// ------------------------------------------------------------------------------------------------
template <int V, int VEC>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const bf16_t* __restrict__ dy, const float* __restrict__ x, RowMap xmap,
                                                      const float* __restrict__ gamma, const float* __restrict__ mean,
                                                      const float* __restrict__ rstd, const float* __restrict__ dres, int64_t rows,
                                                      float* __restrict__ dx, bf16_t* __restrict__ dxb, float* __restrict__ partial) {
    typedef typename LnVec<VEC>::F F;
    typedef typename LnVec<VEC>::H Hh;
    constexpr int DIM = 64 * VEC * V;
    __shared__ float red[4][DIM];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    F g[V], ag[V], ab[V], as[V];
#pragma unroll
    for (int i = 0; i < V; ++i) {
        g[i] = *(const F*)(gamma + (i * 64 + lane) * VEC);
#pragma unroll
        for (int r = 0; r < VEC; ++r) ag[i][r] = ab[i][r] = as[i][r] = 0.f;
    }
    for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < rows; row += (int64_t)gridDim.x * 4) {
        const int64_t pr = xmap(row);
        const float mu = mean[row], rs = rstd[row];
        F xh[V], a[V], rv[V];
        float s1 = 0.f, s2 = 0.f;
        if (dres) {   // issued with the row's other loads: one more 3 KB per wave in flight while the two row sums are taken
#pragma unroll
            for (int i = 0; i < V; ++i) rv[i] = *(const F*)(dres + pr * DIM + (i * 64 + lane) * VEC);
        }
#pragma unroll
        for (int i = 0; i < V; ++i) {
            const F xv = *(const F*)(x + pr * DIM + (i * 64 + lane) * VEC);
            const Hh dv = *(const Hh*)(dy + row * DIM + (i * 64 + lane) * VEC);
#pragma unroll
            for (int r = 0; r < VEC; ++r) {
                const float d = bf2f(dv[r]);
                xh[i][r] = (xv[r] - mu) * rs;
                a[i][r] = d * g[i][r];
                s1 += a[i][r];
                s2 += a[i][r] * xh[i][r];
                ag[i][r] += d * xh[i][r];
                ab[i][r] += d;
            }
        }
        const float m1 = wave_sum(s1) * (1.0f / DIM);
        const float m2 = wave_sum(s2) * (1.0f / DIM);
#pragma unroll
        for (int i = 0; i < V; ++i) {
            F o;
#pragma unroll
            for (int r = 0; r < VEC; ++r) o[r] = rs * (a[i][r] - m1 - xh[i][r] * m2);
            if (dres) {
#pragma unroll
                for (int r = 0; r < VEC; ++r) o[r] += rv[i][r];
            }
#pragma unroll
            for (int r = 0; r < VEC; ++r) as[i][r] += o[r];
            *(F*)(dx + pr * DIM + (i * 64 + lane) * VEC) = o;
            if (dxb) {
                Hh ob;
#pragma unroll
                for (int r = 0; r < VEC; ++r) ob[r] = f2bf(o[r]);
                *(Hh*)(dxb + pr * DIM + (i * 64 + lane) * VEC) = ob;
            }
        }
    }
    // cross-wave reduction of the three column partials, one at a time through LDS
    float* out = partial + (int64_t)blockIdx.x * 3 * DIM;
#pragma unroll
    for (int w = 0; w < 3; ++w) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < V; ++i) *(F*)(&red[wave][(i * 64 + lane) * VEC]) = (w == 0 ? ag[i] : (w == 1 ? ab[i] : as[i]));
        __syncthreads();
        for (int c = threadIdx.x; c < DIM; c += 256) out[w * DIM + c] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
    }
}

// outs[w][c] = sum_s partial[s * slab_stride + w * width + c]   for w < nout (<= 3)
// workgroup = 32 columns x 8 slab lanes; fixed summation order => deterministic
struct ReduceOuts {
    float* o[3];
};
template <int NL>  // slab lanes per column: 8 (256 threads) or 32 (1024 threads, for the 512-slab LayerNorm partials)
__global__ __launch_bounds__(32 * NL) void reduce_partials_kernel(const float* __restrict__ partial, int nslab, int64_t slab_stride, int width,
                                                                   ReduceOuts outs) {
    __shared__ float red[NL][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + tx;
    const int w = blockIdx.y;
    float s = 0.f;
    if (c < width) {
        const float* p = partial + (int64_t)w * width + c;
#pragma unroll 4
        for (int i = ty; i < nslab; i += NL) s += p[(int64_t)i * slab_stride];
    }
    red[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && c < width) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < NL; ++k) t += red[k][tx];
        outs.o[w][c] = t;
    }
}

static void launch_reduce_partials(const float* partial, int nslab, int64_t slab_stride, int width, int nout, const ReduceOuts& ro, hipStream_t s) {
    const dim3 grid((width + 31) / 32, nout);
    if (nslab >= 128) hipLaunchKernelGGL(reduce_partials_kernel<32>, grid, dim3(1024), 0, s, partial, nslab, slab_stride, width, ro);
    else hipLaunchKernelGGL(reduce_partials_kernel<8>, grid, dim3(256), 0, s, partial, nslab, slab_stride, width, ro);
}

// several reductions in one launch (blockIdx.z = item); per item the arithmetic of reduce_partials_kernel<lanes>
struct ReduceGroup {
    vtReduceItem it[VT_REDUCE_MAX_GROUP];
};
__global__ __launch_bounds__(1024) void reduce_grouped_kernel(const ReduceGroup g) {
    __shared__ float red[32][33];
    const vtReduceItem& q = g.it[blockIdx.z];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + tx;
    const int w = blockIdx.y;
    if (w >= q.nout || blockIdx.x * 32 >= q.width) return;      // uniform per workgroup
    float s = 0.f;
    if (c < q.width && ty < q.lanes) {
        const float* p = q.partial + (int64_t)w * q.width + c;
#pragma unroll 4
        for (int i = ty; i < q.nslab; i += q.lanes) s += p[(int64_t)i * q.slab_stride];
    }
    red[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && c < q.width) {
        float t = 0.f;
        for (int k = 0; k < q.lanes; ++k) t += red[k][tx];
        q.o[w][c] = t;
    }
}

// ------------------------------------------------------------------------------------------------
// column sums of a [rows, width] matrix (bf16 or fp32, optional row map) -> partial[slab][width]
// workgroup = 4 waves over one 512-column chunk; lane owns 8 consecutive columns
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ src, int64_t ld, RowMap map, int64_t rows, int width,
                                                      int rows_per_slab, float* __restrict__ partial) {
    __shared__ float red[4][512];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c0 = blockIdx.x * 512 + lane * 8;
    const int slab = blockIdx.y;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const int64_t r_begin = (int64_t)slab * rows_per_slab;
    const int64_t r_end = r_begin + rows_per_slab < rows ? r_begin + rows_per_slab : rows;
    if (c0 < width) {
        // four rows of a wave are requested before the first is added (one load in flight per wave left the kernel at 1.6-2.4 TB/s); the
        // additions keep their order, so the sums are the bits they were
        constexpr int UN = 4;
        int64_t r = r_begin + wave;
        for (; r + 4 * (UN - 1) < r_end; r += 4 * UN) {
            if constexpr (sizeof(T) == 2) {
                bf16x8 v[UN];
#pragma unroll
                for (int u = 0; u < UN; ++u) v[u] = *(const bf16x8*)(src + map(r + 4 * u) * ld + c0);
#pragma unroll
                for (int u = 0; u < UN; ++u)
#pragma unroll
                    for (int k = 0; k < 8; ++k) acc[k] += bf2f(v[u][k]);
            } else {
                f32x4 v0[UN], v1[UN];
#pragma unroll
                for (int u = 0; u < UN; ++u) {
                    const T* p = src + map(r + 4 * u) * ld + c0;
                    v0[u] = *(const f32x4*)p, v1[u] = *(const f32x4*)(p + 4);
                }
#pragma unroll
                for (int u = 0; u < UN; ++u)
#pragma unroll
                    for (int k = 0; k < 4; ++k) { acc[k] += v0[u][k]; acc[4 + k] += v1[u][k]; }
            }
        }
        for (; r < r_end; r += 4) {
            const T* p = src + map(r) * ld + c0;
            if constexpr (sizeof(T) == 2) {
                const bf16x8 v = *(const bf16x8*)p;
#pragma unroll
                for (int k = 0; k < 8; ++k) acc[k] += bf2f(v[k]);
            } else {
                const f32x4 v0 = *(const f32x4*)p, v1 = *(const f32x4*)(p + 4);
#pragma unroll
                for (int k = 0; k < 4; ++k) { acc[k] += v0[k]; acc[4 + k] += v1[k]; }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) red[wave][lane * 8 + k] = acc[k];
    __syncthreads();
    for (int c = threadIdx.x; c < 512; c += 256) {
        const int gc = blockIdx.x * 512 + c;
        if (gc < width) partial[(int64_t)slab * width + gc] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
    }
}

// zero rows map(r), r < rows, of an fp32 [*, dim] matrix and of its bf16 twin
__global__ void zero_rows_kernel(float* __restrict__ a, bf16_t* __restrict__ b, RowMap map, int64_t rows, int dim) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // over rows * dim / 4
    const int d4 = dim >> 2;
    if (idx >= rows * d4) return;
    const int64_t pr = map(idx / d4);
    const int c = (idx % d4) * 4;
    if (a) *(f32x4*)(a + pr * dim + c) = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (b) *(bf16x4*)(b + pr * dim + c) = (bf16x4){f2bf(0.f), f2bf(0.f), f2bf(0.f), f2bf(0.f)};
}

// out[j, :] = sum_b src[map(b * n + j), :]   (fp32)
__global__ void batch_sum_kernel(const float* __restrict__ src, RowMap map, int batch, int n, int dim, float* __restrict__ out) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // over n * dim / 4
    const int d4 = dim >> 2;
    if (idx >= (int64_t)n * d4) return;
    const int j = idx / d4, c = (idx % d4) * 4;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int b = 0; b < batch; ++b) {
        const f32x4 v = *(const f32x4*)(src + map((int64_t)b * n + j) * dim + c);
        s += v;
    }
    *(f32x4*)(out + (int64_t)j * dim + c) = s;
}

// dst[r, :] = bf16(src[map(r), :])
__global__ void cast_rows_kernel(const float* __restrict__ src, RowMap map, int64_t rows, int dim, bf16_t* __restrict__ dst, int64_t ldd) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // over rows * dim / 4
    const int d4 = dim >> 2;
    if (idx >= rows * d4) return;
    const int64_t r = idx / d4;
    const int c = (idx % d4) * 4;
    const f32x4 v = *(const f32x4*)(src + map(r) * dim + c);
    *(bf16x4*)(dst + r * ldd + c) = (bf16x4){f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
}

// dst[b*seq + off + j, :] = (src ? src[(b*n + j), :] : 0) + (table ? table[j, :] : 0) + (vec ? vec[:] : 0)
__global__ void assemble_rows_kernel(float* __restrict__ dst, int64_t seq, int64_t off, int batch, int n, int dim,
                                     const float* __restrict__ src, const float* __restrict__ table, const float* __restrict__ vec) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int d4 = dim >> 2;
    if (idx >= (int64_t)batch * n * d4) return;
    const int c = (idx % d4) * 4;
    const int64_t r = idx / d4;
    const int j = r % n;
    const int64_t b = r / n;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (src) v += *(const f32x4*)(src + r * dim + c);
    if (table) v += *(const f32x4*)(table + (int64_t)j * dim + c);
    if (vec) v += *(const f32x4*)(vec + c);
    *(f32x4*)(dst + (b * seq + off + j) * dim + c) = v;
}

// ------------------------------------------------------------------------------------------------
// weight packing: fp32 W[N,K] -> bf16 Wb[N, ldd] and/or bf16 WT[K, lddT] (32x32 LDS transpose)
// row_perm (optional): packed row r takes source row row_perm[r]
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pack_weight_kernel(const float* __restrict__ w, int N, int K, const int32_t* __restrict__ row_perm,
                                                           bf16_t* __restrict__ wb, int64_t ldd, bf16_t* __restrict__ wt, int64_t lddT) {
    __shared__ float tile[32][33];
    const int n0 = blockIdx.y * 32, k0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int n = n0 + ty + i * 8, k = k0 + tx;
        float v = 0.f;
        if (n < N && k < K) {
            const int sn = row_perm ? row_perm[n] : n;
            v = w[(int64_t)sn * K + k];
            if (wb) wb[(int64_t)n * ldd + k] = f2bf(v);
        }
        tile[ty + i * 8][tx] = v;
    }
    if (!wt) return;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k = k0 + ty + i * 8, n = n0 + tx;
        if (n < N && k < K) wt[(int64_t)k * lddT + n] = f2bf(tile[tx][ty + i * 8]);
    }
}

// the same for up to VT_PACK_MAX_GROUP weights in one launch (after an optimizer step every bf16 operand copy of the
// model is refreshed: ~100 matrices, launch-latency-bound one by one)
struct PackGroup {
    vtPackJob job[VT_PACK_MAX_GROUP];
    int tile_start[VT_PACK_MAX_GROUP + 1];
    int n;
};
__global__ __launch_bounds__(256) void pack_weight_group_kernel(const PackGroup g) {
    __shared__ float tile[32][33];
    int j = 0;
    while (j + 1 < g.n && (int)blockIdx.x >= g.tile_start[j + 1]) ++j;
    const vtPackJob& q = g.job[j];
    const int local = blockIdx.x - g.tile_start[j];
    const int tiles_k = (q.K + 31) / 32;
    const int n0 = (local / tiles_k) * 32, k0 = (local % tiles_k) * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    bf16_t* wb = (bf16_t*)q.wb;
    bf16_t* wt = (bf16_t*)q.wt;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int n = n0 + ty + i * 8, k = k0 + tx;
        float v = 0.f;
        if (n < q.N && k < q.K) {
            const int sn = q.row_perm ? q.row_perm[n] : n;
            v = q.w[(int64_t)sn * q.K + k];
            if (wb) wb[(int64_t)n * q.ldd + k] = f2bf(v);
        }
        tile[ty + i * 8][tx] = v;
    }
    if (!wt) return;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k = k0 + ty + i * 8, n = n0 + tx;
        if (n < q.N && k < q.K) wt[(int64_t)k * q.lddT + n] = f2bf(tile[tx][ty + i * 8]);
    }
}

}  // namespace

int vt_reduce_grouped(const vtReduceItem* items, int n, vtStream stream) {
    VT_CHECK_ARG(items && n > 0 && n <= VT_REDUCE_MAX_GROUP, "vt_reduce_grouped: 1..%d items", VT_REDUCE_MAX_GROUP);
    ReduceGroup g;
    int wmax = 0, nout = 0;
    for (int i = 0; i < n; ++i) {
        g.it[i] = items[i];
        VT_CHECK_ARG(items[i].partial && items[i].nout >= 1 && items[i].nout <= 3 && (items[i].lanes == 8 || items[i].lanes == 32), "vt_reduce_grouped: bad item %d", i);
        wmax = items[i].width > wmax ? items[i].width : wmax;
        nout = items[i].nout > nout ? items[i].nout : nout;
    }
    hipLaunchKernelGGL(reduce_grouped_kernel, dim3((wmax + 31) / 32, nout, n), dim3(1024), 0, (hipStream_t)stream, g);
    VT_CHECK_LAUNCH("vt_reduce_grouped");
    return VT_OK;
}


static inline RowMap to_map(vtRowMap m) { return RowMap{m.grp, m.stride, m.off}; }

extern "C" int vt_layernorm_fwd(const float* x, vtRowMap xmap, const float* gamma, const float* beta, float eps, int64_t rows,
                                int32_t dim, void* y_bf16, float* mean, float* rstd, vtStream stream) {
    VT_CHECK_ARG(x && gamma && beta && y_bf16 && mean && rstd, "vt_layernorm_fwd: null pointer");
    VT_CHECK_ARG(rows > 0 && ((dim % 256 == 0 && dim >= 256 && dim <= 1024) || dim == 128 || dim == 384),
                 "vt_layernorm_fwd: dim=%d must be 128, 256, 384, 512, 768 or 1024", dim);
    const int grid = (int)((rows + 3) / 4 < 2048 ? (rows + 3) / 4 : 2048);
    hipStream_t s = (hipStream_t)stream;
#define LN_FWD(V, VEC) hipLaunchKernelGGL((ln_fwd_kernel<V, VEC>), dim3(grid), dim3(256), 0, s, x, to_map(xmap), gamma, beta, eps, rows, (bf16_t*)y_bf16, mean, rstd)
    switch (dim) {
        case 128: LN_FWD(1, 2); break;
        case 384: LN_FWD(3, 2); break;
        case 256: LN_FWD(1, 4); break;
        case 512: LN_FWD(2, 4); break;
        case 768: LN_FWD(3, 4); break;
        default: LN_FWD(4, 4); break;
    }
#undef LN_FWD
    VT_CHECK_LAUNCH("vt_layernorm_fwd");
    return VT_OK;
}

#define VT_LN_BWD_GRID 512
extern "C" size_t vt_layernorm_bwd_workspace_bytes(int32_t dim) { return (size_t)VT_LN_BWD_GRID * 3 * dim * sizeof(float); }

int vt_layernorm_bwd_partials(const void* dy_bf16, const float* x, vtRowMap xmap, const float* gamma, const float* mean, const float* rstd,
                              const float* dres, int64_t rows, int32_t dim, float* dx, void* dx_bf16, float* part, int* nslab, vtStream stream) {
    VT_CHECK_ARG(dy_bf16 && x && gamma && mean && rstd && dx && part, "vt_layernorm_bwd: null pointer");
    VT_CHECK_ARG(rows > 0 && ((dim % 256 == 0 && dim >= 256 && dim <= 1024) || dim == 128 || dim == 384),
                 "vt_layernorm_bwd: dim=%d must be 128, 256, 384, 512, 768 or 1024", dim);
    const int grid = (int)((rows + 3) / 4 < VT_LN_BWD_GRID ? (rows + 3) / 4 : VT_LN_BWD_GRID);
    hipStream_t s = (hipStream_t)stream;
#define LN_BWD(V, VEC) hipLaunchKernelGGL((ln_bwd_kernel<V, VEC>), dim3(grid), dim3(256), 0, s, (const bf16_t*)dy_bf16, x, to_map(xmap), gamma, mean, rstd, dres, rows, dx, (bf16_t*)dx_bf16, part)
    switch (dim) {
        case 128: LN_BWD(1, 2); break;
        case 384: LN_BWD(3, 2); break;
        case 256: LN_BWD(1, 4); break;
        case 512: LN_BWD(2, 4); break;
        case 768: LN_BWD(3, 4); break;
        default: LN_BWD(4, 4); break;
    }
#undef LN_BWD
    VT_CHECK_LAUNCH("vt_layernorm_bwd");
    *nslab = grid;
    return VT_OK;
}

extern "C" int vt_layernorm_bwd(const void* dy_bf16, const float* x, vtRowMap xmap, const float* gamma, const float* mean,
                                const float* rstd, const float* dres, int64_t rows, int32_t dim, float* dx, void* dx_bf16,
                                float* dgamma, float* dbeta, float* dxsum, void* workspace, vtStream stream) {
    VT_CHECK_ARG(dgamma && dbeta && workspace, "vt_layernorm_bwd: null pointer");
    int grid = 0;
    int rc = vt_layernorm_bwd_partials(dy_bf16, x, xmap, gamma, mean, rstd, dres, rows, dim, dx, dx_bf16, (float*)workspace, &grid, stream);
    if (rc) return rc;
    ReduceOuts ro;
    ro.o[0] = dgamma; ro.o[1] = dbeta; ro.o[2] = dxsum;
    launch_reduce_partials((const float*)workspace, grid, (int64_t)3 * dim, dim, dxsum ? 3 : 2, ro, (hipStream_t)stream);
    VT_CHECK_LAUNCH("vt_layernorm_bwd/reduce");
    return VT_OK;
}

#define VT_COLSUM_SLABS 64
extern "C" size_t vt_colsum_workspace_bytes(int32_t width) { return (size_t)VT_COLSUM_SLABS * width * sizeof(float); }

extern "C" int vt_colsum(const void* src, int32_t src_is_bf16, int64_t ld, vtRowMap map, int64_t rows, int32_t width, float* out,
                         void* workspace, vtStream stream) {
    VT_CHECK_ARG(src && out && workspace, "vt_colsum: null pointer");
    VT_CHECK_ARG(rows > 0 && width > 0 && width % 8 == 0 && ld % 8 == 0, "vt_colsum: width/ld must be multiples of 8");
    int slabs = (int)((rows + 63) / 64);
    if (slabs > VT_COLSUM_SLABS) slabs = VT_COLSUM_SLABS;
    const int rps = (int)((rows + slabs - 1) / slabs);
    const dim3 grid((width + 511) / 512, slabs);
    hipStream_t s = (hipStream_t)stream;
    if (src_is_bf16)
        hipLaunchKernelGGL(colsum_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)src, ld, to_map(map), rows, width, rps, (float*)workspace);
    else
        hipLaunchKernelGGL(colsum_kernel<float>, grid, dim3(256), 0, s, (const float*)src, ld, to_map(map), rows, width, rps, (float*)workspace);
    ReduceOuts ro;
    ro.o[0] = out; ro.o[1] = ro.o[2] = nullptr;
    launch_reduce_partials((const float*)workspace, slabs, (int64_t)width, width, 1, ro, s);
    VT_CHECK_LAUNCH("vt_colsum");
    return VT_OK;
}

extern "C" int vt_zero_rows(float* a, void* b_bf16, vtRowMap map, int64_t rows, int32_t dim, vtStream stream) {
    VT_CHECK_ARG((a || b_bf16) && rows > 0 && dim % 4 == 0, "vt_zero_rows: bad arguments");
    const int64_t total = rows * (dim / 4);
    hipLaunchKernelGGL(zero_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, (bf16_t*)b_bf16, to_map(map), rows, dim);
    VT_CHECK_LAUNCH("vt_zero_rows");
    return VT_OK;
}

// out[c] = sum_s slabs[s * slab_stride + c]  (fixed order; e.g. split-M partial weight gradients)
extern "C" int vt_sum_slabs(const float* slabs, int32_t nslab, int64_t slab_stride, int32_t width, float* out, vtStream stream) {
    VT_CHECK_ARG(slabs && out && nslab > 0 && width > 0, "vt_sum_slabs: bad arguments");
    ReduceOuts ro;
    ro.o[0] = out; ro.o[1] = ro.o[2] = nullptr;
    launch_reduce_partials(slabs, nslab, slab_stride, width, 1, ro, (hipStream_t)stream);
    VT_CHECK_LAUNCH("vt_sum_slabs");
    return VT_OK;
}

extern "C" int vt_batch_sum(const float* src, vtRowMap map, int32_t batch, int32_t n, int32_t dim, float* out, vtStream stream) {
    VT_CHECK_ARG(src && out && batch > 0 && n > 0 && dim % 4 == 0, "vt_batch_sum: bad arguments");
    const int64_t total = (int64_t)n * (dim / 4);
    hipLaunchKernelGGL(batch_sum_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, to_map(map), batch, n, dim, out);
    VT_CHECK_LAUNCH("vt_batch_sum");
    return VT_OK;
}

extern "C" int vt_cast_rows(const float* src, vtRowMap map, int64_t rows, int32_t dim, void* dst_bf16, int64_t ldd, vtStream stream) {
    VT_CHECK_ARG(src && dst_bf16 && rows > 0 && dim % 4 == 0 && ldd % 4 == 0 && ldd >= dim, "vt_cast_rows: bad arguments");
    const int64_t total = rows * (dim / 4);
    hipLaunchKernelGGL(cast_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, to_map(map), rows, dim, (bf16_t*)dst_bf16, ldd);
    VT_CHECK_LAUNCH("vt_cast_rows");
    return VT_OK;
}

extern "C" int vt_assemble_rows(float* dst, int64_t seq, int64_t off, int32_t batch, int32_t n, int32_t dim, const float* src,
                                const float* table, const float* vec, vtStream stream) {
    VT_CHECK_ARG(dst && batch > 0 && n > 0 && dim % 4 == 0 && off >= 0 && off + n <= seq, "vt_assemble_rows: bad arguments");
    const int64_t total = (int64_t)batch * n * (dim / 4);
    hipLaunchKernelGGL(assemble_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dst, seq, off, batch, n, dim, src, table, vec);
    VT_CHECK_LAUNCH("vt_assemble_rows");
    return VT_OK;
}

extern "C" int vt_pack_weight(const float* w, int32_t N, int32_t K, const int32_t* row_perm, void* wb, int64_t ldd, void* wt,
                              int64_t lddT, vtStream stream) {
    VT_CHECK_ARG(w && (wb || wt) && N > 0 && K > 0, "vt_pack_weight: bad arguments");
    VT_CHECK_ARG((!wb || ldd >= K) && (!wt || lddT >= N), "vt_pack_weight: leading dimension too small");
    const dim3 grid((K + 31) / 32, (N + 31) / 32);
    hipLaunchKernelGGL(pack_weight_kernel, grid, dim3(256), 0, (hipStream_t)stream, w, N, K, row_perm, (bf16_t*)wb, ldd, (bf16_t*)wt, lddT);
    VT_CHECK_LAUNCH("vt_pack_weight");
    return VT_OK;
}

// 64x64 tiles, 16 bytes in / 8 bytes out per thread: a row of the tile is a whole 256-B (fp32 read) or 128-B (bf16 write) line in
// both the row-major and the transposed copy; the 32x32 kernel above wrote 64-B half lines and ran at 2.3 TB/s of the 8 B/element
// it moves.  Needs K, N, ldd, lddT multiples of 4 and 16-/8-byte aligned bases (every tokenizer weight); other jobs take the kernel above.
__global__ __launch_bounds__(256) void pack_weight_group64_kernel(const PackGroup g) {
    __shared__ float tile[64][65];
    int j = 0;
    while (j + 1 < g.n && (int)blockIdx.x >= g.tile_start[j + 1]) ++j;
    const vtPackJob& q = g.job[j];
    const int local = blockIdx.x - g.tile_start[j];
    const int tiles_k = (q.K + 63) / 64;
    const int n0 = (local / tiles_k) * 64, k0 = (local % tiles_k) * 64;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;   // 16 x 16
    bf16_t* wb = (bf16_t*)q.wb;
    bf16_t* wt = (bf16_t*)q.wt;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int n = n0 + ty + i * 16, k = k0 + tx * 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (n < q.N && k < q.K) {
            const int sn = q.row_perm ? q.row_perm[n] : n;
            v = *(const f32x4*)(q.w + (int64_t)sn * q.K + k);
            if (wb) *(bf16x4*)(wb + (int64_t)n * q.ldd + k) = (bf16x4){f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) tile[ty + i * 16][tx * 4 + r] = v[r];
    }
    if (!wt) return;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k = k0 + ty + i * 16, n = n0 + tx * 4;
        if (n < q.N && k < q.K)
            *(bf16x4*)(wt + (int64_t)k * q.lddT + n) = (bf16x4){f2bf(tile[tx * 4][ty + i * 16]), f2bf(tile[tx * 4 + 1][ty + i * 16]),
                                                               f2bf(tile[tx * 4 + 2][ty + i * 16]), f2bf(tile[tx * 4 + 3][ty + i * 16])};
    }
}

extern "C" int vt_pack_weights_grouped(const vtPackJob* jobs, int32_t n, vtStream stream) {
    VT_CHECK_ARG(jobs && n > 0, "vt_pack_weights_grouped: bad arguments");
    for (int base = 0; base < n; base += VT_PACK_MAX_GROUP) {
        PackGroup g;
        g.n = n - base < VT_PACK_MAX_GROUP ? n - base : VT_PACK_MAX_GROUP;
        bool wide = true;
        for (int i = 0; i < g.n; ++i) {
            const vtPackJob& q = jobs[base + i];
            VT_CHECK_ARG(q.w && (q.wb || q.wt) && q.N > 0 && q.K > 0, "vt_pack_weights_grouped[%d]: bad arguments", base + i);
            VT_CHECK_ARG((!q.wb || q.ldd >= q.K) && (!q.wt || q.lddT >= q.N), "vt_pack_weights_grouped[%d]: leading dimension too small", base + i);
            wide = wide && q.K % 4 == 0 && q.N % 4 == 0 && q.ldd % 4 == 0 && q.lddT % 4 == 0 && ((uintptr_t)q.w & 15) == 0 &&
                   ((uintptr_t)q.wb & 7) == 0 && ((uintptr_t)q.wt & 7) == 0;
        }
        const int edge = wide ? 64 : 32;
        g.tile_start[0] = 0;
        for (int i = 0; i < g.n; ++i) {
            const vtPackJob& q = jobs[base + i];
            g.job[i] = q;
            g.tile_start[i + 1] = g.tile_start[i] + ((q.N + edge - 1) / edge) * ((q.K + edge - 1) / edge);
        }
        if (wide) hipLaunchKernelGGL(pack_weight_group64_kernel, dim3(g.tile_start[g.n]), dim3(256), 0, (hipStream_t)stream, g);
        else hipLaunchKernelGGL(pack_weight_group_kernel, dim3(g.tile_start[g.n]), dim3(256), 0, (hipStream_t)stream, g);
    }
    VT_CHECK_LAUNCH("vt_pack_weights_grouped");
    return VT_OK;
}
