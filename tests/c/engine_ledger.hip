// TEST INFRASTRUCTURE (never part of libvt_hip.so): a recording stand-in for everything vt_engine.hip calls.
//
// tests/ledger.py links the PRODUCT objects vt_engine.o + vt_api.o against this file instead of the kernel library and the HIP
// runtime (-Wl,-Bsymbolic, so the engine's calls bind here).  The engine's host code then runs unchanged on a box without a
// GPU: every kernel launch, copy, event record and stream wait it issues lands in a log with the stream it was enqueued on and the
// byte ranges it reads and writes.  tests/test_engine_ledger_cpu.py replays that log through a vector-clock happens-before
// analysis: two accesses to overlapping bytes, at least one a write, on different streams and not ordered by an event chain are a
// race -- the check the round-4 verdict asked for ("a host-side write ledger over the flat gradient buffer"), extended from the
// gradient buffer to every buffer the schedule touches.
//
// Footprints are bounding ranges taken from the documented contracts in include/vt_hip.h (row maps: first mapped row .. last
// mapped row); no pointer is ever dereferenced, so the "device" addresses may be any integers.
#include <hip/hip_runtime.h>

#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../video-tokenizer_amd/csrc/vt_common.h"

namespace {

struct Range { uintptr_t lo, hi; bool write; };
struct Op {
    char kind;            // 'K' launch / copy, 'R' event record, 'W' stream waits for event
    std::string name;
    uintptr_t stream;
    uintptr_t event;
    std::vector<Range> ranges;
};
// function-local statics: the engine object's module constructor registers its kernels before this file's globals would exist
std::vector<Op>& log_() { static auto* v = new std::vector<Op>; return *v; }
std::map<const void*, std::string>& names_() { static auto* m = new std::map<const void*, std::string>; return *m; }
uintptr_t g_next_event = 1;
struct CallCfg { dim3 grid, block; size_t shmem; hipStream_t stream; };
std::vector<CallCfg>& cfg_() { static auto* v = new std::vector<CallCfg>; return *v; }
#define g_log log_()
#define g_kernel_names names_()
#define g_cfg cfg_()

struct Rec {
    Op op;
    Rec(const char* name, vtStream s) { op.kind = 'K'; op.name = name; op.stream = (uintptr_t)s; op.event = 0; }
    void add(const void* p, size_t first_byte, size_t end_byte, bool w) {
        if (p && end_byte > first_byte) op.ranges.push_back({(uintptr_t)p + first_byte, (uintptr_t)p + end_byte, w});
    }
    void r(const void* p, size_t bytes) { add(p, 0, bytes, false); }
    void w(const void* p, size_t bytes) { add(p, 0, bytes, true); }
    // rows x cols elements of `es` bytes, leading dimension ld (elements)
    void mat(const void* p, int64_t rows, int64_t cols, int64_t ld, int es, bool wr) {
        if (p && rows > 0 && cols > 0) add(p, 0, (size_t)(((rows - 1) * ld + cols) * es), wr);
    }
    // the same through a row map (vt_hip.h: row r at (r / grp) * stride + off + r % grp)
    void mapped(const void* p, vtRowMap m, int64_t rows, int64_t cols, int64_t ld, int es, bool wr) {
        if (!p || rows <= 0 || cols <= 0) return;
        auto phys = [&](int64_t r) { return m.grp ? (r / m.grp) * m.stride + m.off + (r % m.grp) : r; };
        add(p, (size_t)(phys(0) * ld * es), (size_t)((phys(rows - 1) * ld + cols) * es), wr);
    }
    ~Rec() { g_log.push_back(op); }
};

}  // namespace

// ---- what the test reads ---------------------------------------------------------------------------------------------------
extern "C" void vt_ledger_reset() { g_log.clear(); }
extern "C" int64_t vt_ledger_size() { return (int64_t)g_log.size(); }
// one op as text: kind \t name \t stream \t event \t lo:hi:w,lo:hi:w,...
extern "C" int64_t vt_ledger_get(int64_t i, char* buf, int64_t n) {
    if (i < 0 || i >= (int64_t)g_log.size()) return -1;
    const Op& o = g_log[i];
    std::string s;
    s += o.kind; s += '\t'; s += o.name; s += '\t'; s += std::to_string(o.stream); s += '\t'; s += std::to_string(o.event); s += '\t';
    for (size_t k = 0; k < o.ranges.size(); ++k) {
        if (k) s += ',';
        s += std::to_string(o.ranges[k].lo) + ":" + std::to_string(o.ranges[k].hi) + ":" + (o.ranges[k].write ? "1" : "0");
    }
    if ((int64_t)s.size() + 1 > n) return -2 - (int64_t)s.size();
    memcpy(buf, s.c_str(), s.size() + 1);
    return (int64_t)s.size();
}
// the Python side of the step (the gradient reducer's streams, the optimizer ...) adds its own operations to the same log
extern "C" void vt_ledger_note_access(const char* name, uintptr_t stream, uintptr_t lo, uintptr_t hi, int write) {
    Op o; o.kind = 'K'; o.name = name; o.stream = stream; o.event = 0; o.ranges.push_back({lo, hi, write != 0});
    g_log.push_back(o);
}
extern "C" uintptr_t vt_ledger_note_record(uintptr_t stream) {
    Op o; o.kind = 'R'; o.name = "py_event"; o.stream = stream; o.event = (1ull << 40) + g_next_event++;
    g_log.push_back(o);
    return o.event;
}
extern "C" void vt_ledger_note_wait(uintptr_t stream, uintptr_t event) {
    Op o; o.kind = 'W'; o.name = "py_wait"; o.stream = stream; o.event = event;
    g_log.push_back(o);
}

// ---- HIP runtime entry points the engine uses ----------------------------------------------------------------------------------
extern "C" {
void** __hipRegisterFatBinary(const void*) { static void* h; return &h; }
void __hipUnregisterFatBinary(void**) {}
void __hipRegisterFunction(void**, const void* host_fn, char*, const char* device_name, unsigned, void*, void*, void*, void*, int*) {
    g_kernel_names[host_fn] = device_name ? device_name : "?";
}
void __hipRegisterVar(void**, void*, char*, char*, int, size_t, int, int) {}
hipError_t __hipPushCallConfiguration(dim3 grid, dim3 block, size_t shmem, hipStream_t stream) {
    g_cfg.push_back({grid, block, shmem, stream});
    return hipSuccess;
}
hipError_t __hipPopCallConfiguration(dim3* grid, dim3* block, size_t* shmem, hipStream_t* stream) {
    CallCfg c = g_cfg.back();
    g_cfg.pop_back();
    *grid = c.grid; *block = c.block; *shmem = c.shmem; *stream = c.stream;
    return hipSuccess;
}
hipError_t hipLaunchKernel(const void* fn, dim3, dim3, void** args, size_t, hipStream_t stream) {
    const std::string name = g_kernel_names.count(fn) ? g_kernel_names[fn] : "unknown_kernel";
    Rec rec(name.c_str(), (vtStream)stream);
    auto P = [&](int i) { return *(void**)args[i]; };
    auto I = [&](int i) { return (int64_t) * (int*)args[i]; };
    auto L = [&](int i) { return *(int64_t*)args[i]; };
    if (name.find("gather_f32") != std::string::npos) {            // (src, perm, n, dst): dst[i] = src[perm[i]]
        rec.r(P(0), I(2) * 4); rec.r(P(1), I(2) * 4); rec.w(P(3), I(2) * 4);
    } else if (name.find("scatter_f32") != std::string::npos) {    // (src, perm, n, dst): dst[perm[i]] = src[i]
        rec.r(P(0), I(2) * 4); rec.r(P(1), I(2) * 4); rec.w(P(3), I(2) * 4);
    } else if (name.find("rownorm_mean") != std::string::npos) {   // (x, seq, r0, r1, batch, dim, out)
        rec.r(P(0), (size_t)(I(4) * L(1) * I(5) * 4)); rec.w(P(6), 8);
    } else if (name.find("compact_cols") != std::string::npos) {   // (src, ld, rows, d, dst)
        rec.r(P(0), (size_t)(I(2) * L(1) * 4)); rec.w(P(4), (size_t)(I(2) * I(3) * 4));
    } else {
        rec.op.name = "UNMODELLED:" + name;                          // the test fails on these: every launch must have a footprint
    }
    return hipSuccess;
}
hipError_t hipGetLastError() { return hipSuccess; }
const char* hipGetErrorString(hipError_t) { return "stub"; }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = (hipEvent_t)(g_next_event++); return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s) {
    Op o; o.kind = 'R'; o.name = "hipEventRecord"; o.stream = (uintptr_t)s; o.event = (uintptr_t)e;
    g_log.push_back(o);
    return hipSuccess;
}
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned) {
    Op o; o.kind = 'W'; o.name = "hipStreamWaitEvent"; o.stream = (uintptr_t)s; o.event = (uintptr_t)e;
    g_log.push_back(o);
    return hipSuccess;
}
hipError_t hipMemcpyAsync(void* dst, const void* src, size_t bytes, hipMemcpyKind kind, hipStream_t s) {
    Rec rec("hipMemcpyAsync", (vtStream)s);
    if (kind != hipMemcpyHostToDevice) rec.r(src, bytes);
    rec.w(dst, bytes);
    return hipSuccess;
}
hipError_t hipMemsetAsync(void* dst, int, size_t bytes, hipStream_t s) {
    Rec rec("hipMemsetAsync", (vtStream)s);
    rec.w(dst, bytes);
    return hipSuccess;
}
}  // extern "C"

// ---- the kernel library, by the contracts of include/vt_hip.h ---------------------------------------------------------------
extern "C" size_t vt_gemm_nt_splitk_workspace_bytes(void) { return 4096 + (size_t)512 * 128 * 128 * 4; }
extern "C" size_t vt_layernorm_bwd_workspace_bytes(int32_t dim) { return (size_t)512 * 3 * dim * 4; }
extern "C" size_t vt_colsum_workspace_bytes(int32_t width) { return (size_t)256 * width * 4; }
extern "C" size_t vt_vq_workspace_bytes(int32_t N, int32_t K, int32_t d) { return (size_t)N * 64 * 4 + (size_t)K * d * 4 * 20 + 65536; }

extern "C" int vt_gemm_nt(const vtGemmNT* p, vtStream s) {
    Rec rec(p->epi == VT_EPI_F32 ? "gemm_nt<f32>" : p->epi == VT_EPI_BF16_DGELU ? "gemm_nt<dgelu>" : p->epi == VT_EPI_BF16_GELU ? "gemm_nt<gelu>" : "gemm_nt<bf16>", s);
    rec.mat(p->A, p->M, p->K, p->lda, 2, false);
    rec.mat(p->B, p->N, p->K, p->ldb, 2, false);
    rec.r(p->bias, (size_t)p->N * 4);
    rec.mat(p->rowmod, p->rowmod_period, p->N, p->N, 4, false);
    rec.mat(p->aux, p->M, p->N, p->ldaux, 2, false);
    if (p->epi == VT_EPI_F32) {
        rec.mapped(p->residual, p->omap, p->M, p->N, p->ldr, 4, false);
        rec.mapped(p->out, p->omap, p->M, p->N, p->ldo, 4, true);
        rec.mapped(p->out2, p->omap, p->M, p->N, p->ldo2, 2, true);
    } else {
        rec.mat(p->out, p->M, p->N, p->ldo, 2, true);
        rec.mat(p->out2, p->M, p->N, p->ldo2, 2, true);
    }
    rec.mat(p->colsum_partial, (p->M + 191) / 192, p->N, p->N, 4, true);
    if (p->splitk_ws) { rec.r(p->splitk_ws, (size_t)p->splitk_ws_bytes); rec.w(p->splitk_ws, (size_t)p->splitk_ws_bytes); }
    return VT_OK;
}

extern "C" int vt_gemm_tn_grouped(const vtGemmTN* pr, int32_t n, vtStream s) {
    Rec rec("gemm_tn_grouped", s);
    for (int i = 0; i < n; ++i) {
        const vtGemmTN& p = pr[i];
        rec.mat(p.A, p.M, p.P, p.lda, 2, false);
        rec.mat(p.B, p.M, p.Q, p.ldb, 2, false);
        rec.r(p.row_perm, (size_t)p.p_lim * 4);
        rec.mat(p.out, p.row_perm ? p.P : p.p_lim, p.q_lim, p.ldo, 4, true);
    }
    return VT_OK;
}

int vt_reduce_grouped(const vtReduceItem* it, int n, vtStream s) {
    Rec rec("reduce_grouped", s);
    for (int i = 0; i < n; ++i) {
        rec.r(it[i].partial, (size_t)(((int64_t)(it[i].nslab - 1) * it[i].slab_stride + (int64_t)it[i].nout * it[i].width) * 4));
        for (int w = 0; w < it[i].nout; ++w) rec.w(it[i].o[w], (size_t)it[i].width * 4);
    }
    return VT_OK;
}

extern "C" int vt_layernorm_fwd(const float* x, vtRowMap xmap, const float* gamma, const float* beta, float, int64_t rows, int32_t dim, void* y,
                                float* mean, float* rstd, vtStream s) {
    Rec rec("layernorm_fwd", s);
    rec.mapped(x, xmap, rows, dim, dim, 4, false);
    rec.r(gamma, (size_t)dim * 4); rec.r(beta, (size_t)dim * 4);
    rec.mat(y, rows, dim, dim, 2, true);
    rec.w(mean, (size_t)rows * 4); rec.w(rstd, (size_t)rows * 4);
    return VT_OK;
}

static void ln_bwd_common(Rec& rec, const void* dy, const float* x, vtRowMap xmap, const float* gamma, const float* mean, const float* rstd,
                          const float* dres, int64_t rows, int32_t dim, float* dx, void* dxb) {
    rec.mat(dy, rows, dim, dim, 2, false);
    rec.mapped(x, xmap, rows, dim, dim, 4, false);
    rec.r(gamma, (size_t)dim * 4); rec.r(mean, (size_t)rows * 4); rec.r(rstd, (size_t)rows * 4);
    rec.mapped(dres, xmap, rows, dim, dim, 4, false);
    rec.mapped(dx, xmap, rows, dim, dim, 4, true);
    rec.mapped(dxb, xmap, rows, dim, dim, 2, true);
}
extern "C" int vt_layernorm_bwd(const void* dy, const float* x, vtRowMap xmap, const float* gamma, const float* mean, const float* rstd,
                                const float* dres, int64_t rows, int32_t dim, float* dx, void* dxb, float* dgamma, float* dbeta, float* dxsum,
                                void* ws, vtStream s) {
    Rec rec("layernorm_bwd", s);
    ln_bwd_common(rec, dy, x, xmap, gamma, mean, rstd, dres, rows, dim, dx, dxb);
    rec.w(dgamma, (size_t)dim * 4); rec.w(dbeta, (size_t)dim * 4); rec.w(dxsum, (size_t)dim * 4);
    rec.r(ws, vt_layernorm_bwd_workspace_bytes(dim)); rec.w(ws, vt_layernorm_bwd_workspace_bytes(dim));
    return VT_OK;
}
int vt_layernorm_bwd_partials(const void* dy, const float* x, vtRowMap xmap, const float* gamma, const float* mean, const float* rstd,
                              const float* dres, int64_t rows, int32_t dim, float* dx, void* dxb, float* part, int* nslab, vtStream s) {
    Rec rec("layernorm_bwd_partials", s);
    ln_bwd_common(rec, dy, x, xmap, gamma, mean, rstd, dres, rows, dim, dx, dxb);
    *nslab = 256;
    rec.w(part, (size_t)256 * 3 * dim * 4);
    return VT_OK;
}

extern "C" int vt_colsum(const void* src, int32_t is_bf16, int64_t ld, vtRowMap map, int64_t rows, int32_t width, float* out, void* ws, vtStream s) {
    Rec rec("colsum", s);
    rec.mapped(src, map, rows, width, ld, is_bf16 ? 2 : 4, false);
    rec.w(out, (size_t)width * 4);
    rec.r(ws, vt_colsum_workspace_bytes(width)); rec.w(ws, vt_colsum_workspace_bytes(width));
    return VT_OK;
}
extern "C" int vt_batch_sum(const float* src, vtRowMap map, int32_t batch, int32_t n, int32_t dim, float* out, vtStream s) {
    Rec rec("batch_sum", s);
    rec.mapped(src, map, (int64_t)batch * n, dim, dim, 4, false);
    rec.w(out, (size_t)n * dim * 4);
    return VT_OK;
}
extern "C" int vt_zero_rows(float* a, void* b, vtRowMap map, int64_t rows, int32_t dim, vtStream s) {
    Rec rec("zero_rows", s);
    rec.mapped(a, map, rows, dim, dim, 4, true);
    rec.mapped(b, map, rows, dim, dim, 2, true);
    return VT_OK;
}
extern "C" int vt_sum_slabs(const float* slabs, int32_t nslab, int64_t stride, int32_t width, float* out, vtStream s) {
    Rec rec("sum_slabs", s);
    rec.r(slabs, (size_t)(((int64_t)(nslab - 1) * stride + width) * 4));
    rec.w(out, (size_t)width * 4);
    return VT_OK;
}
extern "C" int vt_cast_rows(const float* src, vtRowMap map, int64_t rows, int32_t dim, void* dst, int64_t ldd, vtStream s) {
    Rec rec("cast_rows", s);
    rec.mapped(src, map, rows, dim, dim, 4, false);
    rec.mat(dst, rows, dim, ldd, 2, true);
    return VT_OK;
}
extern "C" int vt_assemble_rows(float* dst, int64_t seq, int64_t off, int32_t batch, int32_t n, int32_t dim, const float* src, const float* table,
                                const float* vec, vtStream s) {
    Rec rec("assemble_rows", s);
    rec.add(dst, (size_t)(off * dim * 4), (size_t)((((int64_t)batch - 1) * seq + off + n) * dim * 4), true);
    rec.r(src, (size_t)batch * n * dim * 4); rec.r(table, (size_t)n * dim * 4); rec.r(vec, (size_t)dim * 4);
    return VT_OK;
}
extern "C" int vt_pack_weights_grouped(const vtPackJob* j, int32_t n, vtStream s) {
    Rec rec("pack_weights_grouped", s);
    for (int i = 0; i < n; ++i) {
        rec.r(j[i].w, (size_t)j[i].N * j[i].K * 4);
        rec.r(j[i].row_perm, (size_t)j[i].N * 4);
        rec.mat(j[i].wb, j[i].N, j[i].K, j[i].ldd, 2, true);
        rec.mat(j[i].wt, j[i].K, j[i].N, j[i].lddT, 2, true);
    }
    return VT_OK;
}
extern "C" int vt_patchify(const float* video, int32_t B, int32_t C, int32_t T, int32_t S, int32_t pt, int32_t p, void* rows, vtStream s) {
    Rec rec("patchify", s);
    const size_t px = (size_t)B * C * T * S * S;
    rec.r(video, px * 4); rec.w(rows, px * 2);
    (void)pt; (void)p;
    return VT_OK;
}
extern "C" int vt_unpatchify(const float* rows, int32_t B, int32_t C, int32_t T, int32_t S, int32_t, int32_t, float* video, vtStream s) {
    Rec rec("unpatchify", s);
    const size_t px = (size_t)B * C * T * S * S;
    rec.r(rows, px * 4); rec.w(video, px * 4);
    return VT_OK;
}
extern "C" int vt_attention_fwd_rows(const void* qkv, int32_t B, int32_t L, int32_t H, int32_t hd, int32_t q_begin, void* o, float* lse, vtStream s) {
    Rec rec("attention_fwd", s);
    rec.r(qkv, (size_t)B * L * 3 * H * hd * 2);
    rec.w(o, (size_t)B * (L - q_begin) * H * hd * 2);
    rec.w(lse, (size_t)B * H * L * 4);
    return VT_OK;
}
extern "C" int vt_attention_fwd(const void* qkv, int32_t B, int32_t L, int32_t H, int32_t hd, void* o, float* lse, vtStream s) {
    return vt_attention_fwd_rows(qkv, B, L, H, hd, 0, o, lse, s);
}
extern "C" int vt_attention_bwd_rows(const void* qkv, const void* o, const void* dO, const float* lse, int32_t B, int32_t L, int32_t H, int32_t hd,
                                     int32_t q_begin, void* dqkv, float* delta, vtStream s) {
    Rec rec("attention_bwd", s);
    rec.r(qkv, (size_t)B * L * 3 * H * hd * 2);
    rec.r(o, (size_t)B * (L - q_begin) * H * hd * 2); rec.r(dO, (size_t)B * (L - q_begin) * H * hd * 2);
    rec.r(lse, (size_t)B * H * L * 4);
    rec.w(dqkv, (size_t)B * L * 3 * H * hd * 2);
    rec.r(delta, (size_t)B * H * L * 4); rec.w(delta, (size_t)B * H * L * 4);
    return VT_OK;
}
extern "C" int vt_vq_forward_ctr(const float* z, int64_t ldz, const float* cb, int32_t N, int32_t K, int32_t d, int32_t, int32_t, float, float, float,
                                 uint64_t, const uint32_t* ctr, float* E, float* wnorm, float* zn, float* znorm, int64_t* idx, float* rz, void* rzp,
                                 int64_t ldp, float* losses, void* ws, vtStream s) {
    Rec rec("vq_forward", s);
    rec.mat(z, N, d, ldz, 4, false); rec.r(cb, (size_t)K * d * 4); rec.r(ctr, 4);
    rec.w(E, (size_t)K * d * 4); rec.w(wnorm, (size_t)K * 4); rec.w(zn, (size_t)N * d * 4); rec.w(znorm, (size_t)N * 4);
    rec.w(idx, (size_t)N * 8); rec.w(rz, (size_t)N * d * 4); rec.mat(rzp, N, d, ldp, 2, true); rec.w(losses, 16);
    rec.r(ws, vt_vq_workspace_bytes(N, K, d)); rec.w(ws, vt_vq_workspace_bytes(N, K, d));
    return VT_OK;
}
extern "C" int vt_vq_backward(const float* g_rz, int64_t ldg, const float* gscal, float, float, const float* zn, const float* znorm, const float* E,
                              const float* wnorm, const int64_t* idx, int32_t N, int32_t K, int32_t d, int32_t, float* dz, void* dzp, int64_t ldp,
                              float* dW, void* ws, vtStream s) {
    Rec rec("vq_backward", s);
    rec.mat(g_rz, N, d, ldg, 4, false); rec.r(gscal, 12);
    rec.r(zn, (size_t)N * d * 4); rec.r(znorm, (size_t)N * 4); rec.r(E, (size_t)K * d * 4); rec.r(wnorm, (size_t)K * 4); rec.r(idx, (size_t)N * 8);
    rec.w(dz, (size_t)N * d * 4); rec.mat(dzp, N, d, ldp, 2, true); rec.w(dW, (size_t)K * d * 4);
    rec.r(ws, vt_vq_workspace_bytes(N, K, d)); rec.w(ws, vt_vq_workspace_bytes(N, K, d));
    return VT_OK;
}
extern "C" int vt_vq_prep_codebook(const float* cb, int32_t K, int32_t d, int32_t, float* E, float* wnorm, void* ws, vtStream s) {
    Rec rec("vq_prep_codebook", s);
    rec.r(cb, (size_t)K * d * 4); rec.w(E, (size_t)K * d * 4); rec.w(wnorm, (size_t)K * 4); rec.w(ws, 65536);
    return VT_OK;
}
extern "C" int vt_vq_gather(const float* E, const int64_t* idx, int32_t N, int32_t K, int32_t d, float* out, void* outp, int64_t ldp, vtStream s) {
    Rec rec("vq_gather", s);
    rec.r(E, (size_t)K * d * 4); rec.r(idx, (size_t)N * 8); rec.w(out, (size_t)N * d * 4); rec.mat(outp, N, d, ldp, 2, true);
    return VT_OK;
}
