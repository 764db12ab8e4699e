// Whole-model engine for the LARP tokenizer step on gfx950: plans one workspace, then enqueues the
// complete forward (encode -> VQ -> decode) or a range of backward stages on a stream by calling the
// kernels of this library back to back.  Host code only does pointer arithmetic: no allocation, no
// synchronisation, no per-op Python -- the sequence is hipGraph-capturable.
//
// Follows /root/reference/models/larp_tokenizer.py:400-428 (encode), :456-469 (decode), :489-496
// (forward); models/transformer.py:62-70 (cat -> blocks -> last len(query) rows); timm Block
// (pre-LN attention + MLP, see oracle/larp_oracle.py block()); models/bottleneck.py:170-188.
//
// HBM layout (all row-major, rows padded to a multiple of 128 and zero-initialised once so that the
// TN weight-gradient GEMM can contract over padded rows):
//   residual stream x      fp32 [B*L, D]   one buffer per block boundary (saved for LayerNorm backward)
//   MFMA operands          bf16            LayerNorm outputs, qkv, attention out, fc1 pre/post GELU
//   weights                bf16 [N,K] and [K,N] packed copies of the fp32 masters (vt_tokenizer_pack)
// With 288 GB of HBM3E nothing is recomputed except the attention probabilities.
#include <vector>

#include "vt_common.h"

static const bool g_no_splitk = [] { const char* e = getenv("VT_GEMM_SPLITK"); return e && strcmp(e, "0") == 0; }();   // A/B switch, read once

namespace {

__global__ void gather_f32_kernel(const float* __restrict__ src, const int32_t* __restrict__ perm, int n, float* __restrict__ dst) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[perm[i]];
}
// dst[perm[i]] = src[i]
__global__ void scatter_f32_kernel(const float* __restrict__ src, const int32_t* __restrict__ perm, int n, float* __restrict__ dst) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[perm[i]] = src[i];
}

// out[0] = mean_b |x[b*seq + r0, :]|, out[1] = mean_b |x[b*seq + r1, :]|   (bottleneck.py:171-172)
// One workgroup per statistic, one WAVE per clip (the serial loop over the batch took 27 us of dependent loads for 48 KB of input);
// per row the same lane-strided partial sums and wave reduction as before, and the clips are added in index order: same bits.
__global__ __launch_bounds__(512) void rownorm_mean_kernel(const float* __restrict__ x, int64_t seq, int64_t r0, int64_t r1, int batch, int dim, float* __restrict__ out) {
    __shared__ float norms[8];
    const int which = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t r = which == 0 ? r0 : r1;
    float tot = 0.f;
    for (int b0 = 0; b0 < batch; b0 += 8) {
        const int b = b0 + wave;
        if (b < batch) {
            const float* p = x + ((int64_t)b * seq + r) * dim;
            float s = 0.f;
            for (int c = lane; c < dim; c += 64) s += p[c] * p[c];
            s = wave_sum(s);
            if (lane == 0) norms[wave] = sqrtf(s);
        }
        __syncthreads();
        if (threadIdx.x == 0)
            for (int k = 0; k < 8 && b0 + k < batch; ++k) tot += norms[k];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[which] = tot / (float)batch;
}

// dst[r, 0..d) = float(bf16(src[r, 0..d)))   compact copy of the in_linear output ('projected_z')
__global__ void compact_cols_kernel(const float* __restrict__ src, int64_t lds_, int rows, int d, float* __restrict__ dst) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < rows * d) dst[i] = src[(int64_t)(i / d) * lds_ + (i % d)];
}

struct Arena {
    size_t off = 0;
    size_t take(size_t bytes) {
        const size_t o = off;
        off += (bytes + 255) & ~(size_t)255;
        return o;
    }
};

struct BlockBufs {
    // packed weights
    size_t qkv_wb, qkv_wt, proj_wb, proj_wt, fc1_wb, fc1_wt, fc2_wb, fc2_wt;
    // saved activations
    size_t x_mid, h1, mean1, rstd1, qkv, lse, o, h2, mean2, rstd2, u, g;
};

inline int round_up(int a, int b) { return (a + b - 1) / b * b; }

}  // namespace

struct vtTokenizer {
    vtTokenizerConfig c;
    int Nv, L, M, Mp, Mv, Mvp, Mq, Mqp, Kp, D3, D4;
    size_t ws_bytes;
    std::vector<int32_t> perm_host;  // packed head row j  <-  reference row perm[j]
    // ---- workspace offsets
    size_t perm, head_b_perm, dec_query_sum;
    size_t pe_wb, in_wb, in_wt, out_wb, out_wt, head_wb, head_wt;
    std::vector<BlockBufs> enc, dec;
    std::vector<size_t> x_enc, x_dec;  // residual stream at block boundaries (depth+1 each)
    size_t patches, zb, zproj, vq_E, vq_wnorm, vq_zn, vq_znorm, vq_idx, vq_rz, vq_rzpad, vq_losses, vq_ws, encoded_int;
    size_t hN, meanH, rstdH, yrows;
    // backward scratch
    size_t dX, dh, dob, delta, ln_ws, cs_ws, cs_part, dY, dhN, dEncb, d_rz, dz_pad, dTok, tmp_vec, wg_slabs;
    const uint32_t* seed_ctr = nullptr;   // device-side per-call counter of the stochastic VQ (graph replay), see vt_vq_forward_ctr
    bool splitk_on = !g_no_splitk;           // vt_tokenizer_set_split_k / vt_stack_set_split_k; VT_GEMM_SPLITK=0 starts it off
    int wg_batch = 4;                        // vt_tokenizer_set_wgrad_batch: blocks per grouped weight-gradient launch (1..WG_BATCH)
    int wg_tail = 0;                         // vt_tokenizer_set_wgrad_tail: the encoder's first wg_tail blocks (the LAST of the backward) flush their weight gradients block by block
    bool data_parallel = false;              // vt_tokenizer_set_data_parallel: a collective runs next to the backward (see nt())
    bool in_backward = false;                // set by the entry points: nt() hands the split-K workspace to backward GEMMs only
    size_t splitk = 0, splitk_bytes = 0;     // vt_gemm_nt's split-K partial sums + arrival counters (zeroed by *_init_workspace)
    // bf16 gradient operands that a block's weight-gradient GEMMs read.  The wgrads of WG_BATCH consecutive blocks are
    // deferred into one grouped launch, so these rotate over WG_BATCH + 1 sets (the set a block writes its dx_in to
    // is the next block's dx_out set).
    // The LAST block of a stack only has to produce the rows the stack returns (transformer.py:69: h[:, -len(query):]): the
    // last nk rows of every sequence.  Its MLP half, the attention queries and the matching backward work run on those rows
    // only (compact buffers); K/V, the qkv GEMM and LayerNorm1 still cover all rows.  Enabled per stack when the first kept
    // row (L - nk) is a multiple of 64.
    struct LastBlock { int enabled, nk, q_begin, Mk, Mkp; size_t dxa, dxm, du; };
    LastBlock last_enc, last_dec;
    struct GradSet { size_t dx_out, dx_mid, du, dqkv, ln_part1, ln_part2, cs_part; };   // + the block's partial sums awaiting the grouped reduction
    static constexpr int WG_BATCH = 4;
    static constexpr int NSETS_MAX = 2 * WG_BATCH + 1;
    GradSet gs[NSETS_MAX];    // WG_BATCH + 1 are in rotation; all of them when the weight gradients run on their own stream (below)
    // Data-parallel runs: the deferred weight-gradient launches (and the partial-sum reductions of the same blocks) go to a SECOND stream
    // (vt_tokenizer_set_wgrad_stream).  They are off the backward's critical path, and with a collective's workgroups resident the
    // exact-fit GEMM launches of the main stream leave most of the chip idle in their extra round: work of an
    // independent stream fills it.  Ordering: the side stream starts a group behind an event of the main stream (operands written);
    // a gradient set is rewritten by the main stream only behind the event of the group that read it (twice the sets, so that wait is
    // normally over); the last stage of backward joins the side stream.
    hipStream_t wg_stream = nullptr;
    static constexpr int NEV = 16;
    hipEvent_t ev_fork[NEV] = {}, ev_done[NEV] = {};
    int flush_id = 0;                 // groups flushed to the side stream so far (event slot = id % NEV)
    int set_flush[NSETS_MAX];         // id of the side-stream group that last read the set, -1 = none pending
    std::vector<int> sets_pending;    // sets used by the blocks whose weight gradients are queued
    int nsets() const { return wg_stream ? NSETS_MAX : WG_BATCH + 1; }
    ~vtTokenizer() {
        for (int i = 0; i < NEV; ++i) {
            if (ev_fork[i]) (void)hipEventDestroy(ev_fork[i]);
            if (ev_done[i]) (void)hipEventDestroy(ev_done[i]);
        }
    }
    // host-side state of an in-flight backward
    std::vector<vtGemmTN> pending;
    std::vector<vtReduceItem> pending_red;   // partial-sum reductions of the same blocks (one grouped launch at the flush)
    int pending_blocks = 0;   // blocks whose wgrads sit in `pending`
    int set_idx = 0;          // gradient set of the block processed next
    int final_through = 0;    // stages [0, final_through) have complete gradients
    int pending_first_stage = 0;
};

#define WS(T, off) ((T*)((char*)ws + (off)))

static void plan_blocks(vtTokenizer* t, Arena& a, std::vector<BlockBufs>& v, int depth) {
    const size_t Mp = t->Mp, D = t->c.D, D3 = t->D3, D4 = t->D4;
    v.resize(depth);
    for (int i = 0; i < depth; ++i) {
        BlockBufs& b = v[i];
        b.qkv_wb = a.take(D3 * D * 2); b.qkv_wt = a.take(D * D3 * 2);
        b.proj_wb = a.take(D * D * 2); b.proj_wt = a.take(D * D * 2);
        b.fc1_wb = a.take(D4 * D * 2); b.fc1_wt = a.take(D * D4 * 2);
        b.fc2_wb = a.take(D * D4 * 2); b.fc2_wt = a.take(D4 * D * 2);
        b.x_mid = a.take(Mp * D * 4);
        b.h1 = a.take(Mp * D * 2); b.mean1 = a.take(Mp * 4); b.rstd1 = a.take(Mp * 4);
        b.qkv = a.take(Mp * D3 * 2); b.lse = a.take((size_t)t->c.B * t->c.H * t->L * 4);
        b.o = a.take(Mp * D * 2);
        b.h2 = a.take(Mp * D * 2); b.mean2 = a.take(Mp * 4); b.rstd2 = a.take(Mp * 4);
        b.u = a.take(Mp * D4 * 2); b.g = a.take(Mp * D4 * 2);
    }
}

extern "C" int vt_tokenizer_create(const vtTokenizerConfig* cfg, vtTokenizer** out) {
    VT_CHECK_ARG(cfg && out, "vt_tokenizer_create: null pointer");
    const vtTokenizerConfig& c = *cfg;
    VT_CHECK_ARG(c.B > 0 && c.C > 0 && c.T > 0 && c.S > 0 && c.pt > 0 && c.p > 0, "vt_tokenizer_create: bad geometry");
    VT_CHECK_ARG(c.T % c.pt == 0 && c.S % c.p == 0 && c.p % 8 == 0, "vt_tokenizer_create: T%%pt, S%%p, p%%8 must be 0");
    VT_CHECK_ARG(c.D % 256 == 0 && c.D <= 1024 && c.H * 64 == c.D, "vt_tokenizer_create: D=%d H=%d unsupported (head_dim must be 64, D in 256..1024)", c.D, c.H);
    VT_CHECK_ARG(c.depth_enc > 0 && c.depth_dec > 0 && c.Nq > 0 && c.K > 0, "vt_tokenizer_create: bad depth/Nq/K");
    VT_CHECK_ARG(c.d == 8 || c.d == 16 || c.d == 24 || c.d == 32, "vt_tokenizer_create: bottleneck_dim %d unsupported (8,16,24,32)", c.d);
    VT_CHECK_ARG((c.C * c.pt * c.p * c.p) % 64 == 0, "vt_tokenizer_create: patch volume must be a multiple of 64");
    vtTokenizer* t = new vtTokenizer();
    t->c = c;
    t->Nv = (c.T / c.pt) * (c.S / c.p) * (c.S / c.p);
    t->L = t->Nv + c.Nq;
    t->M = c.B * t->L; t->Mp = round_up(t->M, 128);
    t->Mv = c.B * t->Nv; t->Mvp = round_up(t->Mv, 128);
    t->Mq = c.B * c.Nq; t->Mqp = round_up(t->Mq, 128);
    t->Kp = c.C * c.pt * c.p * c.p;
    t->D3 = 3 * c.D; t->D4 = 4 * c.D;
    // head row permutation: packed order (c,dt,dy,dx)  <-  reference order (dt,dy,dx,c)  (larp_tokenizer.py:452-453)
    t->perm_host.resize(t->Kp);
    for (int ch = 0; ch < c.C; ++ch)
        for (int dt = 0; dt < c.pt; ++dt)
            for (int dy = 0; dy < c.p; ++dy)
                for (int dx = 0; dx < c.p; ++dx) {
                    const int packed = ((ch * c.pt + dt) * c.p + dy) * c.p + dx;
                    const int ref = ((dt * c.p + dy) * c.p + dx) * c.C + ch;
                    t->perm_host[packed] = ref;
                }
    Arena a;
    const size_t D = c.D, Mp = t->Mp, Mvp = t->Mvp, Mqp = t->Mqp, Kp = t->Kp;
    t->perm = a.take(Kp * 4); t->head_b_perm = a.take(Kp * 4); t->dec_query_sum = a.take((size_t)t->Nv * D * 4);
    t->pe_wb = a.take(D * Kp * 2);
    t->in_wb = a.take((size_t)64 * D * 2); t->in_wt = a.take(D * 64 * 2);
    t->out_wb = a.take(D * 64 * 2); t->out_wt = a.take((size_t)64 * D * 2);
    t->head_wb = a.take(Kp * D * 2); t->head_wt = a.take(D * Kp * 2);
    plan_blocks(t, a, t->enc, c.depth_enc);
    plan_blocks(t, a, t->dec, c.depth_dec);
    t->x_enc.resize(c.depth_enc + 1);
    for (auto& x : t->x_enc) x = a.take(Mp * D * 4);
    t->x_dec.resize(c.depth_dec + 1);
    for (auto& x : t->x_dec) x = a.take(Mp * D * 4);
    t->patches = a.take(Mvp * Kp * 2);
    t->zb = a.take(Mqp * D * 2);
    t->zproj = a.take(Mqp * 64 * 4);
    t->vq_E = a.take((size_t)c.K * c.d * 4); t->vq_wnorm = a.take((size_t)c.K * 4);
    t->vq_zn = a.take(Mqp * c.d * 4); t->vq_znorm = a.take(Mqp * 4); t->vq_idx = a.take(Mqp * 8);
    t->vq_rz = a.take(Mqp * c.d * 4); t->vq_rzpad = a.take(Mqp * 64 * 2); t->vq_losses = a.take(64);
    t->vq_ws = a.take(vt_vq_workspace_bytes(t->Mq, c.K, c.d));
    t->encoded_int = a.take(Mqp * D * 4);
    t->hN = a.take(Mvp * D * 2); t->meanH = a.take(Mvp * 4); t->rstdH = a.take(Mvp * 4);
    t->yrows = a.take(Mvp * Kp * 4);
    t->dX = a.take(Mp * D * 4);
    for (auto& g : t->gs) {
        g.dx_out = a.take(Mp * D * 2); g.dx_mid = a.take(Mp * D * 2);
        g.du = a.take(Mp * t->D4 * 2); g.dqkv = a.take(Mp * t->D3 * 2);
        g.ln_part1 = a.take(vt_layernorm_bwd_workspace_bytes((int)D)); g.ln_part2 = a.take(vt_layernorm_bwd_workspace_bytes((int)D));
        g.cs_part = a.take((size_t)((t->M + 191) / 192) * t->D4 * 4);
    }
    t->dh = a.take(Mp * D * 2); t->dob = a.take(Mp * D * 2);
    for (int which = 0; which < 2; ++which) {
        vtTokenizer::LastBlock& lb = which == 0 ? t->last_enc : t->last_dec;
        lb.nk = which == 0 ? c.Nq : t->Nv;
        lb.q_begin = t->L - lb.nk;
        lb.Mk = c.B * lb.nk;
        lb.Mkp = round_up(lb.Mk, 128);
        lb.enabled = (lb.q_begin % 64 == 0 && lb.q_begin > 0) ? 1 : 0;
        lb.dxa = a.take((size_t)lb.Mkp * D * 2); lb.dxm = a.take((size_t)lb.Mkp * D * 2); lb.du = a.take((size_t)lb.Mkp * t->D4 * 2);
    }
    t->delta = a.take((size_t)c.B * c.H * t->L * 4);
    t->splitk_bytes = vt_gemm_nt_splitk_workspace_bytes();
    t->splitk = a.take(t->splitk_bytes);
    t->ln_ws = a.take(vt_layernorm_bwd_workspace_bytes(c.D));
    t->cs_ws = a.take(vt_colsum_workspace_bytes((int)(Kp > (size_t)t->D4 ? Kp : t->D4)));
    t->cs_part = a.take((size_t)((t->M + 191) / 192) * t->D4 * 4);  // per-M-tile column sums out of the fc2-dgrad epilogue
    t->dY = a.take(Mvp * Kp * 2); t->dhN = a.take(Mvp * D * 2);
    t->dEncb = a.take(Mqp * D * 2); t->d_rz = a.take(Mqp * 64 * 4); t->dz_pad = a.take(Mqp * 64 * 2);
    t->dTok = a.take(Mvp * D * 2);
    t->tmp_vec = a.take((Kp > (size_t)t->D4 ? Kp : t->D4) * 4);
    t->wg_slabs = a.take((size_t)16 * D * 64 * 4);  // split-M partials of the two skinny (d-wide) weight gradients
    t->ws_bytes = a.off;
    *out = t;
    return VT_OK;
}

extern "C" void vt_tokenizer_destroy(vtTokenizer* t) { delete t; }
extern "C" size_t vt_tokenizer_workspace_bytes(const vtTokenizer* t) { return t ? t->ws_bytes : 0; }
extern "C" int32_t vt_tokenizer_num_backward_stages(const vtTokenizer* t) { return t ? 3 + t->c.depth_enc + t->c.depth_dec : 0; }
extern "C" int vt_tokenizer_set_split_k(vtTokenizer* t, int32_t on) {
    VT_CHECK_ARG(t, "vt_tokenizer_set_split_k: null handle");
    t->splitk_on = on != 0;
    return VT_OK;
}
extern "C" int vt_stack_set_split_k(vtStack* t, int32_t on) { return vt_tokenizer_set_split_k(t, on); }
// Data-parallel runs: the weight-gradient launches on a stream of their own (see the members above); NULL = single-stream schedule.
extern "C" int vt_tokenizer_set_wgrad_stream(vtTokenizer* t, vtStream side) {
    VT_CHECK_ARG(t, "vt_tokenizer_set_wgrad_stream: null handle");
    VT_CHECK_ARG(t->pending.empty() && t->pending_red.empty(), "vt_tokenizer_set_wgrad_stream: a backward is in flight");
    t->wg_stream = (hipStream_t)side;
    for (int i = 0; i < vtTokenizer::NSETS_MAX; ++i) t->set_flush[i] = -1;
    t->sets_pending.clear();
    if (side) {
        for (int i = 0; i < vtTokenizer::NEV; ++i) {
            if (!t->ev_fork[i] && hipEventCreateWithFlags(&t->ev_fork[i], hipEventDisableTiming) != hipSuccess) { vt_set_error("vt_tokenizer_set_wgrad_stream: event creation failed"); return VT_ERR_LAUNCH; }
            if (!t->ev_done[i] && hipEventCreateWithFlags(&t->ev_done[i], hipEventDisableTiming) != hipSuccess) { vt_set_error("vt_tokenizer_set_wgrad_stream: event creation failed"); return VT_ERR_LAUNCH; }
        }
    }
    return VT_OK;
}
// Data-parallel runs: the gradients of a 4-block group only become final -- and their all-reduce can only start -- at the group's grouped
// weight-gradient launch, so with the default schedule the encoder's blocks 3..0 (113 MB of fp32 gradients at config B) are reduced after
// the last backward kernel, fully exposed.  With n > 0 the encoder's blocks below n are flushed block by block (n = 3: groups 3-2 | 1 | 0),
// which leaves one block's 28 MB for the tail and costs three launches that do not fill whole rounds of the chip (+ ~0.1 ms of compute).
// Same kernels on the same operands: gradients are bit-identical to the default schedule.
extern "C" int vt_tokenizer_set_wgrad_tail(vtTokenizer* t, int32_t n) {
    VT_CHECK_ARG(t && n >= 0, "vt_tokenizer_set_wgrad_tail: null handle or negative count");
    t->wg_tail = n;
    return VT_OK;
}
// Data-parallel runs: a gradient all-reduce's persistent workgroups hold CUs during the backward; nt() then launches the backward's
// GEMMs that have more 192x192 tiles than the chip has CUs one tile per workgroup (vtGemmNT.tile = 6).  Same bits.
extern "C" int vt_tokenizer_set_data_parallel(vtTokenizer* t, int32_t on) {
    VT_CHECK_ARG(t, "vt_tokenizer_set_data_parallel: null handle");
    t->data_parallel = on != 0;
    return VT_OK;
}
// blocks per grouped weight-gradient launch, 1..4 (default 4: 768 tiles = three whole rounds of the chip).  Smaller groups hand the gradient
// reducer finished slices sooner and, with the weight gradients on their own stream, keep that stream supplied with work throughout the backward.
extern "C" int vt_tokenizer_set_wgrad_batch(vtTokenizer* t, int32_t n) {
    VT_CHECK_ARG(t && n >= 1 && n <= vtTokenizer::WG_BATCH, "vt_tokenizer_set_wgrad_batch: null handle or n outside 1..%d", vtTokenizer::WG_BATCH);
    VT_CHECK_ARG(t->pending.empty(), "vt_tokenizer_set_wgrad_batch: a backward is in flight");
    t->wg_batch = n;
    return VT_OK;
}
extern "C" int vt_tokenizer_set_seed_counter(vtTokenizer* t, const uint32_t* seed_counter) {
    VT_CHECK_ARG(t, "vt_tokenizer_set_seed_counter: null handle");
    t->seed_ctr = seed_counter;
    return VT_OK;
}
extern "C" int vt_tokenizer_init_workspace(vtTokenizer* t, void* ws, vtStream stream) {
    VT_CHECK_ARG(t && ws, "vt_tokenizer_init_workspace: null pointer");
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(ws, 0, t->ws_bytes, s) != hipSuccess) { vt_set_error("vt_tokenizer_init_workspace: memset failed"); return VT_ERR_LAUNCH; }
    if (hipMemcpyAsync(WS(int32_t, t->perm), t->perm_host.data(), (size_t)t->Kp * 4, hipMemcpyHostToDevice, s) != hipSuccess) {
        vt_set_error("vt_tokenizer_init_workspace: perm upload failed");
        return VT_ERR_LAUNCH;
    }
    return VT_OK;
}

// Attention backward of one block: the two-kernel form (dQ kernel + dK/dV kernel, 7 products).  The five-product kernel with an ordered
// dQ hand-off that round 3 built was correct and bit-reproducible but slower (346 vs 244 us at the step's shape) and left the tree in
// round 4; its record: DESIGN.md section 5 "Attention backward, round 3", profiles/r03_attention_bwd_*.
static int attn_bwd(vtTokenizer* t, void* ws, const void* qkv, const void* o, const void* dO, const float* lse, int q_begin, void* dqkv, vtStream s) {
    const vtTokenizerConfig& c = t->c;
    return vt_attention_bwd_rows(qkv, o, dO, lse, c.B, t->L, c.H, c.D / c.H, q_begin, dqkv, WS(float, t->delta), s);
}

static int copy_d2d(void* dst, const void* src, size_t bytes, hipStream_t s) {
    if (hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, s) != hipSuccess) {
        vt_set_error("device-to-device copy of %zu bytes failed", bytes);
        return VT_ERR_LAUNCH;
    }
    return VT_OK;
}

#define TRY(x)                 \
    do {                       \
        int rc__ = (x);        \
        if (rc__) return rc__; \
    } while (0)

static vtPackJob pack_job(const float* w, int N, int K, const int32_t* perm, void* wb, int64_t ldd, void* wt, int64_t lddT) {
    vtPackJob q;
    q.w = w; q.N = N; q.K = K; q.row_perm = perm; q.wb = wb; q.ldd = ldd; q.wt = wt; q.lddT = lddT;
    return q;
}

static void pack_block_jobs(vtTokenizer* t, const std::vector<BlockBufs>& v, const vtBlockTensors* bl, void* ws, std::vector<vtPackJob>& jobs) {
    const int D = t->c.D, D3 = t->D3, D4 = t->D4;
    for (size_t i = 0; i < v.size(); ++i) {
        const BlockBufs& b = v[i];
        jobs.push_back(pack_job(bl[i].qkv_w, D3, D, nullptr, WS(void, b.qkv_wb), D, WS(void, b.qkv_wt), D3));
        jobs.push_back(pack_job(bl[i].proj_w, D, D, nullptr, WS(void, b.proj_wb), D, WS(void, b.proj_wt), D));
        jobs.push_back(pack_job(bl[i].fc1_w, D4, D, nullptr, WS(void, b.fc1_wb), D, WS(void, b.fc1_wt), D4));
        jobs.push_back(pack_job(bl[i].fc2_w, D, D4, nullptr, WS(void, b.fc2_wb), D4, WS(void, b.fc2_wt), D));
    }
}

static int pack_blocks(vtTokenizer* t, const std::vector<BlockBufs>& v, const vtBlockTensors* bl, void* ws, vtStream s) {
    std::vector<vtPackJob> jobs;
    pack_block_jobs(t, v, bl, ws, jobs);
    return vt_pack_weights_grouped(jobs.data(), (int)jobs.size(), s);
}

extern "C" int vt_tokenizer_pack(vtTokenizer* t, const vtTokenizerTensors* P, void* ws, vtStream s) {
    VT_CHECK_ARG(t && P && ws && P->enc_blocks && P->dec_blocks, "vt_tokenizer_pack: null pointer");
    const vtTokenizerConfig& c = t->c;
    const int D = c.D, Kp = t->Kp;
    std::vector<vtPackJob> jobs;   // every bf16 operand copy of the model: 4 grouped launches instead of ~100 single ones
    jobs.push_back(pack_job(P->pe_w, D, Kp, nullptr, WS(void, t->pe_wb), Kp, nullptr, 0));
    jobs.push_back(pack_job(P->in_w, c.d, D, nullptr, WS(void, t->in_wb), D, WS(void, t->in_wt), 64));    // [d,D] and [D,64]
    jobs.push_back(pack_job(P->out_w, D, c.d, nullptr, WS(void, t->out_wb), 64, WS(void, t->out_wt), D));  // [D,64] and [64,D]
    jobs.push_back(pack_job(P->head_w, Kp, D, WS(int32_t, t->perm), WS(void, t->head_wb), D, WS(void, t->head_wt), Kp));
    pack_block_jobs(t, t->enc, P->enc_blocks, ws, jobs);
    pack_block_jobs(t, t->dec, P->dec_blocks, ws, jobs);
    TRY(vt_pack_weights_grouped(jobs.data(), (int)jobs.size(), s));
    hipLaunchKernelGGL(gather_f32_kernel, dim3((Kp + 255) / 256), dim3(256), 0, (hipStream_t)s, P->head_b, WS(int32_t, t->perm), Kp, WS(float, t->head_b_perm));
    TRY(vt_assemble_rows(WS(float, t->dec_query_sum), t->Nv, 0, 1, t->Nv, D, nullptr, P->dec_patch_query, P->dec_token_type, s));
    VT_CHECK_LAUNCH("vt_tokenizer_pack");
    return VT_OK;
}

// the input-gradient GEMMs of the engine carry the split-K workspace: at one or two clips per GPU the N = D GEMMs are 72 / 144 tiles
// and vt_gemm_nt splits their K (vtGemmNT.splitk_ws); at the headline batch nothing qualifies and the field is ignored
static vtGemmNT nt(const vtTokenizer* t, void* ws, const void* A, int64_t lda, const void* B, int64_t ldb, int M, int N, int K, int epi, void* out,
                   int64_t ldo) {
    vtGemmNT p;
    memset(&p, 0, sizeof(p));
    p.A = A; p.lda = lda; p.B = B; p.ldb = ldb; p.M = M; p.N = N; p.K = K; p.epi = epi; p.out = out; p.ldo = ldo;
    // backward only: the forward pass of a clip stays bit-identical whatever batch it runs in (sampled token ids, reconstructions);
    // its gradients already differ across batch sizes in the last fp32 bits (summation over the batch's rows)
    if (t->splitk_bytes && t->in_backward && t->splitk_on) { p.splitk_ws = WS(void, t->splitk); p.splitk_ws_bytes = (int64_t)t->splitk_bytes; }
    // data-parallel runs (vt_tokenizer_set_data_parallel): launches with more 192x192 tiles than one round of the chip go out one tile per
    // workgroup instead of as 256 persistent workgroups with fixed tile lists.  Equal speed on a free chip (62.9 vs 63.2 us on fc1 forward,
    // measured at whole rounds: 768 / 1024 tiles); with a collective's workgroups holding CUs the hardware dispatcher hands the remaining
    // tiles to whichever CU is free (4 rounds -> 5), where a persistent workgroup that could not start runs its whole list after the
    // others have finished theirs (x 1.67 measured with a single-GPU stand-in, tools/cu_thief_stats.sh -- not yet under a real collective)
    // (backward only: that is where the gradient all-reduce runs, and the forward's GELU launch would fill its look-up table once per tile)
    if (t->data_parallel && t->in_backward && M % 192 == 0 && N % 192 == 0 && (long)(M / 192) * (N / 192) > 256) p.tile = 6;
    return p;
}

// one timm Block forward: x_in -> x_out (both fp32 [M,D])
static int block_forward(vtTokenizer* t, const BlockBufs& b, const vtBlockTensors& w, const float* x_in, float* x_out, void* ws, vtStream s) {
    const vtTokenizerConfig& c = t->c;
    const int M = t->M, D = c.D, D3 = t->D3, D4 = t->D4;
    const vtRowMap id = {0, 0, 0};
    TRY(vt_layernorm_fwd(x_in, id, w.norm1_w, w.norm1_b, 1e-5f, M, D, WS(void, b.h1), WS(float, b.mean1), WS(float, b.rstd1), s));
    vtGemmNT g = nt(t, ws, WS(void, b.h1), D, WS(void, b.qkv_wb), D, M, D3, D, VT_EPI_BF16, WS(void, b.qkv), D3);
    TRY(vt_gemm_nt(&g, s));
    TRY(vt_attention_fwd(WS(void, b.qkv), c.B, t->L, c.H, c.D / c.H, WS(void, b.o), WS(float, b.lse), s));
    g = nt(t, ws, WS(void, b.o), D, WS(void, b.proj_wb), D, M, D, D, VT_EPI_F32, WS(void, b.x_mid), D);
    g.bias = w.proj_b; g.residual = x_in; g.ldr = D;
    TRY(vt_gemm_nt(&g, s));
    TRY(vt_layernorm_fwd(WS(float, b.x_mid), id, w.norm2_w, w.norm2_b, 1e-5f, M, D, WS(void, b.h2), WS(float, b.mean2), WS(float, b.rstd2), s));
    g = nt(t, ws, WS(void, b.h2), D, WS(void, b.fc1_wb), D, M, D4, D, VT_EPI_BF16_GELU, WS(void, b.u), D4);
    g.out2 = WS(void, b.g); g.ldo2 = D4; g.bias = w.fc1_b;
    TRY(vt_gemm_nt(&g, s));
    g = nt(t, ws, WS(void, b.g), D4, WS(void, b.fc2_wb), D4, M, D, D4, VT_EPI_F32, x_out, D);
    g.bias = w.fc2_b; g.residual = WS(float, b.x_mid); g.ldr = D;
    TRY(vt_gemm_nt(&g, s));
    return VT_OK;
}

// The last block of a stack (see vtTokenizer::LastBlock): rows outside the kept suffix of x_out are NOT written.
static int block_forward_last(vtTokenizer* t, const vtTokenizer::LastBlock& lb, const BlockBufs& b, const vtBlockTensors& w, const float* x_in,
                              float* x_out, void* ws, vtStream s) {
    const vtTokenizerConfig& c = t->c;
    const int M = t->M, D = c.D, D3 = t->D3, D4 = t->D4, Mk = lb.Mk;
    const vtRowMap id = {0, 0, 0};
    const vtRowMap kmap = {lb.nk, t->L, lb.q_begin};
    TRY(vt_layernorm_fwd(x_in, id, w.norm1_w, w.norm1_b, 1e-5f, M, D, WS(void, b.h1), WS(float, b.mean1), WS(float, b.rstd1), s));
    vtGemmNT g = nt(t, ws, WS(void, b.h1), D, WS(void, b.qkv_wb), D, M, D3, D, VT_EPI_BF16, WS(void, b.qkv), D3);
    TRY(vt_gemm_nt(&g, s));
    TRY(vt_attention_fwd_rows(WS(void, b.qkv), c.B, t->L, c.H, c.D / c.H, lb.q_begin, WS(void, b.o), WS(float, b.lse), s));
    g = nt(t, ws, WS(void, b.o), D, WS(void, b.proj_wb), D, Mk, D, D, VT_EPI_F32, WS(void, b.x_mid), D);
    g.bias = w.proj_b; g.residual = x_in; g.ldr = D; g.omap = kmap;
    TRY(vt_gemm_nt(&g, s));
    TRY(vt_layernorm_fwd(WS(float, b.x_mid), kmap, w.norm2_w, w.norm2_b, 1e-5f, Mk, D, WS(void, b.h2), WS(float, b.mean2), WS(float, b.rstd2), s));
    g = nt(t, ws, WS(void, b.h2), D, WS(void, b.fc1_wb), D, Mk, D4, D, VT_EPI_BF16_GELU, WS(void, b.u), D4);
    g.out2 = WS(void, b.g); g.ldo2 = D4; g.bias = w.fc1_b;
    TRY(vt_gemm_nt(&g, s));
    g = nt(t, ws, WS(void, b.g), D4, WS(void, b.fc2_wb), D4, Mk, D, D4, VT_EPI_F32, x_out, D);
    g.bias = w.fc2_b; g.residual = WS(float, b.x_mid); g.ldr = D; g.omap = kmap;
    TRY(vt_gemm_nt(&g, s));
    return VT_OK;
}

extern "C" int vt_tokenizer_encode(vtTokenizer* t, const vtTokenizerTensors* P, const float* video, void* ws,
                                   const vtTokenizerOutputs* out, uint64_t seed, vtStream s) {
    VT_CHECK_ARG(t && P && video && ws && out, "vt_tokenizer_encode: null pointer");
    t->in_backward = false;
    VT_CHECK_ARG(out->encoded && out->indices && out->losses, "vt_tokenizer_encode: encoded/indices/losses outputs are required");
    const vtTokenizerConfig& c = t->c;
    const int D = c.D, L = t->L, Nv = t->Nv, Nq = c.Nq;
    // 1. patchify + patch-embed GEMM (+bias +sincos PE) written straight into rows [0,Nv) of every sequence
    TRY(vt_patchify(video, c.B, c.C, c.T, c.S, c.pt, c.p, WS(void, t->patches), s));
    float* x0 = WS(float, t->x_enc[0]);
    vtGemmNT g = nt(t, ws, WS(void, t->patches), t->Kp, WS(void, t->pe_wb), t->Kp, t->Mv, D, t->Kp, VT_EPI_F32, x0, D);
    g.bias = P->pe_b; g.rowmod = P->enc_patch_pe; g.rowmod_period = Nv; g.omap = vtRowMap{Nv, L, 0};
    g.round_bf16 = 1;  // conv output is bf16 under autocast before the fp32 PE add
    TRY(vt_gemm_nt(&g, s));
    // 2. learned latent queries broadcast into rows [Nv, L)   (larp_tokenizer.py:410, transformer.py:64)
    TRY(vt_assemble_rows(x0, L, Nv, c.B, Nq, D, nullptr, P->enc_query, nullptr, s));
    // 3. encoder blocks
    for (int i = 0; i < c.depth_enc; ++i) {
        if (i == c.depth_enc - 1 && t->last_enc.enabled)
            TRY(block_forward_last(t, t->last_enc, t->enc[i], P->enc_blocks[i], WS(float, t->x_enc[i]), WS(float, t->x_enc[i + 1]), ws, s));
        else
            TRY(block_forward(t, t->enc[i], P->enc_blocks[i], WS(float, t->x_enc[i]), WS(float, t->x_enc[i + 1]), ws, s));
    }
    const float* xe = WS(float, t->x_enc[c.depth_enc]);
    const vtRowMap qmap = {Nq, L, Nv};  // the last Nq rows of every sequence (transformer.py:69)
    // 4. bottleneck: norm stats, in_linear, VQ, out_linear
    if (out->input_norms)
        hipLaunchKernelGGL(rownorm_mean_kernel, dim3(2), dim3(512), 0, (hipStream_t)s, xe, (int64_t)L, (int64_t)Nv, (int64_t)L - 1, c.B, D, out->input_norms);
    TRY(vt_cast_rows(xe, qmap, t->Mq, D, WS(void, t->zb), D, s));
    g = nt(t, ws, WS(void, t->zb), D, WS(void, t->in_wb), D, t->Mq, c.d, D, VT_EPI_F32, WS(void, t->zproj), 64);
    g.bias = P->in_b; g.round_bf16 = 1;
    TRY(vt_gemm_nt(&g, s));
    if (out->projected_z)
        hipLaunchKernelGGL(compact_cols_kernel, dim3((t->Mq * c.d + 255) / 256), dim3(256), 0, (hipStream_t)s, WS(float, t->zproj), (int64_t)64, t->Mq, c.d, out->projected_z);
    TRY(vt_vq_forward_ctr(WS(float, t->zproj), 64, P->codebook, t->Mq, c.K, c.d, c.vq_mode, c.l2_normalized, c.inv_tau, c.beta, c.codebook_w,
                      seed, t->seed_ctr, WS(float, t->vq_E), WS(float, t->vq_wnorm), WS(float, t->vq_zn), WS(float, t->vq_znorm), WS(int64_t, t->vq_idx),
                      WS(float, t->vq_rz), WS(void, t->vq_rzpad), 64, WS(float, t->vq_losses), WS(void, t->vq_ws), s));
    hipStream_t hs = (hipStream_t)s;
    TRY(copy_d2d(out->indices, WS(void, t->vq_idx), (size_t)t->Mq * 8, hs));
    TRY(copy_d2d(out->losses, WS(void, t->vq_losses), 16, hs));
    if (out->unregularized_z) TRY(copy_d2d(out->unregularized_z, WS(void, t->vq_zn), (size_t)t->Mq * c.d * 4, hs));
    if (out->regularized_z) TRY(copy_d2d(out->regularized_z, WS(void, t->vq_rz), (size_t)t->Mq * c.d * 4, hs));
    if (out->emb) TRY(copy_d2d(out->emb, WS(void, t->vq_E), (size_t)c.K * c.d * 4, hs));
    g = nt(t, ws, WS(void, t->vq_rzpad), 64, WS(void, t->out_wb), 64, t->Mq, D, 64, VT_EPI_F32, out->encoded, D);
    g.bias = P->out_b; g.round_bf16 = 1;
    TRY(vt_gemm_nt(&g, s));
    VT_CHECK_LAUNCH("vt_tokenizer_encode");
    return VT_OK;
}

extern "C" int vt_tokenizer_codes_to_encoded(vtTokenizer* t, const vtTokenizerTensors* P, const int64_t* indices, void* ws,
                                             float* encoded, vtStream s) {
    VT_CHECK_ARG(t && P && indices && ws && encoded, "vt_tokenizer_codes_to_encoded: null pointer");
    t->in_backward = false;
    const vtTokenizerConfig& c = t->c;
    TRY(vt_vq_prep_codebook(P->codebook, c.K, c.d, c.l2_normalized, WS(float, t->vq_E), WS(float, t->vq_wnorm), WS(void, t->vq_ws), s));
    TRY(vt_vq_gather(WS(float, t->vq_E), indices, t->Mq, c.K, c.d, nullptr, WS(void, t->vq_rzpad), 64, s));
    vtGemmNT g = nt(t, ws, WS(void, t->vq_rzpad), 64, WS(void, t->out_wb), 64, t->Mq, c.D, 64, VT_EPI_F32, encoded, c.D);
    g.bias = P->out_b; g.round_bf16 = 1;
    TRY(vt_gemm_nt(&g, s));
    return VT_OK;
}

extern "C" int vt_tokenizer_decode(vtTokenizer* t, const vtTokenizerTensors* P, const float* encoded, void* ws, float* pred, vtStream s) {
    VT_CHECK_ARG(t && P && encoded && ws && pred, "vt_tokenizer_decode: null pointer");
    t->in_backward = false;
    const vtTokenizerConfig& c = t->c;
    const int D = c.D, L = t->L, Nv = t->Nv, Nq = c.Nq;
    float* x0 = WS(float, t->x_dec[0]);
    // decoder sequence = [encoded + latent PE | patch queries (+ token type)]   (larp_tokenizer.py:463-466)
    TRY(vt_assemble_rows(x0, L, 0, c.B, Nq, D, encoded, P->dec_latent_pe, nullptr, s));
    TRY(vt_assemble_rows(x0, L, Nq, c.B, Nv, D, nullptr, WS(float, t->dec_query_sum), nullptr, s));
    for (int i = 0; i < c.depth_dec; ++i) {
        if (i == c.depth_dec - 1 && t->last_dec.enabled)
            TRY(block_forward_last(t, t->last_dec, t->dec[i], P->dec_blocks[i], WS(float, t->x_dec[i]), WS(float, t->x_dec[i + 1]), ws, s));
        else
            TRY(block_forward(t, t->dec[i], P->dec_blocks[i], WS(float, t->x_dec[i]), WS(float, t->x_dec[i + 1]), ws, s));
    }
    // head on the last Nv rows: LayerNorm(1e-6) -> Linear (rows permuted to (c,dt,dy,dx)) -> unpatchify
    const vtRowMap vmap = {Nv, L, Nq};
    TRY(vt_layernorm_fwd(WS(float, t->x_dec[c.depth_dec]), vmap, P->head_norm_w, P->head_norm_b, 1e-6f, t->Mv, D, WS(void, t->hN),
                         WS(float, t->meanH), WS(float, t->rstdH), s));
    vtGemmNT g = nt(t, ws, WS(void, t->hN), D, WS(void, t->head_wb), D, t->Mv, t->Kp, D, VT_EPI_F32, WS(void, t->yrows), t->Kp);
    g.bias = WS(float, t->head_b_perm);
    TRY(vt_gemm_nt(&g, s));
    TRY(vt_unpatchify(WS(float, t->yrows), c.B, c.C, c.T, c.S, c.pt, c.p, pred, s));
    return VT_OK;
}

static vtGemmTN tn(const void* A, int64_t lda, const void* B, int64_t ldb, int M, int P, int Q, float* out, int64_t ldo) {
    vtGemmTN p;
    memset(&p, 0, sizeof(p));
    p.A = A; p.lda = lda; p.B = B; p.ldb = ldb; p.M = M; p.P = P; p.Q = Q; p.out = out; p.ldo = ldo; p.p_lim = P; p.q_lim = Q;
    return p;
}

// A weight gradient whose output is only a few tiles (bottleneck in/out_linear: D x d) would occupy a handful of the
// 256 CUs for the whole token dimension.  Split the tokens into up to 16 slabs, run them as one grouped launch into
// partial outputs, and add the slabs in a fixed order.
static int skinny_wgrad(vtTokenizer* t, vtGemmTN w, void* ws, vtStream s) {
    int ns = 16;
    while (ns > 1 && (w.M % (64 * ns)) != 0) ns >>= 1;
    if (ns == 1) return vt_gemm_tn_grouped(&w, 1, s);
    const int rows = w.M / ns;
    const size_t slab = (size_t)w.p_lim * w.q_lim;
    vtGemmTN pr[16];
    float* part = WS(float, t->wg_slabs);
    for (int i = 0; i < ns; ++i) {
        pr[i] = w;
        pr[i].A = (const char*)w.A + (size_t)i * rows * w.lda * 2;
        pr[i].B = (const char*)w.B + (size_t)i * rows * w.ldb * 2;
        pr[i].M = rows;
        pr[i].out = part + i * slab;
        pr[i].ldo = w.q_lim;
    }
    TRY(vt_gemm_tn_grouped(pr, ns, s));
    if (w.ldo == w.q_lim) return vt_sum_slabs(part, ns, (int64_t)slab, (int)slab, w.out, s);
    vt_set_error("skinny_wgrad: output must be dense");
    return VT_ERR_INVALID;
}

// the main stream is about to rewrite gradient set `set`: wait for the side-stream group that still reads it
static int wait_set(vtTokenizer* t, int set, vtStream s) {
    if (t->wg_stream && t->set_flush[set] >= 0) {
        if (hipStreamWaitEvent((hipStream_t)s, t->ev_done[t->set_flush[set] % vtTokenizer::NEV], 0) != hipSuccess) { vt_set_error("wait_set: hipStreamWaitEvent failed"); return VT_ERR_LAUNCH; }
        t->set_flush[set] = -1;
    }
    return VT_OK;
}
// all side-stream groups of this backward are done before the main stream goes on (end of backward)
static int join_wgrad_stream(vtTokenizer* t, vtStream s) {
    if (t->wg_stream && t->flush_id > 0) {
        if (hipStreamWaitEvent((hipStream_t)s, t->ev_done[(t->flush_id - 1) % vtTokenizer::NEV], 0) != hipSuccess) { vt_set_error("join_wgrad_stream: hipStreamWaitEvent failed"); return VT_ERR_LAUNCH; }
        for (int i = 0; i < vtTokenizer::NSETS_MAX; ++i) t->set_flush[i] = -1;   // the side stream runs its groups in order
    }
    return VT_OK;
}

static int flush_wgrads(vtTokenizer* t, int stage_done, vtStream main_s) {
    vtStream s = main_s;
    if (t->wg_stream) {
        hipEvent_t fork = t->ev_fork[t->flush_id % vtTokenizer::NEV];
        if (hipEventRecord(fork, (hipStream_t)main_s) != hipSuccess || hipStreamWaitEvent(t->wg_stream, fork, 0) != hipSuccess) { vt_set_error("flush_wgrads: fork failed"); return VT_ERR_LAUNCH; }
        s = (vtStream)t->wg_stream;
    }
    for (size_t i = 0; i < t->pending.size(); i += VT_TN_MAX_GROUP) {
        const int n = (int)((t->pending.size() - i) < VT_TN_MAX_GROUP ? (t->pending.size() - i) : VT_TN_MAX_GROUP);
        TRY(vt_gemm_tn_grouped(t->pending.data() + i, n, s));
    }
    t->pending.clear();
    for (size_t i = 0; i < t->pending_red.size(); i += VT_REDUCE_MAX_GROUP) {
        const int n = (int)((t->pending_red.size() - i) < VT_REDUCE_MAX_GROUP ? (t->pending_red.size() - i) : VT_REDUCE_MAX_GROUP);
        TRY(vt_reduce_grouped(t->pending_red.data() + i, n, s));
    }
    t->pending_red.clear();
    t->pending_blocks = 0;
    t->final_through = stage_done;
    if (t->wg_stream) {
        if (hipEventRecord(t->ev_done[t->flush_id % vtTokenizer::NEV], t->wg_stream) != hipSuccess) { vt_set_error("flush_wgrads: event record failed"); return VT_ERR_LAUNCH; }
        for (int set : t->sets_pending) t->set_flush[set] = t->flush_id;
        t->flush_id++;
    }
    t->sets_pending.clear();
    return VT_OK;
}

// queue the reduction of a LayerNorm backward's partials / of the gelu' epilogue's column sums
static void queue_ln_reduce(vtTokenizer* t, const float* part, int nslab, int D, float* dgamma, float* dbeta, float* dxsum) {
    vtReduceItem q;
    q.partial = part; q.nslab = nslab; q.width = D; q.nout = dxsum ? 3 : 2; q.lanes = nslab >= 128 ? 32 : 8; q.slab_stride = (int64_t)3 * D;
    q.o[0] = dgamma; q.o[1] = dbeta; q.o[2] = dxsum;
    t->pending_red.push_back(q);
}
static void queue_slab_sum(vtTokenizer* t, const float* part, int nslab, int width, float* out) {
    vtReduceItem q;
    q.partial = part; q.nslab = nslab; q.width = width; q.nout = 1; q.lanes = nslab >= 128 ? 32 : 8; q.slab_stride = width;
    q.o[0] = out; q.o[1] = q.o[2] = nullptr;
    t->pending_red.push_back(q);
}

// Backward of one block.  In: dX (fp32) and gs[set].dx_out (bf16) hold dL/dx_out.  Out: dX holds dL/dx_in and its bf16
// copy goes to gs[next set].dx_out.  The four weight-gradient GEMMs are queued (t->pending), not launched.
// prev_bias_grad: where sum_rows(dL/dx_in) goes (= bias gradient of whatever produced x_in), may be NULL.
static int block_backward(vtTokenizer* t, const BlockBufs& b, const vtBlockTensors& w, const vtBlockTensors& gr, const float* x_in,
                          float* prev_bias_grad, void* ws, vtStream s) {
    const vtTokenizerConfig& c = t->c;
    const int M = t->M, Mp = t->Mp, D = c.D, D3 = t->D3, D4 = t->D4;
    const vtRowMap id = {0, 0, 0};
    const int set_next = (t->set_idx + 1) % t->nsets();
    TRY(wait_set(t, t->set_idx, s));
    TRY(wait_set(t, set_next, s));
    t->sets_pending.push_back(t->set_idx);
    const vtTokenizer::GradSet& g0 = t->gs[t->set_idx];
    const vtTokenizer::GradSet& g1 = t->gs[set_next];
    float* dX = WS(float, t->dX);
    void* dXa = WS(void, g0.dx_out);
    void* dXm = WS(void, g0.dx_mid);
    void* du = WS(void, g0.du);
    void* dqkv = WS(void, g0.dqkv);
    // fc2 dgrad fused with GELU': du = (dx_out . W2) * gelu'(u)
    vtGemmNT g = nt(t, ws, dXa, D, WS(void, b.fc2_wt), D, M, D4, D, VT_EPI_BF16_DGELU, du, D4);
    g.aux = WS(void, b.u); g.ldaux = D4;
    g.colsum_partial = WS(float, g0.cs_part);  // fc1 bias gradient = column sums of du, taken in the epilogue; summed at the flush
    TRY(vt_gemm_nt(&g, s));
    queue_slab_sum(t, WS(float, g0.cs_part), (M + 191) / 192, D4, gr.fc1_b);
    // fc1 dgrad
    g = nt(t, ws, du, D4, WS(void, b.fc1_wt), D4, M, D, D4, VT_EPI_BF16, WS(void, t->dh), D);
    TRY(vt_gemm_nt(&g, s));
    // LayerNorm2 backward + residual: dx_mid = dx_out + ln_bwd(dh2) (in place in dX; bf16 copy -> dXm); column sum = proj bias grad
    int nsl = 0;
    TRY(vt_layernorm_bwd_partials(WS(void, t->dh), WS(float, b.x_mid), id, w.norm2_w, WS(float, b.mean2), WS(float, b.rstd2), dX, M, D, dX, dXm,
                                  WS(float, g0.ln_part2), &nsl, s));
    queue_ln_reduce(t, WS(float, g0.ln_part2), nsl, D, gr.norm2_w, gr.norm2_b, gr.proj_b);
    // proj dgrad
    g = nt(t, ws, dXm, D, WS(void, b.proj_wt), D, M, D, D, VT_EPI_BF16, WS(void, t->dob), D);
    TRY(vt_gemm_nt(&g, s));
    // attention backward
    TRY(attn_bwd(t, ws, WS(void, b.qkv), WS(void, b.o), WS(void, t->dob), WS(float, b.lse), 0, dqkv, s));
    // qkv dgrad
    g = nt(t, ws, dqkv, D3, WS(void, b.qkv_wt), D3, M, D, D3, VT_EPI_BF16, WS(void, t->dh), D);
    TRY(vt_gemm_nt(&g, s));
    // the block's four weight gradients: queued for the grouped launch
    t->pending.push_back(tn(dXa, D, WS(void, b.g), D4, Mp, D, D4, gr.fc2_w, D4));
    t->pending.push_back(tn(du, D4, WS(void, b.h2), D, Mp, D4, D, gr.fc1_w, D));
    t->pending.push_back(tn(dXm, D, WS(void, b.o), D, Mp, D, D, gr.proj_w, D));
    t->pending.push_back(tn(dqkv, D3, WS(void, b.h1), D, Mp, D3, D, gr.qkv_w, D));
    t->pending_blocks++;
    // LayerNorm1 backward + residual: dx_in = dx_mid + ln_bwd(dh) (in place; bf16 copy -> next set's dx_out)
    TRY(vt_layernorm_bwd_partials(WS(void, t->dh), x_in, id, w.norm1_w, WS(float, b.mean1), WS(float, b.rstd1), dX, M, D, dX, WS(void, g1.dx_out),
                                  WS(float, g0.ln_part1), &nsl, s));
    queue_ln_reduce(t, WS(float, g0.ln_part1), nsl, D, gr.norm1_w, gr.norm1_b, prev_bias_grad);
    t->set_idx = set_next;
    return VT_OK;
}

// Backward of the last block of a stack: dL/dx_out is zero outside the kept rows, so the MLP half, proj and the query side of
// attention run on the compact kept rows; qkv input gradient, K/V gradients and LayerNorm1 cover all rows as usual.
static int block_backward_last(vtTokenizer* t, const vtTokenizer::LastBlock& lb, const BlockBufs& b, const vtBlockTensors& w,
                               const vtBlockTensors& gr, const float* x_in, float* prev_bias_grad, void* ws, vtStream s) {
    const vtTokenizerConfig& c = t->c;
    const int M = t->M, Mp = t->Mp, D = c.D, D3 = t->D3, D4 = t->D4, Mk = lb.Mk, Mkp = lb.Mkp;
    const vtRowMap id = {0, 0, 0};
    const vtRowMap kmap = {lb.nk, t->L, lb.q_begin};
    const int set_next = (t->set_idx + 1) % t->nsets();
    TRY(wait_set(t, t->set_idx, s));
    TRY(wait_set(t, set_next, s));
    t->sets_pending.push_back(t->set_idx);
    const vtTokenizer::GradSet& g0 = t->gs[t->set_idx];
    const vtTokenizer::GradSet& g1 = t->gs[set_next];
    float* dX = WS(float, t->dX);
    void* dXa = WS(void, lb.dxa);   // compact bf16 copies of dL/dx_out and dL/dx_mid, and du: rows >= Mk stay zero
    void* dXm = WS(void, lb.dxm);
    void* du = WS(void, lb.du);
    void* dqkv = WS(void, g0.dqkv);
    TRY(vt_cast_rows(dX, kmap, Mk, D, dXa, D, s));
    vtGemmNT g = nt(t, ws, dXa, D, WS(void, b.fc2_wt), D, Mk, D4, D, VT_EPI_BF16_DGELU, du, D4);
    g.aux = WS(void, b.u); g.ldaux = D4;
    g.colsum_partial = WS(float, g0.cs_part);
    TRY(vt_gemm_nt(&g, s));
    queue_slab_sum(t, WS(float, g0.cs_part), (Mk + 191) / 192, D4, gr.fc1_b);
    g = nt(t, ws, du, D4, WS(void, b.fc1_wt), D4, Mk, D, D4, VT_EPI_BF16, WS(void, t->dh), D);
    TRY(vt_gemm_nt(&g, s));
    int nsl = 0;
    TRY(vt_layernorm_bwd_partials(WS(void, t->dh), WS(float, b.x_mid), kmap, w.norm2_w, WS(float, b.mean2), WS(float, b.rstd2), dX, Mk, D, dX, nullptr,
                                  WS(float, g0.ln_part2), &nsl, s));
    queue_ln_reduce(t, WS(float, g0.ln_part2), nsl, D, gr.norm2_w, gr.norm2_b, gr.proj_b);
    TRY(vt_cast_rows(dX, kmap, Mk, D, dXm, D, s));
    g = nt(t, ws, dXm, D, WS(void, b.proj_wt), D, Mk, D, D, VT_EPI_BF16, WS(void, t->dob), D);
    TRY(vt_gemm_nt(&g, s));
    TRY(attn_bwd(t, ws, WS(void, b.qkv), WS(void, b.o), WS(void, t->dob), WS(float, b.lse), lb.q_begin, dqkv, s));
    g = nt(t, ws, dqkv, D3, WS(void, b.qkv_wt), D3, M, D, D3, VT_EPI_BF16, WS(void, t->dh), D);
    TRY(vt_gemm_nt(&g, s));
    t->pending.push_back(tn(dXa, D, WS(void, b.g), D4, Mkp, D, D4, gr.fc2_w, D4));
    t->pending.push_back(tn(du, D4, WS(void, b.h2), D, Mkp, D4, D, gr.fc1_w, D));
    t->pending.push_back(tn(dXm, D, WS(void, b.o), D, Mkp, D, D, gr.proj_w, D));
    t->pending.push_back(tn(dqkv, D3, WS(void, b.h1), D, Mp, D3, D, gr.qkv_w, D));
    t->pending_blocks++;
    TRY(vt_layernorm_bwd_partials(WS(void, t->dh), x_in, id, w.norm1_w, WS(float, b.mean1), WS(float, b.rstd1), dX, M, D, dX, WS(void, g1.dx_out),
                                  WS(float, g0.ln_part1), &nsl, s));
    queue_ln_reduce(t, WS(float, g0.ln_part1), nsl, D, gr.norm1_w, gr.norm1_b, prev_bias_grad);
    t->set_idx = set_next;
    return VT_OK;
}

extern "C" int vt_tokenizer_backward(vtTokenizer* t, const vtTokenizerTensors* P, const float* d_pred, const float* gscal, void* ws,
                                     const vtTokenizerTensors* G, int32_t stage_begin, int32_t stage_end, int32_t* final_through,
                                     vtStream s) {
    VT_CHECK_ARG(t && P && ws && G && G->enc_blocks && G->dec_blocks, "vt_tokenizer_backward: null pointer");
    t->in_backward = true;
    const vtTokenizerConfig& c = t->c;
    const int nstage = vt_tokenizer_num_backward_stages(t);
    VT_CHECK_ARG(stage_begin >= 0 && stage_end <= nstage && stage_begin <= stage_end, "vt_tokenizer_backward: bad stage range");
    const int D = c.D, L = t->L, Nv = t->Nv, Nq = c.Nq, Kp = t->Kp;
    const vtRowMap id = {0, 0, 0};
    const vtRowMap vmap = {Nv, L, Nq};   // decoder: last Nv rows
    const vtRowMap qmap = {Nq, L, Nv};   // encoder: last Nq rows
    const vtRowMap lmap = {Nq, L, 0};    // decoder: first Nq rows (latents)
    const vtRowMap tmap = {Nv, L, 0};    // encoder: first Nv rows (video tokens)
    hipStream_t hs = (hipStream_t)s;
    float* dX = WS(float, t->dX);
    for (int st = stage_begin; st < stage_end; ++st) {
        if (st == 0) {
            t->pending.clear();
            t->pending_red.clear(); t->pending_blocks = 0; t->set_idx = 0; t->final_through = 0;
            t->sets_pending.clear();
            TRY(join_wgrad_stream(t, s));    // (a previous backward that was not run to its last stage)
            // ---- head: d_pred -> patch rows (c,dt,dy,dx) -> dgrad / wgrad -> LayerNorm backward into the last Nv rows
            VT_CHECK_ARG(d_pred, "vt_tokenizer_backward: stage 0 needs d_pred");
            TRY(vt_patchify(d_pred, c.B, c.C, c.T, c.S, c.pt, c.p, WS(void, t->dY), s));
            vtGemmNT g = nt(t, ws, WS(void, t->dY), Kp, WS(void, t->head_wt), Kp, t->Mv, D, Kp, VT_EPI_BF16, WS(void, t->dhN), D);
            TRY(vt_gemm_nt(&g, s));
            vtGemmTN w = tn(WS(void, t->dY), Kp, WS(void, t->hN), D, t->Mvp, Kp, D, G->head_w, D);
            w.row_perm = WS(int32_t, t->perm);
            TRY(vt_gemm_tn_grouped(&w, 1, s));
            TRY(vt_colsum(WS(void, t->dY), 1, Kp, id, t->Mv, Kp, WS(float, t->tmp_vec), WS(void, t->cs_ws), s));
            hipLaunchKernelGGL(scatter_f32_kernel, dim3((Kp + 255) / 256), dim3(256), 0, hs, WS(float, t->tmp_vec), WS(int32_t, t->perm), Kp, G->head_b);
            // rows < Nq of the last decoder block's output are dropped by the slice => zero gradient
            TRY(vt_zero_rows(dX, WS(void, t->gs[0].dx_out), lmap, t->Mq, D, s));
            TRY(vt_layernorm_bwd(WS(void, t->dhN), WS(float, t->x_dec[c.depth_dec]), vmap, P->head_norm_w, WS(float, t->meanH), WS(float, t->rstdH),
                                 nullptr, t->Mv, D, dX, WS(void, t->gs[0].dx_out), G->head_norm_w, G->head_norm_b,
                                 nullptr, WS(void, t->ln_ws), s));
            // fc2 bias grad of the last decoder block = column sum of dL/dx_out (all rows; zero rows add nothing)
            TRY(vt_colsum(dX, 0, D, id, t->M, D, G->dec_blocks[c.depth_dec - 1].fc2_b, WS(void, t->cs_ws), s));
            t->final_through = 1;
        } else if (st <= c.depth_dec) {
            const int i = c.depth_dec - st;  // decoder blocks, last first
            float* prev = i > 0 ? G->dec_blocks[i - 1].fc2_b : nullptr;
            if (i == c.depth_dec - 1 && t->last_dec.enabled)
                TRY(block_backward_last(t, t->last_dec, t->dec[i], P->dec_blocks[i], G->dec_blocks[i], WS(float, t->x_dec[i]), prev, ws, s));
            else
                TRY(block_backward(t, t->dec[i], P->dec_blocks[i], G->dec_blocks[i], WS(float, t->x_dec[i]), prev, ws, s));
            if (t->pending_blocks >= t->wg_batch || i == 0) TRY(flush_wgrads(t, st + 1, s));
        } else if (st == c.depth_dec + 1) {
            // ---- bottleneck.  dX holds dL/d(decoder input sequence)
            if (G->dec_token_type) TRY(vt_colsum(dX, 0, D, vmap, t->Mv, D, G->dec_token_type, WS(void, t->cs_ws), s));
            TRY(vt_cast_rows(dX, lmap, t->Mq, D, WS(void, t->dEncb), D, s));  // d encoded (bf16 under autocast)
            TRY(vt_colsum(WS(void, t->dEncb), 1, D, id, t->Mq, D, G->out_b, WS(void, t->cs_ws), s));
            // out_linear: dgrad -> d regularized_z [Mq,64]; wgrad -> dW_out [D,d]
            vtGemmNT g = nt(t, ws, WS(void, t->dEncb), D, WS(void, t->out_wt), D, t->Mq, 64, D, VT_EPI_F32, WS(void, t->d_rz), 64);
            TRY(vt_gemm_nt(&g, s));
            vtGemmTN w = tn(WS(void, t->dEncb), D, WS(void, t->vq_rzpad), 64, t->Mqp, D, 64, G->out_w, c.d);
            w.q_lim = c.d;
            TRY(skinny_wgrad(t, w, ws, s));
            // VQ backward: straight-through + commitment to z, codebook loss to the embedding
            TRY(vt_vq_backward(WS(float, t->d_rz), 64, gscal, c.beta, c.codebook_w, WS(float, t->vq_zn), WS(float, t->vq_znorm), WS(float, t->vq_E),
                               WS(float, t->vq_wnorm), WS(int64_t, t->vq_idx), t->Mq, c.K, c.d, c.l2_normalized, nullptr, WS(void, t->dz_pad), 64,
                               c.freeze_codebook ? nullptr : G->codebook, WS(void, t->vq_ws), s));
            // in_linear: bias grad, wgrad, dgrad scattered into the last Nq rows of the encoder output gradient
            TRY(vt_colsum(WS(void, t->dz_pad), 1, 64, id, t->Mq, 64, WS(float, t->tmp_vec), WS(void, t->cs_ws), s));
            TRY(copy_d2d(G->in_b, WS(void, t->tmp_vec), (size_t)c.d * 4, hs));
            w = tn(WS(void, t->dz_pad), 64, WS(void, t->zb), D, t->Mqp, 64, D, G->in_w, D);
            w.p_lim = c.d;
            TRY(skinny_wgrad(t, w, ws, s));
            // (the encoder goes on in the rotation of gradient sets where the decoder stopped: with the weight gradients on their own
            // stream, set 0 may still be read by the decoder's last group)
            TRY(wait_set(t, t->set_idx, s));
            TRY(vt_zero_rows(dX, WS(void, t->gs[t->set_idx].dx_out), tmap, t->Mv, D, s));  // video-token rows get no gradient from the bottleneck
            g = nt(t, ws, WS(void, t->dz_pad), 64, WS(void, t->in_wt), 64, t->Mq, D, 64, VT_EPI_F32, dX, D);
            g.out2 = WS(void, t->gs[t->set_idx].dx_out); g.ldo2 = D; g.omap = qmap;
            TRY(vt_gemm_nt(&g, s));
            TRY(vt_colsum(dX, 0, D, id, t->M, D, G->enc_blocks[c.depth_enc - 1].fc2_b, WS(void, t->cs_ws), s));
            t->final_through = st + 1;
        } else if (st <= c.depth_dec + 1 + c.depth_enc) {
            const int i = c.depth_enc - (st - c.depth_dec - 1);
            float* prev = i > 0 ? G->enc_blocks[i - 1].fc2_b : nullptr;
            if (i == c.depth_enc - 1 && t->last_enc.enabled)
                TRY(block_backward_last(t, t->last_enc, t->enc[i], P->enc_blocks[i], G->enc_blocks[i], WS(float, t->x_enc[i]), prev, ws, s));
            else
                TRY(block_backward(t, t->enc[i], P->enc_blocks[i], G->enc_blocks[i], WS(float, t->x_enc[i]), prev, ws, s));
            if (t->pending_blocks >= t->wg_batch || i == 0 || i < t->wg_tail) TRY(flush_wgrads(t, st + 1, s));
        } else {
            // ---- patch embed + learned queries.  dX holds dL/d(encoder input sequence)
            TRY(vt_batch_sum(dX, qmap, c.B, Nq, D, G->enc_query, s));
            TRY(vt_cast_rows(dX, tmap, t->Mv, D, WS(void, t->dTok), D, s));
            TRY(vt_colsum(WS(void, t->dTok), 1, D, id, t->Mv, D, G->pe_b, WS(void, t->cs_ws), s));
            vtGemmTN w = tn(WS(void, t->dTok), D, WS(void, t->patches), Kp, t->Mvp, D, Kp, G->pe_w, Kp);
            TRY(vt_gemm_tn_grouped(&w, 1, s));
            TRY(join_wgrad_stream(t, s));    // every gradient of the step is behind the caller's stream from here on
            t->final_through = st + 1;
        }
    }
    if (final_through) *final_through = t->final_through;
    VT_CHECK_LAUNCH("vt_tokenizer_backward");
    return VT_OK;
}

// The same, for a caller that reduces finished gradient slices between stages (parallel.GradReducer): runs stages from stage_begin until
// *final_through advances (a group of weight gradients was flushed) or the last stage is done, and returns the stage to go on with in
// *stage_next.  One call per reported slice (~8 per backward at 12 + 12 blocks) instead of one per stage (27): the host side of the
// reference's own recipe, one clip per GPU under data parallelism, is what this shortens.
extern "C" int vt_tokenizer_backward_until_flush(vtTokenizer* t, const vtTokenizerTensors* P, const float* d_pred, const float* gscal, void* ws,
                                                 const vtTokenizerTensors* G, int32_t stage_begin, int32_t* stage_next, int32_t* final_through,
                                                 vtStream s) {
    VT_CHECK_ARG(t && stage_next && final_through, "vt_tokenizer_backward_until_flush: null pointer");
    const int nstage = vt_tokenizer_num_backward_stages(t);
    VT_CHECK_ARG(stage_begin >= 0 && stage_begin < nstage, "vt_tokenizer_backward_until_flush: bad first stage");
    const int before = stage_begin == 0 ? 0 : t->final_through;
    int st = stage_begin;
    int32_t ft = before;
    do {
        TRY(vt_tokenizer_backward(t, P, d_pred, gscal, ws, G, st, st + 1, &ft, s));
        ++st;
    } while (st < nstage && ft == before);
    *stage_next = st;
    *final_through = ft;
    return VT_OK;
}


// ------------------------------------------------------------------------------------------------
// A stack of timm Blocks on its own: transformer_encoder_parallel / transformer_encoder_fused called outside the
// tokenizer (models/transformer.py:8-70) and the discriminator's encoder (models/loss.py:150-155).  Same block
// machinery and workspace discipline as the tokenizer (the vtStack IS a vtTokenizer with only the block plan filled
// in); head_dim 64 or 32, any L.  One workspace per forward whose backward is still pending.
// ------------------------------------------------------------------------------------------------
extern "C" int vt_stack_create(const vtStackConfig* cfg, vtStack** out) {
    VT_CHECK_ARG(cfg && out, "vt_stack_create: null pointer");
    const int B = cfg->B, L = cfg->L, D = cfg->D, H = cfg->H, depth = cfg->depth;
    VT_CHECK_ARG(B > 0 && L > 0 && depth > 0 && H > 0 && D % H == 0 && (D / H == 64 || D / H == 32),
                 "vt_stack_create: B=%d L=%d D=%d H=%d depth=%d (head_dim must be 64 or 32)", B, L, D, H, depth);
    VT_CHECK_ARG(D == 128 || D == 384 || (D % 256 == 0 && D <= 1024), "vt_stack_create: width %d unsupported (128, 256, 384, 512, 768, 1024)", D);
    vtTokenizer* t = new vtTokenizer();
    memset(&t->c, 0, sizeof(t->c));
    t->c.B = B; t->c.D = D; t->c.H = H; t->c.depth_enc = depth; t->c.depth_dec = 0;
    t->Nv = 0; t->L = L;
    t->M = B * L; t->Mp = round_up(t->M, 128);
    t->Mv = t->Mvp = t->Mq = t->Mqp = t->Kp = 0;
    t->D3 = 3 * D; t->D4 = 4 * D;
    memset(&t->last_enc, 0, sizeof(t->last_enc));
    memset(&t->last_dec, 0, sizeof(t->last_dec));
    Arena a;
    const size_t Mp = t->Mp;
    plan_blocks(t, a, t->enc, depth);
    t->x_enc.resize(depth + 1);
    for (auto& x : t->x_enc) x = a.take(Mp * D * 4);
    t->dX = a.take(Mp * D * 4);
    for (auto& g : t->gs) {
        g.dx_out = a.take(Mp * D * 2); g.dx_mid = a.take(Mp * D * 2);
        g.du = a.take(Mp * t->D4 * 2); g.dqkv = a.take(Mp * t->D3 * 2);
        g.ln_part1 = a.take(vt_layernorm_bwd_workspace_bytes((int)D)); g.ln_part2 = a.take(vt_layernorm_bwd_workspace_bytes((int)D));
        g.cs_part = a.take((size_t)((t->M + 191) / 192) * t->D4 * 4);
    }
    t->dh = a.take(Mp * D * 2); t->dob = a.take(Mp * D * 2);
    t->delta = a.take((size_t)B * H * L * 4);
    t->splitk_bytes = vt_gemm_nt_splitk_workspace_bytes();
    t->splitk = a.take(t->splitk_bytes);
    t->ln_ws = a.take(vt_layernorm_bwd_workspace_bytes(D));
    t->cs_ws = a.take(vt_colsum_workspace_bytes(t->D4));
    t->cs_part = a.take((size_t)((t->M + 191) / 192) * t->D4 * 4);
    t->ws_bytes = a.off;
    *out = t;
    return VT_OK;
}

extern "C" void vt_stack_destroy(vtStack* t) { delete t; }
extern "C" size_t vt_stack_workspace_bytes(const vtStack* t) { return t ? t->ws_bytes : 0; }

extern "C" int vt_stack_init_workspace(vtStack* t, void* ws, vtStream stream) {
    VT_CHECK_ARG(t && ws, "vt_stack_init_workspace: null pointer");
    if (hipMemsetAsync(ws, 0, t->ws_bytes, (hipStream_t)stream) != hipSuccess) { vt_set_error("vt_stack_init_workspace: memset failed"); return VT_ERR_LAUNCH; }
    return VT_OK;
}

extern "C" int vt_stack_forward(vtStack* t, const vtBlockTensors* blocks, const float* x_in, void* ws, float* x_out, vtStream s) {
    VT_CHECK_ARG(t && blocks && x_in && ws && x_out, "vt_stack_forward: null pointer");
    t->in_backward = false;
    const int depth = t->c.depth_enc;
    const size_t bytes = (size_t)t->M * t->c.D * 4;
    hipStream_t hs = (hipStream_t)s;
    TRY(copy_d2d(WS(void, t->x_enc[0]), x_in, bytes, hs));
    TRY(pack_blocks(t, t->enc, blocks, ws, s));
    for (int i = 0; i < depth; ++i)
        TRY(block_forward(t, t->enc[i], blocks[i], WS(float, t->x_enc[i]), WS(float, t->x_enc[i + 1]), ws, s));
    TRY(copy_d2d(x_out, WS(void, t->x_enc[depth]), bytes, hs));
    VT_CHECK_LAUNCH("vt_stack_forward");
    return VT_OK;
}

extern "C" int vt_stack_backward(vtStack* t, const vtBlockTensors* blocks, const float* dy, void* ws, const vtBlockTensors* grads,
                                 float* dx, int32_t need_wgrad, vtStream s) {
    VT_CHECK_ARG(t && blocks && dy && ws && grads && dx, "vt_stack_backward: null pointer");
    t->in_backward = true;
    const int depth = t->c.depth_enc, D = t->c.D;
    const size_t bytes = (size_t)t->M * D * 4;
    const vtRowMap id = {0, 0, 0};
    hipStream_t hs = (hipStream_t)s;
    float* dX = WS(float, t->dX);
    t->pending.clear();
    t->pending_red.clear(); t->pending_blocks = 0; t->set_idx = 0; t->final_through = 0;
    TRY(copy_d2d(dX, dy, bytes, hs));
    TRY(vt_cast_rows(dX, id, t->M, D, WS(void, t->gs[0].dx_out), D, s));
    TRY(vt_colsum(dX, 0, D, id, t->M, D, grads[depth - 1].fc2_b, WS(void, t->cs_ws), s));
    for (int i = depth - 1; i >= 0; --i) {
        TRY(block_backward(t, t->enc[i], blocks[i], grads[i], WS(float, t->x_enc[i]), i > 0 ? grads[i - 1].fc2_b : nullptr, ws, s));
        if (!need_wgrad) {  // frozen stack (generator-side pass through the discriminator): input gradient only
            t->pending.clear();      // (the parameter gradients of a frozen stack are not wanted either: drop the queued reductions)
            t->pending_red.clear();
            t->pending_blocks = 0;
        } else if (t->pending_blocks >= t->wg_batch || i == 0) {
            TRY(flush_wgrads(t, 0, s));
        }
    }
    TRY(copy_d2d(dx, dX, bytes, hs));
    VT_CHECK_LAUNCH("vt_stack_backward");
    return VT_OK;
}
