// api.hip -- C-ABI of libmdgen_amd.so (include/mdgen_amd.h): context, weight packing, workspace
// layout, the denoiser forward / Euler rollout orchestration and hipGraph capture.
#include "../../include/mdgen_amd.h"
#include "kernels.h"

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <climits>
#include <functional>
#include <map>
#include <set>
#include <string>
#include <vector>

using namespace mdg;

// ---------------------------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
static int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
// Plan mode (mdgen_debug_dispatch_plan): the orchestration code below runs on the host ONLY -- every HIP call and every launch is
// skipped, and each launch site's profile class is appended to *g_dry instead.  The plan is therefore the product's own dispatch
// logic, not a restatement of it.
static thread_local std::vector<std::string>* g_dry = nullptr;
#define HIPCHK(expr)                                                                           \
    do {                                                                                       \
        if (g_dry) break;                                                                      \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) return fail((int)e_, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)
#define LAUNCHCHK()                                                                            \
    do {                                                                                       \
        if (g_dry) break;                                                                      \
        hipError_t e_ = hipGetLastError();                                                     \
        if (e_ != hipSuccess) return fail((int)e_, "kernel launch failed: %s (%s:%d)", hipGetErrorString(e_), __FILE__, __LINE__); \
        if (const char* m_ = k32_take_launch_error()) return fail(-7, "internal: %s (%s:%d)", m_, __FILE__, __LINE__); \
    } while (0)

extern "C" const char* mdgen_last_error(void) { return g_err; }
extern "C" int32_t mdgen_abi_version(void) { return MDGEN_ABI_VERSION; }
// 0 for a product library; 1 when this .so was built with an experiment switch (csrc/dev.h) -- its results may be wrong
extern "C" int32_t mdgen_dev_build(void) {
#ifdef MDGEN_DEV_BUILD
    return 1;
#else
    return 0;
#endif
}

// ---------------------------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------------------------
constexpr int kMaxPos = 8160;        // positions covered by the rotary table / flash mask table
constexpr int kKS = 24;              // k-steps at K = 384
constexpr size_t kPackCC = (size_t)12 * kKS * 64;  // bf16x8 elements of a packed [384][384] matrix

struct MhaW {
    bf16x8 *wq = nullptr, *wk = nullptr, *wv_flash = nullptr, *wv_small = nullptr, *wo = nullptr;
    bf16x8* wo_stream = nullptr;   // W_o as the 288-fragment prefix of the row-owner MLP kernel's weight stream (proj_stream_table)
    float *bq = nullptr, *bk = nullptr, *bv_flash = nullptr, *bv_small = nullptr, *bo = nullptr;
    float *bias_k = nullptr, *bias_v = nullptr;
};
struct FfnW {
    bf16x8 *w1 = nullptr, *w2 = nullptr;
    float *b1 = nullptr, *b2 = nullptr;
    bf16x8* wstream = nullptr;   // both matrices as ONE fragment stream in the consumption order of k_mlp_rows (mlp_stream_table)
    float* w2f = nullptr;        // trunk layers: fp32 fc2.weight [384][1536], the source of the per-step gate-folded streams (option mlp_fold)
};
struct TrunkW {
    MhaW mha_l, mha_t;
    FfnW ffn;
};
struct IpaW {
    float* gamma_beta = nullptr;  // [gamma(C) | beta(C)]
    bf16x8* wproj = nullptr;      // [21 ftile][24][64][8]
    float* bproj = nullptr;       // [672]
    float* head_w = nullptr;      // [4]
    bf16x8* wout = nullptr;       // [12 ftile][16][64][8]
    float* bout = nullptr;
    MhaW mha_l;
    FfnW ffn;
};

struct ProfRec {
    const char* cls;
    hipEvent_t a, b;
};

struct GraphEntry {
    std::vector<uint64_t> key;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
};

struct mdgen_ctx {
    mdgen_model_desc d;
    int nl = 0, D = 0, modrow = 0;
    std::vector<void*> allocs;
    std::vector<std::string> names;
    std::map<std::string, std::function<int(const float*, const int64_t*, int, hipStream_t)>> setters;
    std::map<std::string, bool> provided;
    bool finalized = false;
    // fp32 small weights
    float *wl = nullptr, *bl = nullptr, *wc = nullptr, *bc = nullptr, *mask_emb = nullptr, *aa_emb = nullptr;
    float *wl_pack = nullptr, *wc_pack = nullptr;   // latent_to_emb / cond_to_emb in k_embed's operand order (launch_pack_embed)
    float *wl_rows = nullptr, *wc_rows = nullptr;   // ... and in rows_embed_gemm's (the embedding as the tail of the last layer's MLP kernel)
    float* mask_delta = nullptr;                    // mask_to_emb[1] - mask_to_emb[0]
    bf16x8 *wl_hi = nullptr, *wl_lo = nullptr, *wc_hi = nullptr, *wc_lo = nullptr;   // ... as bf16 pairs, K padded to 32 in kappa order
    int opt_trace_tail = 0;     // measurement: mdgen_profile_phase_trace targets the row-owner MLP launch that carries both tails
    int opt_embed_split = 1;    // the embedding tail's products on the bf16 MFMA with hi + lo operand pairs (0: fp32 MFMA, exact)
    float *pos_embed = nullptr, *t_w0 = nullptr, *t_b0 = nullptr, *t_w2 = nullptr, *t_b2 = nullptr;
    float *wf7 = nullptr, *bf7 = nullptr, *wr7 = nullptr, *br7 = nullptr;
    float *ada_w = nullptr, *ada_b = nullptr;
    float *inv_freq = nullptr, *rope = nullptr;
    bf16x8* wfin = nullptr;
    bf16x8* wfin_k = nullptr;   // the same tile with the K dimension in rows.h's kappa order (k_mlp_rows<., TAIL>)
    float* bfin = nullptr;
    std::vector<TrunkW> trunk;
    std::vector<IpaW> ipa;
    // device index maps
    int *map_nat = nullptr, *map_qk = nullptr, *map_vflash = nullptr, *map_vsmall = nullptr, *map_fin = nullptr;
    int *perm_qk = nullptr, *perm_vsmall = nullptr;
    int* mlp_tab = nullptr;     // device copy of mlp_stream_table()
    int* proj_tab = nullptr;    // device copy of proj_stream_table()
    std::vector<GraphEntry> graphs;
    bool inv_freq_set = false;
    bool prof_on = false;
    // run-time options (mdgen_ctx_set_option)
    int opt_precision = 16;     // GEMM / attention operand precision: 16 = bf16 MFMA path, 32 = fp32 path (k_fp32.hip)
    int opt_keep_fp32 = 0;      // keep an fp32 copy of every weight handed over (required by precision 32)
    std::map<std::string, float*> w32;   // fp32 copies, natural layout, keyed by the reference's state_dict key
    std::map<std::string, bool> w32_bound;   // keys whose w32 entry points into the caller's flat parameter buffer (mdgen_train_bind_params)
    int opt_streams = 2;        // concurrent sub-batch streams of the Euler rollout (1 = caller's stream only)
    bool opt_streams_auto = true;   // not set by the caller: the count also follows the fill of the chip (n_streams)
    int opt_attn_path = 0;      // tiled attention: 0 fixed-anchor fast loop with overflow check + fallback, 1 robust loop always
    int opt_mlp_path = 1;       // MLP block: 0 resident-panel kernel (k_mlp), 1 row-owner kernel (k_mlp_rows) when the launch
                                // fills the chip, 2 row-owner kernel always
    int opt_fuse_proj_qkv = 1;  // tiled residue axis (L > 8): its out-projection + gated residual runs inside the temporal q / k / v kernel
    int opt_fuse_proj = 3;      // the temporal attention's out-projection inside the MLP kernel: 0 off, 1 row-owner kernel (same wall time),
                                // 2 panel kernel always, 3 (default) panel kernel where the launch takes the panel kernel anyway (small N: +2 %)
    int opt_panel_waves = 0;    // 64-row panel kernels with a four- and an eight-wave form (k_mlp / k_mlp8, k_ln_qkv<false> / k_ln_qkv8): 0 (default)
                                // eight waves where a launch is at most one workgroup per CU, 4 / 8 force one form (tests, A/B runs)
    int ncu = 256;              // compute units of the device the context was created on (hipDeviceAttributeMultiprocessorCount)
    int opt_small_split = 1;    // launches far below one workgroup per CU (B = 1, the IPA stack): a panel's work over several workgroups --
                                // k_mlp8<., kMlpSplit> (hidden chunks over 3 workgroups, last arriver finishes; panels <= ncu / 3) and
                                // k_ln_qkv8<true> (q, k | v over 2 workgroups; panels <= ncu / 2).  0 off, 1 (default) on
    bool xcd_round_robin = false;   // placement probe: workgroups with equal blockIdx % 8 share an XCD (k_mlp8's split form relies on it)
    long n_split_launches = 0;  // k_mlp8<., kMlpSplit> launches enqueued (or captured) since the last mdgen_profile_report
    int live_streams = 1;       // sub-batch streams of the call being recorded / run (the workspace's split scratch serves one launch at a time)
    int opt_flash_rotate = 1;   // tiled attention: the 64-query chunks of a sequence start their walk over the key tiles at different tiles (k_flash.hip)
    int opt_flash_proj_form = 0;   // ... 0 (default): k_flash_proj8 (eight waves, 128-row panel, four query tiles per wave) for sequences of >= 512
                                   // positions whose launch gives every CU such a workgroup (cfg-2), else k_flash_proj (four waves, 64-row
                                   // panel; ATLAS); 4 / 8: one form always
    int opt_flash_proj = 1;     // tiled attention + its out-projection + gated residual in ONE launch (k_flash_proj): 0 off (k_flash, then
                                // k_proj<0> or a deferred projection), 1 (default) when the launch has >= 2 x CUs workgroups
                                // of (sequence, 64 queries), 2 always
    int opt_train_precision = 32;   // operands of the training step's linear layers / weight gradients: 32 exact fp32, 16 bf16 MFMA
    int opt_train_attn_form = 1;    // training step, bf16 operands: 1 = axes of 129 .. 256 positions take the sequence-resident attention kernels (k_attn16.hip)
    int opt_train_defer_gate = 1;   // training step, trunk forward: 1 = a sub-layer's gated residual update is formed by the next sub-layer's LayerNorm launch
    int opt_train_y_bf16 = 1;       // training step, bf16 operands: the trunk's taped LayerNorm outputs are stored as bf16 rows (GEMM operands only)
    int opt_train_dqkv_bf16 = 1;    // training step, bf16 operands: the sequence-resident attention backward writes dq | dk | dv as bf16 rows
    int opt_train_du_bf16 = 1;      // training step, bf16 operands: the gated gradient du = gate * dh of a trunk sub-layer is stored as bf16 rows
    int opt_train_dhid_bf16 = 1;    // training step, bf16 operands: d pre = d hid * gelu'(pre) of a trunk MLP is stored as bf16 rows
    int opt_train_streams = 2;      // training step: 2 = weight / bias gradients of the linear layers on a second stream (train.inc)
    hipStream_t train_side = nullptr;   // that stream (created on first use, default priority)
    // Turned weights of the small launches' dX products (train.inc `turned`): the requests of one call in order, their images in
    // tr_buf, computed on the second stream at the start of the next call with the same requests
    struct TurnReq { const float* w[3]; int nseg, rows, cols; size_t off; };
    std::vector<TurnReq> tr_plan;
    float* tr_buf = nullptr;
    size_t tr_buf_floats = 0;
    bool tr_plan_ok = false;
    int opt_train_turn_ahead = 1;       // 1: those images are computed ahead on the second stream (needs train_streams 2)
    std::vector<hipEvent_t> train_ev;   // event pool of that fork / join traffic (created on first use, round-robin)
    size_t train_ev_next = 0;
    int opt_mlp_fold = 1;       // sampling (t shared by the batch): the MLP gate folded into per-(step, layer) fc2 streams, k_mlp_rows starts its
                                // accumulators from the residual rows and only stores (one HBM read of the rows instead of two)
    int opt_mlp_tail = 2;       // ... and the FinalLayer + Euler update run inside the last layer's (folded) MLP kernel, which then does not store
                                // its rows: no k_final launch, 196 MB less traffic per network evaluation
    int opt_residue_l4 = 2;     // residue axis, L == 4: 0 general L <= 8 path, 1 attention fused, 2 whole sub-layer fused
    std::vector<void*> milestone_events;          // mdgen_train_set_milestone_events (hipEvent_t handles, caller-owned)
    unsigned long long* phase_trace = nullptr;   // mdgen_profile_phase_trace target (device), consumed by one launch
    long phase_trace_cap = 0;
    std::vector<ProfRec> prof;
    static constexpr int kMaxSide = 7;
    hipStream_t side[kMaxSide] = {};
    hipEvent_t ev_fork = nullptr, ev_join[kMaxSide] = {};

    std::set<std::string> cls_names;   // storage of composed profile class names (ProfRec keeps a const char*)
    const char* intern(const std::string& n) { return cls_names.insert(n).first->c_str(); }
    template <typename T>
    int dalloc(T** p, size_t count) {
        void* q = nullptr;
        hipError_t e = hipMalloc(&q, count * sizeof(T));
        if (e != hipSuccess) return fail((int)e, "hipMalloc(%zu) failed: %s", count * sizeof(T), hipGetErrorString(e));
        allocs.push_back(q);
        *p = (T*)q;
        return 0;
    }
    int trunk_off(int i) const { return i * 9 * kC; }
    int ipa_off(int i) const { return nl * 9 * kC + i * 6 * kC; }
    int final_off() const { return nl * 15 * kC; }
};

// RAII bracket of one launch with a hipEvent pair on the launch stream (profiling mode only)
struct ProfScope {
    mdgen_ctx* c;
    hipStream_t s;
    ProfRec r;
    bool on;
    ProfScope(mdgen_ctx* c_, const char* cls, hipStream_t s_) : c(c_), s(s_), on(c_->prof_on) {
        if (g_dry) {   // plan mode: the class is the record
            g_dry->push_back(cls);
            on = false;
            return;
        }
        if (on) {
            r.cls = cls;
            on = hipEventCreate(&r.a) == hipSuccess && hipEventCreate(&r.b) == hipSuccess &&
                 hipEventRecord(r.a, s) == hipSuccess;
        }
    }
    ~ProfScope() {
        if (on && hipEventRecord(r.b, s) == hipSuccess) c->prof.push_back(r);
    }
};

// ---- host-side index maps (DESIGN.md "fragment layout") -------------------------------------
// Row rho (0..95) of wave-tile w of a TRANSPOSED projection -> (head, lane-half h, slot e):
//   ft = rho/32, m = rho%32 = b + 8a + 4h, a' = 4ft + a, head = 4w + a'/3, e = 4(a'%3) + b
static void decode_T(int rho, int& hd, int& h, int& e) {
    const int ft = rho / 32, m = rho % 32;
    const int b = m & 3, hh = (m >> 2) & 1, a = m >> 3;
    const int ap = 4 * ft + a;
    hd = ap / 3;
    h = hh;
    e = 4 * (ap % 3) + b;
}
// rotary-pair order for Q/K: slot e of half h is feature 6h + e/2 (+12 for odd e)
static int feat_qk(int w, int rho) {
    int hd, h, e;
    decode_T(rho, hd, h, e);
    return (4 * w + hd) * kDH + 6 * h + (e >> 1) + 12 * (e & 1);
}
// natural order for the SMALL-layout V: slot e of half h is feature 12h + e
static int feat_vsmall(int w, int rho) {
    int hd, h, e;
    decode_T(rho, hd, h, e);
    return (4 * w + hd) * kDH + 12 * h + e;
}
// FLASH-layout V (non-transposed): column col (0..95) of wave-tile w -> head 4w + col/24, V^T row
// d = col%24 carries feature psi(d) = 12*((d>>2)&1) + 4*(d>>3) + (d&3)
static int feat_vflash(int w, int col) {
    const int hd = col / kDH, d = col % kDH;
    return (4 * w + hd) * kDH + 12 * ((d >> 2) & 1) + 4 * (d >> 3) + (d & 3);
}

static void build_maps(std::vector<int>& qk, std::vector<int>& vf, std::vector<int>& vs, std::vector<int>& pqk,
                       std::vector<int>& pvs) {
    qk.assign(kC, 0); vf.assign(kC, 0); vs.assign(kC, 0); pqk.assign(kC, 0); pvs.assign(kC, 0);
    for (int w = 0; w < 4; ++w)
        for (int r = 0; r < 96; ++r) {
            qk[w * 96 + r] = feat_qk(w, r);
            vs[w * 96 + r] = feat_vsmall(w, r);
            vf[w * 96 + r] = feat_vflash(w, r);
        }
    // lane-order bias permutations: index ((w*2+h)*4+hd)*12 + e
    for (int w = 0; w < 4; ++w)
        for (int h = 0; h < 2; ++h)
            for (int hd = 0; hd < 4; ++hd)
                for (int e = 0; e < 12; ++e) {
                    const int i = ((w * 2 + h) * 4 + hd) * 12 + e;
                    pqk[i] = (4 * w + hd) * kDH + 6 * h + (e >> 1) + 12 * (e & 1);
                    pvs[i] = (4 * w + hd) * kDH + 12 * h + e;
                }
}

extern "C" int32_t mdgen_debug_layout_maps(int32_t* map_qk, int32_t* map_vflash, int32_t* map_vsmall,
                                           int32_t* perm_qk, int32_t* perm_vsmall) {
    if (!map_qk || !map_vflash || !map_vsmall || !perm_qk || !perm_vsmall) return fail(-1, "null argument");
    std::vector<int> qk, vf, vs, pqk, pvs;
    build_maps(qk, vf, vs, pqk, pvs);
    for (int i = 0; i < kC; ++i) {
        map_qk[i] = qk[i]; map_vflash[i] = vf[i]; map_vsmall[i] = vs[i]; perm_qk[i] = pqk[i]; perm_vsmall[i] = pvs[i];
    }
    return 0;
}

// Weight stream of k_mlp_rows (csrc/k_rows.hip): 2304 fragments, entry = mat << 16 | row tile << 8 | k-step with mat 0 = fc1
// (48 hidden tiles x 24 k-steps), 1 = fc2 (12 feature tiles x 96 k-steps).  Chunk c = hidden units 64 c .. 64 c + 63 =
// fc1 tiles 2c, 2c + 1 = fc2 k-steps 4c .. 4c + 3.  Order = the kernel's software pipeline:
//   [X(0)] [X(1)] { X(c+1) ks 0-5 | Y(c-1) kk 0 | X ks 6-11 | Y kk 1 | X ks 12-17 | Y kk 2 | X ks 18-23 | Y kk 3 } c = 1..22 [Y(22)] [Y(23)]
// X block: (k-step, tile) pairs, tile fastest; Y block: the 12 feature tiles of one k-step.
static std::vector<int> mlp_stream_table() {
    std::vector<int> t;
    auto xblock = [&](int c, int kx) {
        for (int q = 0; q < 12; ++q) t.push_back(0 << 16 | (2 * c + (q & 1)) << 8 | (6 * kx + (q >> 1)));
    };
    auto yblock = [&](int c, int kk) {
        for (int ft = 0; ft < 12; ++ft) t.push_back(1 << 16 | ft << 8 | (4 * c + kk));
    };
    for (int c = 0; c < 2; ++c)
        for (int kx = 0; kx < 4; ++kx) xblock(c, kx);
    for (int c = 1; c < 23; ++c)
        for (int b = 0; b < 4; ++b) {
            xblock(c + 1, b);
            yblock(c - 1, b);
        }
    for (int c = 22; c < 24; ++c)
        for (int kk = 0; kk < 4; ++kk) yblock(c, kk);
    return t;
}
constexpr int kMlpFrags = 2304;
// Out-projection prefix of the fused kernel (k_mlp_rows<NW, true>): 288 fragments, k-step major (one block = the 12 feature
// tiles of one k-step), natural k order inside a fragment.  Entry = 2 << 16 | feature tile << 8 | k-step.
static std::vector<int> proj_stream_table() {
    std::vector<int> t;
    for (int ks = 0; ks < 24; ++ks)
        for (int ft = 0; ft < 12; ++ft) t.push_back(2 << 16 | ft << 8 | ks);
    return t;
}
constexpr int kProjFrags = 288;
extern "C" int32_t mdgen_debug_mlp_stream_table(int32_t* out, int32_t capacity) {
    const std::vector<int> t = mlp_stream_table();
    if (!out || capacity < (int)t.size()) return fail(-1, "need room for %d entries", (int)t.size());
    for (size_t i = 0; i < t.size(); ++i) out[i] = t[i];
    return (int32_t)t.size();
}

static int upload_ints(mdgen_ctx* c, int** dst, const std::vector<int>& v) {
    if (int r = c->dalloc(dst, v.size())) return r;
    HIPCHK(hipMemcpy(*dst, v.data(), v.size() * sizeof(int), hipMemcpyHostToDevice));
    return 0;
}

static bool shape_is(const int64_t* s, int nd, std::initializer_list<int64_t> want) {
    // accept exact shape, ignoring leading singleton dims (bias_k is (1,1,C), pos_embed (1,crop,C))
    std::vector<int64_t> a(s, s + nd), b(want);
    while (a.size() > b.size() && a.front() == 1) a.erase(a.begin());
    return a == b;
}

#define SETTER(key, body)                                                                        \
    c->names.push_back(key);                                                                     \
    c->setters[key] = [=](const float* data, const int64_t* shp, int nd, hipStream_t s) -> int { \
        (void)shp; (void)nd; body; return 0; }
#define WANT(...) \
    if (!shape_is(shp, nd, {__VA_ARGS__})) return fail(-3, "unexpected shape for weight")

static int copy_f32(float* dst, const float* src, size_t n, hipStream_t s) {
    HIPCHK(hipMemcpyAsync(dst, src, n * sizeof(float), hipMemcpyDeviceToDevice, s));
    return 0;
}

static int register_mha(mdgen_ctx* c, const std::string& pre, MhaW* m) {
    const float qscale = (1.0f / std::sqrt((float)kDH)) * kLog2e;
    if (int r = c->dalloc(&m->wq, kPackCC)) return r;
    if (int r = c->dalloc(&m->wk, kPackCC)) return r;
    if (int r = c->dalloc(&m->wv_flash, kPackCC)) return r;
    if (int r = c->dalloc(&m->wv_small, kPackCC)) return r;
    if (int r = c->dalloc(&m->wo, kPackCC)) return r;
    for (float** p : {&m->bq, &m->bk, &m->bv_flash, &m->bv_small, &m->bo, &m->bias_k, &m->bias_v})
        if (int r = c->dalloc(p, (size_t)kC)) return r;
    SETTER(pre + "q_proj.weight", {
        WANT(kC, kC);
        launch_pack_rows(data, kC, c->map_qk, 12, kKS, qscale, m->wq, s);
    });
    SETTER(pre + "q_proj.bias", { WANT(kC); launch_gather_f32(data, c->perm_qk, qscale, m->bq, kC, s); });
    SETTER(pre + "k_proj.weight", {
        WANT(kC, kC);
        launch_pack_rows(data, kC, c->map_qk, 12, kKS, 1.f, m->wk, s);
    });
    SETTER(pre + "k_proj.bias", { WANT(kC); launch_gather_f32(data, c->perm_qk, 1.f, m->bk, kC, s); });
    SETTER(pre + "v_proj.weight", {
        WANT(kC, kC);
        launch_pack_rows(data, kC, c->map_vflash, 12, kKS, 1.f, m->wv_flash, s);
        launch_pack_rows(data, kC, c->map_vsmall, 12, kKS, 1.f, m->wv_small, s);
    });
    SETTER(pre + "v_proj.bias", {
        WANT(kC);
        launch_gather_f32(data, c->map_vflash, 1.f, m->bv_flash, kC, s);
        launch_gather_f32(data, c->perm_vsmall, 1.f, m->bv_small, kC, s);
    });
    if (int r = c->dalloc(&m->wo_stream, (size_t)kProjFrags * 64)) return r;
    SETTER(pre + "out_proj.weight", {
        WANT(kC, kC);
        launch_pack_rows(data, kC, c->map_nat, 12, kKS, 1.f, m->wo, s);
        launch_pack_stream(data, kC, 2, c->proj_tab, kProjFrags, 1.f, 0, m->wo_stream, s);
    });
    SETTER(pre + "out_proj.bias", { WANT(kC); if (int r = copy_f32(m->bo, data, kC, s)) return r; });
    SETTER(pre + "bias_k", { WANT(kC); if (int r = copy_f32(m->bias_k, data, kC, s)) return r; });
    SETTER(pre + "bias_v", { WANT(kC); if (int r = copy_f32(m->bias_v, data, kC, s)) return r; });
    SETTER(pre + "rot_emb.inv_freq", {
        WANT(12);
        if (int r = copy_f32(c->inv_freq, data, 12, s)) return r;
        c->inv_freq_set = true;
    });
    return 0;
}

static int register_ffn(mdgen_ctx* c, const std::string& pre, FfnW* f, bool trunk) {
    if (trunk)
        if (int r = c->dalloc(&f->w2f, (size_t)kC * kF)) return r;
    if (int r = c->dalloc(&f->w1, (size_t)48 * kKS * 64)) return r;
    if (int r = c->dalloc(&f->w2, (size_t)12 * 96 * 64)) return r;
    if (int r = c->dalloc(&f->b1, (size_t)kF)) return r;
    if (int r = c->dalloc(&f->b2, (size_t)kC)) return r;
    if (int r = c->dalloc(&f->wstream, (size_t)kMlpFrags * 64)) return r;
    SETTER(pre + "fc1.weight", {
        WANT(kF, kC);
        launch_pack_rows(data, kC, c->map_nat, 48, kKS, 1.f, f->w1, s);
        launch_pack_stream(data, kC, 0, c->mlp_tab, kMlpFrags, 1.f, 1, f->wstream, s);
    });
    SETTER(pre + "fc1.bias", { WANT(kF); if (int r = copy_f32(f->b1, data, kF, s)) return r; });
    SETTER(pre + "fc2.weight", {
        WANT(kC, kF);
        launch_pack_rows(data, kF, c->map_nat, 12, 96, 1.f, f->w2, s);
        launch_pack_stream(data, kF, 1, c->mlp_tab, kMlpFrags, 1.f, 1, f->wstream, s);
        if (f->w2f)
            if (int r = copy_f32(f->w2f, data, (size_t)kC * kF, s)) return r;
    });
    SETTER(pre + "fc2.bias", { WANT(kC); if (int r = copy_f32(f->b2, data, kC, s)) return r; });
    return 0;
}

extern "C" int32_t mdgen_ctx_create(mdgen_ctx** out, const mdgen_model_desc* d) {
    if (!out || !d) return fail(-1, "null argument");
    if (d->embed_dim != kC || d->mha_heads != kH) return fail(-2, "this build supports embed_dim=384, mha_heads=16 only");
    if (d->ipa_heads != 4 || d->ipa_head_dim != 32 || d->ipa_qk != 8 || d->ipa_v != 8)
        return fail(-2, "this build supports ipa_heads=4, ipa_head_dim=32, ipa_qk=ipa_v=8 only");
    if (d->num_layers < 1 || d->num_layers > 8) return fail(-2, "num_layers must be in 1..8");
    if (d->latent_dim != 21 && d->latent_dim != 28)   // wrapper.py:57-60: 21, or 28 with two-sided conditioning
        return fail(-2, "latent_dim must be 21 (forward simulation) or 28 (two-sided conditioning)");
    if (d->tps_condition && d->latent_dim != 28) return fail(-2, "tps_condition requires latent_dim 28");
    int ndev = 0;
    HIPCHK(hipGetDeviceCount(&ndev));
    if (ndev < 1) return fail(-9, "no HIP device");
    mdgen_ctx* c = new mdgen_ctx();
    {   // "one workgroup per CU" thresholds (eight-wave panel kernels, stream count) follow the device, not a constant
        int dev = 0, ncu = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && ncu > 0)
            c->ncu = ncu;
    }
    {   // placement probe for k_mlp8's split form: do workgroups with the same blockIdx % 8 run on the same XCD?
        // (a performance hint since round 6: the hand-over is a release / acquire pair at agent scope.)  Private non-blocking stream,
        // no legacy-stream launch; any error is consumed here and reads as "rule does not hold".
        int* dp = nullptr;
        int hx[64];
        hipStream_t ps = nullptr;
        if (hipStreamCreateWithFlags(&ps, hipStreamNonBlocking) == hipSuccess) {
            if (hipMalloc((void**)&dp, sizeof(hx)) == hipSuccess) {
                launch_xcc_probe(dp, 64, ps);
                if (hipGetLastError() == hipSuccess &&
                    hipMemcpyAsync(hx, dp, sizeof(hx), hipMemcpyDeviceToHost, ps) == hipSuccess &&
                    hipStreamSynchronize(ps) == hipSuccess) {
                    bool ok = true;
                    for (int i = 8; i < 64; ++i) ok = ok && hx[i] == hx[i & 7];
                    c->xcd_round_robin = ok;
                }
                (void)hipFree(dp);
            }
            (void)hipStreamDestroy(ps);
        }
        (void)hipGetLastError();   // nothing sticky is left for the next LAUNCHCHK
    }
    c->d = *d;
    c->nl = d->num_layers;
    c->D = d->latent_dim;
    c->modrow = (15 * c->nl + 2) * kC;
    const int D = c->D, nl = c->nl;
    // index maps
    std::vector<int> nat(kF), qk, vf, vs, fin(32), pqk, pvs;
    for (int i = 0; i < kF; ++i) nat[i] = i;
    build_maps(qk, vf, vs, pqk, pvs);
    for (int i = 0; i < 32; ++i) fin[i] = i < D ? i : -1;
#define TRY(x) do { if (int r_ = (x)) { mdgen_ctx_destroy(c); return r_; } } while (0)
    TRY(upload_ints(c, &c->map_nat, nat));
    TRY(upload_ints(c, &c->map_qk, qk));
    TRY(upload_ints(c, &c->map_vflash, vf));
    TRY(upload_ints(c, &c->map_vsmall, vs));
    TRY(upload_ints(c, &c->map_fin, fin));
    TRY(upload_ints(c, &c->mlp_tab, mlp_stream_table()));
    TRY(upload_ints(c, &c->proj_tab, proj_stream_table()));
    TRY(upload_ints(c, &c->perm_qk, pqk));
    TRY(upload_ints(c, &c->perm_vsmall, pvs));
    TRY(c->dalloc(&c->wl, (size_t)kC * D));
    TRY(c->dalloc(&c->wl_pack, (size_t)kEmbPackFloats));
    TRY(c->dalloc(&c->wc_pack, (size_t)kEmbPackFloats));
    TRY(c->dalloc(&c->wl_rows, (size_t)kEmbRowsFloats));
    TRY(c->dalloc(&c->wc_rows, (size_t)kEmbRowsFloats));
    TRY(c->dalloc(&c->mask_delta, (size_t)kC));
    for (bf16x8** q : {&c->wl_hi, &c->wl_lo, &c->wc_hi, &c->wc_lo}) TRY(c->dalloc(q, (size_t)12 * 2 * 64));
    TRY(c->dalloc(&c->bl, (size_t)kC));
    TRY(c->dalloc(&c->wc, (size_t)kC * D));
    TRY(c->dalloc(&c->bc, (size_t)kC));
    TRY(c->dalloc(&c->mask_emb, (size_t)2 * kC));
    TRY(c->dalloc(&c->aa_emb, (size_t)21 * kC));
    TRY(c->dalloc(&c->t_w0, (size_t)kC * 256));
    TRY(c->dalloc(&c->t_b0, (size_t)kC));
    TRY(c->dalloc(&c->t_w2, (size_t)kC * kC));
    TRY(c->dalloc(&c->t_b2, (size_t)kC));
    TRY(c->dalloc(&c->ada_w, (size_t)c->modrow * kC));
    TRY(c->dalloc(&c->ada_b, (size_t)c->modrow));
    TRY(c->dalloc(&c->inv_freq, (size_t)12));
    TRY(c->dalloc(&c->rope, (size_t)(kMaxPos + 1) * kRopeRow));
    // a failing HIP call must not leak the half-built context
#define TRYHIP(expr)                                                                                    \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess) {                                                                         \
            mdgen_ctx_destroy(c);                                                                       \
            return fail((int)e_, "%s failed: %s", #expr, hipGetErrorString(e_));                        \
        }                                                                                               \
    } while (0)
    for (int i = 0; i < mdgen_ctx::kMaxSide; ++i) {
        TRYHIP(hipStreamCreateWithFlags(&c->side[i], hipStreamNonBlocking));
        TRYHIP(hipEventCreateWithFlags(&c->ev_join[i], hipEventDisableTiming));
    }
    TRYHIP(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
    TRY(c->dalloc(&c->wfin, (size_t)kKS * 64));
    TRY(c->dalloc(&c->wfin_k, (size_t)kKS * 64));
    TRY(c->dalloc(&c->bfin, (size_t)32));
    TRYHIP(hipMemset(c->bfin, 0, 32 * sizeof(float)));
#undef TRYHIP
    SETTER("latent_to_emb.weight", { WANT(kC, c->D); if (int r = copy_f32(c->wl, data, (size_t)kC * c->D, s)) return r; launch_pack_embed(c->wl, c->D, c->wl_pack, s); launch_pack_embed_rows(c->wl, c->D, c->wl_rows, s);
        launch_pack_rows(c->wl, c->D, c->map_nat, 12, 2, 1.f, c->wl_hi, s, 1, 0); launch_pack_rows(c->wl, c->D, c->map_nat, 12, 2, 1.f, c->wl_lo, s, 1, 1); });
    SETTER("latent_to_emb.bias", { WANT(kC); if (int r = copy_f32(c->bl, data, kC, s)) return r; });
    SETTER("cond_to_emb.weight", { WANT(kC, c->D); if (int r = copy_f32(c->wc, data, (size_t)kC * c->D, s)) return r; launch_pack_embed(c->wc, c->D, c->wc_pack, s); launch_pack_embed_rows(c->wc, c->D, c->wc_rows, s);
        launch_pack_rows(c->wc, c->D, c->map_nat, 12, 2, 1.f, c->wc_hi, s, 1, 0); launch_pack_rows(c->wc, c->D, c->map_nat, 12, 2, 1.f, c->wc_lo, s, 1, 1); });
    SETTER("cond_to_emb.bias", { WANT(kC); if (int r = copy_f32(c->bc, data, kC, s)) return r; });
    SETTER("mask_to_emb.weight", { WANT(2, kC); if (int r = copy_f32(c->mask_emb, data, 2 * kC, s)) return r; launch_sub_f32(c->mask_emb + kC, c->mask_emb, c->mask_delta, kC, s); });
    SETTER("aatype_to_emb.weight", { WANT(21, kC); if (int r = copy_f32(c->aa_emb, data, 21 * kC, s)) return r; });
    SETTER("t_embedder.mlp.0.weight", { WANT(kC, 256); if (int r = copy_f32(c->t_w0, data, (size_t)kC * 256, s)) return r; });
    SETTER("t_embedder.mlp.0.bias", { WANT(kC); if (int r = copy_f32(c->t_b0, data, kC, s)) return r; });
    SETTER("t_embedder.mlp.2.weight", { WANT(kC, kC); if (int r = copy_f32(c->t_w2, data, (size_t)kC * kC, s)) return r; });
    SETTER("t_embedder.mlp.2.bias", { WANT(kC); if (int r = copy_f32(c->t_b2, data, kC, s)) return r; });
    if (d->abs_pos_emb) {
        if (d->crop < 1) { mdgen_ctx_destroy(c); return fail(-2, "abs_pos_emb requires crop >= 1"); }
        TRY(c->dalloc(&c->pos_embed, (size_t)d->crop * kC));
        SETTER("pos_embed", { WANT(c->d.crop, kC); if (int r = copy_f32(c->pos_embed, data, (size_t)c->d.crop * kC, s)) return r; });
    }
    if (d->tps_condition) {
        TRY(c->dalloc(&c->wf7, (size_t)kC * 7));
        TRY(c->dalloc(&c->bf7, (size_t)kC));
        TRY(c->dalloc(&c->wr7, (size_t)kC * 7));
        TRY(c->dalloc(&c->br7, (size_t)kC));
        SETTER("latent_to_emb_f.weight", { WANT(kC, 7); if (int r = copy_f32(c->wf7, data, kC * 7, s)) return r; });
        SETTER("latent_to_emb_f.bias", { WANT(kC); if (int r = copy_f32(c->bf7, data, kC, s)) return r; });
        SETTER("latent_to_emb_r.weight", { WANT(kC, 7); if (int r = copy_f32(c->wr7, data, kC * 7, s)) return r; });
        SETTER("latent_to_emb_r.bias", { WANT(kC); if (int r = copy_f32(c->br7, data, kC, s)) return r; });
    }
    SETTER("emb_to_latent.linear.weight",
           { WANT(c->D, kC); launch_pack_rows(data, kC, c->map_fin, 1, kKS, 1.f, c->wfin, s); launch_pack_rows(data, kC, c->map_fin, 1, kKS, 1.f, c->wfin_k, s, 1); });
    SETTER("emb_to_latent.linear.bias", { WANT(c->D); if (int r = copy_f32(c->bfin, data, c->D, s)) return r; });
    SETTER("emb_to_latent.adaLN_modulation.1.weight", {
        WANT(2 * kC, kC);
        if (int r = copy_f32(c->ada_w + (size_t)c->final_off() * kC, data, (size_t)2 * kC * kC, s)) return r;
    });
    SETTER("emb_to_latent.adaLN_modulation.1.bias",
           { WANT(2 * kC); if (int r = copy_f32(c->ada_b + c->final_off(), data, 2 * kC, s)) return r; });
    c->trunk.resize(nl);
    c->ipa.resize(nl);
    for (int i = 0; i < nl; ++i) {
        const std::string p = "layers." + std::to_string(i) + ".";
        TrunkW* t = &c->trunk[i];
        SETTER(p + "adaLN_modulation.1.weight", {
            WANT(9 * kC, kC);
            if (int r = copy_f32(c->ada_w + (size_t)c->trunk_off(i) * kC, data, (size_t)9 * kC * kC, s)) return r;
        });
        SETTER(p + "adaLN_modulation.1.bias",
               { WANT(9 * kC); if (int r = copy_f32(c->ada_b + c->trunk_off(i), data, 9 * kC, s)) return r; });
        TRY(register_mha(c, p + "mha_t.attn.", &t->mha_t));
        TRY(register_mha(c, p + "mha_l.attn.", &t->mha_l));
        TRY(register_ffn(c, p, &t->ffn, true));
    }
    for (int i = 0; i < nl; ++i) {
        const std::string p = "ipa_layers." + std::to_string(i) + ".";
        IpaW* w = &c->ipa[i];
        TRY(c->dalloc(&w->gamma_beta, (size_t)2 * kC));
        TRY(c->dalloc(&w->wproj, (size_t)21 * kKS * 64));
        TRY(c->dalloc(&w->bproj, (size_t)kIpaProj));
        TRY(c->dalloc(&w->head_w, (size_t)4));
        TRY(c->dalloc(&w->wout, (size_t)12 * 16 * 64));
        TRY(c->dalloc(&w->bout, (size_t)kC));
        SETTER(p + "adaLN_modulation.1.weight", {
            WANT(6 * kC, kC);
            if (int r = copy_f32(c->ada_w + (size_t)c->ipa_off(i) * kC, data, (size_t)6 * kC * kC, s)) return r;
        });
        SETTER(p + "adaLN_modulation.1.bias",
               { WANT(6 * kC); if (int r = copy_f32(c->ada_b + c->ipa_off(i), data, 6 * kC, s)) return r; });
        SETTER(p + "ipa_norm.weight", { WANT(kC); if (int r = copy_f32(w->gamma_beta, data, kC, s)) return r; });
        SETTER(p + "ipa_norm.bias", { WANT(kC); if (int r = copy_f32(w->gamma_beta + kC, data, kC, s)) return r; });
        SETTER(p + "ipa.head_weights", { WANT(4); if (int r = copy_f32(w->head_w, data, 4, s)) return r; });
        SETTER(p + "ipa.linear_q.weight", { WANT(128, kC); launch_pack_rows(data, kC, c->map_nat, 4, kKS, 1.f, w->wproj, s); });
        SETTER(p + "ipa.linear_q.bias", { WANT(128); if (int r = copy_f32(w->bproj, data, 128, s)) return r; });
        SETTER(p + "ipa.linear_kv.weight",
               { WANT(256, kC); launch_pack_rows(data, kC, c->map_nat, 8, kKS, 1.f, w->wproj + (size_t)4 * kKS * 64, s); });
        SETTER(p + "ipa.linear_kv.bias", { WANT(256); if (int r = copy_f32(w->bproj + 128, data, 256, s)) return r; });
        SETTER(p + "ipa.linear_q_points.weight",
               { WANT(96, kC); launch_pack_rows(data, kC, c->map_nat, 3, kKS, 1.f, w->wproj + (size_t)12 * kKS * 64, s); });
        SETTER(p + "ipa.linear_q_points.bias", { WANT(96); if (int r = copy_f32(w->bproj + 384, data, 96, s)) return r; });
        SETTER(p + "ipa.linear_kv_points.weight",
               { WANT(192, kC); launch_pack_rows(data, kC, c->map_nat, 6, kKS, 1.f, w->wproj + (size_t)15 * kKS * 64, s); });
        SETTER(p + "ipa.linear_kv_points.bias", { WANT(192); if (int r = copy_f32(w->bproj + 480, data, 192, s)) return r; });
        SETTER(p + "ipa.linear_out.weight",
               { WANT(kC, kIpaFeat); launch_pack_rows(data, kIpaFeat, c->map_nat, 12, 16, 1.f, w->wout, s); });
        SETTER(p + "ipa.linear_out.bias", { WANT(kC); if (int r = copy_f32(w->bout, data, kC, s)) return r; });
        TRY(register_mha(c, p + "mha_l.attn.", &w->mha_l));
        TRY(register_ffn(c, p, &w->ffn, false));
    }
#undef TRY
    *out = c;
    return 0;
}

extern "C" int32_t mdgen_ctx_destroy(mdgen_ctx* c) {
    if (!c) return 0;
    for (auto& g : c->graphs) {
        if (g.exec) (void)hipGraphExecDestroy(g.exec);
        if (g.graph) (void)hipGraphDestroy(g.graph);
    }
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    for (hipEvent_t e : c->train_ev) (void)hipEventDestroy(e);
    if (c->train_side) (void)hipStreamDestroy(c->train_side);
    if (c->tr_buf) (void)hipFree(c->tr_buf);
    for (int i = 0; i < mdgen_ctx::kMaxSide; ++i) {
        if (c->ev_join[i]) (void)hipEventDestroy(c->ev_join[i]);
        if (c->side[i]) (void)hipStreamDestroy(c->side[i]);
    }
    for (void* p : c->allocs) (void)hipFree(p);
    delete c;
    return 0;
}

extern "C" int32_t mdgen_ctx_num_weights(const mdgen_ctx* c) { return c ? (int32_t)c->names.size() : 0; }
extern "C" const char* mdgen_ctx_weight_name(const mdgen_ctx* c, int32_t i) {
    if (!c || i < 0 || i >= (int32_t)c->names.size()) return nullptr;
    return c->names[i].c_str();
}

extern "C" int32_t mdgen_ctx_set_weight(mdgen_ctx* c, const char* key, const float* data, const int64_t* shape,
                                        int32_t ndim, void* stream) {
    if (!c || !key || !data || !shape) return fail(-1, "null argument");
    auto it = c->setters.find(key);
    if (it == c->setters.end()) return fail(-4, "unknown weight key '%s'", key);
    const int r = it->second(data, shape, ndim, (hipStream_t)stream);
    if (r) {
        if (r == -3) {
            char buf[160] = "";
            int o = 0;
            for (int i = 0; i < ndim && o < 140; ++i) o += snprintf(buf + o, sizeof(buf) - o, "%lld,", (long long)shape[i]);
            return fail(-3, "unexpected shape (%s) for weight '%s'", buf, key);
        }
        return r;
    }
    LAUNCHCHK();
    if (c->opt_keep_fp32) {
        size_t n = 1;
        for (int i = 0; i < ndim; ++i) n *= (size_t)shape[i];
        float*& dst = c->w32[key];
        if (!dst)
            if (int e = c->dalloc(&dst, n)) return e;
        // a BOUND entry (mdgen_train_bind_params) is the caller's master copy of the parameter: never written from here -- a
        // set_weight with other values (EMA weights swapped in for validation) only re-packs the sampler's operands
        if (dst != data && !c->w32_bound.count(key))
            HIPCHK(hipMemcpyAsync(dst, data, n * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    }
    c->provided[key] = true;
    c->finalized = false;
    return 0;
}

extern "C" int32_t mdgen_ctx_finalize(mdgen_ctx* c, void* stream) {
    if (!c) return fail(-1, "null context");
    for (const auto& n : c->names)
        if (!c->provided.count(n)) return fail(-5, "weight '%s' was not provided", n.c_str());
    if (!c->inv_freq_set) return fail(-5, "rot_emb.inv_freq was not provided");
    launch_rope_table(c->rope, c->inv_freq, kMaxPos + 1, (hipStream_t)stream);
    LAUNCHCHK();
    c->finalized = true;
    return 0;
}

extern "C" int32_t mdgen_ctx_set_option(mdgen_ctx* c, const char* name, int32_t value) {
    if (!c || !name) return fail(-1, "null argument");
    const std::string n(name);
    if (n == "streams") {
        if (value < 1 || value > mdgen_ctx::kMaxSide + 1) return fail(-2, "streams must be in 1..%d", mdgen_ctx::kMaxSide + 1);
        c->opt_streams = value;
        c->opt_streams_auto = false;
    } else if (n == "keep_fp32_weights") {
        if (value != 0 && value != 1) return fail(-2, "keep_fp32_weights must be 0 or 1");
        c->opt_keep_fp32 = value;
    } else if (n == "precision") {
        if (value != 16 && value != 32) return fail(-2, "precision must be 16 (bf16 operands) or 32 (fp32 operands)");
        if (value == 32 && (!c->opt_keep_fp32 || c->w32.empty()))
            return fail(-6, "precision 32 needs the fp32 weight copies: set option keep_fp32_weights = 1 before loading weights");
        c->opt_precision = value;
    } else if (n == "attention_path") {
        if (value != 0 && value != 1) return fail(-2, "attention_path must be 0 (auto) or 1 (robust loop always)");
        c->opt_attn_path = value;
    } else if (n == "fuse_proj") {
        if (value < 0 || value > 3)
            return fail(-2, "fuse_proj must be 0 (off), 1 (inside the row-owner MLP kernel), 2 (inside the panel MLP kernel) or 3 (panel kernel where it runs anyway)");
        c->opt_fuse_proj = value;
    } else if (n == "fuse_proj_qkv") {
        if (value != 0 && value != 1) return fail(-2, "fuse_proj_qkv must be 0 or 1");
        c->opt_fuse_proj_qkv = value;
    } else if (n == "small_split") {
        if (value != 0 && value != 1) return fail(-2, "small_split must be 0 or 1");
        c->opt_small_split = value;
    } else if (n == "panel_waves") {
        if (value != 0 && value != 4 && value != 8) return fail(-2, "panel_waves must be 0 (by launch size), 4 or 8");
        c->opt_panel_waves = value;
    } else if (n == "flash_rotate") {
        if (value != 0 && value != 1) return fail(-2, "flash_rotate must be 0 or 1");
        c->opt_flash_rotate = value;
    } else if (n == "flash_proj_form") {
        if (value != 0 && value != 4 && value != 8) return fail(-2, "flash_proj_form must be 0 (by shape), 4 or 8");
        c->opt_flash_proj_form = value;
    } else if (n == "flash_proj") {
        if (value < 0 || value > 2) return fail(-2, "flash_proj must be 0 (off), 1 (launches that fill the chip) or 2 (always)");
        c->opt_flash_proj = value;
    } else if (n == "mlp_path") {
        if (value < 0 || value > 2) return fail(-2, "mlp_path must be 0 (panel kernel), 1 (row-owner kernel when it fills the chip) or 2 (always)");
        c->opt_mlp_path = value;
    } else if (n == "train_precision") {
        if (value != 16 && value != 32) return fail(-2, "train_precision must be 32 (fp32 operands, exact) or 16 (bf16 operands, fp32 accumulate)");
        c->opt_train_precision = value;
    } else if (n == "train_attn_form") {
        if (value != 0 && value != 1) return fail(-2, "train_attn_form must be 0 (chunked attention kernels) or 1 (sequence-resident for axes of 129 .. 256 positions)");
        c->opt_train_attn_form = value;
    } else if (n == "train_defer_gate") {
        if (value != 0 && value != 1) return fail(-2, "train_defer_gate must be 0 or 1");
        c->opt_train_defer_gate = value;
    } else if (n == "train_turn_ahead") {
        if (value != 0 && value != 1) return fail(-2, "train_turn_ahead must be 0 or 1");
        c->opt_train_turn_ahead = value;
        c->tr_plan_ok = false;
    } else if (n == "train_y_bf16") {
        if (value != 0 && value != 1) return fail(-2, "train_y_bf16 must be 0 or 1");
        c->opt_train_y_bf16 = value;
    } else if (n == "train_dqkv_bf16") {
        if (value != 0 && value != 1) return fail(-2, "train_dqkv_bf16 must be 0 or 1");
        c->opt_train_dqkv_bf16 = value;
    } else if (n == "train_du_bf16") {
        if (value != 0 && value != 1) return fail(-2, "train_du_bf16 must be 0 or 1");
        c->opt_train_du_bf16 = value;
    } else if (n == "train_dhid_bf16") {
        if (value != 0 && value != 1) return fail(-2, "train_dhid_bf16 must be 0 or 1");
        c->opt_train_dhid_bf16 = value;
    } else if (n == "train_streams") {
        if (value != 1 && value != 2) return fail(-2, "train_streams must be 1 (one stream) or 2 (weight gradients on a second stream)");
        c->opt_train_streams = value;
    } else if (n == "mlp_fold") {
        if (value != 0 && value != 1) return fail(-2, "mlp_fold must be 0 or 1");
        c->opt_mlp_fold = value;
    } else if (n == "trace_tail") {
        c->opt_trace_tail = value != 0;
    } else if (n == "embed_split") {
        if (value != 0 && value != 1) return fail(-2, "embed_split must be 0 or 1");
        c->opt_embed_split = value;
    } else if (n == "mlp_tail") {
        if (value < 0 || value > 2) return fail(-2, "mlp_tail must be 0 (off), 1 (FinalLayer + Euler update) or 2 (... + the next step's token embedding)");
        c->opt_mlp_tail = value;
    } else if (n == "residue_l4_path") {
        if (value < 0 || value > 2) return fail(-2, "residue_l4_path must be 0, 1 or 2");
        c->opt_residue_l4 = value;
    } else {
        return fail(-4, "unknown option '%s'", name);
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------
// workspace
// ---------------------------------------------------------------------------------------------
static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

// panels the split scratch of a call covers, and its bytes (counters first: make_run zeroes them in every call): the largest launch that
// can take the split form has ncu / kMlpSplit panels; nothing but the counter block when the option is off
static long split_panels(const mdgen_ctx* c, long maxrows) {
    if (!c->opt_small_split) return 0;
    const long pn = (maxrows + kPanel - 1) / kPanel, cap = c->ncu / kMlpSplit < kMlpSplitMaxPanels ? c->ncu / kMlpSplit : kMlpSplitMaxPanels;
    return pn < cap ? pn : cap;
}
constexpr size_t kSplitCounterBytes = 1024;   // kMlpSplitMaxPanels counters, padded
static_assert(kMlpSplitMaxPanels * sizeof(unsigned) <= kSplitCounterBytes, "counter block");
static size_t split_bytes(long panels) { return kSplitCounterBytes + 2 * (size_t)panels * kMlpSplit * kPanel * kC * 4; }

// Gate fold (option mlp_fold): per (step, trunk layer) one MLP weight stream with that step's gate folded into fc2, and b2' = gate * b2.
// Carved when the call shares t across the batch and its trunk launches can take the row-owner kernel.
static bool mlp_uses_rows(const mdgen_ctx* c, long nrows);
constexpr size_t kFoldStreamBytes = (size_t)kMlpFrags * 1024;
constexpr size_t kFoldMaxBytes = (size_t)8 << 30;   // a call with so many steps that its streams would pass 8 GiB keeps the unfolded kernel
static size_t fold_bytes(const mdgen_ctx* c, int S) { return (size_t)S * c->nl * (kFoldStreamBytes + (size_t)kC * 4); }
static bool fold_on(const mdgen_ctx* c, long N, int t_shared, int S) {
    return t_shared && c->opt_mlp_fold && c->opt_precision == 16 && mlp_uses_rows(c, N) && fold_bytes(c, S) <= kFoldMaxBytes;
}

static size_t frag_bytes(long nseq, int len) { return (size_t)nseq * kH * (len / 32 + 1) * kFragBytes; }

// The panel prologues / epilogues (csrc/panel.h) address the residual stream as a uniform base + a 32-bit byte
// offset token * 1536, so ONE launch may cover at most kMaxViewTokens token rows.  Larger batches are run as
// several contiguous sub-batch views (plan_views); one sample (T * L tokens) must fit a view by itself.
constexpr long kMaxViewTokens = 0xFFFFFFFFL / (kC * 4);   // 2 796 202

// Contiguous sub-batch views of a call: at least `streams` of them (capped by B), and as many as the per-launch
// token limit requires.  Returns the number of views, 0 if a single sample already exceeds the limit.
static int plan_views(long B, long T, long L, int streams) {
    const long tl = T * L;
    if (tl > kMaxViewTokens) return 0;
    const long per = kMaxViewTokens / tl;              // samples per view at most
    long nv = (B + per - 1) / per;
    if (nv < streams) nv = streams;
    if (nv > B) nv = B;
    while ((B + nv - 1) / nv > per) ++nv;              // the largest view (ceil(B / nv) samples) must fit
    return (int)nv;
}

extern "C" int32_t mdgen_debug_view_plan(const mdgen_shape* sh, int32_t streams, int32_t* n_views,
                                         int32_t* max_batch_per_view) {
    if (!sh || !n_views || !max_batch_per_view) return fail(-1, "null argument");
    if (sh->B < 1 || sh->T < 1 || sh->L < 1 || streams < 1) return fail(-2, "B, T, L, streams must be >= 1");
    const int nv = plan_views(sh->B, sh->T, sh->L, streams);
    if (nv == 0) return fail(-2, "one sample (T*L = %ld tokens) exceeds the per-launch limit of %ld tokens",
                             (long)sh->T * sh->L, kMaxViewTokens);
    *n_views = nv;
    *max_batch_per_view = (sh->B + nv - 1) / nv;
    return 0;
}

static int check_shape(const mdgen_ctx* c, const mdgen_shape* sh, int S) {
    if (!c || !sh) return fail(-1, "null argument");
    if (sh->B < 1 || sh->T < 1 || sh->L < 1 || S < 1) return fail(-2, "B, T, L, n_steps must be >= 1");
    if (sh->T > kMaxPos - 1 || sh->L > kMaxPos - 1) return fail(-2, "T and L must be < %d", kMaxPos - 1);
    if ((long)sh->B * sh->T * sh->L > 1500000000L) return fail(-2, "token count too large");
    if ((long)sh->T * sh->L > kMaxViewTokens)
        return fail(-2, "one sample (T*L = %ld tokens) exceeds the per-launch limit of %ld tokens", (long)sh->T * sh->L,
                    kMaxViewTokens);
    if ((long)S * sh->B * sh->L > kMaxViewTokens)
        return fail(-2, "n_steps*B*L = %ld rows of the IPA table exceed the per-launch limit of %ld", (long)S * sh->B * sh->L,
                    kMaxViewTokens);
    if (c->d.abs_pos_emb && sh->L > c->d.crop) return fail(-2, "L=%d exceeds pos_embed crop=%d", sh->L, c->d.crop);
    return 0;
}

extern "C" int32_t mdgen_workspace_layout(const mdgen_ctx* c, const mdgen_shape* sh, int32_t S, int32_t t_shared,
                                          mdgen_ws_layout* o) {
    if (!o) return fail(-1, "null argument");
    if (int r = check_shape(c, sh, S)) return r;
    const long B = sh->B, T = sh->T, L = sh->L;
    const long N = B * T * L, Mp = (long)S * B * L, R = t_shared ? S : (long)S * B;
    const long maxrows = N > Mp ? N : Mp;
    size_t fq = (size_t)maxrows * 3 * kC * 2;  // SMALL layout lives in the q region
    size_t fkv = 0;
    auto upd = [&](size_t b) { if (b > fq) fq = b; if (b > fkv) fkv = b; };
    upd(frag_bytes(B * L, (int)T));
    if (L > 8) {
        upd(frag_bytes(B * T, (int)L));
        upd(frag_bytes((long)S * B, (int)L));
    }
    if (fkv == 0) fkv = 256;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t a = off; off += align256(bytes); return a; };
    o->h = take((size_t)N * kC * 4);
    o->qf = take(fq);
    o->kf = take(fkv);
    o->vf = take(fkv);
    o->obuf = take((size_t)maxrows * kC * 2);
    o->mod = take((size_t)R * c->modrow * 4);
    o->silu_t = take((size_t)R * kC * 4);
    o->ipa_out = take((size_t)Mp * kC * 4);
    o->h_ipa = take((size_t)Mp * kC * 4 * (c->d.tps_condition ? 2 : 1));
    o->ipa_proj = take((size_t)Mp * kIpaProj * 4);
    o->ipa_feat = take((size_t)Mp * kIpaFeat * 2);
    o->mask_bl = take((size_t)B * L * 4);
    o->rel7 = take((size_t)2 * B * L * 7 * 4);
    o->tgrid = take((size_t)S * B * 4);
    // fp32 path scratch: LN output [rows][384] | q,k,v [rows][1152] | attention output [rows][384] | MLP hidden
    // [rows][1536] | IPA features [Mp][256]   (only when the context keeps fp32 weights)
    o->f32_scratch = take(c->opt_keep_fp32 ? (size_t)maxrows * (kC + 3 * kC + kC + kF) * 4 + (size_t)Mp * kIpaFeat * 4 : 0);
    // k_mlp8's split form (launches of <= ncu / kMlpSplit panels): counters [kMlpSplitMaxPanels] | fc2 partials | private residual rows,
    // the latter two [panels][kMlpSplit][64][384] fp32 each
    o->split = take(split_bytes(split_panels(c, maxrows)));
    // per-(step, layer) gate-folded MLP streams [S][nl][2304 KiB] | b2' [S][nl][384] fp32 (0 bytes unless fold_on)
    o->fold = take(fold_on(c, N, t_shared, S) ? fold_bytes(c, S) : 0);
    // base rows of the token embedding per (step, b, l) for the embedding-as-tail form [S][B*L][384] fp32 (0 bytes unless fold and S > 1)
    o->embase = take(fold_on(c, N, t_shared, S) && c->opt_mlp_tail == 2 && S > 1 ? (size_t)Mp * kC * 4 : 0);
    o->total_bytes = off;
    return 0;
}

// ---------------------------------------------------------------------------------------------
// orchestration
// ---------------------------------------------------------------------------------------------
struct Run {
    mdgen_ctx* c;
    int B, T, L, D, S;
    long N, Mp;
    int t_shared;
    long mod_step_stride, mod_group_stride;
    unsigned char* ws;
    mdgen_ws_layout lay;
    const float *mask, *start_rot, *start_trans, *end_rot, *end_trans, *x_cond;
    const float* rel7_in;        // caller-supplied relative-frame 7-vectors (2,B,L,7) of the two-sided model, or null
    const int64_t *x_cond_mask, *aatype;
    hipStream_t s;
    // trunk buffers (a sub-batch view shifts these; see sub_run)
    float* hp;
    unsigned char *qfp, *kfp, *vfp;
    __bf16* obufp;
    float* modp;                 // adaLN table row of (step 0, first batch element of this view)
    const float* ipa_out_p;      // IPA table of (step 0, first batch element of this view)
    long ipa_step_stride;        // floats between consecutive steps of the IPA table
    // split scratch (layout.split): arrival counters, fc2 partials, private residual rows; split_cap panels
    unsigned* split_counters;
    float *split_part, *split_hupd;
    long split_cap;
    // gate fold: streams [S][nl][kFoldStreamBytes], then b2' [S][nl][384]; null when off for this call
    unsigned char* fold_streams;
    float* fold_b2g;
    bool fold_ready;   // prepare() has packed them for this call (its views' MLP launches take the row-owner kernel)
    // embedding-as-tail: base rows [S][B*L][384] (step 0, first batch element of this view), null when off; floats between steps
    const float* embase_p;
    long embase_step_stride;
    float* h() const { return hp; }
    float* mod() const { return modp; }
};

// View of the contiguous sub-batch [b0, b0+Bs) of a prepared call: every trunk buffer is per-token (or per
// sequence) and batch-major, so a sub-batch is a pointer shift.  Used to run two halves of the batch on two
// streams: their kernels are in different phases (HBM-bound prologues/epilogues, VALU-bound attention,
// MFMA-bound GEMMs) and overlap on the CUs instead of queueing behind each other.
static Run sub_run(const Run& r, int b0, int Bs, hipStream_t stream) {
    Run v = r;
    const long tl = (long)r.T * r.L;
    v.B = Bs;
    v.N = (long)Bs * tl;
    v.s = stream;
    v.mask = r.mask + b0 * tl;
    v.x_cond = r.x_cond + b0 * tl * r.D;
    v.x_cond_mask = r.x_cond_mask + b0 * tl;
    v.hp = r.hp + b0 * tl * kC;
    v.obufp = r.obufp + b0 * tl * kC;
    // q region: max(SMALL layout, fragment layouts) per batch element; k/v regions: fragment layouts
    const size_t per_b_small = (size_t)tl * 3 * kC * 2;
    const size_t per_b_fragT = (size_t)r.L * kH * (r.T / 32 + 1) * kFragBytes;
    const size_t per_b_fragL = r.L > 8 ? (size_t)r.T * kH * (r.L / 32 + 1) * kFragBytes : 0;
    const size_t per_b_kv = per_b_fragT > per_b_fragL ? per_b_fragT : per_b_fragL;
    const size_t per_b_q = per_b_small > per_b_kv ? per_b_small : per_b_kv;
    v.qfp = r.qfp + (size_t)b0 * per_b_q;
    v.kfp = r.kfp + (size_t)b0 * per_b_kv;
    v.vfp = r.vfp + (size_t)b0 * per_b_kv;
    v.modp = r.modp + (long)b0 * r.mod_group_stride;
    v.ipa_out_p = r.ipa_out_p + (long)b0 * r.L * kC;
    if (r.embase_p) v.embase_p = r.embase_p + (long)b0 * r.L * kC;
    return v;
}

// ---- fp32-operand path (option "precision" = 32; kernels in k_fp32.hip) ------------------------------------------
struct F32Bufs {
    float *y, *qkv, *att, *hid, *feat;
};
static F32Bufs f32_bufs(const Run& r) {
    const long maxrows = r.N > r.Mp ? r.N : r.Mp;
    float* b = (float*)(r.ws + r.lay.f32_scratch);
    F32Bufs o;
    o.y = b;
    o.qkv = o.y + maxrows * kC;
    o.att = o.qkv + maxrows * 3 * kC;
    o.hid = o.att + maxrows * kC;
    o.feat = o.hid + maxrows * kF;
    return o;
}
static const float* w32(const mdgen_ctx* c, const std::string& key) {
    auto it = c->w32.find(key);
    return it == c->w32.end() ? nullptr : it->second;
}
#define W32(var, key)                                                                      \
    const float* var = w32(r.c, key);                                                      \
    if (!var) return fail(-6, "fp32 copy of weight '%s' is missing (option keep_fp32_weights)", std::string(key).c_str())

// one attention sub-layer, fp32: LN + modulate -> q, k, v -> RoPE -> softmax attention -> out-projection + gated residual
static int attn_sublayer_fp32(const Run& r, const std::string& pre, float* h, long nrows, const AxisMap& ax,
                              const ModMap& mm, int shift, int scale, int gate, const MaskMap& mk, long pos_div, int pos_mod) {
    const F32Bufs b = f32_bufs(r);
    W32(wq, pre + "q_proj.weight"); W32(bq, pre + "q_proj.bias");
    W32(wk, pre + "k_proj.weight"); W32(bk, pre + "k_proj.bias");
    W32(wv, pre + "v_proj.weight"); W32(bv, pre + "v_proj.bias");
    W32(wo, pre + "out_proj.weight"); W32(bo, pre + "out_proj.bias");
    W32(biask, pre + "bias_k"); W32(biasv, pre + "bias_v");
    const ModMap none{nullptr, 1, 1, 0, 0};
    launch32_ln_mod(h, nrows, mm, shift, scale, 0, 1e-6f, b.y, r.s);
    const float qscale = 1.0f / std::sqrt((float)kDH);   // mha.py:263 q *= head_dim ** -0.5
    launch32_linear(b.y, kC, wq, kC, bq, nrows, kC, kC, 4, b.qkv, 3 * kC, 0, none, 0, 0, qscale, r.s);
    launch32_linear(b.y, kC, wk, kC, bk, nrows, kC, kC, 0, b.qkv, 3 * kC, kC, none, 0, 0, 0.f, r.s);
    launch32_linear(b.y, kC, wv, kC, bv, nrows, kC, kC, 0, b.qkv, 3 * kC, 2 * kC, none, 0, 0, 0.f, r.s);
    launch32_rope(b.qkv, nrows, 3 * kC, pos_div, pos_mod, r.c->inv_freq, r.s);
    launch32_attn(b.qkv, 3 * kC, ax, mk, biask, biasv, r.c->inv_freq, b.att, r.s);
    launch32_linear(b.att, kC, wo, kC, bo, nrows, kC, kC, 2, h, kC, 0, mm, gate, 1, 0.f, r.s);
    LAUNCHCHK();
    return 0;
}

static int mlp_sublayer_fp32(const Run& r, const std::string& pre, float* h, long nrows, const ModMap& mm, int shift, int scale,
                             int gate) {
    const F32Bufs b = f32_bufs(r);
    W32(w1, pre + "fc1.weight"); W32(b1, pre + "fc1.bias");
    W32(w2, pre + "fc2.weight"); W32(b2, pre + "fc2.bias");
    const ModMap none{nullptr, 1, 1, 0, 0};
    launch32_ln_mod(h, nrows, mm, shift, scale, 0, 1e-6f, b.y, r.s);
    launch32_linear(b.y, kC, w1, kC, b1, nrows, kF, kC, 1, b.hid, kF, 0, none, 0, 0, 0.f, r.s);
    launch32_linear(b.hid, kF, w2, kF, b2, nrows, kC, kF, 2, h, kC, 0, mm, gate, 1, 0.f, r.s);
    LAUNCHCHK();
    return 0;
}

// Every token-local kernel addresses the residual stream with 32-bit byte offsets (token * 1536; rows.h, panel.h): one
// launch may cover at most kMaxViewTokens rows.  The trunk is cut into views accordingly (plan_views); this guard is for
// whatever reaches a launcher directly (the IPA stack's S * B * L rows).
static int check_launch_rows(long nrows) {
    if (nrows > kMaxViewTokens)
        return fail(-7, "%ld token rows in one launch exceed the 32-bit offset limit of %ld (use a smaller batch per call)", nrows,
                    kMaxViewTokens);
    return 0;
}

// k_flash_proj owns a (sequence, 64-query chunk) for all 16 heads: four times the work of a k_flash workgroup, a quarter of the
// workgroups.  It pays where those still fill the chip (cfg-2: 1024 per launch, ATLAS: 1000 / 1024); small launches (B = 1: 64,
// the IPA stack) keep the finer-grained k_flash + projection.
// (two such workgroups per CU: 512 on the 256-CU part)
static bool flash_proj_on(const mdgen_ctx* c, const AxisMap& ax) {
    return c->opt_precision == 16 && (c->opt_flash_proj == 2 || (c->opt_flash_proj == 1 && flash_proj_jobs(ax) >= 2L * c->ncu));
}

// `defer`: when non-null and the sub-layer takes the tiled-attention path, its out-projection is NOT launched; *defer receives
// what the fused kernel (k_mlp_rows<NW, true>) needs to run it ahead of the MLP (a_bf16 stays null otherwise).
static int attn_sublayer(const Run& r, const MhaW& m, float* h, long nrows, const AxisMap& ax, const ModMap& mm,
                         int shift, int scale, int gate, const MaskMap& mk, bool residue_axis, bool trunk,
                         ProjParams* defer = nullptr, const ProjParams* pre = nullptr) {
    if (int e = check_launch_rows(nrows)) return e;
    const char* c_qkv = !trunk ? "ipa.ln_qkv" : residue_axis ? "ln_qkv_L" : "ln_qkv_T";
    const char* c_att = !trunk ? "ipa.flash" : residue_axis ? "flash_L" : "flash_T";
    const char* c_prj = !trunk ? "ipa.proj" : residue_axis ? "proj_L" : "proj_T";
    QkvParams q{};
    q.h = h;
    q.nrows = nrows;
    q.ax = ax;
    q.mm = mm;
    q.shift_chunk = shift;
    q.scale_chunk = scale;
    q.wq = m.wq;
    q.wk = m.wk;
    q.bq = m.bq;
    q.bk = m.bk;
    q.rope = r.c->rope;
    q.qf = r.qfp;
    q.kf = r.kfp;
    q.vf = r.vfp;
    q.qkv_small = (__bf16*)r.qfp;
    q.panels_per_seq = (ax.len + kPanel - 1) / kPanel;
    ProjParams p{};
    p.h = h;
    p.nrows = nrows;
    p.mm = mm;
    p.gate_chunk = gate;
    p.gated = 1;
    p.w = m.wo;
    p.bias = m.bo;
    const bool small = residue_axis && ax.len <= 8;
    if (small && ax.len == 4 && r.c->opt_residue_l4 != 0) {
        // L == 4: the 5-key attention runs inside the QKV kernel (quad-local), which writes the attention output
        q.wv = m.wv_small;
        q.bv = m.bv_small;
        q.bias_k = m.bias_k;
        q.bias_v = m.bias_v;
        q.mk = mk;
        q.obuf = r.obufp;
        if (r.c->opt_residue_l4 == 2) {   // whole residue-axis sub-layer in one kernel (option 1: attention only)
            q.h_rw = h;
            q.wo = m.wo;
            q.bo = m.bo;
            q.gate_chunk = gate;
            // launches of at most one 32-row workgroup per CU (B <= 2 at T 1000, the IPA stack): half panels, twice the CUs at work
            // (k_ln_qkv_attn4<true, true>; option small_split; tag "@h32")
            const bool half = r.c->opt_small_split && (nrows + 31) / 32 <= r.c->ncu;
            const std::string cls = std::string(residue_axis && trunk ? "attn_L_fused" : c_qkv) + (half ? "@h32" : "");
            { ProfScope ps(r.c, r.c->intern(cls), r.s); if (!g_dry) launch_ln_qkv_attn4(q, true, r.s, half); }
            LAUNCHCHK();
        } else {
            { ProfScope ps(r.c, c_qkv, r.s); if (!g_dry) launch_ln_qkv_attn4(q, false, r.s); }
            LAUNCHCHK();
            p.a_bf16 = r.obufp;
            { ProfScope ps(r.c, c_prj, r.s); if (!g_dry) launch_proj(p, 0, r.s); }
            LAUNCHCHK();
        }
    } else if (small) {
        q.wv = m.wv_small;
        q.bv = m.bv_small;
        { ProfScope ps(r.c, c_qkv, r.s); if (!g_dry) launch_ln_qkv(q, true, r.s); }
        LAUNCHCHK();
        p.qkv_small = q.qkv_small;
        p.ax = ax;
        p.mk = mk;
        p.bias_k = m.bias_k;
        p.bias_v = m.bias_v;
        p.rope = r.c->rope;
        { ProfScope ps(r.c, c_prj, r.s); if (!g_dry) launch_proj(p, 2, r.s); }
        LAUNCHCHK();
    } else {
        q.wv = m.wv_flash;
        q.bv = m.bv_flash;
        q.bias_k = m.bias_k;   // written into key slot `len` of the K / V^T fragments by the sequence's last panel
        q.bias_v = m.bias_v;
        q.mk = mk;   // key-validity words for the attention kernel, in the slack behind the V^T fragments
        q.vmask = (uint32_t*)(q.vf + flash_vmask_offset(ax.nseq, ax.ntile()));
        q.vmask_stride = flash_vmask_stride(ax.ntile());
        if (pre && pre->a_bf16) {
            // the PREVIOUS sub-layer's deferred out-projection + gated residual runs in this kernel's panels first (option
            // fuse_proj_qkv): its rows then come back from L2 for the LayerNorm instead of from HBM in a launch of their own
            q.obuf = const_cast<__bf16*>(pre->a_bf16);
            q.wo = pre->w;
            q.bo = pre->bias;
            q.gate_chunk = pre->gate_chunk;
            q.h_rw = h;
            { ProfScope ps(r.c, "projL_qkvT", r.s); if (!g_dry) launch_ln_qkv(q, false, r.s, true); }
            LAUNCHCHK();
        } else {
            // (profile class "...@p8": the eight-wave form ran -- tests assert which kernel a launch took)
            const int pw = panel_waves_for((long)ax.nseq * q.panels_per_seq, r.c->opt_panel_waves, r.c->ncu);
            const bool split = pw == 8 && r.c->opt_small_split && 2L * ax.nseq * q.panels_per_seq <= r.c->ncu;
            // ... and 32 positions per workgroup pair where even those fit one per CU (B = 1 at T 1000: 256; tag "@h32x2")
            const int pps32 = (ax.len + 31) / 32;
            const bool half = split && 2L * ax.nseq * pps32 <= r.c->ncu;
            if (half) q.panels_per_seq = pps32;
            const std::string cls = std::string(c_qkv) + (half ? "@h32x2" : split ? "@p8x2" : pw == 8 ? "@p8" : "");
            { ProfScope ps(r.c, r.c->intern(cls), r.s); if (!g_dry) launch_ln_qkv(q, false, r.s, false, pw, split, half); }
            LAUNCHCHK();
        }
        FlashParams f{};
        f.ax = ax;
        f.mk = mk;
        f.qf = q.qf;
        f.kf = q.kf;
        f.vf = q.vf;
        f.bias_k = m.bias_k;
        f.bias_v = m.bias_v;
        f.rope = r.c->rope;
        f.obuf = r.obufp;
        f.force_robust = r.c->opt_attn_path;
        f.rotate = r.c->opt_flash_rotate;
        f.vmask = q.vmask;
        f.vmask_stride = q.vmask_stride;
        if (flash_proj_on(r.c, ax)) {
            // attention of all heads + out-projection + gated residual in one launch: nothing is deferred, no k_proj<0>
            FlashProjParams fp{};
            fp.f = f;
            fp.h = h;
            fp.mm = mm;
            fp.gate_chunk = gate;
            fp.wo = m.wo;
            fp.bo = m.bo;
            // the 128-row form where it still gives every CU a workgroup (cfg-2: 256 per sub-batch stream)
            const long jobs8 = (long)ax.nseq * ((ax.len + 2 * kPanel - 1) / (2 * kPanel));
            const int form = r.c->opt_flash_proj_form ? r.c->opt_flash_proj_form : (ax.len >= 512 && jobs8 >= r.c->ncu) ? 8 : 4;
            // (class "...@q64" / "@q128": which fused form ran -- k_flash_proj / k_flash_proj8; tests assert it)
            const std::string cls = std::string(!trunk ? "ipa.flash_proj" : residue_axis ? "flash_proj_L" : "flash_proj_T") + (form == 8 ? "@q128" : "@q64");
            { ProfScope ps(r.c, r.c->intern(cls), r.s); if (!g_dry) launch_flash_proj(fp, form, r.s); }
            LAUNCHCHK();
            return 0;
        }
        { ProfScope ps(r.c, c_att, r.s); if (!g_dry) launch_flash(f, r.s); }
        LAUNCHCHK();
        p.a_bf16 = f.obuf;
        if (defer) {
            *defer = p;
            return 0;
        }
        { ProfScope ps(r.c, c_prj, r.s); if (!g_dry) launch_proj(p, 0, r.s); }
        LAUNCHCHK();
    }
    return 0;
}

// Row-owner kernel: one workgroup = 4 waves x 32 rows and one workgroup per CU, so it needs ~200 workgroups to fill the
// chip; smaller launches (IPA stack, B = 1 tetrapeptides) stay on the 64-row panel kernel.
static bool mlp_uses_rows(const mdgen_ctx* c, long nrows) {
    const long tiles = (nrows + 31) / 32;
    return c->opt_mlp_path == 2 || (c->opt_mlp_path == 1 && tiles >= 3L * c->ncu);   // (768 row tiles = 192 four-wave workgroups on the 256-CU part)
}

// `proj`: a deferred out-projection (attn_sublayer) to run inside the MLP kernel, ahead of the MLP
// `fold_sl` >= 0 (trunk, gate fold active): index step * nl + layer of the folded stream / b2' of this launch
// `tail` (last trunk layer; nullable): the FinalLayer's parameters; *tail_done = true when the launch ran it (folded row-owner form only)
static int mlp_sublayer(const Run& r, const FfnW& f, float* h, long nrows, const ModMap& mm, int shift, int scale,
                        int gate, bool trunk, const ProjParams* proj = nullptr, const bf16x8* wo_stream = nullptr, long fold_sl = -1,
                        const FinalParams* tail = nullptr, bool* tail_done = nullptr, int next_step = -1, bool* next_h0 = nullptr) {
    if (int e = check_launch_rows(nrows)) return e;
    const bool panel_fused = proj && proj->a_bf16 && r.c->opt_fuse_proj >= 2;   // (3: only handed a projection when the panel kernel runs anyway)
    if (!panel_fused && mlp_uses_rows(r.c, nrows)) {
        MlpRowsParams q{};
        bool tail_on = false, emb_on = false;
        q.h = h;
        q.nrows = nrows;
        q.mm = mm;
        q.shift_chunk = shift;
        q.scale_chunk = scale;
        q.gate_chunk = gate;
        q.wstream = (const unsigned char*)f.wstream;
        q.b1 = f.b1;
        q.b2 = f.b2;
        if (proj && proj->a_bf16) {
            q.o = proj->a_bf16;
            q.wo_stream = (const unsigned char*)wo_stream;
            q.bo = proj->bias;
            q.gate_chunk_o = proj->gate_chunk;
        } else if (fold_sl >= 0 && r.fold_ready && mm.group_stride == 0 && mm.step_stride == 0) {
            q.wstream = r.fold_streams + (size_t)fold_sl * kFoldStreamBytes;
            q.b2g = r.fold_b2g + (size_t)fold_sl * kC;
            if (tail && tail_done && r.c->opt_mlp_tail && tail->mm.group_stride == 0 && tail->mm.step_stride == 0) {
                tail_on = true;
                q.tail_w = r.c->wfin_k;
                q.tail_b = tail->bias;
                q.tail_mod = tail->mm.mod;
                q.tail_D = tail->D;
                q.tail_euler = tail->euler;
                q.tail_dt = tail->dt;
                q.tail_x = tail->x;
                q.tail_out = tail->out;
                *tail_done = true;
                if (next_step >= 0 && next_h0 && r.embase_p && tail->euler) {   // ... and the next step's token embedding
                    q.emb_wl = r.c->wl_rows;
                    q.emb_wc = r.c->wc_rows;
                    if (r.c->opt_embed_split) {
                        q.emb_wl_hi = r.c->wl_hi;
                        q.emb_wl_lo = r.c->wl_lo;
                        q.emb_wc_hi = r.c->wc_hi;
                        q.emb_wc_lo = r.c->wc_lo;
                    }
                    q.emb_base = r.embase_p + (long)next_step * r.embase_step_stride;
                    q.emb_mdelta = r.c->mask_delta;
                    q.emb_xcond = r.x_cond;
                    q.emb_cmask = r.x_cond_mask;
                    q.emb_T = r.T;
                    q.emb_L = r.L;
                    emb_on = true;
                    *next_h0 = true;
                }
            }
        }
        if (trunk && r.c->phase_trace && (!r.c->opt_trace_tail || emb_on)) {   // (trace_tail: the launch with both tails is the one traced)
            q.trace = r.c->phase_trace;
            q.trace_cap = r.c->phase_trace_cap;
            r.c->phase_trace = nullptr;
        }
        // (class "mlp@fold": the folded form ran -- tests assert it)
        // ("mlp@fold+final": ... with the FinalLayer + Euler update as its tail)
        { ProfScope ps(r.c, !trunk ? "ipa.mlp" : q.o ? "proj_mlp" : emb_on ? "mlp@fold+final+embed" : tail_on ? "mlp@fold+final" : q.b2g ? "mlp@fold" : "mlp", r.s); if (!g_dry) launch_mlp_rows(q, 4, r.s); }
        LAUNCHCHK();
        return 0;
    }
    MlpParams p{};
    if (proj && proj->a_bf16) {   // fuse_proj = 2: the deferred out-projection runs in the panel kernel's prologue
        p.o = proj->a_bf16;
        p.wo = proj->w;
        p.bo = proj->bias;
        p.gate_chunk_o = proj->gate_chunk;
    }
    p.h = h;
    p.nrows = nrows;
    p.mm = mm;
    p.shift_chunk = shift;
    p.scale_chunk = scale;
    p.gate_chunk = gate;
    p.w1 = f.w1;
    p.w2 = f.w2;
    p.b1 = f.b1;
    p.b2 = f.b2;
    if (trunk && r.c->phase_trace) {   // one-shot: the next trunk MLP launch records its phase stamps
        p.trace = r.c->phase_trace;
        p.trace_cap = r.c->phase_trace_cap;
        r.c->phase_trace = nullptr;
    }
    const long panels = (nrows + kPanel - 1) / kPanel;
    const int pw = p.trace ? 4 : panel_waves_for(panels, r.c->opt_panel_waves, r.c->ncu);
    // the split form: the context's scratch serves one launch at a time (one stream), and its workgroups must meet in one L2
    const bool split = pw == 8 && r.c->opt_small_split && r.c->xcd_round_robin && r.c->live_streams <= 1 &&
                       panels * kMlpSplit <= r.c->ncu && panels <= r.split_cap;
    if (split) {
        ++r.c->n_split_launches;
        p.part = r.split_part;
        p.hupd = r.split_hupd;
        p.counters = r.split_counters;
    }
    const std::string cls = std::string(!trunk ? "ipa.mlp" : p.o ? "proj_mlp" : "mlp") + (split ? "@p8x3" : pw == 8 ? "@p8" : "@p4");
    { ProfScope ps(r.c, r.c->intern(cls), r.s); if (!g_dry) launch_mlp(p, r.s, pw); }
    LAUNCHCHK();
    return 0;
}

// IPA stack for all prepared steps at once (latent_model.py:175-210): tokens (step, b, l).
static int ipa_stack(const Run& r, float* hbuf, const float* rel7, const float* w7, const float* b7, const float* rot,
                     const float* trans) {
    mdgen_ctx* c = r.c;
    const int G = r.S * r.B;
    if (!g_dry) launch_ipa_init(c->aa_emb, r.aatype, rel7, w7, b7, hbuf, G, r.B, r.L, r.s);
    LAUNCHCHK();
    AxisMap ax{G, r.L, G, 0, r.L, 1};
    MaskMap mk{(const float*)(r.ws + r.lay.mask_bl), (long)r.B * r.L};
    for (int i = 0; i < c->nl; ++i) {
        const IpaW& w = c->ipa[i];
        ModMap mm{r.mod() + c->ipa_off(i), r.L, r.B, r.mod_step_stride, r.mod_group_stride};
        if (c->opt_precision == 32) {   // ---- fp32 operands: ipa_norm -> four projections -> point attention -> linear_out
            const std::string pre = "ipa_layers." + std::to_string(i) + ".";
            const F32Bufs fb = f32_bufs(r);
            W32(wq, pre + "ipa.linear_q.weight"); W32(bq, pre + "ipa.linear_q.bias");
            W32(wkv, pre + "ipa.linear_kv.weight"); W32(bkv, pre + "ipa.linear_kv.bias");
            W32(wqp, pre + "ipa.linear_q_points.weight"); W32(bqp, pre + "ipa.linear_q_points.bias");
            W32(wkp, pre + "ipa.linear_kv_points.weight"); W32(bkp, pre + "ipa.linear_kv_points.bias");
            W32(wout, pre + "ipa.linear_out.weight"); W32(bout, pre + "ipa.linear_out.bias");
            const ModMap none{nullptr, 1, 1, 0, 0};
            float* proj = (float*)(r.ws + r.lay.ipa_proj);
            launch32_ln_mod(hbuf, r.Mp, ModMap{w.gamma_beta, 1, 1, 0, 0}, 1, 0, 1, 1e-5f, fb.y, r.s);
            launch32_linear(fb.y, kC, wq, kC, bq, r.Mp, 128, kC, 0, proj, kIpaProj, 0, none, 0, 0, 0.f, r.s);
            launch32_linear(fb.y, kC, wkv, kC, bkv, r.Mp, 256, kC, 0, proj, kIpaProj, 128, none, 0, 0, 0.f, r.s);
            launch32_linear(fb.y, kC, wqp, kC, bqp, r.Mp, 96, kC, 0, proj, kIpaProj, 384, none, 0, 0, 0.f, r.s);
            launch32_linear(fb.y, kC, wkp, kC, bkp, r.Mp, 192, kC, 0, proj, kIpaProj, 480, none, 0, 0, 0.f, r.s);
            IpaAttnParams ap{};
            ap.proj = proj;
            ap.rot = rot;
            ap.trans = trans;
            ap.mask_bl = (const float*)(r.ws + r.lay.mask_bl);
            ap.head_w = w.head_w;
            ap.feat = nullptr;
            ap.feat32 = fb.feat;
            ap.ngroups = G;
            ap.B = r.B;
            ap.L = r.L;
            launch_ipa_attn(ap, r.s);
            launch32_linear(fb.feat, kIpaFeat, wout, kIpaFeat, bout, r.Mp, kC, kIpaFeat, 2, hbuf, kC, 0, none, 0, 0, 0.f, r.s);
            LAUNCHCHK();
            if (int e = attn_sublayer_fp32(r, pre + "mha_l.attn.", hbuf, r.Mp, ax, mm, 0, 1, 2, mk, 1, r.L)) return e;
            if (int e = mlp_sublayer_fp32(r, pre, hbuf, r.Mp, mm, 3, 4, 5)) return e;
            continue;
        }
        LnLinearParams lp{};
        lp.h = hbuf;
        lp.nrows = r.Mp;
        lp.mm = ModMap{w.gamma_beta, 1, 1, 0, 0};
        lp.w = w.wproj;
        lp.bias = w.bproj;
        lp.out = (float*)(r.ws + r.lay.ipa_proj);
        lp.nout = kIpaProj;
        { ProfScope ps(c, "ipa.ln_linear", r.s); if (!g_dry) launch_ln_linear(lp, r.s); }
        LAUNCHCHK();
        IpaAttnParams ap{};
        ap.proj = lp.out;
        ap.rot = rot;
        ap.trans = trans;
        ap.mask_bl = (const float*)(r.ws + r.lay.mask_bl);
        ap.head_w = w.head_w;
        ap.feat = (__bf16*)(r.ws + r.lay.ipa_feat);
        ap.ngroups = G;
        ap.B = r.B;
        ap.L = r.L;
        { ProfScope ps(c, "ipa.point_attn", r.s); if (!g_dry) launch_ipa_attn(ap, r.s); }
        LAUNCHCHK();
        ProjParams pp{};
        pp.h = hbuf;
        pp.nrows = r.Mp;
        pp.mm = mm;
        pp.gated = 0;
        pp.w = w.wout;
        pp.bias = w.bout;
        pp.a_bf16 = ap.feat;
        { ProfScope ps(c, "ipa.linear_out", r.s); if (!g_dry) launch_proj(pp, 1, r.s); }
        LAUNCHCHK();
        if (int e = attn_sublayer(r, w.mha_l, hbuf, r.Mp, ax, mm, 0, 1, 2, mk, true, false)) return e;
        if (int e = mlp_sublayer(r, w.ffn, hbuf, r.Mp, mm, 3, 4, 5, false)) return e;
    }
    return 0;
}

// Step-invariant work: adaLN table for every prepared time row, compact mask, IPA table.
// t values: t_dev (device, [S][B]) when non-null, else t_host[step] baked into the launch.
// view_rows: token rows of the call's largest trunk launch (a sub-batch view)
static int prepare(Run& r, const float* t_dev, const float* t_host, long view_rows) {
    mdgen_ctx* c = r.c;
    float* silu = (float*)(r.ws + r.lay.silu_t);
    const int R = r.t_shared ? r.S : r.S * r.B;
    // K/V fragment regions: key slots past a sequence's end are never written by k_ln_qkv but are read (and
    // masked to P = 0) by k_flash, so they must hold FINITE values: zero them once per call.
    HIPCHK(hipMemsetAsync(r.ws + r.lay.kf, 0, r.lay.obuf - r.lay.kf, r.s));
    if (t_dev) {
        if (r.t_shared && r.B > 1) return fail(-2, "device t rows require t_shared == 0 or B == 1");
        if (!g_dry) launch_temb(t_dev, R, c->d.time_multiplier, c->t_w0, c->t_b0, c->t_w2, c->t_b2, silu, r.s);
        LAUNCHCHK();
    } else {
        // the (tiny) host time grid travels as kernel arguments: capturable, no host buffer lifetime issue
        float* tg = (float*)(r.ws + r.lay.tgrid);
        if (!g_dry) launch_write_floats(t_host, r.S, tg, r.s);
        LAUNCHCHK();
        if (!g_dry) launch_temb(tg, r.S, c->d.time_multiplier, c->t_w0, c->t_b0, c->t_w2, c->t_b2, silu, r.s);
        LAUNCHCHK();
    }
    { ProfScope ps(c, "adaln_table", r.s); if (!g_dry) launch_adaln(silu, R, c->ada_w, c->ada_b, c->modrow, r.mod(), r.s); }
    LAUNCHCHK();
    r.fold_ready = r.fold_streams && mlp_uses_rows(c, view_rows);
    if (r.fold_ready) {   // the steps' MLP gates folded into per-(step, layer) fc2 streams (t_shared: R == S rows)
        int goff[8];
        const float *w2[8], *b2[8];
        const bf16x8* base[8];
        for (int i = 0; i < c->nl; ++i) {
            goff[i] = c->trunk_off(i) + 8 * kC;
            w2[i] = c->trunk[i].ffn.w2f;
            b2[i] = c->trunk[i].ffn.b2;
            base[i] = c->trunk[i].ffn.wstream;
        }
        { ProfScope ps(c, "fold_pack", r.s); if (!g_dry) launch_pack_fold(r.mod(), r.mod_step_stride, r.S, c->nl, goff, w2, b2, base, c->mlp_tab, (bf16x8*)r.fold_streams, r.fold_b2g, r.s); }
        LAUNCHCHK();
    }
    // mask_bl[b][l] = mask[b][0][l]  (latent_model.py:246 passes mask[:,0])
    HIPCHK(hipMemcpy2DAsync(r.ws + r.lay.mask_bl, (size_t)r.L * 4, r.mask, (size_t)r.T * r.L * 4, (size_t)r.L * 4, r.B,
                            hipMemcpyDeviceToDevice, r.s));
    float* ipa_out = (float*)(r.ws + r.lay.ipa_out);
    if (!c->d.tps_condition) {
        if (int e = ipa_stack(r, ipa_out, nullptr, nullptr, nullptr, r.start_rot, r.start_trans)) return e;
    } else {
        if (!r.end_rot || !r.end_trans) return fail(-2, "tps_condition requires end frames");
        float* rel = (float*)(r.ws + r.lay.rel7);
        const long BL = (long)r.B * r.L;
        // x_f = (start^-1 o end).to_tensor_7(), x_r = (end^-1 o start).to_tensor_7()   (latent_model.py:194-195)
        // The quaternion's SIGN is whatever torch.linalg.eigh returns in the reference (rigid_utils.py:191-210) and it does
        // reach a Linear: a caller that needs the reference's exact inputs passes its own to_tensor_7() outputs (rel7_in);
        // otherwise the library computes them with the sign fixed to w >= 0.
        if (r.rel7_in) {
            HIPCHK(hipMemcpyAsync(rel, r.rel7_in, (size_t)2 * BL * 7 * 4, hipMemcpyDeviceToDevice, r.s));
        } else {
            if (!g_dry) launch_rel7(r.start_rot, r.start_trans, r.end_rot, r.end_trans, rel, BL, r.s);
            if (!g_dry) launch_rel7(r.end_rot, r.end_trans, r.start_rot, r.start_trans, rel + BL * 7, BL, r.s);
            LAUNCHCHK();
        }
        float* h2 = (float*)(r.ws + r.lay.h_ipa);
        // x_r stream runs on the start frames, x_f stream on the end frames (latent_model.py:203-205)
        if (int e = ipa_stack(r, ipa_out, rel + BL * 7, c->wr7, c->br7, r.start_rot, r.start_trans)) return e;
        if (int e = ipa_stack(r, h2, rel, c->wf7, c->bf7, r.end_rot, r.end_trans)) return e;
        if (!g_dry) launch_add_inplace(ipa_out, h2, r.Mp * kC, r.s);
        LAUNCHCHK();
    }
    if (r.fold_ready && c->opt_mlp_tail == 2 && r.S > 1) {
        // steps 1 .. S-1 take their token embedding from the previous step's last MLP launch (rows_embed_tail): the part of it that does
        // not depend on x, per (step, b, l)
        float* eb = (float*)(r.ws + r.lay.embase);
        { ProfScope ps(c, "embed_base", r.s); if (!g_dry) launch_embed_base(c->bl, c->bc, c->mask_emb, c->d.abs_pos_emb ? c->pos_embed : nullptr, ipa_out, r.S, r.B * r.L, r.L, eb, r.s); }
        LAUNCHCHK();
        r.embase_p = eb;
    }
    return 0;
}

// One network evaluation at prepared step `step`: x -> velocity (out) or Euler update of x in place.
// h0_ready: the previous step's last MLP launch has already written this step's token embedding into h (no k_embed launch);
// next_h0 (nullable): ask this step to do the same for step + 1; *next_h0 = true when it did.
static int denoise_step(const Run& r, int step, float* x, float* out, int euler, float dt, float* trace_h, bool h0_ready = false,
                        bool* next_h0 = nullptr) {
    mdgen_ctx* c = r.c;
    float* h = r.h();
    EmbedParams e{};
    e.x = x;
    e.x_cond = r.x_cond;
    e.x_cond_mask = r.x_cond_mask;
    e.wl = c->wl;
    e.wl_pack = c->wl_pack;
    e.wc_pack = c->wc_pack;
    e.bl = c->bl;
    e.wc = c->wc;
    e.bc = c->bc;
    e.mask_emb = c->mask_emb;
    e.pos_embed = c->d.abs_pos_emb ? c->pos_embed : nullptr;
    e.ipa_out = r.ipa_out_p + (long)step * r.ipa_step_stride;
    e.h = h;
    e.N = r.N;
    e.T = r.T;
    e.L = r.L;
    e.D = r.D;
    if (!h0_ready) {
        { ProfScope ps(c, "embed", r.s); if (!g_dry) launch_embed(e, r.s); }
        LAUNCHCHK();
    }
    const size_t hbytes = (size_t)r.N * kC * 4;
    if (trace_h) HIPCHK(hipMemcpyAsync(trace_h, h, hbytes, hipMemcpyDeviceToDevice, r.s));
    const float* modstep = r.mod() + (long)step * r.mod_step_stride;
    AxisMap axL{r.B * r.T, r.L, r.B * r.T, 0, r.L, 1};
    AxisMap axT{r.B * r.L, r.T, r.L, r.T * r.L, 1, r.L};
    MaskMap mk{r.mask, 0};
    if (c->opt_precision == 32) {   // ---- fp32 operands (k_fp32.hip): same dataflow, one kernel per reference op group
        for (int i = 0; i < c->nl; ++i) {
            const std::string pre = "layers." + std::to_string(i) + ".";
            ModMap mm{modstep + c->trunk_off(i), r.T * r.L, r.B, 0, r.mod_group_stride};
            if (int er = attn_sublayer_fp32(r, pre + "mha_l.attn.", h, r.N, axL, mm, 0, 1, 2, mk, 1, r.L)) return er;
            if (int er = attn_sublayer_fp32(r, pre + "mha_t.attn.", h, r.N, axT, mm, 3, 4, 5, mk, r.L, r.T)) return er;
            if (int er = mlp_sublayer_fp32(r, pre, h, r.N, mm, 6, 7, 8)) return er;
            if (trace_h) HIPCHK(hipMemcpyAsync(trace_h + (size_t)(i + 1) * r.N * kC, h, hbytes, hipMemcpyDeviceToDevice, r.s));
        }
        W32(wfin, "emb_to_latent.linear.weight");
        W32(bfin, "emb_to_latent.linear.bias");
        const F32Bufs fb = f32_bufs(r);
        const ModMap fm{modstep + c->final_off(), r.T * r.L, r.B, 0, r.mod_group_stride};
        const ModMap none{nullptr, 1, 1, 0, 0};
        launch32_ln_mod(h, r.N, fm, 0, 1, 0, 1e-6f, fb.y, r.s);
        launch32_linear(fb.y, kC, wfin, kC, bfin, r.N, r.D, kC, euler ? 3 : 0, euler ? x : out, r.D, 0, none, 0, 0, dt, r.s);
        LAUNCHCHK();
        return 0;
    }
    FinalParams f{};
    f.h = h;
    f.nrows = r.N;
    f.mm = ModMap{modstep + c->final_off(), r.T * r.L, r.B, 0, r.mod_group_stride};
    f.shift_chunk = 0;
    f.scale_chunk = 1;
    f.w = c->wfin;
    f.bias = c->bfin;
    f.D = r.D;
    f.euler = euler;
    f.dt = dt;
    f.x = x;
    f.out = out;
    bool tail_done = false;
    for (int i = 0; i < c->nl; ++i) {
        const TrunkW& w = c->trunk[i];
        ModMap mm{modstep + c->trunk_off(i), r.T * r.L, r.B, 0, r.mod_group_stride};
        // residue axis on the tiled-attention path (L > 8): its out-projection may run inside the temporal q / k / v kernel
        ProjParams def_l{};
        {
            const bool fuse_lt = c->opt_fuse_proj_qkv && r.L > 8 && r.T > 8;
            if (int er = attn_sublayer(r, w.mha_l, h, r.N, axL, mm, 0, 1, 2, mk, true, true, fuse_lt ? &def_l : nullptr)) return er;
        }
        ProjParams deferred{};
        const bool fuse = c->opt_fuse_proj == 2 || (c->opt_fuse_proj == 1 && mlp_uses_rows(c, r.N)) ||
                          (c->opt_fuse_proj == 3 && !mlp_uses_rows(c, r.N));
        if (int er = attn_sublayer(r, w.mha_t, h, r.N, axT, mm, 3, 4, 5, mk, false, true, fuse ? &deferred : nullptr,
                                   def_l.a_bf16 ? &def_l : nullptr))
            return er;
        // the last layer's MLP may run the FinalLayer as its tail (then h is NOT written: not with a residual-stream trace)
        const bool last = i == c->nl - 1 && !trace_h;
        if (int er = mlp_sublayer(r, w.ffn, h, r.N, mm, 6, 7, 8, true, &deferred, w.mha_t.wo_stream, (long)step * c->nl + i,
                                  last ? &f : nullptr, last ? &tail_done : nullptr, next_h0 ? step + 1 : -1, next_h0))
            return er;
        if (trace_h) HIPCHK(hipMemcpyAsync(trace_h + (size_t)(i + 1) * r.N * kC, h, hbytes, hipMemcpyDeviceToDevice, r.s));
    }
    if (tail_done) return 0;
    { ProfScope ps(c, "final_euler", r.s); if (!g_dry) launch_final(f, r.s); }
    LAUNCHCHK();
    return 0;
}

static int make_run(Run* r, mdgen_ctx* c, const mdgen_shape* sh, int S, int t_shared, void* ws, size_t ws_bytes,
                    void* stream) {
    if (int e = check_shape(c, sh, S)) return e;
    if (!c->finalized) return fail(-6, "context not finalized (mdgen_ctx_finalize)");
    if (!ws) return fail(-1, "null workspace");
    if (int e = mdgen_workspace_layout(c, sh, S, t_shared, &r->lay)) return e;
    if (ws_bytes < r->lay.total_bytes)
        return fail(-7, "workspace too small: %zu < %zu bytes", ws_bytes, r->lay.total_bytes);
    if (((uintptr_t)ws & 255) != 0) return fail(-7, "workspace must be 256-byte aligned");
    if (c->opt_precision == 32 && !c->opt_keep_fp32) return fail(-6, "precision 32 requires option keep_fp32_weights");
    if (g_dry && c->opt_precision == 32) return fail(-2, "the dispatch plan covers the bf16 path");
    r->c = c;
    r->B = sh->B;
    r->T = sh->T;
    r->L = sh->L;
    r->D = c->D;
    r->S = S;
    r->N = (long)sh->B * sh->T * sh->L;
    r->Mp = (long)S * sh->B * sh->L;
    r->t_shared = t_shared;
    r->mod_step_stride = t_shared ? c->modrow : (long)sh->B * c->modrow;
    r->mod_group_stride = t_shared ? 0 : c->modrow;
    r->ws = (unsigned char*)ws;
    r->s = (hipStream_t)stream;
    r->hp = (float*)(r->ws + r->lay.h);
    r->qfp = r->ws + r->lay.qf;
    r->kfp = r->ws + r->lay.kf;
    r->vfp = r->ws + r->lay.vf;
    r->obufp = (__bf16*)(r->ws + r->lay.obuf);
    r->modp = (float*)(r->ws + r->lay.mod);
    r->ipa_out_p = (const float*)(r->ws + r->lay.ipa_out);
    r->ipa_step_stride = (long)sh->B * sh->L * kC;
    {
        const long maxrows = r->N > r->Mp ? r->N : r->Mp;
        r->split_cap = split_panels(c, maxrows);
        r->split_counters = (unsigned*)(r->ws + r->lay.split);
        r->split_part = (float*)(r->ws + r->lay.split + kSplitCounterBytes);
        r->split_hupd = r->split_part + (size_t)r->split_cap * kMlpSplit * kPanel * kC;
        // the counters must be zero when a launch starts (every launch leaves them so); the caller's workspace holds anything.
        // Eagerly on the call's stream, ahead of the (possibly replayed) graph.
        HIPCHK(hipMemsetAsync(r->split_counters, 0, kSplitCounterBytes, r->s));
    }
    r->fold_streams = nullptr;
    r->fold_b2g = nullptr;
    r->fold_ready = false;
    r->embase_p = nullptr;
    r->embase_step_stride = (long)sh->B * sh->L * kC;
    if (fold_on(c, r->N, t_shared, S)) {
        r->fold_streams = r->ws + r->lay.fold;
        r->fold_b2g = (float*)(r->fold_streams + (size_t)S * c->nl * kFoldStreamBytes);
    }
    return 0;
}

extern "C" int32_t mdgen_denoiser_forward(mdgen_ctx* c, const mdgen_shape* sh, const float* x, const float* t,
                                          const float* mask, const float* start_rot, const float* start_trans,
                                          const float* end_rot, const float* end_trans, const float* rel7,
                                          const float* x_cond, const int64_t* x_cond_mask, const int64_t* aatype, float* out,
                                          float* trace_h, float* trace_ipa, void* ws, size_t ws_bytes, void* stream) {
    if (!x || !t || !mask || !start_rot || !start_trans || !x_cond || !x_cond_mask || !aatype || !out)
        return fail(-1, "null tensor argument");
    Run r{};
    const int t_shared = (sh && sh->B == 1) ? 1 : 0;
    if (int e = make_run(&r, c, sh, 1, t_shared, ws, ws_bytes, stream)) return e;
    r.mask = mask;
    r.start_rot = start_rot;
    r.start_trans = start_trans;
    r.end_rot = end_rot;
    r.end_trans = end_trans;
    r.rel7_in = rel7;
    r.x_cond = x_cond;
    r.x_cond_mask = x_cond_mask;
    r.aatype = aatype;
    const int nv = c->opt_precision == 32 ? 1 : plan_views(r.B, r.T, r.L, 1);
    if (int e = prepare(r, t, nullptr, nv > 0 ? (long)((r.B + nv - 1) / nv) * r.T * r.L : r.N)) return e;
    if (trace_ipa)
        HIPCHK(hipMemcpyAsync(trace_ipa, r.ws + r.lay.ipa_out, (size_t)r.B * r.L * kC * 4, hipMemcpyDeviceToDevice, r.s));
    if (nv <= 1) return denoise_step(r, 0, const_cast<float*>(x), out, 0, 0.f, trace_h);
    if (trace_h) return fail(-2, "trace_h is not available when the batch needs more than one launch view");
    int b0 = 0;
    for (int i = 0; i < nv; ++i) {   // sequential sub-batch views on the caller's stream
        const int Bs = r.B / nv + (i < r.B % nv ? 1 : 0);
        const Run v = sub_run(r, b0, Bs, r.s);
        const long o = (long)b0 * r.T * r.L * r.D;
        if (int e = denoise_step(v, 0, const_cast<float*>(x) + o, out + o, 0, 0.f, nullptr)) return e;
        b0 += Bs;
    }
    return 0;
}

// torch.linspace(0, 1, n) in fp32 (ATen RangeFactories: symmetric evaluation around the midpoint)
static void linspace01(int n, std::vector<float>* out) {
    out->resize(n);
    const float step = 1.0f / (float)(n - 1);
    const int half = n / 2;
    for (int i = 0; i < n; ++i) (*out)[i] = i < half ? 0.0f + step * (float)i : 1.0f - step * (float)(n - 1 - i);
}

// Number of concurrent sub-batch streams for the Euler rollout (option "streams", default 2; 1 disables).
// Sub-batch streams pay when every piece still fills the chip by itself (cfg-2: 2 x 500 panels, +9 ... 12 %); pieces of
// fewer than 256 64-row panels leave CUs idle in BOTH streams (TPS shard B32 T100: 2 x 100 panels 68.3k frames/s, one stream
// 73.4k, profiles/r04_experiments.txt): the batch is then cut into fewer pieces, down to one.
static int n_streams(const Run& r) {
    int n = r.c->opt_streams;
    if (n > r.B) n = r.B;
    const long fill = (long)r.c->ncu * kPanel;
    if (r.c->opt_streams_auto && (long)n * fill > r.N) n = (int)(r.N / fill);
    if (n < 2 || r.c->prof_on || r.N < 4096 || r.c->opt_precision == 32) return 1;
    return n;
}

static int euler_steps(const Run& v, const std::vector<float>& tg, float* x) {
    if (g_dry) g_dry->push_back("@view");   // plan mode: a sub-batch view's launches start here
    bool h0_ready = false;
    for (int i = 0; i < v.S; ++i) {
        const float dt = tg[i + 1] - tg[i];
        bool next = false;
        if (int e = denoise_step(v, i, x, nullptr, 1, dt, nullptr, h0_ready, i + 1 < v.S ? &next : nullptr)) return e;
        h0_ready = next;
    }
    return 0;
}


static int euler_body(const Run& r_in, const std::vector<float>& tg, float* x) {
    Run r = r_in;
    const int ns = n_streams(r);
    // >= ns views; more when a view would exceed kMaxViewTokens (the fp32 kernels index with 64 bits: one view)
    const int nv = r.c->opt_precision == 32 ? 1 : plan_views(r.B, r.T, r.L, ns);
    if (nv == 0) return fail(-2, "sample too large for one launch");
    if (int e = prepare(r, nullptr, tg.data(), (long)((r.B + nv - 1) / nv) * r.T * r.L)) return e;
    r.c->live_streams = 1;
    if (nv == 1) return euler_steps(r, tg, x);
    struct Live {   // the views below run on ns streams at once: kernels that use context-owned scratch stay off meanwhile
        mdgen_ctx* c;
        ~Live() { c->live_streams = 1; }
    } live{r.c};
    r.c->live_streams = ns;
    // contiguous sub-batch views, view i on stream i % ns (fork after the shared preparation, join at the end)
    mdgen_ctx* c = r.c;
    if (ns > 1) HIPCHK(hipEventRecord(c->ev_fork, r.s));
    for (int i = 1; i < ns; ++i) HIPCHK(hipStreamWaitEvent(c->side[i - 1], c->ev_fork, 0));
    int b0 = 0;
    for (int i = 0; i < nv; ++i) {
        const int Bs = r.B / nv + (i < r.B % nv ? 1 : 0);
        hipStream_t st = (i % ns) == 0 ? r.s : c->side[i % ns - 1];
        const Run v = sub_run(r, b0, Bs, st);
        if (int e = euler_steps(v, tg, x + (long)b0 * r.T * r.L * r.D)) return e;
        b0 += Bs;
    }
    for (int i = 1; i < ns; ++i) {
        HIPCHK(hipEventRecord(c->ev_join[i - 1], c->side[i - 1]));
        HIPCHK(hipStreamWaitEvent(r.s, c->ev_join[i - 1], 0));
    }
    return 0;
}

// Replay the cached hipGraph for `key`, or capture `body` (which enqueues work on `s`, possibly forking onto the
// context's side streams and joining back) into a new one, cache it (LRU, 8 entries) and launch it.
static int replay_or_capture(mdgen_ctx* c, const std::vector<uint64_t>& key, hipStream_t s, const std::function<int()>& body) {
    if (!s) return fail(-8, "use_graph requires a non-default stream");
    for (size_t gi = 0; gi < c->graphs.size(); ++gi)
        if (c->graphs[gi].key == key) {
            if (gi + 1 != c->graphs.size()) {   // most recently used goes to the back (eviction takes the front)
                GraphEntry hit = c->graphs[gi];
                c->graphs.erase(c->graphs.begin() + gi);
                c->graphs.push_back(hit);
            }
            HIPCHK(hipGraphLaunch(c->graphs.back().exec, s));
            return 0;
        }
    GraphEntry ge;
    ge.key = key;
    HIPCHK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    const int e = body();
    hipError_t ce = hipStreamEndCapture(s, &ge.graph);
    if (e) {
        if (ge.graph) (void)hipGraphDestroy(ge.graph);
        return e;
    }
    if (ce != hipSuccess) {
        if (ge.graph) (void)hipGraphDestroy(ge.graph);
        return fail((int)ce, "hipStreamEndCapture failed: %s", hipGetErrorString(ce));
    }
    {
        hipError_t ie = hipGraphInstantiate(&ge.exec, ge.graph, nullptr, nullptr, 0);
        if (ie != hipSuccess) {
            (void)hipGraphDestroy(ge.graph);
            return fail((int)ie, "hipGraphInstantiate failed: %s", hipGetErrorString(ie));
        }
    }
    if (c->graphs.size() >= 8) {
        (void)hipGraphExecDestroy(c->graphs.front().exec);
        (void)hipGraphDestroy(c->graphs.front().graph);
        c->graphs.erase(c->graphs.begin());
    }
    c->graphs.push_back(ge);
    HIPCHK(hipGraphLaunch(ge.exec, s));
    return 0;
}

extern "C" int32_t mdgen_sample_euler(mdgen_ctx* c, const mdgen_shape* sh, int32_t S, float* x, const float* mask,
                                      const float* start_rot, const float* start_trans, const float* end_rot,
                                      const float* end_trans, const float* rel7, const float* x_cond,
                                      const int64_t* x_cond_mask, const int64_t* aatype, void* ws, size_t ws_bytes,
                                      int32_t use_graph, void* stream) {
    if (!x || !mask || !start_rot || !start_trans || !x_cond || !x_cond_mask || !aatype)
        return fail(-1, "null tensor argument");
    Run r{};
    if (int e = make_run(&r, c, sh, S, 1, ws, ws_bytes, stream)) return e;
    r.mask = mask;
    r.start_rot = start_rot;
    r.start_trans = start_trans;
    r.end_rot = end_rot;
    r.end_trans = end_trans;
    r.rel7_in = rel7;
    r.x_cond = x_cond;
    r.x_cond_mask = x_cond_mask;
    r.aatype = aatype;
    std::vector<float> tg;
    linspace01(S + 1, &tg);   // integrators.py:88  th.linspace(t0, t1, num_steps)
    if (!use_graph || c->prof_on) return euler_body(r, tg, x);
    std::vector<uint64_t> key = {0u, (uint64_t)sh->B, (uint64_t)sh->T, (uint64_t)sh->L, (uint64_t)S, (uint64_t)x,
                                 (uint64_t)mask, (uint64_t)start_rot, (uint64_t)start_trans, (uint64_t)end_rot,
                                 (uint64_t)end_trans, (uint64_t)x_cond, (uint64_t)x_cond_mask, (uint64_t)aatype,
                                 (uint64_t)ws, (uint64_t)n_streams(r), (uint64_t)(c->opt_residue_l4 | c->opt_mlp_path << 8 | c->opt_fuse_proj << 12 | c->opt_fuse_proj_qkv << 20 | c->opt_flash_proj << 24 | (uint64_t)c->opt_panel_waves << 32 | (uint64_t)c->opt_flash_rotate << 36 | (uint64_t)c->opt_flash_proj_form << 40 | (uint64_t)c->opt_small_split << 44 | (uint64_t)c->opt_mlp_fold << 45 | (uint64_t)c->opt_mlp_tail << 46 | (uint64_t)c->opt_embed_split << 48), (uint64_t)c->opt_precision,
                                 (uint64_t)c->opt_attn_path, (uint64_t)rel7};
    return replay_or_capture(c, key, r.s, [&]() { return euler_body(r, tg, x); });
}

// Multi-block rollout (sim_inference.py:61-98, 110-113) as ONE call / ONE hipGraph: per block
//   conditioning frame (B, L) expanded over T -> prep_batch latents (wrapper.py:298-342)
//   -> S Euler steps from the block's noise (wrapper.py:439-447) -> atom14 (wrapper.py:456-478)
//   -> last frame -> next block's conditioning frame (sim_inference.py:91-96),
// nothing returning to the host in between.
extern "C" int32_t mdgen_rollout_euler(mdgen_ctx* c, const mdgen_shape* sh, int32_t S, int32_t n_blocks, float* zs,
                                       const float* mask, float* cond_rots, float* cond_trans, float* cond_torsions,
                                       const int64_t* seqres, float* x_cond, int64_t* x_cond_mask,
                                       const mdgen_residue_tables* tb, float* atom14, void* ws, size_t ws_bytes,
                                       int32_t use_graph, void* stream) {
    if (!zs || !mask || !cond_rots || !cond_trans || !cond_torsions || !seqres || !x_cond || !x_cond_mask || !tb || !atom14)
        return fail(-1, "null tensor argument");
    if (!tb->default_frames || !tb->lit_positions || !tb->atom14_group || !tb->atom14_mask || !tb->atom37_to_atom14 ||
        !tb->atom37_mask || !tb->chi_atom_indices || !tb->chi_angles_mask)
        return fail(-1, "null residue table");
    if (n_blocks < 1) return fail(-2, "n_blocks must be >= 1");
    if (c && c->d.tps_condition) return fail(-2, "the block rollout is defined for forward-simulation models (sim_condition)");
    Run r{};
    if (int e = make_run(&r, c, sh, S, 1, ws, ws_bytes, stream)) return e;
    if ((long)n_blocks * sh->T > 2000000000L / ((long)sh->L * 42)) return fail(-2, "trajectory too long for one call");
    r.mask = mask;
    r.start_rot = cond_rots;
    r.start_trans = cond_trans;
    r.end_rot = nullptr;
    r.end_trans = nullptr;
    r.x_cond = x_cond;
    r.x_cond_mask = x_cond_mask;
    r.aatype = seqres;
    std::vector<float> tg;
    linspace01(S + 1, &tg);
    const mdgen_residue_tables t = *tb;
    const long blk = (long)sh->B * sh->T * sh->L * r.D;
    float* tmask_scratch = (float*)(r.ws + r.lay.rel7);   // (B, L, 7) fp32: fits the (unused, non-TPS) 2 x (B, L, 7) slot
    auto body = [&]() -> int {
        for (int b = 0; b < n_blocks; ++b) {
            float* x = zs + (long)b * blk;
            launch_prep_latents(r.B, r.T, r.L, 0, 1, 0, cond_rots, cond_trans, cond_torsions, nullptr, x_cond, x_cond_mask, r.s);
            LAUNCHCHK();
            if (int e = euler_body(r, tg, x)) return e;
            launch_samples_to_atom14(r.B, r.T, r.L, r.D, 0, x, cond_rots, cond_trans, seqres, t.default_frames,
                                     t.lit_positions, t.atom14_group, t.atom14_mask, atom14, n_blocks * r.T, b * r.T, r.s);
            LAUNCHCHK();
            // last frame of this block (frame b*T + T-1 of every trajectory) -> conditioning frame of the next
            const float* last = atom14 + ((long)(b + 1) * r.T - 1) * r.L * 42;
            launch_atom14_to_cond(r.B, r.L, last, (long)n_blocks * r.T * r.L * 42, seqres, t.atom37_to_atom14, t.atom37_mask,
                                  t.chi_atom_indices, t.chi_angles_mask, cond_rots, cond_trans, cond_torsions,
                                  tmask_scratch, r.s);
            LAUNCHCHK();
        }
        return 0;
    };
    if (!use_graph || c->prof_on) return body();
    std::vector<uint64_t> key = {1u, (uint64_t)sh->B, (uint64_t)sh->T, (uint64_t)sh->L, (uint64_t)S, (uint64_t)n_blocks,
                                 (uint64_t)zs, (uint64_t)mask, (uint64_t)cond_rots, (uint64_t)cond_trans,
                                 (uint64_t)cond_torsions, (uint64_t)seqres, (uint64_t)x_cond, (uint64_t)x_cond_mask,
                                 (uint64_t)atom14, (uint64_t)ws, (uint64_t)n_streams(r), (uint64_t)(c->opt_residue_l4 | c->opt_mlp_path << 8 | c->opt_fuse_proj << 12 | c->opt_fuse_proj_qkv << 20 | c->opt_flash_proj << 24 | (uint64_t)c->opt_panel_waves << 32 | (uint64_t)c->opt_flash_rotate << 36 | (uint64_t)c->opt_flash_proj_form << 40 | (uint64_t)c->opt_small_split << 44 | (uint64_t)c->opt_mlp_fold << 45 | (uint64_t)c->opt_mlp_tail << 46 | (uint64_t)c->opt_embed_split << 48),
                                 (uint64_t)t.default_frames, (uint64_t)t.atom37_to_atom14, (uint64_t)c->opt_precision,
                                 (uint64_t)c->opt_attn_path};
    return replay_or_capture(c, key, r.s, body);
}

// ---------------------------------------------------------------------------------------------
// dispatch plan (host only)
// ---------------------------------------------------------------------------------------------
// Which kernel classes a call of this shape launches, and how often: the sampler's orchestration code above, run in plan mode
// (g_dry) on a context that owns no device memory.  mode 0: mdgen_sample_euler as the product runs it (sub-batch streams);
// 1: mdgen_denoiser_forward; 2: mdgen_sample_euler as mdgen_profile_enable sees it (one stream); 3: mdgen_denoiser_forward with trace_h.  options: "name=value,..."
// (mdgen_ctx_set_option names).  ncu / xcd_round_robin: what mdgen_ctx_create would have found on the device.
// Output: {"streams": n, "prepare": {"<class>": launches, ...}, "views": [{"B": samples of the view, "classes": {...}}, ...]}.
extern "C" int32_t mdgen_debug_dispatch_plan(const mdgen_shape* sh, int32_t n_steps, int32_t mode, int32_t tps_condition,
                                             int32_t num_layers, int32_t ncu, int32_t xcd_round_robin, const char* options,
                                             char* buf, size_t buflen) {
    if (!sh || !buf || buflen < 64) return fail(-1, "null argument");
    if (mode < 0 || mode > 3) return fail(-2, "mode: 0 sample_euler, 1 forward, 2 sample_euler under the profiler, 3 forward with trace_h");
    if (num_layers < 1 || num_layers > 8 || ncu < 1) return fail(-2, "num_layers in 1..8, ncu >= 1");
    mdgen_ctx ctx;   // no device memory, no streams: plan mode never touches them
    mdgen_ctx* c = &ctx;
    c->nl = num_layers;
    c->D = tps_condition ? 28 : 21;
    c->d.num_layers = num_layers;
    c->d.latent_dim = c->D;
    c->d.tps_condition = tps_condition;
    c->d.abs_pos_emb = 0;
    c->modrow = (15 * c->nl + 2) * kC;
    c->trunk.resize(c->nl);
    c->ipa.resize(c->nl);
    c->finalized = true;
    c->ncu = ncu;
    c->xcd_round_robin = xcd_round_robin != 0;
    c->prof_on = mode == 2;
    for (std::string rest = options ? options : ""; !rest.empty();) {
        const size_t comma = rest.find(',');
        const std::string kv = rest.substr(0, comma);
        rest = comma == std::string::npos ? "" : rest.substr(comma + 1);
        const size_t eq = kv.find('=');
        if (eq == std::string::npos) return fail(-2, "options: name=value[,name=value...]");
        if (int e = mdgen_ctx_set_option(c, kv.substr(0, eq).c_str(), std::atoi(kv.c_str() + eq + 1))) return e;
    }
    std::vector<std::string> plan;
    struct Dry {
        Dry(std::vector<std::string>* p) { g_dry = p; }
        ~Dry() { g_dry = nullptr; }
    } dry(&plan);
    const bool fwd = mode == 1 || mode == 3;
    const int S = fwd ? 1 : n_steps;
    const int t_shared = fwd ? (sh->B == 1 ? 1 : 0) : 1;
    Run r{};
    if (int e = make_run(&r, c, sh, S, t_shared, (void*)4096, (size_t)1 << 60, nullptr)) return e;
    float* fake = (float*)4096;   // never dereferenced: plan mode launches nothing
    r.mask = r.start_rot = r.start_trans = r.end_rot = r.end_trans = r.x_cond = fake;
    r.x_cond_mask = r.aatype = (const int64_t*)fake;
    int ns = 1, nv = 1;
    if (fwd) {
        nv = plan_views(r.B, r.T, r.L, 1);
        if (int e = prepare(r, fake, nullptr, (long)((r.B + nv - 1) / nv) * r.T * r.L)) return e;
        int b0 = 0;
        for (int i = 0; i < nv; ++i) {
            const int Bs = r.B / nv + (i < r.B % nv ? 1 : 0);
            const Run v = nv > 1 ? sub_run(r, b0, Bs, r.s) : r;
            plan.push_back("@view");
            if (int e = denoise_step(v, 0, fake, fake, 0, 0.f, mode == 3 ? fake : nullptr)) return e;
            b0 += Bs;
        }
    } else {
        std::vector<float> tg;
        linspace01(S + 1, &tg);
        ns = n_streams(r);
        nv = plan_views(r.B, r.T, r.L, ns);
        if (int e = euler_body(r, tg, fake)) return e;
    }
    // {"streams": n, "prepare": {class: launches}, "views": [{"B": samples, "classes": {class: launches}}, ...]}
    auto dump = [](const std::map<std::string, long>& agg) {
        std::string o = "{";
        bool first = true;
        for (const auto& kv : agg) {
            o += std::string(first ? "" : ", ") + "\"" + kv.first + "\": " + std::to_string(kv.second);
            first = false;
        }
        return o + "}";
    };
    std::vector<std::map<std::string, long>> parts(1);
    for (const auto& k : plan) {
        if (k == "@view") parts.emplace_back();
        else ++parts.back()[k];
    }
    if ((int)parts.size() != nv + 1) return fail(-7, "internal: %d views planned, %d recorded", nv, (int)parts.size() - 1);
    std::string js = "{\"streams\": " + std::to_string(ns) + ", \"prepare\": " + dump(parts[0]) + ", \"views\": [";
    for (int i = 0; i < nv; ++i)
        js += std::string(i ? ", " : "") + "{\"B\": " + std::to_string(r.B / nv + (i < r.B % nv ? 1 : 0)) + ", \"classes\": " + dump(parts[i + 1]) + "}";
    js += "]}";
    if (js.size() + 1 > buflen) return fail(-7, "plan buffer too small");
    std::memcpy(buf, js.c_str(), js.size() + 1);
    return 0;
}

// ---------------------------------------------------------------------------------------------
// measurement
// ---------------------------------------------------------------------------------------------
extern "C" int32_t mdgen_profile_enable(mdgen_ctx* c, int32_t on) {
    if (!c) return fail(-1, "null context");
    c->prof_on = on != 0;
    return 0;
}

extern "C" int32_t mdgen_profile_phase_trace(mdgen_ctx* c, uint64_t* dev_buf, int64_t capacity_words) {
    if (!c) return fail(-1, "null context");
    c->phase_trace = (unsigned long long*)dev_buf;
    c->phase_trace_cap = dev_buf ? capacity_words : 0;
    return 0;
}

extern "C" int32_t mdgen_profile_report(mdgen_ctx* c, void* stream, char* buf, size_t buflen) {
    if (!c || !buf || buflen < 8) return fail(-1, "null argument");
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    std::map<std::string, std::pair<long, double>> agg;
    for (auto& r : c->prof) {
        float ms = 0.f;
        if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
            auto& a = agg[r.cls];
            a.first += 1;
            a.second += ms;
        }
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
    }
    c->prof.clear();
    std::string js = "{";
    bool first = true;
    for (auto& kv : agg) {
        char tmp[160];
        snprintf(tmp, sizeof(tmp), "%s\"%s\": {\"count\": %ld, \"ms\": %.6f}", first ? "" : ", ", kv.first.c_str(),
                 kv.second.first, kv.second.second);
        js += tmp;
        first = false;
    }
    {   // not a kernel class: the context's placement-probe result and the split-form launches since the last report
        char tmp[200];
        snprintf(tmp, sizeof(tmp), "%s\"@context\": {\"count\": %ld, \"ms\": 0.0, \"xcd_round_robin\": %d, \"ncu\": %d}", first ? "" : ", ",
                 c->n_split_launches, c->xcd_round_robin ? 1 : 0, c->ncu);
        js += tmp;
        c->n_split_launches = 0;
    }
    js += "}";
    if (js.size() + 1 > buflen) return fail(-7, "profile buffer too small");
    std::memcpy(buf, js.c_str(), js.size() + 1);
    return 0;
}

// ---------------------------------------------------------------------------------------------
// SE(3) / pre / post
// ---------------------------------------------------------------------------------------------
static bool any_null(std::initializer_list<const void*> ps) {
    for (const void* p : ps)
        if (!p) return true;
    return false;
}
#define NONNULL(...) \
    if (any_null({__VA_ARGS__})) return fail(-1, "null tensor argument")

extern "C" int32_t mdgen_rigid_compose(int64_t n, const float* r1, const float* t1, const float* r2, const float* t2,
                                       float* ro, float* to, void* stream) {
    NONNULL(r1, t1, r2, t2, ro, to);
    if (n <= 0) return 0;
    launch_rigid_compose(n, r1, t1, r2, t2, ro, to, (hipStream_t)stream);
    LAUNCHCHK();
    return 0;
}
extern "C" int32_t mdgen_rigid_invert(int64_t n, const float* r, const float* t, float* ro, float* to, void* stream) {
    NONNULL(r, t, ro, to);
    if (n <= 0) return 0;
    launch_rigid_invert(n, r, t, ro, to, (hipStream_t)stream);
    LAUNCHCHK();
    return 0;
}
extern "C" int32_t mdgen_rigid_apply(int64_t n, int64_t ppf, const float* r, const float* t, const float* pts,
                                     float* out, int32_t inverse, void* stream) {
    NONNULL(r, t, pts, out);
    if (n <= 0 || ppf <= 0) return 0;
    launch_rigid_apply(n, ppf, r, t, pts, out, inverse, (hipStream_t)stream);
    LAUNCHCHK();
    return 0;
}
extern "C" int32_t mdgen_quat_to_rot(int64_t n, const float* q, int32_t normalize, float* rot, void* stream) {
    NONNULL(q, rot);
    if (n <= 0) return 0;
    launch_quat_to_rot(n, q, normalize, rot, (hipStream_t)stream);
    LAUNCHCHK();
    return 0;
}
extern "C" int32_t mdgen_rot_to_quat(int64_t n, const float* rot, float* q, void* stream) {
    NONNULL(rot, q);
    if (n <= 0) return 0;
    launch_rot_to_quat(n, rot, q, (hipStream_t)stream);
    LAUNCHCHK();
    return 0;
}
extern "C" int32_t mdgen_prep_latents(const mdgen_shape* sh, int32_t tps, int32_t cond_interval, const float* rots, const float* trans,
                                      const float* torsions, float* latents, float* x_cond, int64_t* x_cond_mask,
                                      void* stream) {
    if (!sh) return fail(-1, "null shape");
    NONNULL(rots, trans, torsions, latents, x_cond, x_cond_mask);
    if (sh->B < 1 || sh->T < 1 || sh->L < 1) return fail(-2, "B, T, L must be >= 1");
    if (cond_interval < 0) return fail(-2, "cond_interval must be >= 0 (0 = none)");
    launch_prep_latents(sh->B, sh->T, sh->L, tps, 0, cond_interval, rots, trans, torsions, latents, x_cond, x_cond_mask,
                        (hipStream_t)stream);
    LAUNCHCHK();
    return 0;
}
extern "C" int32_t mdgen_samples_to_atom14(const mdgen_shape* sh, int32_t D, int32_t tps, const float* samples,
                                           const float* rot0, const float* trans0, const int64_t* seqres,
                                           const float* default_frames, const float* lit_positions,
                                           const int64_t* atom14_group, const float* atom14_mask, float* atom14,
                                           void* stream) {
    if (!sh) return fail(-1, "null shape");
    NONNULL(samples, rot0, trans0, seqres, default_frames, lit_positions, atom14_group, atom14_mask, atom14);
    if (sh->B < 1 || sh->T < 1 || sh->L < 1) return fail(-2, "B, T, L must be >= 1");
    if (D < (tps ? 28 : 21)) return fail(-2, "latent_dim too small");
    launch_samples_to_atom14(sh->B, sh->T, sh->L, D, tps, samples, rot0, trans0, seqres, default_frames, lit_positions,
                             atom14_group, atom14_mask, atom14, sh->T, 0, (hipStream_t)stream);
    LAUNCHCHK();
    return 0;
}
extern "C" int32_t mdgen_atom14_to_cond(int32_t B, int32_t L, const float* atom14, const int64_t* seqres,
                                        const int64_t* a37to14, const float* a37mask, const int64_t* chi_idx,
                                        const float* chi_mask, float* rots, float* trans, float* tors, float* tmask,
                                        void* stream) {
    NONNULL(atom14, seqres, a37to14, a37mask, chi_idx, chi_mask, rots, trans, tors, tmask);
    if (B < 1 || L < 1) return fail(-2, "B, L must be >= 1");
    launch_atom14_to_cond(B, L, atom14, (long)L * 42, seqres, a37to14, a37mask, chi_idx, chi_mask, rots, trans, tors, tmask,
                          (hipStream_t)stream);
    LAUNCHCHK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// flow-matching training target and loss (forward only)
// ---------------------------------------------------------------------------------------------
extern "C" int32_t mdgen_path_plan(int64_t B, int64_t per_sample, int32_t path_type, const float* t, const float* x0,
                                   const float* x1, float* xt, float* ut, void* stream) {
    NONNULL(t, x0, x1, xt, ut);
    if (B < 1 || per_sample < 1) return fail(-2, "B, per_sample must be >= 1");
    if (path_type != 0 && path_type != 1) return fail(-2, "path_type: 0 = Linear, 1 = GVP");
    launch_path_plan(t, x0, x1, xt, ut, per_sample, B, path_type, (hipStream_t)stream);
    LAUNCHCHK();
    return 0;
}

extern "C" int32_t mdgen_masked_mse(int64_t B, int64_t per_sample, const float* pred, const float* target,
                                    const float* mask, float* loss, void* stream) {
    NONNULL(pred, target, mask, loss);
    if (B < 1 || per_sample < 1) return fail(-2, "B, per_sample must be >= 1");
    launch_masked_mse(pred, target, mask, loss, per_sample, B, (hipStream_t)stream);
    LAUNCHCHK();
    return 0;
}

extern "C" int32_t mdgen_from_3_points(int64_t n, const float* p_neg_x, const float* origin, const float* p_xy,
                                       float* rot, float* trans, void* stream) {
    NONNULL(p_neg_x, origin, p_xy, rot, trans);
    if (n < 1) return fail(-2, "n must be >= 1");
    launch_from_3_points(n, p_neg_x, origin, p_xy, rot, trans, (hipStream_t)stream);
    LAUNCHCHK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// optimiser side of the training step (flat fp32 buffers; csrc/k_optim.hip)
// ---------------------------------------------------------------------------------------------
extern "C" int32_t mdgen_grad_sumsq(int64_t n, const float* grads, float scale, float* scratch, int32_t scratch_floats,
                                    float* out, void* stream) {
    NONNULL(grads, scratch, out);
    if (n < 1) return fail(-2, "n must be >= 1");
    if (scratch_floats < 2) return fail(-2, "scratch_floats must be >= 2");
    if (((uintptr_t)scratch & 7) != 0) return fail(-2, "scratch must be 8-byte aligned");
    const int nb = scratch_floats / 2 < 512 ? scratch_floats / 2 : 512;   // fp64 partial sums
    launch_sumsq(grads, n, scale, scratch, nb, out, (hipStream_t)stream);
    LAUNCHCHK();
    return 0;
}

extern "C" int32_t mdgen_adam_step(int64_t n, float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                                   int32_t step, float lr, float beta1, float beta2, float eps, float weight_decay,
                                   int32_t adamw, float grad_scale, const float* sumsq, float max_norm, void* stream) {
    NONNULL(params, grads, exp_avg, exp_avg_sq);
    if (n < 1 || step < 1) return fail(-2, "n and step must be >= 1");
    if (!(beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f) || !(eps > 0.f) || !(lr >= 0.f))
        return fail(-2, "invalid Adam hyper-parameters");
    const double bc1 = 1.0 - std::pow((double)beta1, (double)step);          // torch: 1 - beta ** step (python floats)
    const double bc2 = 1.0 - std::pow((double)beta2, (double)step);
    launch_adam(params, grads, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, adamw, (float)bc1,
                (float)std::sqrt(bc2), grad_scale, sumsq, max_norm, (hipStream_t)stream);
    LAUNCHCHK();
    return 0;
}

extern "C" int32_t mdgen_ema_update(int64_t n, float* ema, const float* params, float decay, void* stream) {
    NONNULL(ema, params);
    if (n < 1) return fail(-2, "n must be >= 1");
    launch_ema(ema, params, n, 1.0f - decay, (hipStream_t)stream);
    LAUNCHCHK();
    return 0;
}

#include "train.inc"
