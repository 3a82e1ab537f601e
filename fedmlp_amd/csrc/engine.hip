// FedMLP engine: ResNet-18 and EfficientNet-B0 training/eval graphs over the HIP kernels + the
// C ABI of include/fedmlp_hip.h.  Host code only (kernels live in igemm/wgrad/elementwise/
// heads/effnet.hip).  One engine per process/GPU.
//
// Data layout in HBM (all fp32):
//   state   : [conv weights OHWI (stem padded to [64][7][8][4])][fc.W][fc.b][pad]
//             [gamma of all 20 BN][beta of all 20 BN] | [running_mean all][running_var all]
//             `- trainable (Adam, grads mirror this part) -'   `- FedAvg'd, not trained -'
//   activations NHWC; train-mode forward keeps per conv the raw output y, per block
//   the post-BN/ReLU z1 and the block output; the two views of a FedMLP step are
//   one batch of 2B images with per-view ("group") BN statistics.
//   EfficientNet-B0 (model 1): channel counts are padded to multiples of 16 inside the engine
//   (24->32, 40->48; padded weights/gamma/beta are 0 and stay 0 under Adam, so padded channels
//   carry exact zeros); depthwise weights are [k*k][C]; squeeze-excite W1 [Cs][C], W2 kept transposed [Cs][C].
#include <math.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "../../include/fedmlp_hip.h"
#include "../../include/fedmlp_hip_debug.h"
#include "comm.h"
#include "common.h"
#include "kernels.h"
#include "pwconv.h"

static thread_local std::string g_err;
const char* fm_last_error(void) { return g_err.c_str(); }
const char* fm_version(void) { return "fedmlp_hip 0.2 (gfx950)"; }

#define HIPCHK(x)                                                                              \
    do {                                                                                       \
        hipError_t e_ = (x);                                                                   \
        if (e_ != hipSuccess) {                                                                \
            char b_[512];                                                                      \
            snprintf(b_, sizeof b_, "%s:%d %s -> %s", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            g_err = b_;                                                                        \
            return FM_ERR_HIP;                                                                 \
        }                                                                                      \
    } while (0)
#define ARGCHK(c, msg)                                                       \
    do {                                                                     \
        if (!(c)) { g_err = std::string("bad argument: ") + msg; return FM_ERR_ARG; } \
    } while (0)

namespace {

struct DgradClass {
    TapList taps;
    int dh[9], dw[9];
    int ph, pw;
    int nsteps;
    float* wpack = nullptr;
    long long sp_off = -1;    // bf16 planes of wpack inside fm_engine::wsp_d (2-byte units), -1 = none
    long long bm_off = -1;    // block-major planes of wpack inside fm_engine::wbm_d (pconv.hip), -1 = none
};

struct Conv {
    int cin, cout, k, stride, pad;
    int cin_p, kw_p;          // padded input channels / kernel width of the engine layout
    bool stem3 = false;       // ResNet's 7x7 stem, packed: K = 7 rows x (8 taps x 3 channels) + 8 zeros = 176 (engine.hip add_conv)
    int Hp = 0, Wp = 0;       // stem3: framed input sizes (hin + 6, win + 8)
    int cout_p;               // rows of the engine weight matrix (= cout for ResNet, padded to 16 for EfficientNet)
    int hin, win, hout, wout;
    size_t w_off;             // offset into the state arena
    size_t w_numel;           // cout_p*k*kw_p*cin_p
    int Kw;                   // k*kw_p*cin_p  (GEMM K of fwd, N of wgrad)
    int nsteps;               // Kw/16 (igemm K steps)
    int4* tab = nullptr;      // [Kw/4]
    int ncls = 0;
    DgradClass cls[4];
    int bn;                   // index of the BatchNorm that follows
    long long sp_off = -1;    // bf16 planes of the forward weights inside fm_engine::wsp_f / twsp_f (2-byte units), -1 = none
    long long bm_off = -1;    // block-major planes of the forward weights inside fm_engine::wbm_f / twbm_f (pconv.hip), -1 = none
    long long wb_off = -1, wbt_off = -1;   // bf16 shadow of a 1x1 conv's weights [cout_p][cin_p] / transposed (bf16 mode)
    double macs_per_img;      // algorithmic MACs (real k, real cin)
    float* y = nullptr;       // raw conv output (train) [max_images][hout][wout][cout]
};

struct Bn {
    int C;                    // engine channels (padded)
    int C_real;               // state_dict channels
    int ch_off;               // channel offset inside the all-BN vectors
    float *mean, *istd, *scale, *shift;   // [2][C] train-mode per-group
};

struct Block {
    int c1, c2, ds;
    float* z1 = nullptr;
    float* out = nullptr;
    float *t_z1 = nullptr, *t_out = nullptr, *t_dsy = nullptr;     // the side-stream teacher's set (see fm_engine::st2)
    // planes mode: block-major bf16 planes of z1 / out (what the conv GEMMs read), and the teacher's set
    unsigned short *z1p = nullptr, *outp = nullptr, *t_z1p = nullptr, *t_outp = nullptr;
};

struct MBConv {               // EfficientNet block (efficientnet-pytorch MBConvBlock)
    int k, s, expand, cin, cout, cin_p, ce, ce_p, cout_p, cs;
    int hin, win, hout, wout, pad_t, pad_l;
    bool skip;
    int c_exp = -1, c_proj = -1;          // 1x1 conv indices
    int bn0 = -1, bn1 = -1, bn2 = -1;
    size_t dw_off, w1_off, b1_off, w2_off, b2_off;
    float *a_e = nullptr, *y_d = nullptr, *a_s = nullptr, *out = nullptr;
    float *sq = nullptr, *rpre = nullptr, *gate = nullptr;
    // second set for the teacher's eval forward when it runs on the side stream next to the student's train forward
    float *t_a_e = nullptr, *t_y_d = nullptr, *t_a_s = nullptr, *t_out = nullptr;
    float *t_sq = nullptr, *t_rpre = nullptr, *t_gate = nullptr;
};

struct StateEntry {           // one state_dict entry, in reference key order
    int kind;                 // 0 OIHW weight (re-laid out to O,H,W(pad),I(pad)), 1 float vector, 2 int64 counter
    int conv;                 // conv index (kind 0, -1 for depthwise / squeeze-excite weights)
    size_t eng_off;           // engine arena offset
    size_t n;                 // elements in state_dict form
    int bn;                   // counter's BN index (kind 2)
    int O = 0, I = 0, KH = 0, KW = 0, Wpad = 0, Ipad = 0;   // kind 0 geometry
    int Ostride = 0;          // engine row stride when it exceeds KH*Wpad*Ipad (packed stem: 176), 0 = dense
};

struct EvPair { hipEvent_t a, b; int family; double flops; };
struct OpEv { hipEvent_t a, b; int id; };

}  // namespace

struct fm_engine {
    fm_config cfg;
    hipStream_t st = nullptr;
    int C = 0, D = 512, H = 0, W = 0, maxB = 0;
    std::vector<Conv> convs;
    std::vector<Bn> bns;
    std::vector<Block> blocks;
    std::vector<MBConv> mbs;
    int model = 0, c_stem = 0, c_head = -1, bn_stem = 0, bn_head = -1;
    float bn_eps = 1e-5f, bn_mom = 0.1f;
    int maxC = 512;                   // widest BN (sizes ca/cb/cc and the partial-sum workspace)
    float *a0 = nullptr;              // EfficientNet: swish(bn(stem))
    float *T_small = nullptr, *T_mid = nullptr, *T_big = nullptr;
    float *se_dgp = nullptr, *se_drp = nullptr, *se_ds = nullptr, *se_pool = nullptr, *hfeat = nullptr;
    const float *dc_dev = nullptr, *drop_dev = nullptr;   // caller-owned stochastic multipliers (or null)
    int pending_views = 0, pending_B = 0;                 // fm_forward_train awaiting fm_backward_step
    std::vector<int64_t> tcounters;                       // the teacher's num_batches_tracked
    std::vector<StateEntry> entries;
    int n_bn_ch = 0;
    size_t NP = 0, NS = 0;            // trainable floats (padded to 4), whole state floats
    size_t off_fcw = 0, off_fcb = 0, off_gamma = 0, off_beta = 0, off_rm = 0, off_rv = 0;
    int64_t nf_sd = 0, ni_sd = 0;     // state_dict sizes
    float *state = nullptr, *tstate = nullptr, *grad = nullptr, *adam_m = nullptr, *adam_v = nullptr;
    float *ev_scale = nullptr, *ev_shift = nullptr, *tev_scale = nullptr, *tev_shift = nullptr;
    float* stage_sd = nullptr;        // device staging buffer in state_dict order
    std::vector<int64_t> counters;    // num_batches_tracked per BN
    fm_adam hp{3e-5f, 0.9f, 0.999f, 1e-8f, 5e-4f};
    int64_t adam_t = 0;
    bool ev_dirty = true, tev_dirty = true;
    // workspaces
    float *x4 = nullptr, *p0 = nullptr, *dyh0 = nullptr;
    float* x3 = nullptr;      // packed stem: zero-framed NHWC3 input [max_images][H + 6][W + 8][3]
    // planes mode at 112-pixel stem rows (stem_rows.hip): the framed input as bf16 planes of four channels per pixel
    // [3][max_images][H + 6][W + 8][4], and the stem's weight planes [7][3][64][32] of the student and of the teacher
    bool stem_rows = false;
    unsigned short *x3p = nullptr, *wst = nullptr, *twst = nullptr;
    long long x3p_plane_elems = 0;
    uint8_t* idx0 = nullptr;
    float *GA = nullptr, *GB = nullptr, *GC = nullptr, *GD = nullptr, *GE = nullptr;
    float *ws_stats = nullptr, *ws_part = nullptr, *ws_slab = nullptr;
    // what ws_stats holds: the BN partial sums of conv `stats_conv`, `stats_tiles_n` tiles per group -- written by the conv_fwd
    // that launched the GEMM (the count depends on the kernel and on that call's operand prologue), read by the finalize
    int stats_conv = -1, stats_tiles_n = 0;
    size_t slab_floats = 0;
    float *ca = nullptr, *cb = nullptr, *cc = nullptr;
    float *feat = nullptr, *logits = nullptr, *tfeat = nullptr, *tlogits = nullptr, *dlogits = nullptr;
    // prototypes
    float* psum = nullptr;
    int64_t *pcnt = nullptr, *tcnt = nullptr;
    float* zeros = nullptr;
    float* sk_slab = nullptr;
    int* sk_counters = nullptr;
    int *sel_counts = nullptr, *sel_top = nullptr, *sel_bot = nullptr, *cls_dev = nullptr;
    int sel_cap = 0;
    int* tag_buf = nullptr;           // fm_select_topk_rows: [pool sizes ncls | counts 2 ncls | top ncls*cap | bot ncls*cap | rows ncls*stride]
    size_t tag_buf_ints = 0;
    // profiling
    bool prof = false, prof_fail = false;
    hipError_t soft_err = hipSuccess;      // first failed event record / stream wait of the current step (soft())
    // a kernel's own failure report (pconv.hip: a stream-K part that never arrived): dev_err is read by the optimizer kernel
    // (the step does not touch the weights), host_err is its host-mapped twin that STEP_DONE checks at every call
    int* dev_err = nullptr;
    int* host_err = nullptr;
    int* host_err_dev = nullptr;
    // per-op timing (FM profile leg of tools/op_profile.py): label = "<op>@<block>"
    bool oprof = false;
    int ctx = -1;
    std::vector<std::string> op_names;
    std::vector<double> op_ms;
    std::vector<int64_t> op_n;
    std::vector<OpEv> opevs;
    std::vector<EvPair> evs;
    std::vector<hipEvent_t> ev_free;
    double prof_ms[FM_PROFILE_FAMILIES] = {}, prof_flops[FM_PROFILE_FAMILIES] = {};
    int64_t prof_n[FM_PROFILE_FAMILIES] = {};
    std::vector<void*> allocs;
    // dgrad weight packs: rebuilt by ONE launch after every optimizer step (and lazily after any
    // external change of the state), not per convolution call
    PackJob* pack_jobs = nullptr;
    int n_pack_jobs = 0, n_pack_blocks = 0;
    bool wpack_dirty = true;
    // bf16 planes of the conv weights for the split-product GEMMs (split3.h; ResNet, fp32 mode): student forward, teacher
    // forward, student data-gradient packs; rebuilt with the packs (wpack_dirty) / the teacher shadows (twb_dirty)
    unsigned short *wsp_f = nullptr, *twsp_f = nullptr, *wsp_d = nullptr;
    // planes mode (ResNet-18, split product forms): the 3x3 / 1x1 conv GEMMs take BOTH operands as block-major bf16 planes
    // (pconv.hip); the activation planes are written by the kernels that produce the tensors
    bool planes = false;
    unsigned short *wbm_f = nullptr, *twbm_f = nullptr, *wbm_d = nullptr;
    SplitJobBM *bm_f = nullptr, *bm_d = nullptr;
    int n_bm_f = 0, n_bm_f_blocks = 0, n_bm_d = 0, n_bm_d_blocks = 0;
    unsigned short *p0p = nullptr, *t_p0p = nullptr;                 // planes of the stem's pooled output
    unsigned short *GBp = nullptr, *GCp = nullptr, *GDp = nullptr;   // planes of d y2, d y_ds, d y1 (what the data gradients read)
    unsigned short *GB2p = nullptr, *GC2p = nullptr, *GD2p = nullptr;
    unsigned short* xp_scratch = nullptr;       // planes of an fp32 operand nobody produced planes for (test hooks)
    unsigned short* xp_scratch2 = nullptr;      // ... and of the second operand of a weight gradient
    size_t xp_scratch_elems = 0;
    const float* xp_scratch_src = nullptr;
    long long xp_scratch_npix = 0;
    SplitJob *split_f = nullptr, *split_d = nullptr;
    int n_split_f = 0, n_split_f_blocks = 0, n_split_d = 0, n_split_d_blocks = 0;
    // RCCL (comm.hip)
    void* comm = nullptr;
    int comm_rank = 0, comm_world = 0;
    double* comm_buf = nullptr;       // device scratch of the small all-reduces
    size_t comm_buf_n = 0;
    int precision = 0;                // 0 fp32 activations, 1 bf16 activations (EfficientNet-B0 only)
    int products = 6;                 // fm_config.reserved[2]: how the fp32 conv GEMMs form their products (0 fp32 pipe, 6 / 9 bf16 partials)
    int stream_mode = 0;              // fm_config.reserved[1]: 0 side stream for teacher + weight gradients, 1 one stream, 2 teacher only
    int dt = DT_F32;                  // storage type of activations / their gradients (DT_F32 or DT_BF16)
    bf16 *wb = nullptr, *twb = nullptr;   // bf16 weight shadows of the student / the teacher (1x1 convs, W and W^T)
    size_t wb_numel = 0;
    CastJob* cast_jobs = nullptr;
    int n_cast_jobs = 0, n_cast_blocks = 0;
    bool wb_dirty = true, twb_dirty = true;
    bool fuse_gate = false;           // squeeze-excite gate applied on the project conv's operand load (a_s never stored)
    // stage-1 steps of EfficientNet-B0: the frozen teacher's forward is independent of the student's until the loss, so it is
    // enqueued on a side stream with its own activation / workspace set (stream_mode 1: one stream, shared buffers)
    hipStream_t st2 = nullptr;
    hipEvent_t ev_in = nullptr, ev_t = nullptr;
    bool side_ok = false;
    float *t_a0 = nullptr, *t_Tmid = nullptr, *t_se_pool = nullptr, *t_rec = nullptr;
    float *t_c0y = nullptr, *t_p0 = nullptr;                        // ResNet-18: stem output / pooled stem of the teacher
    // EfficientNet backward: the weight gradients of the 1x1 and depthwise convs run on the side stream next to the
    // data-gradient chain; the gradient tensors they read are double-buffered by block parity (stream_mode 1 or 2: inline)
    bool side_w = false;
    float *T_small2 = nullptr, *T_mid2 = nullptr, *T_big2 = nullptr, *ws_slab2 = nullptr;
    float *GB2 = nullptr, *GC2 = nullptr, *GD2 = nullptr;           // ResNet-18 (stream_mode 0): the same for d y2 / d y_ds / d y1
    hipEvent_t ev_p[4][2] = {}, ev_c[4][2] = {}, ev_wdone = nullptr;    // [tensor: d y_p, d y_d, d y_e, SE vectors][block parity]
    float *se_dgp2 = nullptr, *se_drp2 = nullptr;
    float* sk_slab2 = nullptr;                                       // stream-K fix-up workspace of igemm launches on st2
    int* sk_counters2 = nullptr;
    float* stem_col = nullptr;        // bf16 mode: [images][hout][wout][k][4][4] bf16 im2col of the input (the stem's X operand)
};
static inline void soft(fm_engine* e, hipError_t rc)
{
    if (rc != hipSuccess && e->soft_err == hipSuccess) e->soft_err = rc;
}
// end of a step's launch sequence: a failed launch (hipGetLastError) or a failed cross-stream event operation surfaces here
#define STEP_DONE(e)                                                 \
    do {                                                             \
        const hipError_t se_ = (e)->soft_err;                        \
        (e)->soft_err = hipSuccess;                                  \
        HIPCHK(se_);                                                 \
        HIPCHK(hipGetLastError());                                   \
        if ((e)->host_err && *(volatile int*)(e)->host_err) {        \
            /* drain the failed step (its later kernels may report too), then clear both words: it is reported ONCE */ \
            (void)hipStreamSynchronize((e)->st);                     \
            if ((e)->st2) (void)hipStreamSynchronize((e)->st2);      \
            *(volatile int*)(e)->host_err = 0;                       \
            (void)hipMemsetAsync((e)->dev_err, 0, sizeof(int), (e)->st); \
            g_err = "a stream-K part never arrived (pconv): the step's optimizer update was skipped"; \
            return FM_ERR_HIP;                                       \
        }                                                            \
    } while (0)


namespace {

template <typename T>
int dalloc(fm_engine* e, T** p, size_t n)
{
    void* q = nullptr;
    HIPCHK(hipMalloc(&q, std::max<size_t>(n, 1) * sizeof(T)));
    e->allocs.push_back(q);
    *p = reinterpret_cast<T*>(q);
    return FM_OK;
}
#define DALLOC(p, n)                                   \
    do {                                               \
        int rc_ = dalloc(e, &(p), (n));                \
        if (rc_ != FM_OK) return rc_;                  \
    } while (0)

// activation buffer of n elements in the engine's storage type (kept behind float* handles)
int aalloc(fm_engine* e, float** p, size_t n) { return dalloc(e, p, e->precision ? (n + 1) / 2 : n); }
#define AALLOC(p, n)                                   \
    do {                                               \
        int rc_ = aalloc(e, &(p), (n));                \
        if (rc_ != FM_OK) return rc_;                  \
    } while (0)

// The engine's side stream runs at the LOWEST priority: the caller's stream carries the dependent chain (student forward,
// BatchNorm / data-gradient chain) whose short bandwidth-bound kernels should be dispatched first; the side stream's
// teacher forward / weight gradients fill what is left (ResNet-18 stage-1 step 37.7 -> 37.1 ms; highest priority: 38.3)
hipError_t create_side_stream(hipStream_t* st)
{
    int least = 0, greatest = 0;
    const int pr = fm_tune("FM_SIDE_PRIO", 1);         // tuning builds: 1 lowest (shipped), 0 default, -1 highest
    if (!pr || hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) return hipStreamCreateWithFlags(st, hipStreamNonBlocking);
    return hipStreamCreateWithPriority(st, hipStreamNonBlocking, pr > 0 ? least : greatest);
}

int upload_tab(fm_engine* e, const std::vector<int4>& h, int4** d)
{
    DALLOC(*d, h.size());
    HIPCHK(hipMemcpy(*d, h.data(), h.size() * sizeof(int4), hipMemcpyHostToDevice));
    return FM_OK;
}

// ---- model construction -------------------------------------------------------
int add_conv(fm_engine* e, int cin, int cout, int k, int stride, int pad, int hin, int win, size_t& off)
{
    Conv c{};
    c.cin = cin; c.cout = cout; c.k = k; c.stride = stride; c.pad = pad;
    const bool eff = e->model == 1;
    c.cin_p = (cin == 3) ? 4 : (eff ? (cin + 15) / 16 * 16 : cin);
    c.cout_p = eff ? (cout + 15) / 16 * 16 : cout;
    c.kw_p = (cin == 3) ? (k <= 4 ? 4 : 8) : k;
    c.hin = hin; c.win = win;
    if (eff) {                       // TF-"same": out = ceil(in/stride); pad here = top/left padding
        c.hout = (hin + stride - 1) / stride;
        c.wout = (win + stride - 1) / stride;
    } else {
        c.hout = (hin + 2 * pad - k) / stride + 1;
        c.wout = (win + 2 * pad - k) / stride + 1;
    }
    c.Kw = k * c.kw_p * c.cin_p;
    const char* pk = getenv("FM_STEM_PACKED");
    if (cin == 3 && k == 7 && !eff && !(pk && atoi(pk) == 0)) {
        // packed 7x7 stem: a kernel row is the 24 contiguous floats (7 taps x 3 channels + one zero tap) of an NHWC3 row
        // of the zero-framed input; 7 x 24 = 168 -> 176 = 11 K-steps of 16 (the [7][8][4] form, FM_STEM_PACKED=0,
        // needs 14 and wastes 43 % of the weight gradient's tile columns)
        c.stem3 = true;
        c.cin_p = 3; c.kw_p = 8;
        c.Kw = 176;
        c.Hp = hin + 6; c.Wp = win + 8;
    }
    c.nsteps = c.Kw / 16;
    c.w_numel = (size_t)c.cout_p * c.Kw;
    c.w_off = off;
    off += c.w_numel;
    c.macs_per_img = (double)c.hout * c.wout * cout * cin * k * k;
    c.bn = (int)e->convs.size();
    e->convs.push_back(c);
    return (int)e->convs.size() - 1;
}

int build_tables(fm_engine* e)
{
    if (e->planes) {
        // planes mode needs every non-stem conv GEMM (forward and data gradient) inside pconv.hip's limits at max_images
        for (auto& c : e->convs) {
            if (c.cin == 3) continue;
            if (!pconv_takes(c.cout_p, c.cin_p, (long long)e->maxB * c.hin * c.win, c.win) ||
                !pconv_takes(c.cin_p, c.cout_p, (long long)e->maxB * c.hout * c.wout, c.wout) ||
                !pwgrad_takes(c.cout_p, c.cin_p, c.k, (long long)e->maxB * c.hout * c.wout, (long long)e->maxB * c.hin * c.win, c.win,
                              c.pad))
                e->planes = false;
        }
    }
    for (auto& c : e->convs) {
        if (c.Kw % 16 != 0) { g_err = "conv K not a multiple of 16"; return FM_ERR_ARG; }
        // forward / wgrad table: chunk q -> (kh, kw, ci0)
        std::vector<int4> t(c.Kw / 4);
        if (c.stem3) {                      // chunk q = floats 4j.. of kernel row kh's window at framed pixel (2 oh + kh, 2 ow)
            for (int q = 0; q < c.Kw / 4; ++q) t[q] = make_int4(q / 6, 0, 4 * (q % 6), q < 42 ? 1 : 0);
            int rc = upload_tab(e, t, &c.tab);
            if (rc) return rc;
            continue;
        }
        for (int q = 0; q < c.Kw / 4; ++q) {
            const int n = 4 * q;
            const int tap = n / c.cin_p, ci0 = n % c.cin_p;
            const int kh = tap / c.kw_p, kw = tap % c.kw_p;
            t[q] = make_int4(kh - c.pad, kw - c.pad, ci0, kw < c.k ? 1 : 0);
        }
        int rc = upload_tab(e, t, &c.tab);
        if (rc) return rc;
        if (c.cin == 3) continue;          // the stem needs no input gradient
        // data-gradient parity classes
        const int s = c.stride;
        for (int ph = 0; ph < s; ++ph)
            for (int pw = 0; pw < s; ++pw) {
                DgradClass d{};
                d.ph = ph; d.pw = pw; d.taps.n = 0;
                for (int kh = 0; kh < c.k; ++kh)
                    for (int kw = 0; kw < c.k; ++kw) {
                        if ((ph + c.pad - kh) % s != 0 || (pw + c.pad - kw) % s != 0) continue;
                        const int j = d.taps.n++;
                        d.taps.t[j] = kh * c.k + kw;
                        d.dh[j] = (ph + c.pad - kh) / s;     // exact (divisible), may be negative
                        d.dw[j] = (pw + c.pad - kw) / s;
                    }
                if (d.taps.n == 0) continue;
                const int K = d.taps.n * c.cout_p;
                d.nsteps = K / 16;
                DALLOC(d.wpack, (size_t)c.cin_p * K);
                c.cls[c.ncls++] = d;
            }
    }
    // one job per (conv, parity class) for the batched pack kernel
    std::vector<PackJob> jobs;
    int blk = 0;
    for (auto& c : e->convs)
        for (int k = 0; k < c.ncls; ++k) {
            const DgradClass& d = c.cls[k];
            PackJob j{};
            j.w_off = (long long)c.w_off; j.out = d.wpack; j.Co = c.cout_p; j.T = c.k * c.k; j.Ci = c.cin_p;
            j.ntaps = d.taps.n;
            for (int t = 0; t < d.taps.n; ++t) j.taps[t] = d.taps.t[t];
            j.blk0 = blk;
            blk += pack_job_blocks(c.cout_p, c.cin_p, d.taps.n);
            jobs.push_back(j);
        }
    if (e->precision) {        // bf16 shadows (W and W^T) of every 1x1 convolution
        std::vector<CastJob> cj;
        long long off = 0;
        int cb = 0;
        for (auto& c : e->convs) {
            const bool stem = c.cin == 3 && e->model == 1;    // the stem runs as a K = k*kw_p*4 pointwise conv on its im2col
            if (c.k != 1 && !stem) continue;
            const int Kc = stem ? c.Kw : c.cin_p;
            const long long n = (long long)c.cout_p * Kc;
            c.wb_off = off; c.wbt_off = off + n;
            cj.push_back({(long long)c.w_off, c.wb_off, c.wbt_off, c.cout_p, Kc, cb});
            off += 2 * n;
            cb += (int)((n + 255) / 256);
        }
        e->wb_numel = (size_t)off;
        e->n_cast_jobs = (int)cj.size(); e->n_cast_blocks = cb;
        DALLOC(e->cast_jobs, cj.size());
        HIPCHK(hipMemcpy(e->cast_jobs, cj.data(), cj.size() * sizeof(CastJob), hipMemcpyHostToDevice));
        DALLOC(e->wb, e->wb_numel); DALLOC(e->twb, e->wb_numel);
    }
    e->n_pack_jobs = (int)jobs.size();
    e->n_pack_blocks = blk;
    DALLOC(e->pack_jobs, jobs.size());
    if (!jobs.empty())
        HIPCHK(hipMemcpy(e->pack_jobs, jobs.data(), jobs.size() * sizeof(PackJob), hipMemcpyHostToDevice));
    if (e->model == 0 && !e->precision) {
        // weight planes for the GEMMs with whole 64-row tiles and whole 32-k blocks per tap (igemm.hip, WP form)
        std::vector<SplitJob> jf, jd;
        long long off_f = 0, off_d = 0;
        int bf = 0, bd = 0;
        for (auto& c : e->convs) {
            if (c.cin == 3 || e->planes) continue;      // (planes mode: the fp32-operand kernels never run on these convs)
            if (c.cout_p % 64 == 0 && c.cin_p % 32 == 0) {
                c.sp_off = off_f;
                jf.push_back({(long long)c.w_off, nullptr, off_f, c.cout_p, c.Kw / 32, bf});
                bf += split_job_blocks(c.cout_p, c.Kw / 32);
                off_f += (long long)c.cout_p * c.Kw * 3;
            }
            if (c.cin_p % 64 == 0 && c.cout_p % 32 == 0)
                for (int k = 0; k < c.ncls; ++k) {
                    DgradClass& d = c.cls[k];
                    const int K = d.taps.n * c.cout_p;
                    d.sp_off = off_d;
                    jd.push_back({0, d.wpack, off_d, c.cin_p, K / 32, bd});
                    bd += split_job_blocks(c.cin_p, K / 32);
                    off_d += (long long)c.cin_p * K * 3;
                }
        }
        if (e->planes) {
            // block-major planes (pconv.hip): every non-stem conv of ResNet-18 and every parity class of its data gradients
            std::vector<SplitJobBM> bf_, bd_;
            long long o_f = 0, o_d = 0;
            int nbf = 0, nbd = 0;
            for (auto& c : e->convs) {
                if (c.cin == 3) continue;
                if (c.cout_p % 64 == 0 && c.cin_p % 32 == 0) {
                    c.bm_off = o_f;
                    bf_.push_back({(long long)c.w_off, nullptr, o_f, c.cout_p, c.k * c.k, c.cin_p / 32, nbf});
                    nbf += split_job_bm_blocks(c.cout_p, c.Kw / 32);
                    o_f += (long long)c.cout_p * c.Kw * 3;
                }
                if (c.cin_p % 64 == 0 && c.cout_p % 32 == 0)
                    for (int k = 0; k < c.ncls; ++k) {
                        DgradClass& d = c.cls[k];
                        const int K = d.taps.n * c.cout_p;
                        d.bm_off = o_d;
                        bd_.push_back({0, d.wpack, o_d, c.cin_p, d.taps.n, c.cout_p / 32, nbd});
                        nbd += split_job_bm_blocks(c.cin_p, K / 32);
                        o_d += (long long)c.cin_p * K * 3;
                    }
            }
            float* t2 = nullptr;
            if (o_f) {
                DALLOC(t2, (size_t)(o_f + 1) / 2); e->wbm_f = reinterpret_cast<unsigned short*>(t2);
                DALLOC(t2, (size_t)(o_f + 1) / 2); e->twbm_f = reinterpret_cast<unsigned short*>(t2);
            }
            if (o_d) { DALLOC(t2, (size_t)(o_d + 1) / 2); e->wbm_d = reinterpret_cast<unsigned short*>(t2); }
            e->n_bm_f = (int)bf_.size(); e->n_bm_f_blocks = nbf; e->n_bm_d = (int)bd_.size(); e->n_bm_d_blocks = nbd;
            if (!bf_.empty()) {
                DALLOC(e->bm_f, bf_.size());
                HIPCHK(hipMemcpy(e->bm_f, bf_.data(), bf_.size() * sizeof(SplitJobBM), hipMemcpyHostToDevice));
            }
            if (!bd_.empty()) {
                DALLOC(e->bm_d, bd_.size());
                HIPCHK(hipMemcpy(e->bm_d, bd_.data(), bd_.size() * sizeof(SplitJobBM), hipMemcpyHostToDevice));
            }
        }
        float* tmp = nullptr;
        if (off_f) {
            DALLOC(tmp, (size_t)(off_f + 1) / 2); e->wsp_f = reinterpret_cast<unsigned short*>(tmp);
            DALLOC(tmp, (size_t)(off_f + 1) / 2); e->twsp_f = reinterpret_cast<unsigned short*>(tmp);
        }
        if (off_d) { DALLOC(tmp, (size_t)(off_d + 1) / 2); e->wsp_d = reinterpret_cast<unsigned short*>(tmp); }
        e->n_split_f = (int)jf.size(); e->n_split_f_blocks = bf; e->n_split_d = (int)jd.size(); e->n_split_d_blocks = bd;
        if (!jf.empty()) {
            DALLOC(e->split_f, jf.size());
            HIPCHK(hipMemcpy(e->split_f, jf.data(), jf.size() * sizeof(SplitJob), hipMemcpyHostToDevice));
        }
        if (!jd.empty()) {
            DALLOC(e->split_d, jd.size());
            HIPCHK(hipMemcpy(e->split_d, jd.data(), jd.size() * sizeof(SplitJob), hipMemcpyHostToDevice));
        }
    }
    return FM_OK;
}

int build_resnet18(fm_engine* e)
{
    size_t off = 0;
    int h = e->H, w = e->W;
    add_conv(e, 3, 64, 7, 2, 3, h, w, off);
    h /= 2; w /= 2;          // conv1
    h /= 2; w /= 2;          // maxpool
    int cin = 64;
    const int widths[4] = {64, 128, 256, 512};
    for (int li = 0; li < 4; ++li)
        for (int b = 0; b < 2; ++b) {
            const int wd = widths[li];
            const int stride = (li > 0 && b == 0) ? 2 : 1;
            Block blk{};
            blk.c1 = add_conv(e, cin, wd, 3, stride, 1, h, w, off);
            const int ho = e->convs[blk.c1].hout, wo = e->convs[blk.c1].wout;
            blk.c2 = add_conv(e, wd, wd, 3, 1, 1, ho, wo, off);
            blk.ds = (stride != 1 || cin != wd) ? add_conv(e, cin, wd, 1, stride, 0, h, w, off) : -1;
            e->blocks.push_back(blk);
            cin = wd; h = ho; w = wo;
        }
    e->off_fcw = off; off += (size_t)e->C * 512;
    e->off_fcb = off; off += e->C;
    off = (off + 3) & ~(size_t)3;
    int ch = 0;
    for (auto& c : e->convs) {
        Bn b{};
        b.C = c.cout; b.C_real = c.cout; b.ch_off = ch; ch += c.cout;
        e->bns.push_back(b);
    }
    e->n_bn_ch = ch;
    e->off_gamma = off; off += ch;
    e->off_beta = off; off += ch;
    e->NP = off;             // ch is a multiple of 64 -> NP multiple of 4
    e->off_rm = off; off += ch;
    e->off_rv = off; off += ch;
    e->NS = off;
    // state_dict entry table (torchvision key order)
    auto push_conv = [&](int ci) {
        const Conv& c = e->convs[ci];
        e->entries.push_back({0, ci, c.w_off, (size_t)c.cout * c.cin * c.k * c.k, -1, c.cout, c.cin, c.k, c.k, c.kw_p,
                              c.cin_p, c.stem3 ? c.Kw : 0});
    };
    auto push_bn = [&](int bi) {
        const Bn& b = e->bns[bi];
        e->entries.push_back({1, -1, e->off_gamma + b.ch_off, (size_t)b.C, -1});
        e->entries.push_back({1, -1, e->off_beta + b.ch_off, (size_t)b.C, -1});
        e->entries.push_back({1, -1, e->off_rm + b.ch_off, (size_t)b.C, -1});
        e->entries.push_back({1, -1, e->off_rv + b.ch_off, (size_t)b.C, -1});
        e->entries.push_back({2, -1, 0, 1, bi});
    };
    push_conv(0); push_bn(0);
    for (auto& blk : e->blocks) {
        push_conv(blk.c1); push_bn(blk.c1);
        push_conv(blk.c2); push_bn(blk.c2);
        if (blk.ds >= 0) { push_conv(blk.ds); push_bn(blk.ds); }
    }
    e->entries.push_back({1, -1, e->off_fcw, (size_t)e->C * 512, -1});
    e->entries.push_back({1, -1, e->off_fcb, (size_t)e->C, -1});
    e->nf_sd = 0; e->ni_sd = 0;
    for (auto& en : e->entries) (en.kind == 2 ? e->ni_sd : e->nf_sd) += (int64_t)en.n;
    e->counters.assign(e->bns.size(), 0);
    e->tcounters = e->counters;
    return FM_OK;
}

// EfficientNet-B0 (efficientnet-pytorch 0.7.1 'efficientnet-b0', width/depth 1.0): stem 3x3 s2 ->
// 16 MBConv blocks -> head 1x1 -> pool -> dropout -> fc.  State entries follow the package's
// state_dict order (fedmlp_amd/spec.py:efficientnet_b0_entries is the host-side mirror).
int build_effnet_b0(fm_engine* e)
{
    static const int stages[7][6] = {{1, 3, 1, 1, 32, 16},  {2, 3, 2, 6, 16, 24},  {2, 5, 2, 6, 24, 40},
                                     {3, 3, 2, 6, 40, 80},  {3, 5, 1, 6, 80, 112}, {4, 5, 2, 6, 112, 192},
                                     {1, 3, 1, 6, 192, 320}};
    e->bn_eps = 1e-3f; e->bn_mom = 0.01f; e->D = 1280;
    auto r16 = [](int c) { return (c + 15) / 16 * 16; };
    size_t off = 0;
    int ch = 0;
    auto add_bn = [&](int c_real) {
        Bn b{};
        b.C = r16(c_real); b.C_real = c_real; b.ch_off = ch; ch += b.C;
        e->bns.push_back(b);
        return (int)e->bns.size() - 1;
    };
    auto same_pad = [](int size, int k, int s) {
        const int o = (size + s - 1) / s;
        const int p = std::max((o - 1) * s + k - size, 0);
        return p / 2;
    };
    int h = e->H, w = e->W;
    e->c_stem = add_conv(e, 3, 32, 3, 2, same_pad(h, 3, 2), h, w, off);
    e->bn_stem = add_bn(32);
    e->convs[e->c_stem].bn = e->bn_stem;
    h = e->convs[e->c_stem].hout; w = e->convs[e->c_stem].wout;
    for (int st = 0; st < 7; ++st)
        for (int r = 0; r < stages[st][0]; ++r) {
            MBConv m{};
            m.k = stages[st][1];
            m.s = r == 0 ? stages[st][2] : 1;
            m.expand = stages[st][3];
            m.cin = r == 0 ? stages[st][4] : stages[st][5];
            m.cout = stages[st][5];
            m.cin_p = r16(m.cin); m.ce = m.cin * m.expand; m.ce_p = r16(m.ce); m.cout_p = r16(m.cout);
            m.cs = std::max(1, m.cin / 4);
            m.hin = h; m.win = w;
            m.hout = (h + m.s - 1) / m.s; m.wout = (w + m.s - 1) / m.s;
            m.pad_t = same_pad(h, m.k, m.s); m.pad_l = same_pad(w, m.k, m.s);
            m.skip = m.s == 1 && m.cin == m.cout;
            if (m.expand != 1) {
                m.c_exp = add_conv(e, m.cin, m.ce, 1, 1, 0, h, w, off);
                m.bn0 = add_bn(m.ce);
                e->convs[m.c_exp].bn = m.bn0;
            }
            m.dw_off = off; off += (size_t)m.k * m.k * m.ce_p;
            m.bn1 = add_bn(m.ce);
            m.w1_off = off; off += (size_t)m.cs * m.ce_p;
            m.b1_off = off; off += (size_t)(m.cs + 3) / 4 * 4;
            m.w2_off = off; off += (size_t)m.ce_p * m.cs;
            m.b2_off = off; off += m.ce_p;
            off = (off + 3) & ~(size_t)3;
            m.c_proj = add_conv(e, m.ce, m.cout, 1, 1, 0, m.hout, m.wout, off);
            m.bn2 = add_bn(m.cout);
            e->convs[m.c_proj].bn = m.bn2;
            e->mbs.push_back(m);
            h = m.hout; w = m.wout;
        }
    e->c_head = add_conv(e, 320, 1280, 1, 1, 0, h, w, off);
    e->bn_head = add_bn(1280);
    e->convs[e->c_head].bn = e->bn_head;
    e->off_fcw = off; off += (size_t)e->C * 1280;
    e->off_fcb = off; off += e->C;
    off = (off + 3) & ~(size_t)3;
    e->n_bn_ch = ch;
    e->off_gamma = off; off += ch;
    e->off_beta = off; off += ch;
    e->NP = off;
    e->off_rm = off; off += ch;
    e->off_rv = off; off += ch;
    e->NS = off;
    e->maxC = 1280;
    // ---- state_dict entries ----------------------------------------------------------------
    auto push_conv = [&](int ci) {
        const Conv& c = e->convs[ci];
        e->entries.push_back({0, ci, c.w_off, (size_t)c.cout * c.cin * c.k * c.k, -1, c.cout, c.cin, c.k, c.k, c.kw_p,
                              c.cin_p});
    };
    auto push_bn = [&](int bi) {
        const Bn& b = e->bns[bi];
        e->entries.push_back({1, -1, e->off_gamma + b.ch_off, (size_t)b.C_real, -1});
        e->entries.push_back({1, -1, e->off_beta + b.ch_off, (size_t)b.C_real, -1});
        e->entries.push_back({1, -1, e->off_rm + b.ch_off, (size_t)b.C_real, -1});
        e->entries.push_back({1, -1, e->off_rv + b.ch_off, (size_t)b.C_real, -1});
        e->entries.push_back({2, -1, 0, 1, bi});
    };
    push_conv(e->c_stem); push_bn(e->bn_stem);
    for (auto& m : e->mbs) {
        if (m.c_exp >= 0) { push_conv(m.c_exp); push_bn(m.bn0); }
        // depthwise [ce][1][k][k] -> [k][k][ce_p]: an OIHW->OHWI re-layout with O=1, I=ce
        e->entries.push_back({0, -1, m.dw_off, (size_t)m.ce * m.k * m.k, -1, 1, m.ce, m.k, m.k, m.k, m.ce_p});
        push_bn(m.bn1);
        e->entries.push_back({0, -1, m.w1_off, (size_t)m.cs * m.ce, -1, m.cs, m.ce, 1, 1, 1, m.ce_p});
        e->entries.push_back({1, -1, m.b1_off, (size_t)m.cs, -1});
        // _se_expand.weight [ce][cs][1][1] is kept TRANSPOSED, [cs][ce_p] like W1 (every squeeze-excite kernel then reads both
        // matrices with the channel index on the lanes): an "OIHW" view O = 1, I = ce, H = cs, W = 1 -> [1][cs][1][ce_p]
        e->entries.push_back({0, -1, m.w2_off, (size_t)m.ce * m.cs, -1, 1, m.ce, m.cs, 1, 1, m.ce_p});
        e->entries.push_back({1, -1, m.b2_off, (size_t)m.ce, -1});
        push_conv(m.c_proj); push_bn(m.bn2);
    }
    push_conv(e->c_head); push_bn(e->bn_head);
    e->entries.push_back({1, -1, e->off_fcw, (size_t)e->C * 1280, -1});
    e->entries.push_back({1, -1, e->off_fcb, (size_t)e->C, -1});
    e->nf_sd = 0; e->ni_sd = 0;
    for (auto& en : e->entries) (en.kind == 2 ? e->ni_sd : e->nf_sd) += (int64_t)en.n;
    e->counters.assign(e->bns.size(), 0);
    e->tcounters = e->counters;
    return FM_OK;
}

int alloc_workspaces(fm_engine* e)
{
    const size_t B = e->maxB;
    DALLOC(e->state, e->NS); DALLOC(e->tstate, e->NS);
    DALLOC(e->grad, e->NP); DALLOC(e->adam_m, e->NP); DALLOC(e->adam_v, e->NP);
    HIPCHK(hipMemset(e->state, 0, e->NS * 4)); HIPCHK(hipMemset(e->tstate, 0, e->NS * 4));
    HIPCHK(hipMemset(e->grad, 0, e->NP * 4));
    HIPCHK(hipMemset(e->adam_m, 0, e->NP * 4)); HIPCHK(hipMemset(e->adam_v, 0, e->NP * 4));
    DALLOC(e->ev_scale, e->n_bn_ch); DALLOC(e->ev_shift, e->n_bn_ch);
    DALLOC(e->tev_scale, e->n_bn_ch); DALLOC(e->tev_shift, e->n_bn_ch);
    DALLOC(e->stage_sd, (size_t)e->nf_sd);
    DALLOC(e->x4, B * e->H * e->W * 4);
    if (e->convs[0].stem3) {
        const size_t n3 = B * e->convs[0].Hp * e->convs[0].Wp * 3 + 64;      // + slack for the last window's over-read
        DALLOC(e->x3, n3);
        HIPCHK(hipMemset(e->x3, 0, n3 * 4));                                   // the frame stays zero for the engine's life
        const Conv& c0 = e->convs[0];
        e->x3p_plane_elems = (long long)B * c0.Hp * c0.Wp * 4;
        e->stem_rows = e->planes && stem_rows_takes(c0.k, c0.stride, c0.cout_p, c0.hout, c0.wout, e->x3p_plane_elems * 2);
        if (e->stem_rows) {
            DALLOC(e->x3p, (size_t)3 * e->x3p_plane_elems + 64);
            HIPCHK(hipMemset(e->x3p, 0, ((size_t)3 * e->x3p_plane_elems + 64) * 2));
            DALLOC(e->wst, 7 * 3 * 64 * 32); DALLOC(e->twst, 7 * 3 * 64 * 32);
        }
    }
    size_t max_stats = 0, max_slab = 0;
    for (auto& c : e->convs) {
        AALLOC(c.y, B * c.hout * c.wout * c.cout_p);
        if (c.cin == 3 && e->precision) AALLOC(e->stem_col, B * c.hout * c.wout * c.Kw);   // bf16 im2col of the input batch
        size_t tiles = (B * c.hout * c.wout + igemm_tile_n(c.cout_p, c.cin == 3) - 1) / igemm_tile_n(c.cout_p, c.cin == 3) + 2;
        if (e->precision) tiles = std::max<size_t>(tiles, B * c.hout * c.wout / 128 + 4);   // pw_blocks(): >= 128 pixels per block
        max_stats = std::max(max_stats, (tiles + 2 * 32) * 2 * c.cout_p);   // + folded partials
        max_slab = std::max(max_slab, c.w_numel);
    }
    for (auto& b : e->bns) {
        DALLOC(b.mean, 2 * b.C); DALLOC(b.istd, 2 * b.C); DALLOC(b.scale, 2 * b.C); DALLOC(b.shift, 2 * b.C);
    }
    const Conv& c0 = e->convs[0];
    if (e->model == 0) {
        const size_t pooled = B * (c0.hout / 2) * (c0.wout / 2) * 64;
        DALLOC(e->p0, pooled); DALLOC(e->idx0, pooled);
        DALLOC(e->dyh0, B * c0.hout * c0.wout * 64);
        for (auto& blk : e->blocks) {
            const Conv& c = e->convs[blk.c1];
            const size_t n = B * c.hout * c.wout * c.cout;
            DALLOC(blk.z1, n); DALLOC(blk.out, n);
        }
        DALLOC(e->GA, pooled); DALLOC(e->GB, pooled); DALLOC(e->GC, pooled); DALLOC(e->GD, pooled); DALLOC(e->GE, pooled);
    } else {
        size_t g_io = B * c0.hout * c0.wout * c0.cout_p, t_small = 0, t_mid = 0, t_big = 0, max_ce = 0, max_cs = 0;
        AALLOC(e->a0, g_io);
        for (auto& m : e->mbs) {
            const size_t nin = B * m.hin * m.win, nout = B * m.hout * m.wout;
            if (m.c_exp >= 0) AALLOC(m.a_e, nin * m.ce_p);
            AALLOC(m.y_d, nout * m.ce_p); AALLOC(m.a_s, nout * m.ce_p);
            AALLOC(m.out, nout * m.cout_p);
            DALLOC(m.sq, B * m.ce_p); DALLOC(m.rpre, B * m.cs); DALLOC(m.gate, B * m.ce_p);
            g_io = std::max(g_io, std::max(nin * m.cin_p, nout * m.cout_p));
            t_small = std::max(t_small, nout * m.cout_p);
            t_mid = std::max(t_mid, nout * m.ce_p);
            t_big = std::max(t_big, nin * m.ce_p);
            max_ce = std::max<size_t>(max_ce, m.ce_p); max_cs = std::max<size_t>(max_cs, m.cs);
        }
        const Conv& chd = e->convs[e->c_head];
        t_mid = std::max(t_mid, B * chd.hout * chd.wout * chd.cout_p);
        AALLOC(e->GA, g_io); AALLOC(e->GB, g_io);
        AALLOC(e->T_small, t_small); AALLOC(e->T_mid, t_mid); AALLOC(e->T_big, t_big);
        DALLOC(e->se_dgp, B * max_ce); DALLOC(e->se_drp, B * max_cs); DALLOC(e->se_ds, B * max_ce);
        DALLOC(e->se_pool, B * 16 * 5 * max_ce);       // [imgs][<=16 chunks][5 sums][C]
        {
            int side = e->stream_mode != 1;          // fm_config.reserved[1]: 0 two streams, 1 one stream, 2 teacher only
            if (side) {
                // the second buffer sets roughly double the activation footprint: keep one stream when they would not fit
                // next to what is still to be allocated (slabs, statistics: < 2 GB) with 4 GB to spare
                size_t need = 0, free_b = 0, total_b = 0;
                const size_t es = e->precision ? 2 : 4;
                for (auto& m : e->mbs) {
                    const size_t nin = B * m.hin * m.win, nout = B * m.hout * m.wout;
                    need += ((m.c_exp >= 0 ? nin * m.ce_p : 0) + 2 * nout * m.ce_p + nout * m.cout_p) * es;
                }
                need += (t_small + 2 * t_mid + t_big) * es + 2 * ((size_t)48 << 22);
                if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b < need + ((size_t)6 << 30)) side = 0;
            }
            if (side) {
                for (auto& m : e->mbs) {
                    const size_t nin = B * m.hin * m.win, nout = B * m.hout * m.wout;
                    if (m.c_exp >= 0) AALLOC(m.t_a_e, nin * m.ce_p);
                    AALLOC(m.t_y_d, nout * m.ce_p); AALLOC(m.t_a_s, nout * m.ce_p);
                    AALLOC(m.t_out, nout * m.cout_p);
                    DALLOC(m.t_sq, B * m.ce_p); DALLOC(m.t_rpre, B * m.cs); DALLOC(m.t_gate, B * m.ce_p);
                }
                AALLOC(e->t_a0, B * c0.hout * c0.wout * c0.cout_p);
                AALLOC(e->t_Tmid, t_mid);
                DALLOC(e->t_se_pool, B * 16 * max_ce);
                DALLOC(e->t_rec, (size_t)16 << 20);           // pooling records of the eval depthwise forward (<= 33 MB)
                HIPCHK(create_side_stream(&e->st2));
                HIPCHK(hipEventCreateWithFlags(&e->ev_in, hipEventDisableTiming));
                HIPCHK(hipEventCreateWithFlags(&e->ev_t, hipEventDisableTiming));
                e->side_ok = true;
                const int sidew = e->stream_mode == 0;
                if (sidew) {
                    AALLOC(e->T_small2, t_small); AALLOC(e->T_mid2, t_mid); AALLOC(e->T_big2, t_big);
                    DALLOC(e->se_dgp2, B * max_ce); DALLOC(e->se_drp2, B * max_cs);
                    for (int k = 0; k < 4; ++k)
                        for (int q = 0; q < 2; ++q) {
                            HIPCHK(hipEventCreateWithFlags(&e->ev_p[k][q], hipEventDisableTiming));
                            HIPCHK(hipEventCreateWithFlags(&e->ev_c[k][q], hipEventDisableTiming));
                        }
                    HIPCHK(hipEventCreateWithFlags(&e->ev_wdone, hipEventDisableTiming));
                    e->side_w = true;          // ws_slab2 is allocated with ws_slab below
                }
            }
        }
        DALLOC(e->hfeat, B * e->D);
    }
    DALLOC(e->ws_stats, max_stats);
    DALLOC(e->ws_part, (size_t)2 * (1024 + 32) * 2 * e->maxC);   // per-block partials + folded partials
    e->slab_floats = std::max<size_t>(max_slab * 8, (size_t)48 << 20);   // >= 192 MB of partial slabs
    DALLOC(e->ws_slab, e->slab_floats);
    if (e->side_w) DALLOC(e->ws_slab2, e->slab_floats);
    DALLOC(e->ca, 2 * e->maxC); DALLOC(e->cb, 2 * e->maxC); DALLOC(e->cc, 2 * e->maxC);
    DALLOC(e->feat, B * e->D); DALLOC(e->tfeat, B * e->D);
    DALLOC(e->logits, B * e->C); DALLOC(e->tlogits, B * e->C); DALLOC(e->dlogits, B * e->C);
    DALLOC(e->psum, (size_t)2 * e->C * e->D); DALLOC(e->pcnt, 2 * e->C); DALLOC(e->tcnt, e->C);
    DALLOC(e->sel_counts, 2); DALLOC(e->cls_dev, FM_MAX_CLASSES);
    DALLOC(e->zeros, 64);
    if (e->planes) {
        auto palloc = [&](unsigned short** pp, size_t elems) -> int {       // planes of `elems` fp32 values: 3 x 2 B each
            float* t = nullptr;
            DALLOC(t, (elems * 3 + 1) / 2);
            *pp = reinterpret_cast<unsigned short*>(t);
            return FM_OK;
        };
        const Conv& cs = e->convs[0];
        const size_t pooled = B * (cs.hout / 2) * (cs.wout / 2) * 64;
        if (palloc(&e->p0p, pooled) || palloc(&e->GBp, pooled) || palloc(&e->GCp, pooled) || palloc(&e->GDp, pooled)) return FM_ERR_HIP;
        for (auto& blk : e->blocks) {
            const Conv& c = e->convs[blk.c1];
            const size_t n = B * c.hout * c.wout * c.cout;
            if (palloc(&blk.z1p, n) || palloc(&blk.outp, n)) return FM_ERR_HIP;
        }
        // scratch planes for operands that arrive as fp32 (test hooks): the largest conv input / output-gradient tensor
        size_t mx = 0;
        for (auto& c : e->convs) {
            if (c.cin == 3) continue;
            mx = std::max(mx, std::max(B * c.hin * c.win * c.cin_p, B * c.hout * c.wout * c.cout_p));
        }
        e->xp_scratch_elems = mx * 3;
        float* t3 = nullptr;
        DALLOC(t3, (e->xp_scratch_elems + 1) / 2);
        e->xp_scratch = reinterpret_cast<unsigned short*>(t3);
        DALLOC(t3, (e->xp_scratch_elems + 1) / 2);
        e->xp_scratch2 = reinterpret_cast<unsigned short*>(t3);
    }
    const size_t sk_floats = std::max((size_t)igemm_max_blocks() * 2 * 16384, pconv_slab_floats());
    DALLOC(e->sk_slab, sk_floats);   // [blocks][2][BM*BN]
    DALLOC(e->dev_err, 16);
    HIPCHK(hipMemset(e->dev_err, 0, 16 * sizeof(int)));
    HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&e->host_err), 64, hipHostMallocMapped));
    *e->host_err = 0;
    HIPCHK(hipHostGetDevicePointer(reinterpret_cast<void**>(&e->host_err_dev), e->host_err, 0));
    DALLOC(e->sk_counters, (size_t)1 << 20);
    HIPCHK(hipMemset(e->sk_counters, 0, ((size_t)1 << 20) * 4));
    {
        int side = e->stream_mode != 1;
        if (side && e->model == 0) {
            // the teacher's activation set, the second gradient / slab buffers: keep one stream when they would not fit with
            // 4 GB to spare (the EfficientNet branch above guards its own second set the same way); fm_stream_mode() reports
            // the mode that was actually set up
            const Conv& c0 = e->convs[0];
            const size_t pooled = B * (c0.hout / 2) * (c0.wout / 2) * 64;
            size_t need = (B * c0.hout * c0.wout * c0.cout_p + 4 * pooled + e->slab_floats) * 4, free_b = 0, total_b = 0;
            for (auto& blk : e->blocks) {
                const Conv& c = e->convs[blk.c1];
                need += (size_t)(blk.ds >= 0 ? 3 : 2) * B * c.hout * c.wout * c.cout * 4;
            }
            need += sk_floats * 4 + ((size_t)4 << 20);
            if (e->planes) need += need * 3 / 2;       // the planes of the teacher's activations and of the second gradient set
            if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b < need + ((size_t)4 << 30)) side = 0;
        }
        if (side) {
            DALLOC(e->sk_slab2, sk_floats);
            DALLOC(e->sk_counters2, (size_t)1 << 20);
            HIPCHK(hipMemset(e->sk_counters2, 0, ((size_t)1 << 20) * 4));
            // ResNet-18 (MFMA-bound, persistent 512-block kernels): 39.0 -> 37.7 ms per stage-1 step with the teacher and the
            // weight gradients on the side stream, same bits.  Co-running kernels stretch each other's launch windows, so
            // per-kernel durations (bench.py's roofline, rocprofv3) are taken from a one-stream engine (stream mode 1).
            if (e->model == 0) {
                const Conv& c0 = e->convs[0];
                const size_t pooled = B * (c0.hout / 2) * (c0.wout / 2) * 64;
                DALLOC(e->t_c0y, B * c0.hout * c0.wout * c0.cout_p);
                DALLOC(e->t_p0, pooled);
                for (auto& blk : e->blocks) {
                    const Conv& c = e->convs[blk.c1];
                    const size_t n = B * c.hout * c.wout * c.cout;
                    DALLOC(blk.t_z1, n); DALLOC(blk.t_out, n);
                    if (blk.ds >= 0) DALLOC(blk.t_dsy, n);
                }
                HIPCHK(create_side_stream(&e->st2));
                HIPCHK(hipEventCreateWithFlags(&e->ev_in, hipEventDisableTiming));
                HIPCHK(hipEventCreateWithFlags(&e->ev_t, hipEventDisableTiming));
                e->side_ok = true;
                const int sidew = e->stream_mode == 0;
                if (sidew) {
                    DALLOC(e->GB2, pooled); DALLOC(e->GC2, pooled); DALLOC(e->GD2, pooled);
                    DALLOC(e->ws_slab2, e->slab_floats);
                    for (int k = 0; k < 4; ++k)
                        for (int q = 0; q < 2; ++q) {
                            HIPCHK(hipEventCreateWithFlags(&e->ev_p[k][q], hipEventDisableTiming));
                            HIPCHK(hipEventCreateWithFlags(&e->ev_c[k][q], hipEventDisableTiming));
                        }
                    HIPCHK(hipEventCreateWithFlags(&e->ev_wdone, hipEventDisableTiming));
                    e->side_w = true;
                }
            }
        }
    }
    if (e->planes) {       // the side stream's sets of planes (teacher activations, second gradient buffers)
        auto palloc = [&](unsigned short** pp, size_t elems) -> int {
            float* t = nullptr;
            DALLOC(t, (elems * 3 + 1) / 2);
            *pp = reinterpret_cast<unsigned short*>(t);
            return FM_OK;
        };
        const Conv& cs = e->convs[0];
        const size_t pooled = B * (cs.hout / 2) * (cs.wout / 2) * 64;
        if (e->side_ok) {
            if (palloc(&e->t_p0p, pooled)) return FM_ERR_HIP;
            for (auto& blk : e->blocks) {
                const Conv& c = e->convs[blk.c1];
                const size_t n = B * c.hout * c.wout * c.cout;
                if (palloc(&blk.t_z1p, n) || palloc(&blk.t_outp, n)) return FM_ERR_HIP;
            }
        }
        if (e->side_w && (palloc(&e->GB2p, pooled) || palloc(&e->GC2p, pooled) || palloc(&e->GD2p, pooled))) return FM_ERR_HIP;
    }
    HIPCHK(hipMemset(e->zeros, 0, 64 * 4));
    return FM_OK;
}

// ---- profiling helpers -----------------------------------------------------------
hipEvent_t get_ev(fm_engine* e)
{
    if (!e->ev_free.empty()) { hipEvent_t v = e->ev_free.back(); e->ev_free.pop_back(); return v; }
    hipEvent_t v = nullptr;
    if (hipEventCreate(&v) != hipSuccess) { e->prof_fail = true; return nullptr; }
    return v;
}
struct ProfScope {
    fm_engine* e; hipEvent_t a{}, b{}; int fam; double fl; bool on;
    ProfScope(fm_engine* e_, int family, double flops) : e(e_), fam(family), fl(flops), on(e_->prof)
    {
        if (on) {
            a = get_ev(e); b = get_ev(e);
            on = a && b;
            if (on && hipEventRecord(a, e->st) != hipSuccess) { e->prof_fail = true; on = false; }
        }
    }
    ~ProfScope()
    {
        if (on) {
            if (hipEventRecord(b, e->st) != hipSuccess) e->prof_fail = true;
            e->evs.push_back({a, b, fam, fl});
        }
    }
};

struct OpScope {
    fm_engine* e; hipEvent_t a{}, b{}; int id = -1; bool on;
    OpScope(fm_engine* e_, const char* op) : e(e_), on(e_->oprof)
    {
        if (!on) return;
        char lab[96];
        snprintf(lab, sizeof lab, "%s@%d", op, e->ctx);
        for (size_t i = 0; i < e->op_names.size(); ++i)
            if (e->op_names[i] == lab) { id = (int)i; break; }
        if (id < 0) { id = (int)e->op_names.size(); e->op_names.push_back(lab); e->op_ms.push_back(0); e->op_n.push_back(0); }
        a = get_ev(e); b = get_ev(e);
        on = a && b && hipEventRecord(a, e->st) == hipSuccess;
    }
    ~OpScope()
    {
        if (on && hipEventRecord(b, e->st) == hipSuccess) e->opevs.push_back({a, b, id});
    }
};
#define OP(name) OpScope os_(e, name)

// ---- layer launchers --------------------------------------------------------------
// operand prologue of a pointwise conv (bf16 mode): Xe = swish(X*psc+psh)*gate, psc == null: X*gate
struct Prologue { const float* psc; const float* psh; const float* gate; };

const bf16* shadow_of(fm_engine* e, const float* S) { return S == e->tstate ? e->twb : e->wb; }

// planes of an fp32 NHWC operand nobody produced planes for (test hooks): [C/32][3][npix][32] in the scratch buffer
const unsigned short* scratch_planes(fm_engine* e, const float* x, long long npix, int C)
{
    if ((size_t)npix * C * 3 > e->xp_scratch_elems) { soft(e, hipErrorInvalidValue); return nullptr; }
    // FM_DEBUG_REUSE_PLANES=1 (tools/probe_conv.py): time the GEMM alone -- the planes of the same operand are made once
    static const bool reuse = getenv("FM_DEBUG_REUSE_PLANES") && atoi(getenv("FM_DEBUG_REUSE_PLANES")) != 0;
    if (reuse && e->xp_scratch_src == x && e->xp_scratch_npix == npix) return e->xp_scratch;
    k_split_planes(x, e->xp_scratch, npix, C, e->st);
    e->xp_scratch_src = x; e->xp_scratch_npix = npix;
    return e->xp_scratch;
}
bool conv_uses_pconv(const fm_engine* e, const Conv& c, int imgs)
{
    return e->planes && c.bm_off >= 0 && pconv_takes(c.cout_p, c.cin_p, (long long)imgs * c.hin * c.win, c.win);
}

// xp: the input's block-major planes (planes mode; null = made here from x); yp: also / only write the output's planes
// (eval epilogue; y may then be null)
// partial-sum tiles per group the forward GEMM of conv c leaves in `stats` (pro_gate: this call carries an operand prologue)
static int stats_tiles_for(fm_engine* e, const Conv& c, int imgs_per_group, int groups, bool pro_gate)
{
    if (e->stem_rows && c.stem3) return stem_rows_stats_tiles(imgs_per_group, c.hout);
    if (e->precision && (c.k == 1 || c.cin == 3))
        return pw_blocks(imgs_per_group * c.hout * c.wout, groups, c.cout_p, c.cin == 3 ? c.Kw : c.cin_p, pro_gate, c.hout * c.wout);
    if (conv_uses_pconv(e, c, imgs_per_group * groups)) return (imgs_per_group * c.hout * c.wout + pconv_tile_n(c.cout_p) - 1) / pconv_tile_n(c.cout_p);
    const int bn = igemm_tile_n(c.cout_p, c.cin == 3);
    return (imgs_per_group * c.hout * c.wout + bn - 1) / bn;
}

void conv_fwd(fm_engine* e, int ci, const float* S, const float* x, float* y, int imgs, int groups,
              const float* scale, const float* shift, const float* res, int relu, float* stats,
              const Prologue* pro = nullptr, const unsigned short* xp = nullptr, unsigned short* yp = nullptr,
              const unsigned short* resp = nullptr)
{
    Conv& c = e->convs[ci];
    if (stats) {
        e->stats_conv = ci;
        e->stats_tiles_n = stats_tiles_for(e, c, imgs / groups, groups, pro && pro->gate);
    }
    if (e->stem_rows && c.stem3) {           // `x` is ignored: the operand is the framed image's planes (to_nhwc4)
        StemRowsParams q{};
        q.Xp = e->x3p; q.Wst = S == e->tstate ? e->twst : e->wst; q.Y = y;
        q.scale = scale; q.shift = shift; q.stats = stats; q.relu = relu;
        q.imgs = imgs; q.imgs_per_group = imgs / groups; q.Hp = c.Hp; q.Wp = c.Wp; q.Ho = c.hout; q.Wo = c.wout;
        q.sp = e->products; q.plane_bytes = e->x3p_plane_elems * 2;
        ProfScope ps(e, 2, 2.0 * c.macs_per_img * imgs);
        launch_stem_rows(q, e->st);
        return;
    }
    if (conv_uses_pconv(e, c, imgs)) {
        IgemmParams p{};
        p.xp_pix = (long long)imgs * c.hin * c.win;
        p.Xp = xp ? xp : scratch_planes(e, x, p.xp_pix, c.cin_p);
        p.Wsp = (S == e->tstate ? e->twbm_f : e->wbm_f) + c.bm_off;
        p.Y = y; p.Yp = yp; p.yp_pix = (long long)imgs * c.hout * c.wout;
        p.slab = e->sk_slab; p.counters = e->sk_counters; p.err = e->dev_err; p.err_host = e->host_err_dev; p.sp = e->products;
        p.ntaps = c.k * c.k;
        for (int t = 0; t < c.k * c.k; ++t) { p.dh[t] = t / c.k - c.pad; p.dw[t] = t % c.k - c.pad; }
        p.res = res; p.resp = resp; p.scale = scale; p.shift = shift; p.stats = stats;
        p.M = c.cout_p;
        p.Hi = c.hin; p.Wi = c.win; p.Ci = c.cin_p;
        p.Hg = c.hout; p.Wg = c.wout; p.sg = c.stride;
        p.Ho = c.hout; p.Wo = c.wout; p.Co = c.cout_p;
        p.os = 1; p.oh0 = 0; p.ow0 = 0;
        p.imgs_per_group = imgs / groups;
        p.tilesM = c.cout_p / pconv_tile_m(c.cout_p);
        p.tilesN = (p.imgs_per_group * c.hout * c.wout + pconv_tile_n(c.cout_p) - 1) / pconv_tile_n(c.cout_p);
        p.relu = relu;
        // measurement families follow the kernel symbols: 0 / 1 = pconv_kernel<4 | 2, ..., true> (tap rows shared), 6 / 7 = <..., false>
        ProfScope ps(e, (c.cout_p >= 128 ? 0 : 1) + (pconv_uses_ts(p) ? 0 : 6), 2.0 * c.macs_per_img * imgs);
        launch_pconv(p, groups, e->st);
        return;
    }
    // planes mode keeps no fp32 operands for these convs (activations exist only as planes, no row-major weight planes): a shape
    // the planes kernel does not take is a graph error, not a reason to run the fp32-operand kernel on unwritten buffers
    if (e->planes && c.cin != 3) { soft(e, hipErrorInvalidValue); return; }
    const bool stem16 = e->precision && c.cin == 3;     // `x` is ignored: the operand is the im2col matrix
    if (e->precision && (c.k == 1 || stem16)) {          // bf16 storage + bf16 MFMA (pwconv_bf16.hip)
        PwParams q{};
        q.zeros = e->zeros;
        q.W = shadow_of(e, S) + c.wb_off;
        q.X = reinterpret_cast<const bf16*>(stem16 ? e->stem_col : x); q.Y = reinterpret_cast<bf16*>(y);
        q.M = c.cout_p; q.K = stem16 ? c.Kw : c.cin_p;
        q.npix = (imgs / groups) * c.hout * c.wout; q.groups = groups;
        q.scale = scale; q.shift = shift; q.res = reinterpret_cast<const bf16*>(res); q.act = relu;
        q.stats = stats;
        if (pro && pro->gate) { q.psc = pro->psc; q.psh = pro->psh; q.gate = pro->gate; }
        q.HW = c.hout * c.wout;
        launch_pw_conv(q, e->st);
        return;
    }
    IgemmParams p{};
    p.W = S + c.w_off; p.X = x; p.Y = y; p.zeros = e->zeros; p.slab = e->sk_slab; p.counters = e->sk_counters; p.err = e->dev_err; p.err_host = e->host_err_dev;
    p.sp = e->products;
    if (c.sp_off >= 0 && e->wsp_f) p.Wsp = (S == e->tstate ? e->twsp_f : e->wsp_f) + c.sp_off;
    if (c.cin == 3) {
        p.stem_kw = c.k; p.stem_pad = c.pad; p.stem_h2 = c.kw_p == 8 ? 1 : 0; p.stem3 = c.stem3 ? 1 : 0;
        if (c.stem3) p.X = e->x3;               // `x` is ignored: the operand is the framed NHWC3 image
    } else {
        p.ntaps = c.k * c.k;
        for (int t = 0; t < c.k * c.k; ++t) { p.dh[t] = t / c.k - c.pad; p.dw[t] = t % c.k - c.pad; }
    }
    p.res = res; p.scale = scale; p.shift = shift; p.stats = stats;
    if (pro && pro->gate) {       // fp32 project conv: only the streaming kernel (conv1x1.hip) applies the gate (eval) or BN1 + Swish + gate (train)
        p.gate = pro->gate; p.gate_HW = c.hout * c.wout; p.psc = pro->psc; p.psh = pro->psh;
        // the streaming kernel has a gate-only eval form (no statistics) and a BN1 + Swish + gate train form (statistics): a
        // gate-only prologue WITH statistics, or a full prologue without them, has no instantiation
        if (!conv1x1_stream_takes(c.cin_p, c.cout_p, c.cout_p) || c.k != 1 || c.stride != 1 || (pro->psc && !stats) ||
            (!pro->psc && stats))
            soft(e, hipErrorInvalidValue);
    }
    p.M = c.cout_p; p.nsteps = c.nsteps;
    p.Hi = c.hin; p.Wi = c.win; p.Ci = c.cin_p;
    if (c.stem3) { p.Hi = c.Hp; p.Wi = c.Wp; }
    p.Hg = c.hout; p.Wg = c.wout; p.sg = c.stride;
    p.Ho = c.hout; p.Wo = c.wout; p.Co = c.cout_p;
    p.os = 1; p.oh0 = 0; p.ow0 = 0;
    p.imgs_per_group = imgs / groups;
    p.tilesM = (c.cout_p + igemm_tile_m(c.cout_p) - 1) / igemm_tile_m(c.cout_p);
    const int bn = igemm_tile_n(c.cout_p, c.cin == 3);
    p.tilesN = (p.imgs_per_group * c.hout * c.wout + bn - 1) / bn;
    p.relu = relu;           // 0 none, 1 relu, 2 swish
    ProfScope ps(e, c.cin == 3 ? 2 : (c.cout_p >= 128 ? 0 : 1), 2.0 * c.macs_per_img * imgs);
    launch_igemm(p, groups, e->st);
}

// tiles per group of the partial sums conv ci's forward left in ws_stats (set by that conv_fwd call; asking for another conv's
// is a graph error: the buffer holds one conv's partials at a time)
int stats_tiles(fm_engine* e, int ci, int imgs_per_group, int groups)
{
    (void)imgs_per_group; (void)groups;
    if (e->stats_conv != ci) soft(e, hipErrorInvalidValue);
    return e->stats_tiles_n;
}

// dx[imgs][hin][win][cin] = dgrad(dy[imgs][hout][wout][cout]); res: optional residual added
// (for stride-2 convs `acc_cls0` adds the existing dx contents for parity class (0,0))
void conv_dgrad(fm_engine* e, int ci, const float* S, const float* dy, float* dx, int imgs, const float* res,
                bool acc_cls0, const unsigned short* dyp = nullptr)
{
    Conv& c = e->convs[ci];
    if (e->planes && c.ncls > 0 && c.cls[0].bm_off >= 0 &&
        pconv_takes(c.cin_p, c.cout_p, (long long)imgs * c.hout * c.wout, c.wout)) {
        const long long xp_pix = (long long)imgs * c.hout * c.wout;
        if (!dyp) dyp = scratch_planes(e, dy, xp_pix, c.cout_p);
        for (int k = 0; k < c.ncls; ++k) {
            DgradClass& d = c.cls[k];
            IgemmParams p{};
            p.Xp = dyp; p.xp_pix = xp_pix; p.Wsp = e->wbm_d + d.bm_off;
            p.Y = dx; p.slab = e->sk_slab; p.counters = e->sk_counters; p.err = e->dev_err; p.err_host = e->host_err_dev; p.sp = e->products;
            p.ntaps = d.taps.n;
            for (int t = 0; t < d.taps.n; ++t) { p.dh[t] = d.dh[t]; p.dw[t] = d.dw[t]; }
            p.res = res ? res : ((acc_cls0 && d.ph == 0 && d.pw == 0) ? dx : nullptr);
            p.M = c.cin_p;
            p.Hi = c.hout; p.Wi = c.wout; p.Ci = c.cout_p;
            p.Hg = (c.hin - d.ph + c.stride - 1) / c.stride;
            p.Wg = (c.win - d.pw + c.stride - 1) / c.stride;
            p.sg = 1;
            p.Ho = c.hin; p.Wo = c.win; p.Co = c.cin_p;
            p.os = c.stride; p.oh0 = d.ph; p.ow0 = d.pw;
            p.imgs_per_group = imgs;
            p.tilesM = c.cin_p / pconv_tile_m(c.cin_p);
            p.tilesN = (imgs * p.Hg * p.Wg + pconv_tile_n(c.cin_p) - 1) / pconv_tile_n(c.cin_p);
            p.relu = 0;
            ProfScope ps(e, (c.cin_p >= 128 ? 0 : 1) + (pconv_uses_ts(p) ? 0 : 6), 2.0 * c.macs_per_img * imgs * d.taps.n / (double)(c.k * c.k));
            launch_pconv(p, 1, e->st);
        }
        return;
    }
    if (e->planes && c.cin != 3) { soft(e, hipErrorInvalidValue); return; }      // (see conv_fwd: no fp32-operand fallback in planes mode)
    if (e->precision && c.k == 1) {          // dX = dY W: the same streaming kernel with the transposed bf16 shadow
        PwParams q{};
        q.zeros = e->zeros;
        q.W = shadow_of(e, S) + c.wbt_off;
        q.X = reinterpret_cast<const bf16*>(dy); q.Y = reinterpret_cast<bf16*>(dx);
        q.M = c.cin_p; q.K = c.cout_p;
        q.npix = imgs * c.hout * c.wout; q.groups = 1;
        q.res = reinterpret_cast<const bf16*>(res);
        q.HW = c.hout * c.wout;
        launch_pw_conv(q, e->st);
        return;
    }
    for (int k = 0; k < c.ncls; ++k) {
        DgradClass& d = c.cls[k];
        IgemmParams p{};
        p.W = d.wpack; p.X = dy; p.Y = dx; p.zeros = e->zeros; p.slab = e->sk_slab; p.counters = e->sk_counters; p.err = e->dev_err; p.err_host = e->host_err_dev;
        p.sp = e->products;
        if (d.sp_off >= 0 && e->wsp_d) p.Wsp = e->wsp_d + d.sp_off;
        p.ntaps = d.taps.n;
        for (int t = 0; t < d.taps.n; ++t) { p.dh[t] = d.dh[t]; p.dw[t] = d.dw[t]; }
        p.res = res ? res : ((acc_cls0 && d.ph == 0 && d.pw == 0) ? dx : nullptr);
        p.M = c.cin_p; p.nsteps = d.nsteps;
        p.Hi = c.hout; p.Wi = c.wout; p.Ci = c.cout_p;
        p.Hg = (c.hin - d.ph + c.stride - 1) / c.stride;
        p.Wg = (c.win - d.pw + c.stride - 1) / c.stride;
        p.sg = 1;
        p.Ho = c.hin; p.Wo = c.win; p.Co = c.cin_p;
        p.os = c.stride; p.oh0 = d.ph; p.ow0 = d.pw;
        p.imgs_per_group = imgs;
        p.tilesM = (c.cin_p + igemm_tile_m(c.cin_p) - 1) / igemm_tile_m(c.cin_p);
        const int bn = igemm_tile_n(c.cin_p);
        p.tilesN = (imgs * p.Hg * p.Wg + bn - 1) / bn;
        p.relu = 0;
        ProfScope ps(e, c.cin_p >= 128 ? 0 : 1, 2.0 * c.macs_per_img * imgs * d.taps.n / (double)(c.k * c.k));
        launch_igemm(p, 1, e->st);
    }
}

// xp / dyp (planes mode): block-major planes of x / dy (null = made here from the fp32 tensors: test hooks)
void conv_wgrad(fm_engine* e, int ci, const float* x, const float* dy, int imgs, const Prologue* pro = nullptr,
                int pix_per_group = 0, const unsigned short* xp = nullptr, const unsigned short* dyp = nullptr)
{
    const Conv& c = e->convs[ci];
    if (e->planes && c.cin != 3 &&
        pwgrad_takes(c.cout_p, c.cin_p, c.k, (long long)imgs * c.hout * c.wout, (long long)imgs * c.hin * c.win, c.win, c.pad)) {
        PwgradParams q{};
        q.npix = (long long)imgs * c.hout * c.wout; q.xpix = (long long)imgs * c.hin * c.win;
        if (!xp) xp = scratch_planes(e, x, q.xpix, c.cin_p);
        if (!dyp) {
            if ((size_t)q.npix * c.cout_p * 3 > e->xp_scratch_elems) { soft(e, hipErrorInvalidValue); return; }
            k_split_planes(dy, e->xp_scratch2, q.npix, c.cout_p, e->st);
            dyp = e->xp_scratch2;
        }
        q.dYp = dyp; q.Xp = xp; q.slab = e->ws_slab;
        q.M = c.cout_p; q.Nw = c.Kw;
        q.Ho = c.hout; q.Wo = c.wout; q.Hi = c.hin; q.Wi = c.win; q.Ci = c.cin_p; q.stride = c.stride; q.pad = c.pad; q.ksz = c.k;
        q.sp = e->products;
        int sk;
        {
            const bool ring = pwgrad_ring_takes(q);
            ProfScope ps(e, ring ? 8 : (c.cout_p >= 128 ? 3 : 4), 2.0 * c.macs_per_img * imgs);
            sk = ring ? launch_pwgrad_ring(q, e->slab_floats, e->st) : launch_pwgrad(q, e->slab_floats, e->st);
        }
        if (sk > 0) k_reduce_slabs(e->ws_slab, e->grad + c.w_off, sk, (int64_t)c.w_numel, e->st);
        else soft(e, hipErrorInvalidValue);
        return;
    }
    if (e->planes && c.cin != 3) { soft(e, hipErrorInvalidValue); return; }      // (see conv_fwd: no fp32-operand fallback in planes mode)
    const bool stem16 = e->precision && c.cin == 3;
    if (e->precision && (c.k == 1 || stem16)) {
        PwWgradParams q{};
        q.dY = reinterpret_cast<const bf16*>(dy); q.X = reinterpret_cast<const bf16*>(stem16 ? e->stem_col : x);
        q.slab = e->ws_slab; q.M = c.cout_p; q.K = stem16 ? c.Kw : c.cin_p; q.npix = imgs * c.hout * c.wout;
        if (pro && pro->gate) { q.psc = pro->psc; q.psh = pro->psh; q.gate = pro->gate; }
        q.HW = c.hout * c.wout; q.pix_per_group = pix_per_group ? pix_per_group : q.npix;
        const int sk = launch_pw_wgrad(q, e->slab_floats, e->st);
        if (sk > 0) k_reduce_slabs(e->ws_slab, e->grad + c.w_off, sk, (int64_t)c.w_numel, e->st);
        else g_err = "pw_wgrad: shape not handled";
        return;
    }
    WgradParams p{};
    p.dY = dy; p.X = x; p.slab = e->ws_slab; p.tab = c.tab; p.zeros = e->zeros;
    p.sp = e->products;
    p.M = c.cout_p; p.Nw = c.Kw;
    p.Ho = c.hout; p.Wo = c.wout; p.Hi = c.hin; p.Wi = c.win; p.Ci = c.cin_p; p.stride = c.stride;
    if (c.stem3) { p.X = e->x3; p.Hi = c.Hp; p.Wi = c.Wp; }      // gather table of build_tables: framed rows, no bounds
    p.npix = imgs * c.hout * c.wout;
    if (e->model == 1 && (c.k == 1 || c.cin == 3)) {
        if (c.cin == 3) { p.gather_k = c.k; p.gather_pad = c.pad; p.gather_kw_p = c.kw_p; }
        int sk;
        {
            ProfScope ps(e, 4, 2.0 * c.macs_per_img * imgs);
            sk = launch_wgrad_skinny(p, e->slab_floats, e->st);
        }
        if (sk > 0) {
            k_reduce_slabs(e->ws_slab, e->grad + c.w_off, sk, (int64_t)c.w_numel, e->st);
            return;
        }
    }
    const int bm = c.cout_p >= 128 ? 128 : 64, bn = wgrad_tile_n(c.cout_p, c.Kw);
    p.bn = bn;
    p.tilesM = (c.cout_p + bm - 1) / bm;
    p.tilesN = (c.Kw + bn - 1) / bn;
    const int tiles = p.tilesM * p.tilesN;
    // tiles * splits <= 512 = ONE round of the 512 block slots: with the weight gradients on the side stream next to the
    // BN-backward / data-gradient chain, a second round of their blocks holds slots the main stream's next GEMM is waiting for
    // (two-stream step 36.8 -> 36.3 ms; one stream: 38.5 either way; 384: 37.3, 640: 37.4, 1024 = round 2's choice)
    static const int wg_slots = fm_tune("FM_WGRAD_SLOTS", 512);
    int splits = std::max(1, wg_slots / tiles);
    const int max_by_pix = std::max(1, p.npix / 256);
    const int max_by_mem = (int)std::max<size_t>(1, e->slab_floats / c.w_numel);
    splits = std::max(1, std::min(splits, std::min(max_by_pix, max_by_mem)));
    p.pix_per_split = (((p.npix + splits - 1) / splits) + 31) & ~31;
    splits = (p.npix + p.pix_per_split - 1) / p.pix_per_split;
    {
        ProfScope ps(e, c.cout_p >= 128 ? 3 : (c.cin == 3 ? 5 : 4), 2.0 * c.macs_per_img * imgs);
        if (!launch_wgrad(p, splits, e->st)) soft(e, hipErrorInvalidValue);     // surfaces through STEP_DONE as FM_ERR_HIP
    }
    k_reduce_slabs(e->ws_slab, e->grad + c.w_off, splits, (int64_t)c.w_numel, e->st);
    if (c.stem3) k_stem3_mask_grad(e->grad + c.w_off, c.cout_p, e->st);
}

// BN statistics of conv `ci`'s output (partials left in ws_stats by the conv epilogue)
void bn_fwd_finalize(fm_engine* e, int ci, int groups, int imgs_per_group)
{
    const Conv& c = e->convs[ci];
    const int bi = c.bn;
    Bn& b = e->bns[bi];
    k_bn_finalize(e->ws_stats, groups, stats_tiles(e, ci, imgs_per_group, groups), b.C, imgs_per_group * c.hout * c.wout,
                  e->state + e->off_gamma + b.ch_off, e->state + e->off_beta + b.ch_off,
                  e->state + e->off_rm + b.ch_off, e->state + e->off_rv + b.ch_off, b.mean, b.istd, b.scale,
                  b.shift, e->bn_eps, e->bn_mom, e->st, e->dev_err);
    e->counters[bi] += groups;
}

// backward through BN bi: dz (+ optional relu mask source z) -> dy ; optional masked grad out
// z_is_relu_of_bn: z = relu(bn(y)) of THIS BatchNorm with nothing added (bn1 of a basic block): the ReLU mask is then
// recomputed from y, which both passes read anyway, and z is not read (FM_BN_MASK_FROM_Y=0 reads z as before)
// dyp (planes mode): also write dy's block-major planes (what the data gradient reads)
// zh (planes mode): z is kept only as planes: the mask is the sign of its h plane (z is then null)
void bn_bwd(fm_engine* e, int bi, const float* dz, const float* z, float* dy, float* dyh_out, int groups,
            int imgs_per_group, bool z_is_relu_of_bn = false, unsigned short* dyp = nullptr, const unsigned short* zh = nullptr)
{
    const Conv& c = e->convs[bi];
    Bn& b = e->bns[bi];
    const int pix = imgs_per_group * c.hout * c.wout;
    const char* fy = getenv("FM_BN_MASK_FROM_Y");          // read per call: tests compare both forms in one process
    const int from_y = fy ? atoi(fy) : 1;
    const float *msc = nullptr, *msh = nullptr;
    if (z_is_relu_of_bn && (z || zh) && (from_y || !z)) { msc = b.scale; msh = b.shift; z = nullptr; zh = nullptr; }
    k_bn_bwd_reduce(dz, z, c.y, b.mean, b.istd, e->ws_part, groups, pix, b.C, e->st, msc, msh, zh);
    k_bn_bwd_finalize(e->ws_part, groups, bn_bwd_blocks(pix), b.C, pix, e->state + e->off_gamma + b.ch_off, b.mean,
                      b.istd, e->ca, e->cb, e->cc, e->grad + e->off_gamma + b.ch_off,
                      e->grad + e->off_beta + b.ch_off, e->st);
    if (dyp) k_bn_bwd_apply_planes(dz, z, c.y, e->ca, e->cb, e->cc, dy, dyp, dyh_out, groups, pix, b.C, e->st, msc, msh, zh);
    else k_bn_bwd_apply(dz, z, c.y, e->ca, e->cb, e->cc, dy, dyh_out, groups, pix, b.C, e->st, msc, msh);
}

void to_nhwc4(fm_engine* e, const float* const* xs, int groups, int B)
{
    if (e->precision && e->model == 1) {      // bf16 EfficientNet: the stem reads an im2col matrix (teacher, student and wgrad share it)
        const Conv& c = e->convs[e->c_stem];
        for (int g = 0; g < groups; ++g)
            k_stem_im2col(xs[g], reinterpret_cast<bf16*>(e->stem_col) + (size_t)g * B * c.hout * c.wout * c.Kw, B, e->H, e->W, c.hout,
                          c.wout, c.k, c.stride, c.pad, c.pad, e->st);
        return;
    }
    if (e->convs[0].stem3) {
        const Conv& c = e->convs[0];
        for (int g = 0; g < groups; ++g)
            k_frame_nhwc3(xs[g], e->x3 + (size_t)g * B * c.Hp * c.Wp * 3, B, e->H, e->W, c.Hp, c.Wp, 3, 3, 0, e->st,
                          e->stem_rows ? e->x3p + (size_t)g * B * c.Hp * c.Wp * 4 : nullptr, e->x3p_plane_elems);
        return;
    }
    for (int g = 0; g < groups; ++g)
        k_nchw_to_nhwc4(xs[g], e->x4 + (size_t)g * B * e->H * e->W * 4, B, e->H, e->W, e->st);
}

// train-mode forward of groups*B images already in e->x4; fills feat/logits
void forward_train(fm_engine* e, int groups, int B)
{
    const int imgs = groups * B;
    const float* S = e->state;
    Conv& c0 = e->convs[0];
    conv_fwd(e, 0, S, e->x4, c0.y, imgs, groups, nullptr, nullptr, nullptr, 0, e->ws_stats);
    bn_fwd_finalize(e, 0, groups, B);
    const bool pm = e->planes;
    if (pm) k_stem_pool_planes(c0.y, e->bns[0].scale, e->bns[0].shift, e->p0, e->idx0, e->p0p, groups, B, c0.hout, c0.wout, 64, e->st);
    else k_stem_pool(c0.y, e->bns[0].scale, e->bns[0].shift, e->p0, e->idx0, groups, B, c0.hout, c0.wout, 64, e->st);
    // BatchNorm apply.  Planes mode: the activations z1 / out exist ONLY as planes (what the conv GEMMs read; the residual add
    // and the ReLU masks of the backward re-form the fp32 value / its sign from them) -- except the last block's `out`, which
    // feeds the average pool: fp32 only
    auto apply = [&](const float* y, int b1, const float* res, const unsigned short* resp, const float* y2, int b2, float* out,
                     unsigned short* outp, int pix, int C) {
        const float *s2 = y2 ? e->bns[b2].scale : nullptr, *h2 = y2 ? e->bns[b2].shift : nullptr;
        if (pm && outp)
            k_bn_apply_planes(y, e->bns[b1].scale, e->bns[b1].shift, res, y2, s2, h2, nullptr, outp, groups, pix, C, 1, e->st, resp);
        else
            k_bn_apply(y, e->bns[b1].scale, e->bns[b1].shift, res, y2, s2, h2, out, groups, pix, C, 1, e->st);
    };
    const float* cur = e->p0;
    const unsigned short* curp = e->p0p;
    bool cur_f32 = true;                 // is `cur` valid as fp32?  (p0 is written both ways)
    for (size_t bi = 0; bi < e->blocks.size(); ++bi) {
        Block& blk = e->blocks[bi];
        Conv& c1 = e->convs[blk.c1];
        Conv& c2 = e->convs[blk.c2];
        const int pix = B * c1.hout * c1.wout;
        const bool last = bi + 1 == e->blocks.size();
        conv_fwd(e, blk.c1, S, cur, c1.y, imgs, groups, nullptr, nullptr, nullptr, 0, e->ws_stats, nullptr, curp);
        bn_fwd_finalize(e, blk.c1, groups, B);
        apply(c1.y, blk.c1, nullptr, nullptr, nullptr, -1, blk.z1, blk.z1p, pix, c1.cout);
        conv_fwd(e, blk.c2, S, blk.z1, c2.y, imgs, groups, nullptr, nullptr, nullptr, 0, e->ws_stats, nullptr, blk.z1p);
        bn_fwd_finalize(e, blk.c2, groups, B);
        unsigned short* outp = (pm && !last) ? blk.outp : nullptr;
        if (blk.ds >= 0) {
            Conv& cd = e->convs[blk.ds];
            conv_fwd(e, blk.ds, S, cur, cd.y, imgs, groups, nullptr, nullptr, nullptr, 0, e->ws_stats, nullptr, curp);
            bn_fwd_finalize(e, blk.ds, groups, B);
            apply(c2.y, blk.c2, nullptr, nullptr, cd.y, blk.ds, blk.out, outp, pix, c2.cout);
        } else if (pm && !cur_f32) {
            if (last) {      // the last block adds a planes-only residual into an fp32 output: re-form the residual first
                k_planes_to_f32(curp, e->blocks[bi - 1].out, (long long)imgs * c1.hin * c1.win, c1.cin, e->st);
                apply(c2.y, blk.c2, e->blocks[bi - 1].out, nullptr, nullptr, -1, blk.out, nullptr, pix, c2.cout);
            } else
                apply(c2.y, blk.c2, nullptr, curp, nullptr, -1, blk.out, outp, pix, c2.cout);
        } else {
            apply(c2.y, blk.c2, cur, nullptr, nullptr, -1, blk.out, outp, pix, c2.cout);
        }
        cur = blk.out;
        curp = blk.outp;
        cur_f32 = !(pm && !last);
    }
    const Conv& cl = e->convs[e->blocks.back().c2];
    k_avgpool(cur, DT_F32, e->feat, imgs, cl.hout * cl.wout, 512, e->st);
    k_fc_fwd(e->feat, S + e->off_fcw, S + e->off_fcb, e->logits, imgs, 512, e->C, e->st);
    e->ev_dirty = true;      // running stats moved
}

// eval-mode forward (BN folded into the conv epilogue) of `imgs` images in e->x4
void forward_eval(fm_engine* e, const float* S, float* evs, float* evh, bool& dirty, int imgs, float* feat,
                  float* logits)
{
    if (dirty) {
        k_bn_eval_affine(S + e->off_gamma, S + e->off_beta, S + e->off_rm, S + e->off_rv, evs, evh, e->n_bn_ch,
                         e->bn_eps, e->st);
        dirty = false;
    }
    auto sc = [&](int bi) { return evs + e->bns[bi].ch_off; };
    auto sh = [&](int bi) { return evh + e->bns[bi].ch_off; };
    Conv& c0 = e->convs[0];
    conv_fwd(e, 0, S, e->x4, c0.y, imgs, 1, sc(0), sh(0), nullptr, 1, nullptr);
    const bool pm = e->planes;
    if (pm) k_stem_pool_planes(c0.y, nullptr, nullptr, e->p0, nullptr, e->p0p, 1, imgs, c0.hout, c0.wout, 64, e->st);
    else k_stem_pool(c0.y, nullptr, nullptr, e->p0, nullptr, 1, imgs, c0.hout, c0.wout, 64, e->st);
    const float* cur = e->p0;
    const unsigned short* curp = e->p0p;
    bool cur_f32 = true;
    for (size_t bi = 0; bi < e->blocks.size(); ++bi) {
        Block& blk = e->blocks[bi];
        // planes mode: z1 and `out` exist only as planes (the next convs read them; the next block's residual add re-forms the fp32
        // values in the epilogue) -- except the last block's `out`, which feeds the average pool: fp32 only
        const bool last = bi + 1 == e->blocks.size();
        conv_fwd(e, blk.c1, S, cur, pm ? nullptr : blk.z1, imgs, 1, sc(blk.c1), sh(blk.c1), nullptr, 1, nullptr, nullptr, curp,
                 pm ? blk.z1p : nullptr);
        const float* idt = cur;
        const unsigned short* idtp = nullptr;
        if (blk.ds >= 0) {
            Conv& cd = e->convs[blk.ds];
            conv_fwd(e, blk.ds, S, cur, cd.y, imgs, 1, sc(blk.ds), sh(blk.ds), nullptr, 0, nullptr, nullptr, curp);
            idt = cd.y;
        } else if (!cur_f32) { idt = nullptr; idtp = curp; }
        const bool planes_only = pm && !last;
        conv_fwd(e, blk.c2, S, blk.z1, planes_only ? nullptr : blk.out, imgs, 1, sc(blk.c2), sh(blk.c2), idt, 1, nullptr, nullptr,
                 blk.z1p, planes_only ? blk.outp : nullptr, idtp);
        cur = blk.out;
        curp = blk.outp;
        cur_f32 = !planes_only;
    }
    const Conv& cl = e->convs[e->blocks.back().c2];
    k_avgpool(cur, DT_F32, feat, imgs, cl.hout * cl.wout, 512, e->st);
    k_fc_fwd(feat, S + e->off_fcw, S + e->off_fcb, logits, imgs, 512, e->C, e->st);
}

// transposed weight packs of every data-gradient GEMM, all in ONE launch
void ensure_packed(fm_engine* e)
{
    if (!e->wpack_dirty) return;
    if (e->precision) launch_cast_weights(e->state, e->wb, e->cast_jobs, e->n_cast_jobs, e->n_cast_blocks, e->st);
    else if (e->n_pack_jobs) k_pack_dgrad_all(e->state, e->pack_jobs, e->n_pack_jobs, e->n_pack_blocks, e->st);
    if (e->planes) {
        k_split_weights_bm(e->state, e->wbm_f, e->bm_f, e->n_bm_f, e->n_bm_f_blocks, e->st);
        k_split_weights_bm(nullptr, e->wbm_d, e->bm_d, e->n_bm_d, e->n_bm_d_blocks, e->st);      // planes of the packs just made
        if (e->stem_rows) k_stem_weight_planes(e->state + e->convs[0].w_off, e->wst, e->convs[0].Kw, e->st);
    } else {
        k_split_weights(e->state, e->wsp_f, e->split_f, e->n_split_f, e->n_split_f_blocks, e->st);
        k_split_weights(nullptr, e->wsp_d, e->split_d, e->n_split_d, e->n_split_d_blocks, e->st);     // planes of the packs just made
    }
    e->wpack_dirty = false;
}
void ensure_teacher_shadow(fm_engine* e)
{
    if (!e->twb_dirty) return;
    if (e->precision) launch_cast_weights(e->tstate, e->twb, e->cast_jobs, e->n_cast_jobs, e->n_cast_blocks, e->st);
    else if (e->planes) {
        k_split_weights_bm(e->tstate, e->twbm_f, e->bm_f, e->n_bm_f, e->n_bm_f_blocks, e->st);
        if (e->stem_rows) k_stem_weight_planes(e->tstate + e->convs[0].w_off, e->twst, e->convs[0].Kw, e->st);
    }
    else k_split_weights(e->tstate, e->twsp_f, e->split_f, e->n_split_f, e->n_split_f_blocks, e->st);
    e->twb_dirty = false;
}

// optimizer.step(): one fused kernel over the whole trainable arena (torch Adam with coupled L2)
void adam_step(fm_engine* e)
{
    e->adam_t += 1;
    const double bc1 = 1.0 - pow((double)e->hp.beta1, (double)e->adam_t);
    const double bc2 = 1.0 - pow((double)e->hp.beta2, (double)e->adam_t);
    k_adam(e->state, e->grad, e->adam_m, e->adam_v, (int64_t)e->NP, e->hp.lr, e->hp.beta1, e->hp.beta2, e->hp.eps,
           e->hp.weight_decay, (float)bc1, (float)sqrt(bc2), e->st, e->dev_err);
    e->ev_dirty = true;
    e->wpack_dirty = true;
    ensure_packed(e);        // the next step's data gradients read the packed (transposed) weights
}

// backward from e->dlogits through the graph saved by forward_train, then Adam
void backward_and_step(fm_engine* e, int groups, int B)
{
    const int imgs = groups * B;
    const float* S = e->state;
    const Conv& cl = e->convs[e->blocks.back().c2];
    // side_w (stream_mode 0): the weight gradients run on the side stream next to the BN-backward / data-gradient chain
    // (d y2, d y_ds, d y1 double-buffered by block parity, own slab workspace) -- same scheme as eff_backward_and_step
    const bool sw = e->side_w;
    float* GBp[2] = {e->GB, sw ? e->GB2 : e->GB};
    float* GCp[2] = {e->GC, sw ? e->GC2 : e->GC};
    float* GDp[2] = {e->GD, sw ? e->GD2 : e->GD};
    const bool pm = e->planes;
    unsigned short* PB[2] = {e->GBp, sw ? e->GB2p : e->GBp};
    unsigned short* PC[2] = {e->GCp, sw ? e->GC2p : e->GCp};
    unsigned short* PD[2] = {e->GDp, sw ? e->GD2p : e->GDp};
    hipStream_t main_st = e->st;
    auto side_begin = [&](int k, int par) {
        if (!sw) return;
        soft(e, hipEventRecord(e->ev_p[k][par], main_st));
        soft(e, hipStreamWaitEvent(e->st2, e->ev_p[k][par], 0));
        e->st = e->st2;
        std::swap(e->ws_slab, e->ws_slab2);
    };
    auto side_end = [&](int k, int par) {
        if (!sw) return;
        std::swap(e->ws_slab, e->ws_slab2);
        soft(e, hipEventRecord(e->ev_c[k][par], e->st2));
        e->st = main_st;
    };
    auto guard = [&](int k, int par) { if (sw) soft(e, hipStreamWaitEvent(main_st, e->ev_c[k][par], 0)); };
    k_fc_bwd(e->dlogits, e->feat, S + e->off_fcw, nullptr, e->grad + e->off_fcw, e->grad + e->off_fcb, e->GA, DT_F32, imgs,
             512, e->C, cl.hout * cl.wout, e->st);
    float *ga = e->GA, *ge = e->GE;
    for (int b = (int)e->blocks.size() - 1; b >= 0; --b) {
        Block& blk = e->blocks[b];
        const int par = b & 1;
        float *GB = GBp[par], *GC = GCp[par], *GD = GDp[par];
        unsigned short *gbp = pm ? PB[par] : nullptr, *gcp = pm ? PC[par] : nullptr, *gdp = pm ? PD[par] : nullptr;
        const float* in = b == 0 ? e->p0 : e->blocks[b - 1].out;
        // out = relu(bn2(y2) + identity): masked grad dyh goes to bn2 and to the identity path
        // planes mode: d y2 / d y_ds / d y1 exist only as planes (read by the data gradients and the weight gradients); the ReLU
        // mask of `out` is the sign of its h plane (the last block's `out` is fp32: it feeds the average pool)
        const bool last = b + 1 == (int)e->blocks.size();
        guard(0, par);
        if (pm && !last) bn_bwd(e, blk.c2, ga, nullptr, nullptr, ga, groups, B, false, gbp, blk.outp);
        else bn_bwd(e, blk.c2, ga, blk.out, pm ? nullptr : GB, ga, groups, B, false, gbp);
        if (blk.ds >= 0) { guard(1, par); bn_bwd(e, blk.ds, ga, nullptr, pm ? nullptr : GC, nullptr, groups, B, false, gcp); }
        // Where a weight gradient enters the side stream.  lag = 0: as soon as its dy exists, i.e. beside the data gradient of the
        // SAME conv -- two GEMMs share the CUs, and the BatchNorm-backward passes that follow run with nothing beside them.
        // lag = 1 (round 6): behind that data gradient, i.e. beside the NEXT BatchNorm-backward passes of the main stream: a
        // streaming kernel lives beside a GEMM's waves on a CU (the GEMMs leave it the registers since their K-steps run row-major),
        // two GEMMs do not (LDS).  Same kernels, same bits; the buffer guards are the same events.
        static const int lag = fm_tune("FM_WGRAD_LAG", 1);
        const unsigned short* inp = pm ? (b == 0 ? e->p0p : e->blocks[b - 1].outp) : nullptr;
        auto wgrad_c2 = [&]() {
            side_begin(0, par);
            conv_wgrad(e, blk.c2, blk.z1, GB, imgs, nullptr, 0, pm ? blk.z1p : nullptr, gbp);
            side_end(0, par);
        };
        auto wgrad_c1 = [&]() {
            side_begin(2, par);
            conv_wgrad(e, blk.c1, in, GD, imgs, nullptr, 0, inp, gdp);
            side_end(2, par);
        };
        auto wgrad_ds = [&]() {
            side_begin(1, par);
            conv_wgrad(e, blk.ds, in, GC, imgs, nullptr, 0, inp, gcp);
            side_end(1, par);
        };
        if (!lag) wgrad_c2();
        guard(2, par);
        conv_dgrad(e, blk.c2, S, GB, GD, imgs, nullptr, false, gbp);
        if (lag) wgrad_c2();
        if (pm) bn_bwd(e, blk.c1, GD, nullptr, nullptr, nullptr, groups, B, true, gdp, blk.z1p);      // mask from y1 (z1 = relu(bn1(y1)))
        else bn_bwd(e, blk.c1, GD, blk.z1, GD, nullptr, groups, B, true);
        if (!lag) { wgrad_c1(); if (blk.ds >= 0) wgrad_ds(); }
        if (blk.ds >= 0) {
            conv_dgrad(e, blk.ds, S, GC, ge, imgs, nullptr, false, gcp);   // writes parity class (0,0)
            conv_dgrad(e, blk.c1, S, GD, ge, imgs, nullptr, true, gdp);    // all classes, (0,0) accumulates
        } else {
            conv_dgrad(e, blk.c1, S, GD, ge, imgs, ga, false, gdp);
        }
        if (lag) { wgrad_c1(); if (blk.ds >= 0) wgrad_ds(); }
        std::swap(ga, ge);
    }
    const Conv& c0 = e->convs[0];
    {
        // max-pool backward + BatchNorm backward of the stem in two passes, without the dense intermediate (elementwise.hip)
        Bn& b0 = e->bns[0];
        const int pooled_pg = B * (c0.hout / 2) * (c0.wout / 2), pix = B * c0.hout * c0.wout;
        static const int from_pooled = fm_tune("FM_STEM_XHAT_FROM_POOLED", 1);
        k_stem_pool_bn_reduce(ga, e->p0, e->idx0, c0.y, b0.mean, b0.istd, e->ws_part, groups, B, c0.hout, c0.wout, 64, e->st,
                              from_pooled ? e->state + e->off_gamma + b0.ch_off : nullptr,
                              from_pooled ? e->state + e->off_beta + b0.ch_off : nullptr);
        k_bn_bwd_finalize(e->ws_part, groups, stem_pool_bn_blocks(pooled_pg), 64, pix, e->state + e->off_gamma + b0.ch_off,
                          b0.mean, b0.istd, e->ca, e->cb, e->cc, e->grad + e->off_gamma + b0.ch_off,
                          e->grad + e->off_beta + b0.ch_off, e->st);
        k_stem_pool_bn_apply(ga, e->p0, e->idx0, c0.y, e->ca, e->cb, e->cc, e->dyh0, groups, B, c0.hout, c0.wout, 64, e->st);
    }
    conv_wgrad(e, 0, e->x4, e->dyh0, imgs);
    if (sw) {
        soft(e, hipEventRecord(e->ev_wdone, e->st2));
        soft(e, hipStreamWaitEvent(main_st, e->ev_wdone, 0));
    }
    adam_step(e);      // optimizer.step()
}

// =============================== EfficientNet-B0 graph =================================
// Squeeze-excite gate on the project conv's operand load: only where the conv reads its input once or twice
// (<= 2 M-tiles).  The late blocks (K = 672, 1152: 32-row M-tiles, 6-10 of them) would redo the BN + Swish
// prologue per M-tile on tensors that are tiny anyway: they keep the materialised a_s.
// fp32 storage: where the project conv's backward is fused (pw_proj_bwd_f32_kernel needs no a_s) and the train forward streams
// through conv1x1.hip, which then applies BN1 + Swish + gate on its operand load.
bool fuse_for(fm_engine* e, const MBConv& m)
{
    if (e->precision) return e->fuse_gate && pw_tiles_m(m.cout_p, m.ce_p) <= 2;
    static const int on = fm_tune("FM_F32_TRAIN_GATE", 1);
    return on && pw_proj_bwd_f32_nch(m.ce_p, m.cout_p, 1, m.hout * m.wout) > 0 && conv1x1_stream_takes(m.ce_p, m.cout_p, m.cout_p);
}

// BN over an arbitrary NHWC tensor (depthwise output): statistics by chan_reduce, then the same finalize
// sums_ready: ws_part already holds dw_stats_tiles() partials per group (left by the depthwise forward)
void bn_fwd_tensor(fm_engine* e, int bi, const float* y, int groups, int pix_per_group, int HW, bool sums_ready = false)
{
    Bn& b = e->bns[bi];
    if (!sums_ready)
        k_chan_reduce(nullptr, e->dt, y, e->dt, nullptr, nullptr, nullptr, nullptr, nullptr, e->ws_part, groups, pix_per_group,
                      HW, b.C, 0, 0, nullptr, nullptr, e->st);
    k_bn_finalize(e->ws_part, groups, sums_ready ? dw_stats_tiles() : bn_bwd_blocks(pix_per_group), b.C, pix_per_group,
                  e->state + e->off_gamma + b.ch_off, e->state + e->off_beta + b.ch_off,
                  e->state + e->off_rm + b.ch_off, e->state + e->off_rv + b.ch_off, b.mean, b.istd, b.scale, b.shift,
                  e->bn_eps, e->bn_mom, e->st);
    e->counters[bi] += groups;
}

// backward through act(bn(y))*rowscale: dz -> dy (may alias dz); writes dgamma/dbeta
void bnact_bwd(fm_engine* e, int bi, const float* dz, const float* y, float* dy, const float* rowscale, int groups,
               int pix_per_group, int HW, int act, const float* gate = nullptr, const float* dsv = nullptr,
               int ty = -1, int sums_ready = 0)
{
    Bn& b = e->bns[bi];
    if (ty < 0) ty = e->dt;                  // storage type of y / dy (the stem's are fp32 in every mode)
    // sums_ready > 0: ws_part already holds the two backward sums as that many partials per group (k_se_bwd_bn1)
    if (!sums_ready)
        k_chan_reduce(dz, e->dt, y, ty, b.mean, b.istd, b.scale, b.shift, rowscale, e->ws_part, groups, pix_per_group, HW, b.C,
                      1, act, gate, dsv, e->st);
    k_bn_bwd_finalize(e->ws_part, groups, sums_ready ? sums_ready : bn_bwd_blocks(pix_per_group), b.C, pix_per_group,
                      e->state + e->off_gamma + b.ch_off, b.mean, b.istd, e->ca, e->cb, e->cc,
                      e->grad + e->off_gamma + b.ch_off, e->grad + e->off_beta + b.ch_off, e->st);
    k_bnact_bwd_apply(dz, e->dt, y, ty, e->ca, e->cb, e->cc, b.scale, b.shift, rowscale, dy, groups, pix_per_group, HW,
                      b.C, act, gate, dsv, e->st);
}

void eff_forward_train(fm_engine* e, int groups, int B)
{
    const int imgs = groups * B;
    const float* S = e->state;
    Conv& cs = e->convs[e->c_stem];
    e->ctx = -1;
    { OP("conv_fwd"); conv_fwd(e, e->c_stem, S, e->x4, cs.y, imgs, groups, nullptr, nullptr, nullptr, 0, e->ws_stats); }
    { OP("bn_fwd_finalize"); bn_fwd_finalize(e, e->c_stem, groups, B); }
    {
        Bn& b = e->bns[e->bn_stem];
        { OP("k_bnact_apply"); k_bnact_apply(cs.y, e->dt, b.scale, b.shift, nullptr, nullptr, e->a0, e->dt, groups, B * cs.hout * cs.wout,
                      cs.hout * cs.wout, b.C, 2, e->st); }
    }
    const float* cur = e->a0;
    for (size_t i = 0; i < e->mbs.size(); ++i) {
        MBConv& m = e->mbs[i];
        e->ctx = (int)i;
        const int HWi = m.hin * m.win, HWo = m.hout * m.wout;
        const float* a_e = cur;
        if (m.c_exp >= 0) {
            Conv& ce = e->convs[m.c_exp];
            { OP("exp_fwd"); conv_fwd(e, m.c_exp, S, cur, ce.y, imgs, groups, nullptr, nullptr, nullptr, 0, e->ws_stats); }
            { OP("bn_fwd_finalize"); bn_fwd_finalize(e, m.c_exp, groups, B); }
            Bn& b = e->bns[m.bn0];
            { OP("k_bnact_apply"); k_bnact_apply(ce.y, e->dt, b.scale, b.shift, nullptr, nullptr, m.a_e, e->dt, groups, B * HWi, HWi, b.C, 2, e->st); }
            a_e = m.a_e;
        }
        bool st_done;
        { OP("k_dw_fwd"); st_done = k_dw_fwd(a_e, S + m.dw_off, m.y_d, e->dt, nullptr, nullptr, imgs, m.hin, m.win, m.hout, m.wout, m.ce_p,
                 m.k, m.s, m.pad_t, m.pad_l, 0, e->st, e->ws_slab, e->ws_part, groups); }   // + BN1 batch statistics
        { OP("bn_fwd_tensor"); bn_fwd_tensor(e, m.bn1, m.y_d, groups, B * HWo, HWo, st_done); }
        {
            // a_d = swish(bn1(y_d)) is never written: the pooling and the gating pass form it on load
            Bn& b = e->bns[m.bn1];
            { OP("k_se_fwd"); k_se_fwd(m.y_d, e->dt, b.scale, b.shift, B, e->se_pool, S + m.w1_off, S + m.b1_off, S + m.w2_off, S + m.b2_off,
                     m.sq, m.rpre, m.gate, imgs, HWo, m.ce_p, m.cs, e->st); }
            if (!fuse_for(e, m)) k_se_scale(m.y_d, e->dt, b.scale, b.shift, B, m.gate, m.a_s, imgs, HWo, m.ce_p, e->st);
        }
        Conv& cp = e->convs[m.c_proj];
        if (fuse_for(e, m)) {          // a_s = swish(bn1(y_d)) * gate is formed on the project conv's operand load
            const Prologue pro{e->bns[m.bn1].scale, e->bns[m.bn1].shift, m.gate};
            { OP("proj_fwd"); conv_fwd(e, m.c_proj, S, m.y_d, cp.y, imgs, groups, nullptr, nullptr, nullptr, 0, e->ws_stats, &pro); }
        } else
            { OP("proj_fwd"); conv_fwd(e, m.c_proj, S, m.a_s, cp.y, imgs, groups, nullptr, nullptr, nullptr, 0, e->ws_stats); }
        { OP("bn_fwd_finalize"); bn_fwd_finalize(e, m.c_proj, groups, B); }
        {
            Bn& b = e->bns[m.bn2];
            const float* dc = (m.skip && e->dc_dev) ? e->dc_dev + i * (size_t)imgs : nullptr;
            { OP("k_bnact_apply"); k_bnact_apply(cp.y, e->dt, b.scale, b.shift, m.skip ? cur : nullptr, dc, m.out, e->dt, groups, B * HWo, HWo,
                          b.C, 0, e->st); }
        }
        cur = m.out;
    }
    Conv& ch = e->convs[e->c_head];
    const int HWh = ch.hout * ch.wout;
    e->ctx = 100;
    { OP("conv_fwd"); conv_fwd(e, e->c_head, S, cur, ch.y, imgs, groups, nullptr, nullptr, nullptr, 0, e->ws_stats); }
    { OP("bn_fwd_finalize"); bn_fwd_finalize(e, e->c_head, groups, B); }
    {
        Bn& b = e->bns[e->bn_head];
        { OP("k_bnact_apply"); k_bnact_apply(ch.y, e->dt, b.scale, b.shift, nullptr, nullptr, e->T_mid, e->dt, groups, B * HWh, HWh, b.C, 2, e->st); }
    }
    { OP("k_avgpool"); k_avgpool(e->T_mid, e->dt, e->feat, imgs, HWh, e->D, e->st); }
    const float* h = e->feat;
    if (e->drop_dev) {
        { OP("k_mul"); k_mul(e->feat, e->drop_dev, e->hfeat, (int64_t)imgs * e->D, e->st); }
        h = e->hfeat;
    }
    { OP("k_fc_fwd"); k_fc_fwd(h, S + e->off_fcw, S + e->off_fcb, e->logits, imgs, e->D, e->C, e->st); }
    e->ev_dirty = true;
}

void eff_forward_eval(fm_engine* e, const float* S, float* evs, float* evh, bool& dirty, int imgs, float* feat,
                      float* logits)
{
    if (dirty) {
        { OP("k_bn_eval_affine"); k_bn_eval_affine(S + e->off_gamma, S + e->off_beta, S + e->off_rm, S + e->off_rv, evs, evh, e->n_bn_ch,
                         e->bn_eps, e->st); }
        dirty = false;
    }
    auto sc = [&](int bi) { return evs + e->bns[bi].ch_off; };
    auto sh = [&](int bi) { return evh + e->bns[bi].ch_off; };
    { OP("conv_fwd"); conv_fwd(e, e->c_stem, S, e->x4, e->a0, imgs, 1, sc(e->bn_stem), sh(e->bn_stem), nullptr, 2, nullptr); }
    const float* cur = e->a0;
    e->ctx = 200;
    for (auto& m : e->mbs) {
        ++e->ctx;                                  // eval-mode ops are labelled @201..@216 (head @300)
        const int HWo = m.hout * m.wout;
        const float* a_e = cur;
        if (m.c_exp >= 0) {
            { OP("exp_fwd"); conv_fwd(e, m.c_exp, S, cur, m.a_e, imgs, 1, sc(m.bn0), sh(m.bn0), nullptr, 2, nullptr); }
            a_e = m.a_e;
        }
        bool pooled;                      // eval: y_d holds swish(bn1(.)) directly; its per-image channel sums come with it
        { OP("k_dw_fwd"); pooled = k_dw_fwd(a_e, S + m.dw_off, m.y_d, e->dt, sc(m.bn1), sh(m.bn1), imgs, m.hin, m.win, m.hout, m.wout, m.ce_p,
                 m.k, m.s, m.pad_t, m.pad_l, 2, e->st, e->ws_slab, nullptr, 1, e->se_pool); }
        { OP("k_se_fwd"); k_se_fwd(m.y_d, e->dt, nullptr, nullptr, 1, e->se_pool, S + m.w1_off, S + m.b1_off, S + m.w2_off, S + m.b2_off, m.sq,
                 m.rpre, m.gate, imgs, HWo, m.ce_p, m.cs, e->st, pooled); }
        // the gate multiplies the activation on the project conv's operand load: bf16 wherever the conv reads it once or twice,
        // fp32 where the conv streams through conv1x1.hip (K <= 256: the high-resolution blocks)
        const Conv& cpj = e->convs[m.c_proj];
        if (fuse_for(e, m) || (!e->precision && conv1x1_stream_takes(cpj.cin_p, cpj.cout_p, cpj.cout_p))) {
            const Prologue pro{nullptr, nullptr, m.gate};
            { OP("proj_fwd"); conv_fwd(e, m.c_proj, S, m.y_d, m.out, imgs, 1, sc(m.bn2), sh(m.bn2), m.skip ? cur : nullptr, 0, nullptr, &pro); }
        } else {
            { OP("k_se_scale"); k_se_scale(m.y_d, e->dt, nullptr, nullptr, 1, m.gate, m.a_s, imgs, HWo, m.ce_p, e->st); }
            { OP("proj_fwd"); conv_fwd(e, m.c_proj, S, m.a_s, m.out, imgs, 1, sc(m.bn2), sh(m.bn2), m.skip ? cur : nullptr, 0, nullptr); }
        }
        cur = m.out;
    }
    Conv& ch = e->convs[e->c_head];
    e->ctx = 300;
    { OP("conv_fwd"); conv_fwd(e, e->c_head, S, cur, e->T_mid, imgs, 1, sc(e->bn_head), sh(e->bn_head), nullptr, 2, nullptr); }
    { OP("k_avgpool"); k_avgpool(e->T_mid, e->dt, feat, imgs, ch.hout * ch.wout, e->D, e->st); }
    { OP("k_fc_fwd"); k_fc_fwd(feat, S + e->off_fcw, S + e->off_fcb, logits, imgs, e->D, e->C, e->st); }
}


void eff_backward_and_step(fm_engine* e, int groups, int B)
{
    const int imgs = groups * B;
    const float* S = e->state;
    float* G = e->grad;
    Conv& ch = e->convs[e->c_head];
    const int HWh = ch.hout * ch.wout;
    const float* h = e->drop_dev ? e->hfeat : e->feat;
    // Weight gradients on the side stream (side_w): each reads a gradient tensor the data-gradient chain has just produced
    // -- T_small (d y_p) for the project conv, T_mid (d y_d) for the depthwise conv, T_big (d y_e) for the expand conv --
    // and nothing downstream waits for it before Adam.  The three tensors are double-buffered by block parity; "produced"
    // events release the side stream, "consumed" events guard the overwrite two blocks later; the side stream has its own
    // slab workspace.  Arithmetic and summation orders are those of the inline order: results are bit-identical.
    const bool sw = e->side_w;
    float* Tsm[2] = {e->T_small, sw ? e->T_small2 : e->T_small};
    float* Tmd[2] = {e->T_mid, sw ? e->T_mid2 : e->T_mid};
    float* Tbg[2] = {e->T_big, sw ? e->T_big2 : e->T_big};
    hipStream_t main_st = e->st;
    auto side_begin = [&](int k, int par) {          // main has produced tensor k of parity par
        if (!sw) return;
        soft(e, hipEventRecord(e->ev_p[k][par], main_st));
        soft(e, hipStreamWaitEvent(e->st2, e->ev_p[k][par], 0));
        e->st = e->st2;
        std::swap(e->ws_slab, e->ws_slab2);
    };
    auto side_end = [&](int k, int par) {
        if (!sw) return;
        std::swap(e->ws_slab, e->ws_slab2);
        soft(e, hipEventRecord(e->ev_c[k][par], e->st2));
        e->st = main_st;
    };
    auto guard = [&](int k, int par) { if (sw) soft(e, hipStreamWaitEvent(main_st, e->ev_c[k][par], 0)); };
    e->ctx = 500;
    { OP("k_fc_bwd"); k_fc_bwd(e->dlogits, h, S + e->off_fcw, e->drop_dev, G + e->off_fcw, G + e->off_fcb, e->T_mid, e->dt, imgs, e->D, e->C,
             HWh, e->st); }
    { OP("bnact_bwd"); bnact_bwd(e, e->bn_head, e->T_mid, ch.y, e->T_mid, nullptr, groups, B * HWh, HWh, 2); }
    { OP("conv_wgrad"); conv_wgrad(e, e->c_head, e->mbs.back().out, e->T_mid, imgs); }
    float *go = e->GA, *gi = e->GB;
    { OP("conv_dgrad"); conv_dgrad(e, e->c_head, S, e->T_mid, go, imgs, nullptr, false); }
    for (int i = (int)e->mbs.size() - 1; i >= 0; --i) {
        MBConv& m = e->mbs[i];
        e->ctx = 400 + i;                          // backward ops @400..@415 (head @500, stem @399)
        const int par = i & 1;
        float *T_small = Tsm[par], *T_mid = Tmd[par], *T_big = Tbg[par];
        const float* in = i == 0 ? e->a0 : e->mbs[i - 1].out;
        const int HWi = m.hin * m.win, HWo = m.hout * m.wout;
        Conv& cp = e->convs[m.c_proj];
        const float* dc = (m.skip && e->dc_dev) ? e->dc_dev + (size_t)i * imgs : nullptr;
        // out = bn2(y_p)*dc + in
        guard(0, par);
        { OP("bnact_bwd"); bnact_bwd(e, m.bn2, go, cp.y, T_small, dc, groups, B * HWo, HWo, 0); }
        Bn& b1 = e->bns[m.bn1];
        float* dgp = (sw && par) ? e->se_dgp2 : e->se_dgp;       // the squeeze-excite gradient vectors alternate too
        float* drp = (sw && par) ? e->se_drp2 : e->se_drp;
        // early blocks (half of the depthwise-resolution bytes): d a_s is never stored -- pw_proj_bwd_kernel forms it twice on the
        // matrix pipe, once for the five per-image sums + the weight gradient, once for the BN1-backward apply (7 passes -> 3)
        const int nch5 = e->precision ? pw_proj_bwd_nch(m.ce_p, cp.cout_p, imgs, HWo) : pw_proj_bwd_f32_nch(m.ce_p, cp.cout_p, imgs, HWo);
        bool pfused = false;
        if (nch5) {
            PwProjBwdParams q{};
            PwProjBwdF32Params qf{};
            if (e->precision) {
                q.dYp = reinterpret_cast<const bf16*>(T_small); q.Yd = reinterpret_cast<const bf16*>(m.y_d);
                q.Wt = shadow_of(e, S) + cp.wbt_off; q.dYd = reinterpret_cast<bf16*>(T_mid);
                q.slab = e->ws_slab; q.pool5 = e->se_pool;
                q.sc = b1.scale; q.sh = b1.shift; q.mean = b1.mean; q.istd = b1.istd; q.ca = e->ca; q.cb = e->cb; q.cc = e->cc;
                q.gate = m.gate; q.ds = e->se_ds;
                q.L = m.ce_p; q.S = cp.cout_p; q.imgs = imgs; q.HW = HWo; q.ipg = B; q.nch = nch5;
            } else {
                qf.dYp = T_small; qf.Yd = m.y_d; qf.W = S + cp.w_off; qf.dYd = T_mid;
                qf.slab = e->ws_slab; qf.pool5 = e->se_pool;
                qf.sc = b1.scale; qf.sh = b1.shift; qf.mean = b1.mean; qf.istd = b1.istd; qf.ca = e->ca; qf.cb = e->cb; qf.cc = e->cc;
                qf.gate = m.gate; qf.ds = e->se_ds;
                qf.L = m.ce_p; qf.S = cp.cout_p; qf.imgs = imgs; qf.HW = HWo; qf.ipg = B; qf.nch = nch5;
            }
            auto launch = [&](int phase) {
                return e->precision ? launch_pw_proj_bwd(q, phase, e->slab_floats, e->st) : launch_pw_proj_bwd_f32(qf, phase, e->slab_floats, e->st);
            };
            int sk;
            { OP("proj_bwd_sums"); sk = launch(0);
              if (sk > 0) k_reduce_slabs(e->ws_slab, G + cp.w_off, sk, (int64_t)cp.w_numel, e->st); }
            if (sk > 0) {
                guard(3, par);
                { OP("k_se_bwd"); k_se_bwd_bn1(nullptr, m.y_d, e->dt, b1.scale, b1.shift, b1.mean, b1.istd, B, e->se_pool, m.gate, m.rpre,
                             S + m.w1_off, S + m.w2_off, dgp, drp, e->se_ds, e->ws_part, imgs, HWo, m.ce_p, m.cs, e->st, nch5); }
                side_begin(3, par);
                { OP("k_se_wgrad"); k_se_wgrad(dgp, drp, m.rpre, m.sq, e->ws_slab, G + m.w1_off, imgs, m.ce_p, m.cs, e->st); }
                side_end(3, par);
                guard(1, par);
                { OP("proj_bwd_apply");
                  k_bn_bwd_finalize(e->ws_part, groups, se_bwd_bn1_splits(B), b1.C, B * HWo, e->state + e->off_gamma + b1.ch_off, b1.mean,
                                    b1.istd, e->ca, e->cb, e->cc, e->grad + e->off_gamma + b1.ch_off, e->grad + e->off_beta + b1.ch_off,
                                    e->st);
                  if (launch(1) != 1) soft(e, hipErrorInvalidValue); }      // d y_d
                pfused = true;
            }
        }
        if (!pfused) {
            if (!e->precision && fuse_for(e, m)) soft(e, hipErrorInvalidValue);     // fp32: a_s was not stored for this block and only
                                                                                    // the fused backward can do without it
            side_begin(0, par);
            if (fuse_for(e, m)) {      // the project conv's operand a_s was never stored: re-formed from y_d on load
                const Prologue pro{b1.scale, b1.shift, m.gate};
                { OP("proj_wgrad"); conv_wgrad(e, m.c_proj, m.y_d, T_small, imgs, &pro, B * HWo); }
            } else
                { OP("proj_wgrad"); conv_wgrad(e, m.c_proj, m.a_s, T_small, imgs); }
            side_end(0, par);
            guard(1, par);
            { OP("proj_dgrad"); conv_dgrad(e, m.c_proj, S, T_small, T_mid, imgs, nullptr, false); }          // d a_s
            // a_s = a_d * gate(a_d)
            // ONE pass over (d a_s, y_d) yields the squeeze-excite backward's pooled sums and the BN1-backward sums
            guard(3, par);
            { OP("k_se_bwd"); k_se_bwd_bn1(T_mid, m.y_d, e->dt, b1.scale, b1.shift, b1.mean, b1.istd, B, e->se_pool, m.gate, m.rpre,
                         S + m.w1_off, S + m.w2_off, dgp, drp, e->se_ds, e->ws_part, imgs, HWo, m.ce_p, m.cs, e->st); }
            side_begin(3, par);
            { OP("k_se_wgrad"); k_se_wgrad(dgp, drp, m.rpre, m.sq, e->ws_slab, G + m.w1_off, imgs, m.ce_p, m.cs, e->st); }
            side_end(3, par);
            // d a_d = d a_s * gate + ds/HW is formed on load inside the BN backward's apply pass
            { OP("bnact_bwd"); bnact_bwd(e, m.bn1, T_mid, m.y_d, T_mid, nullptr, groups, B * HWo, HWo, 2, m.gate, e->se_ds, -1,
                                         se_bwd_bn1_splits(B)); }   // d y_d
        }
        const float* a_e = m.c_exp >= 0 ? m.a_e : in;
        side_begin(1, par);
        { OP("k_dw_wgrad"); k_dw_wgrad(T_mid, a_e, e->dt, e->ws_slab, G + m.dw_off, imgs, m.hin, m.win, m.hout, m.wout, m.ce_p, m.k, m.s,
                   m.pad_t, m.pad_l, e->st); }
        side_end(1, par);
        if (m.c_exp >= 0) {
            Conv& ce = e->convs[m.c_exp];
            Bn& b0 = e->bns[m.bn0];
            bool sums;                           // d a_e, and the BN0-backward sums from the same registers
            guard(2, par);
            { OP("k_dw_dgrad"); sums = k_dw_dgrad(T_mid, S + m.dw_off, T_big, e->dt, imgs, m.hin, m.win, m.hout, m.wout, m.ce_p, m.k, m.s,
                       m.pad_t, m.pad_l, e->st, ce.y, b0.mean, b0.istd, b0.scale, b0.shift, e->ws_slab, e->ws_part, groups); }
            bool fused = false;
            if ((ce.cout_p == 96 || ce.cout_p == 144) && ce.cin_p <= 32) {
                // early blocks (62 % of the expanded-tensor bytes): BN0-backward apply + weight gradient + data gradient of the
                // expand conv in ONE kernel that reads d a_e and y_e once (pw_exp_bwd_kernel / pw_exp_bwd_f32_kernel) instead of
                // five passes over them
                OP("exp_bwd_fused");
                const int pix = B * HWi;
                if (!sums)
                    k_chan_reduce(T_big, e->dt, ce.y, e->dt, b0.mean, b0.istd, b0.scale, b0.shift, nullptr, e->ws_part, groups, pix,
                                  HWi, b0.C, 1, 2, nullptr, nullptr, e->st);
                k_bn_bwd_finalize(e->ws_part, groups, sums ? dw_stats_tiles() : bn_bwd_blocks(pix), b0.C, pix,
                                  e->state + e->off_gamma + b0.ch_off, b0.mean, b0.istd, e->ca, e->cb, e->cc,
                                  e->grad + e->off_gamma + b0.ch_off, e->grad + e->off_beta + b0.ch_off, e->st);
                int sk;
                if (e->precision) {
                    PwExpBwdParams q{};
                    q.dA = reinterpret_cast<const bf16*>(T_big); q.Ye = reinterpret_cast<const bf16*>(ce.y);
                    q.X = reinterpret_cast<const bf16*>(in); q.Wt = shadow_of(e, S) + ce.wbt_off;
                    q.res = reinterpret_cast<const bf16*>(m.skip ? go : nullptr); q.dX = reinterpret_cast<bf16*>(gi);
                    q.slab = e->ws_slab; q.ca = e->ca; q.cb = e->cb; q.cc = e->cc; q.sc = b0.scale; q.sh = b0.shift;
                    q.L = ce.cout_p; q.S = ce.cin_p; q.npix = imgs * HWi; q.pix_per_group = pix; q.groups = groups;
                    sk = launch_pw_exp_bwd(q, e->slab_floats, e->st);
                } else {
                    PwExpBwdF32Params q{};
                    q.dA = T_big; q.Ye = ce.y; q.X = in; q.W = S + ce.w_off; q.res = m.skip ? go : nullptr; q.dX = gi;
                    q.slab = e->ws_slab; q.ca = e->ca; q.cb = e->cb; q.cc = e->cc; q.sc = b0.scale; q.sh = b0.shift;
                    q.L = ce.cout_p; q.S = ce.cin_p; q.npix = imgs * HWi; q.pix_per_group = pix; q.groups = groups;
                    sk = launch_pw_exp_bwd_f32(q, e->slab_floats, e->st);
                }
                if (sk > 0) {
                    k_reduce_slabs(e->ws_slab, G + ce.w_off, sk, (int64_t)ce.w_numel, e->st);
                    fused = true;
                }
            }
            if (!fused) {
                // (a refused fused launch has left ca / cb / cc and the BN gradients exactly as bnact_bwd is about to)
                { OP("bnact_bwd"); bnact_bwd(e, m.bn0, T_big, ce.y, T_big, nullptr, groups, B * HWi, HWi, 2, nullptr, nullptr, -1,
                                             sums ? dw_stats_tiles() : 0); }
                side_begin(2, par);
                { OP("exp_wgrad"); conv_wgrad(e, m.c_exp, in, T_big, imgs); }
                side_end(2, par);
                { OP("exp_dgrad"); conv_dgrad(e, m.c_exp, S, T_big, gi, imgs, m.skip ? go : nullptr, false); }
            }
        } else {
            { OP("k_dw_dgrad"); k_dw_dgrad(T_mid, S + m.dw_off, gi, e->dt, imgs, m.hin, m.win, m.hout, m.wout, m.ce_p, m.k, m.s, m.pad_t,
                       m.pad_l, e->st); }
            if (m.skip) k_add_inplace(gi, go, e->dt, (int64_t)imgs * HWi * m.cin_p, e->st);
        }
        std::swap(go, gi);
    }
    Conv& cs = e->convs[e->c_stem];
    e->ctx = 399;
    { OP("bnact_bwd"); bnact_bwd(e, e->bn_stem, go, cs.y, go, nullptr, groups, B * cs.hout * cs.wout, cs.hout * cs.wout, 2); }
    { OP("conv_wgrad"); conv_wgrad(e, e->c_stem, e->x4, go, imgs); }
    if (sw) {                                    // every weight gradient is in G before the optimizer reads it
        soft(e, hipEventRecord(e->ev_wdone, e->st2));
        soft(e, hipStreamWaitEvent(main_st, e->ev_wdone, 0));
    }
    { OP("adam_step"); adam_step(e); }
}

// model dispatch
void net_forward_train(fm_engine* e, int groups, int B)
{
    ensure_packed(e);
    if (e->model == 1) eff_forward_train(e, groups, B);
    else forward_train(e, groups, B);
}
void net_forward_eval(fm_engine* e, bool teacher, int imgs)
{
    const float* S = teacher ? e->tstate : e->state;
    float* evs = teacher ? e->tev_scale : e->ev_scale;
    float* evh = teacher ? e->tev_shift : e->ev_shift;
    bool& dirty = teacher ? e->tev_dirty : e->ev_dirty;
    float* feat = teacher ? e->tfeat : e->feat;
    float* logits = teacher ? e->tlogits : e->logits;
    if (teacher) ensure_teacher_shadow(e);
    else ensure_packed(e);
    if (e->model == 1) eff_forward_eval(e, S, evs, evh, dirty, imgs, feat, logits);
    else forward_eval(e, S, evs, evh, dirty, imgs, feat, logits);
}
// teacher forward on the side stream: same code, the teacher's buffer set and stream swapped in while it is enqueued
void swap_teacher_ws(fm_engine* e)
{
    for (auto& m : e->mbs) {
        std::swap(m.a_e, m.t_a_e); std::swap(m.y_d, m.t_y_d); std::swap(m.a_s, m.t_a_s); std::swap(m.out, m.t_out);
        std::swap(m.sq, m.t_sq); std::swap(m.rpre, m.t_rpre); std::swap(m.gate, m.t_gate);
    }
    std::swap(e->sk_slab, e->sk_slab2); std::swap(e->sk_counters, e->sk_counters2);
    if (e->model == 0) {
        std::swap(e->convs[0].y, e->t_c0y); std::swap(e->p0, e->t_p0); std::swap(e->p0p, e->t_p0p);
        for (auto& blk : e->blocks) {
            std::swap(blk.z1, blk.t_z1); std::swap(blk.out, blk.t_out);
            std::swap(blk.z1p, blk.t_z1p); std::swap(blk.outp, blk.t_outp);
            if (blk.ds >= 0) std::swap(e->convs[blk.ds].y, blk.t_dsy);
        }
        return;
    }
    std::swap(e->a0, e->t_a0); std::swap(e->T_mid, e->t_Tmid); std::swap(e->se_pool, e->t_se_pool);
    std::swap(e->ws_slab, e->t_rec);
}
int teacher_forward_side(fm_engine* e, int imgs)
{
    ensure_teacher_shadow(e);                                   // weight shadows on the main stream, before the fork
    HIPCHK(hipEventRecord(e->ev_in, e->st));
    HIPCHK(hipStreamWaitEvent(e->st2, e->ev_in, 0));
    hipStream_t main_st = e->st;
    e->st = e->st2;
    swap_teacher_ws(e);
    net_forward_eval(e, true, imgs);
    swap_teacher_ws(e);
    e->st = main_st;
    HIPCHK(hipEventRecord(e->ev_t, e->st2));
    return FM_OK;
}
void net_backward_and_step(fm_engine* e, int groups, int B)
{
    ensure_packed(e);
    if (e->model == 1) eff_backward_and_step(e, groups, B);
    else backward_and_step(e, groups, B);
}

ClassVec to_cv(const float* h, int C)
{
    ClassVec v{};
    for (int i = 0; i < C; ++i) v.v[i] = h[i];
    return v;
}

}  // namespace

// =============================== C ABI =======================================
extern "C" {

int fm_create(const fm_config* cfg, fm_engine** out)
{
    ARGCHK(cfg && out, "null cfg/out");
    ARGCHK(cfg->model == 0 || cfg->model == 1, "model must be 0 (ResNet-18) or 1 (EfficientNet-B0)");
    ARGCHK(cfg->n_classes >= 1 && cfg->n_classes <= FM_MAX_CLASSES, "n_classes out of range");
    ARGCHK(cfg->in_h >= 32 && cfg->in_w >= 32 && cfg->in_h % 32 == 0 && cfg->in_w % 32 == 0,
           "in_h/in_w must be multiples of 32");
    ARGCHK(cfg->max_images >= 1, "max_images");
    ARGCHK(cfg->reserved[0] == 0 || (cfg->reserved[0] == 1 && cfg->model == 1),
           "precision (reserved[0]) must be 0 (fp32) or, for EfficientNet-B0, 1 (bf16 activations)");
    ARGCHK(cfg->reserved[1] >= 0 && cfg->reserved[1] <= 2, "reserved[1] (stream mode) must be 0, 1 or 2");
    ARGCHK(cfg->reserved[2] >= 0 && cfg->reserved[2] <= 3,
           "reserved[2] (product form of the fp32 conv GEMMs) must be 0 (library default: six bf16 partial products), 1 (fp32 "
           "matrix pipe), 2 (nine bf16 partial products) or 3 (exactly six, whatever the default)");
    fm_engine* e = new fm_engine();
    e->precision = cfg->reserved[0];
    e->stream_mode = cfg->reserved[1];
    // the product form belongs to the handle: fixed here, carried to every launch in the kernels' parameter blocks.  0 resolves
    // to the library default, which the test-only FM_MFMA_SPLIT overrides (read once, here)
    e->products = cfg->reserved[2] == 1 ? 0 : (cfg->reserved[2] == 2 ? 9 : (cfg->reserved[2] == 3 ? 6 : fm_mfma_split()));
    // planes mode: ResNet-18 in a split product form (FM_PLANES=0 keeps the fp32-operand kernels of igemm.hip: the A/B arm)
    e->planes = cfg->model == 0 && cfg->reserved[0] == 0 && e->products != 0 && !(getenv("FM_PLANES") && atoi(getenv("FM_PLANES")) == 0);
    e->dt = e->precision ? DT_BF16 : DT_F32;
    e->fuse_gate = e->precision && !(getenv("FM_FUSE_GATE") && atoi(getenv("FM_FUSE_GATE")) == 0);
    e->cfg = *cfg;
    e->st = reinterpret_cast<hipStream_t>(cfg->stream);
    e->C = cfg->n_classes; e->H = cfg->in_h; e->W = cfg->in_w; e->maxB = cfg->max_images;
    e->model = cfg->model;
    int rc = e->model == 1 ? build_effnet_b0(e) : build_resnet18(e);
    if (rc == FM_OK) rc = build_tables(e);
    if (rc == FM_OK) rc = alloc_workspaces(e);
    if (rc != FM_OK) { fm_destroy(e); return rc; }
    *out = e;
    return FM_OK;
}

int fm_destroy(fm_engine* e)
{
    if (!e) return FM_OK;
    (void)hipStreamSynchronize(e->st);
    if (e->st2) { (void)hipStreamSynchronize(e->st2); (void)hipStreamDestroy(e->st2); }
    if (e->ev_in) (void)hipEventDestroy(e->ev_in);
    if (e->ev_t) (void)hipEventDestroy(e->ev_t);
    for (int k = 0; k < 4; ++k)
        for (int q = 0; q < 2; ++q) {
            if (e->ev_p[k][q]) (void)hipEventDestroy(e->ev_p[k][q]);
            if (e->ev_c[k][q]) (void)hipEventDestroy(e->ev_c[k][q]);
        }
    if (e->ev_wdone) (void)hipEventDestroy(e->ev_wdone);
    if (e->comm) { (void)fmcomm_destroy(e->comm); e->comm = nullptr; }
    if (e->host_err) (void)hipHostFree(e->host_err);
    for (void* p : e->allocs) (void)hipFree(p);
    for (auto& p : e->evs) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
    for (auto v : e->ev_free) (void)hipEventDestroy(v);
    delete e;
    return FM_OK;
}

int fm_sync(fm_engine* e)
{
    ARGCHK(e, "null engine");
    HIPCHK(hipStreamSynchronize(e->st));
    STEP_DONE(e);
    return FM_OK;
}

int fm_state_sizes(fm_engine* e, int64_t* n_f32, int64_t* n_i64)
{
    ARGCHK(e, "null engine");
    if (n_f32) *n_f32 = e->nf_sd;
    if (n_i64) *n_i64 = e->ni_sd;
    return FM_OK;
}

int fm_set_state(fm_engine* e, const float* host_f32, const int64_t* host_i64)
{
    ARGCHK(e && host_f32, "null engine/state");
    HIPCHK(hipMemcpyAsync(e->stage_sd, host_f32, (size_t)e->nf_sd * 4, hipMemcpyHostToDevice, e->st));
    size_t off = 0;
    int ic = 0;
    for (auto& en : e->entries) {
        if (en.kind == 0) {
            k_oihw_to_ohwi(e->stage_sd + off, e->state + en.eng_off, en.O, en.I, en.KH, en.KW, en.Wpad, en.Ipad, e->st, en.Ostride);
            off += en.n;
        } else if (en.kind == 1) {
            HIPCHK(hipMemcpyAsync(e->state + en.eng_off, e->stage_sd + off, en.n * 4, hipMemcpyDeviceToDevice, e->st));
            off += en.n;
        } else {
            e->counters[en.bn] = host_i64 ? host_i64[ic] : 0;
            ++ic;
        }
    }
    HIPCHK(hipStreamSynchronize(e->st));
    e->ev_dirty = true;
    e->wpack_dirty = true;
    return FM_OK;
}

int fm_get_state(fm_engine* e, float* host_f32, int64_t* host_i64)
{
    ARGCHK(e && host_f32, "null engine/state");
    size_t off = 0;
    int ic = 0;
    for (auto& en : e->entries) {
        if (en.kind == 0) {
            k_ohwi_to_oihw(e->state + en.eng_off, e->stage_sd + off, en.O, en.I, en.KH, en.KW, en.Wpad, en.Ipad, e->st, en.Ostride);
            off += en.n;
        } else if (en.kind == 1) {
            HIPCHK(hipMemcpyAsync(e->stage_sd + off, e->state + en.eng_off, en.n * 4, hipMemcpyDeviceToDevice, e->st));
            off += en.n;
        } else {
            if (host_i64) host_i64[ic] = e->counters[en.bn];
            ++ic;
        }
    }
    HIPCHK(hipMemcpyAsync(host_f32, e->stage_sd, (size_t)e->nf_sd * 4, hipMemcpyDeviceToHost, e->st));
    HIPCHK(hipStreamSynchronize(e->st));
    return FM_OK;
}

int fm_state_device(fm_engine* e, float** dev_ptr, int64_t* numel)
{
    ARGCHK(e && dev_ptr && numel, "null");
    *dev_ptr = e->state;
    *numel = (int64_t)e->NS;
    e->ev_dirty = true;       // the caller is about to overwrite it (all-reduce)
    e->wpack_dirty = true;
    return FM_OK;
}

int fm_counters(fm_engine* e, int64_t* host_i64, int32_t set)
{
    ARGCHK(e && host_i64, "null");
    int ic = 0;
    for (auto& en : e->entries)
        if (en.kind == 2) {
            if (set) e->counters[en.bn] = host_i64[ic];
            else host_i64[ic] = e->counters[en.bn];
            ++ic;
        }
    return FM_OK;
}

int fm_state_scale(fm_engine* e, float w)
{
    ARGCHK(e, "null engine");
    k_scale(e->state, w, (int64_t)e->NS, e->st);
    e->ev_dirty = true;
    e->wpack_dirty = true;
    return FM_OK;
}

int fm_stream_mode(fm_engine* e)
{
    if (!e) return -1;
    return e->side_ok ? (e->side_w ? 0 : 2) : 1;
}

int fm_mfma_products(void) { return fm_mfma_split(); }
int fm_products(fm_engine* e) { return e ? e->products : FM_ERR_ARG; }
int fm_planes_mode(fm_engine* e) { return e ? (e->planes ? 1 : 0) : FM_ERR_ARG; }

int fm_fedavg_fold(fm_engine* e, const float* const* states_dev, const float* n_host, int32_t K, float* out_dev)
{
    ARGCHK(e && states_dev && n_host && out_dev, "null argument");
    ARGCHK(K >= 1 && K <= FM_FOLD_MAX, "fm_fedavg_fold: 1 <= K <= FM_FOLD_MAX");
    FoldArgs a{};
    float tot = 0.f;                                    // sum(dict_len) as the reference's Python int sum, exact below 2^24
    double tot_d = 0.0;
    for (int k = 0; k < K; ++k) {
        ARGCHK(states_dev[k], "fm_fedavg_fold: null state pointer");
        a.s[k] = states_dev[k];
        a.n[k] = n_host[k];
        tot_d += (double)n_host[k];
    }
    tot = (float)tot_d;
    k_fedavg_fold(a, K, tot, out_dev, (int64_t)e->NS, e->st);
    if (out_dev == e->state) { e->ev_dirty = true; e->wpack_dirty = true; e->wb_dirty = true; }
    HIPCHK(hipGetLastError());
    return FM_OK;
}

/* ---- RCCL inside the library ------------------------------------------------------------- */
#define COMMCHK(x)                                       \
    do {                                                 \
        if (!(x)) { g_err = fmcomm_error(); return FM_ERR_HIP; } \
    } while (0)

int fm_comm_preflight(void)
{
    COMMCHK(fmcomm_preflight());
    return FM_OK;
}

int fm_comm_unique_id(uint8_t* id128)
{
    ARGCHK(id128, "null id");
    COMMCHK(fmcomm_unique_id(id128));
    return FM_OK;
}

int fm_comm_init(fm_engine* e, const uint8_t* id128, int32_t rank, int32_t world)
{
    ARGCHK(e && id128 && world >= 1 && rank >= 0 && rank < world, "comm arguments");
    if (e->comm) { COMMCHK(fmcomm_destroy(e->comm)); e->comm = nullptr; }
    COMMCHK(fmcomm_init(&e->comm, id128, rank, world));
    e->comm_rank = rank; e->comm_world = world;
    const size_t need = std::max<size_t>((size_t)2 * e->C * e->D + 2 * e->C, e->counters.size()) + 64;
    if (e->comm_buf_n < need) { DALLOC(e->comm_buf, need); e->comm_buf_n = need; }
    return FM_OK;
}

int fm_comm_destroy(fm_engine* e)
{
    ARGCHK(e, "null engine");
    if (e->comm) { HIPCHK(hipStreamSynchronize(e->st)); COMMCHK(fmcomm_destroy(e->comm)); }
    e->comm = nullptr; e->comm_world = 0;
    return FM_OK;
}

int fm_comm_size(fm_engine* e) { return (e && e->comm) ? e->comm_world : 0; }

int fm_fedavg_allreduce(fm_engine* e, float w)
{
    ARGCHK(e, "null engine");
    k_scale(e->state, w, (int64_t)e->NS, e->st);
    e->ev_dirty = true; e->wpack_dirty = true;
    if (!e->comm) return FM_OK;       // no communicator: a single client, FedAvg of one = w * state
    // (1) the whole fp32 arena, in place, on the engine stream: the next round's first kernels queue behind it
    COMMCHK(fmcomm_allreduce_sum(e->comm, e->state, e->NS, false, e->st));
    // (2) num_batches_tracked: weighted mean in float64, truncated on load like utils/FedAvg.py:13 + load_state_dict
    const size_t nb = e->counters.size();
    std::vector<double> h(nb);
    for (size_t i = 0; i < nb; ++i) h[i] = (double)w * (double)e->counters[i];
    HIPCHK(hipMemcpyAsync(e->comm_buf, h.data(), nb * 8, hipMemcpyHostToDevice, e->st));
    COMMCHK(fmcomm_allreduce_sum(e->comm, e->comm_buf, nb, true, e->st));
    HIPCHK(hipMemcpyAsync(h.data(), e->comm_buf, nb * 8, hipMemcpyDeviceToHost, e->st));
    HIPCHK(hipStreamSynchronize(e->st));
    for (size_t i = 0; i < nb; ++i) e->counters[i] = (int64_t)trunc(h[i] + 1e-9);
    return FM_OK;
}

int fm_fedavg_tao(fm_engine* e, const double* t_host, double n_i, const float* negative_mask_host, double* out_host)
{
    ARGCHK(e && t_host && negative_mask_host && out_host, "null");
    const int C = e->C;
    std::vector<double> h(2 * C);
    for (int c = 0; c < C; ++c) {
        const double m = (double)negative_mask_host[c];      // 0/1 for one client; a weight when a rank folds several
        h[c] = m != 0.0 ? t_host[c] * n_i * m : 0.0;
        h[C + c] = n_i * m;
    }
    if (e->comm) {
        HIPCHK(hipMemcpyAsync(e->comm_buf, h.data(), h.size() * 8, hipMemcpyHostToDevice, e->st));
        COMMCHK(fmcomm_allreduce_sum(e->comm, e->comm_buf, h.size(), true, e->st));
        HIPCHK(hipMemcpyAsync(h.data(), e->comm_buf, h.size() * 8, hipMemcpyDeviceToHost, e->st));
        HIPCHK(hipStreamSynchronize(e->st));
    }
    for (int c = 0; c < C; ++c) out_host[c] = h[C + c] == 0.0 ? 1.0 : h[c] / h[C + c];   // no active client: 1.0 (:66-67)
    return FM_OK;
}

int fm_fedavg_proto(fm_engine* e, const float* proto_host, double n_i, const float* active_mask_host, float* out_host)
{
    ARGCHK(e && proto_host && active_mask_host && out_host, "null");
    const int C = e->C, D = e->D;
    const size_t np = (size_t)2 * C * D;
    std::vector<float> h(np + 2 * C);
    for (int r = 0; r < 2 * C; ++r) {
        const float wr = (float)(n_i * (double)active_mask_host[r / 2]);   // mask 0/1, or a per-class weight
        for (int d = 0; d < D; ++d) h[(size_t)r * D + d] = wr != 0.f ? proto_host[(size_t)r * D + d] * wr : 0.f;
        h[np + r] = wr;
    }
    if (e->comm) {
        float* buf = reinterpret_cast<float*>(e->comm_buf);
        HIPCHK(hipMemcpyAsync(buf, h.data(), h.size() * 4, hipMemcpyHostToDevice, e->st));
        COMMCHK(fmcomm_allreduce_sum(e->comm, buf, h.size(), false, e->st));
        HIPCHK(hipMemcpyAsync(h.data(), buf, h.size() * 4, hipMemcpyDeviceToHost, e->st));
        HIPCHK(hipStreamSynchronize(e->st));
    }
    for (int r = 0; r < 2 * C; ++r)
        for (int d = 0; d < D; ++d) out_host[(size_t)r * D + d] = h[(size_t)r * D + d] / h[np + r];   // 0/0 = NaN row (:85-86)
    return FM_OK;
}

int fm_teacher_snapshot(fm_engine* e)
{
    ARGCHK(e, "null engine");
    HIPCHK(hipMemcpyAsync(e->tstate, e->state, e->NS * 4, hipMemcpyDeviceToDevice, e->st));
    e->tcounters = e->counters;
    e->tev_dirty = true;
    e->twb_dirty = true;
    return FM_OK;
}

int fm_adam_reset(fm_engine* e, const fm_adam* hp)
{
    ARGCHK(e, "null engine");
    if (hp) e->hp = *hp;
    e->adam_t = 0;
    HIPCHK(hipMemsetAsync(e->adam_m, 0, e->NP * 4, e->st));
    HIPCHK(hipMemsetAsync(e->adam_v, 0, e->NP * 4, e->st));
    return FM_OK;
}

int fm_forward_eval(fm_engine* e, const float* x_dev, int32_t B, int32_t use_teacher, float* feat_dev,
                    float* logits_dev)
{
    ARGCHK(e && x_dev, "null");
    ARGCHK(B >= 1 && B <= e->maxB, "B exceeds max_images");
    const float* xs[1] = {x_dev};
    to_nhwc4(e, xs, 1, B);
    net_forward_eval(e, use_teacher != 0, B);
    if (feat_dev)
        HIPCHK(hipMemcpyAsync(feat_dev, use_teacher ? e->tfeat : e->feat, (size_t)B * e->D * 4,
                              hipMemcpyDeviceToDevice, e->st));
    if (logits_dev)
        HIPCHK(hipMemcpyAsync(logits_dev, use_teacher ? e->tlogits : e->logits, (size_t)B * e->C * 4,
                              hipMemcpyDeviceToDevice, e->st));
    HIPCHK(hipGetLastError());
    return FM_OK;
}

int fm_step_bce(fm_engine* e, const float* x_dev, const float* y_dev, int32_t B, const float* pos_weight_host,
                int32_t bs_norm, float* loss_dev)
{
    ARGCHK(e && x_dev && y_dev && pos_weight_host && loss_dev, "null");
    ARGCHK(B >= 1 && B <= e->maxB, "B exceeds max_images");
    const float* xs[1] = {x_dev};
    to_nhwc4(e, xs, 1, B);
    net_forward_train(e, 1, B);
    k_loss_bce(e->logits, y_dev, to_cv(pos_weight_host, e->C), B, e->C, 1.f / ((float)bs_norm * (float)e->C),
               e->dlogits, loss_dev, e->st);
    net_backward_and_step(e, 1, B);
    STEP_DONE(e);
    return FM_OK;
}

int fm_step_stage1(fm_engine* e, const float* x1_dev, const float* x2_dev, const float* y_dev, int32_t B,
                   const float* active_mask_host, int32_t annotation_num, int32_t bs_norm, float* loss_dev)
{
    ARGCHK(e && x1_dev && x2_dev && y_dev && active_mask_host && loss_dev, "null");
    ARGCHK(B >= 1 && 2 * B <= e->maxB, "2*B exceeds max_images");
    int n_neg = 0;
    for (int c = 0; c < e->C; ++c) n_neg += active_mask_host[c] == 0.f;
    const float* xs[2] = {x1_dev, x2_dev};
    to_nhwc4(e, xs, 2, B);
    if (e->side_ok) {
        // frozen teacher on the side stream (own buffers), student on the main stream; they meet at the loss
        const int rc = teacher_forward_side(e, 2 * B);
        if (rc != FM_OK) return rc;
        net_forward_train(e, 2, B);
        HIPCHK(hipStreamWaitEvent(e->st, e->ev_t, 0));
    } else {
        // frozen teacher first (eval mode; its activations may be overwritten by the student)
        net_forward_eval(e, true, 2 * B);
        net_forward_train(e, 2, B);
    }
    k_loss_stage1(e->logits, e->tlogits, y_dev, to_cv(active_mask_host, e->C), B, e->C,
                  1.f / ((float)bs_norm * (float)annotation_num),
                  n_neg ? 1.f / ((float)bs_norm * (float)n_neg) : 0.f, e->dlogits, loss_dev, e->st);
    net_backward_and_step(e, 2, B);
    STEP_DONE(e);
    return FM_OK;
}

int fm_step_stage2(fm_engine* e, const float* x_dev, const float* y_dev, const float* distill_dev, int32_t B,
                   float* loss_dev)
{
    ARGCHK(e && x_dev && y_dev && distill_dev && loss_dev, "null");
    ARGCHK(B >= 1 && B <= e->maxB, "B exceeds max_images");
    const float* xs[1] = {x_dev};
    to_nhwc4(e, xs, 1, B);
    net_forward_train(e, 1, B);
    k_loss_stage2(e->logits, y_dev, distill_dev, B, e->C, e->dlogits, loss_dev, e->st);
    net_backward_and_step(e, 1, B);
    STEP_DONE(e);
    return FM_OK;
}

int fm_step_fixmatch(fm_engine* e, const float* xw_dev, const float* xs_dev, const float* y_dev, int32_t B,
                     const float* pos_weight_host, const float* pos_weight_unk_host, const float* active_mask_host,
                     int32_t annotation_num, int32_t bs_norm, float* loss_dev)
{
    ARGCHK(e && xw_dev && xs_dev && y_dev && pos_weight_host && pos_weight_unk_host && active_mask_host && loss_dev,
           "null");
    ARGCHK(B >= 1 && 2 * B <= e->maxB && B <= 2048, "2*B exceeds max_images");
    int n_neg = 0;
    for (int c = 0; c < e->C; ++c) n_neg += active_mask_host[c] == 0.f;
    const float* xs[2] = {xw_dev, xs_dev};
    to_nhwc4(e, xs, 2, B);
    net_forward_train(e, 2, B);
    k_loss_fixmatch(e->logits, y_dev, to_cv(pos_weight_host, e->C), to_cv(pos_weight_unk_host, e->C),
                    to_cv(active_mask_host, e->C), B, e->C, n_neg, 1.f / ((float)bs_norm * (float)annotation_num),
                    e->C - annotation_num, e->dlogits, loss_dev, e->st);
    net_backward_and_step(e, 2, B);
    STEP_DONE(e);
    return FM_OK;
}

int fm_proto_reset(fm_engine* e)
{
    ARGCHK(e, "null engine");
    HIPCHK(hipMemsetAsync(e->psum, 0, (size_t)2 * e->C * e->D * 4, e->st));
    HIPCHK(hipMemsetAsync(e->pcnt, 0, (size_t)2 * e->C * 8, e->st));
    HIPCHK(hipMemsetAsync(e->tcnt, 0, (size_t)e->C * 8, e->st));
    return FM_OK;
}

int fm_proto_accumulate(fm_engine* e, const float* feat_dev, const float* logits_dev, const float* labels_dev,
                        int32_t B, const float* active_mask_host, const float* negative_mask_host, float L, float U)
{
    ARGCHK(e && feat_dev && logits_dev && labels_dev && active_mask_host && negative_mask_host, "null");
    k_proto_accumulate(feat_dev, logits_dev, labels_dev, B, e->D, e->C, to_cv(active_mask_host, e->C),
                       to_cv(negative_mask_host, e->C), L, U, e->psum, e->pcnt, e->tcnt, e->st);
    return FM_OK;
}

int fm_proto_finalize(fm_engine* e, int32_t zero_guard, int64_t n_local, const float* active_mask_host,
                      float* proto_host, double* t_host)
{
    ARGCHK(e && active_mask_host && proto_host && t_host, "null");
    std::vector<int64_t> pc(2 * e->C), tc(e->C);
    const int D = e->D;
    HIPCHK(hipMemcpyAsync(proto_host, e->psum, (size_t)2 * e->C * D * 4, hipMemcpyDeviceToHost, e->st));
    HIPCHK(hipMemcpyAsync(pc.data(), e->pcnt, pc.size() * 8, hipMemcpyDeviceToHost, e->st));
    HIPCHK(hipMemcpyAsync(tc.data(), e->tcnt, tc.size() * 8, hipMemcpyDeviceToHost, e->st));
    HIPCHK(hipStreamSynchronize(e->st));
    for (int c = 0; c < e->C; ++c) {
        t_host[c] = (double)tc[c] / (double)n_local;
        if (active_mask_host[c] == 0.f) continue;
        for (int v = 0; v < 2; ++v) {
            const int r = 2 * c + v;
            if (zero_guard && pc[r] == 0) continue;
            const float den = (float)pc[r];
            for (int d = 0; d < D; ++d) proto_host[(size_t)r * D + d] = proto_host[(size_t)r * D + d] / den;
        }
    }
    return FM_OK;
}

int fm_cos_tag(fm_engine* e, const float* feat_dev, int64_t N, const float* proto_dev, const int32_t* classes_host,
               int32_t n_cls, float* sim_dev)
{
    ARGCHK(e && feat_dev && proto_dev && classes_host && sim_dev, "null");
    ARGCHK(n_cls >= 0 && n_cls <= FM_MAX_CLASSES, "n_cls");
    if (n_cls == 0 || N == 0) return FM_OK;
    HIPCHK(hipMemcpyAsync(e->cls_dev, classes_host, (size_t)n_cls * 4, hipMemcpyHostToDevice, e->st));
    k_cos_tag(feat_dev, N, e->D, proto_dev, e->cls_dev, n_cls, sim_dev, e->st);
    return FM_OK;
}

int fm_select_topk(fm_engine* e, const float* sim_dev, int64_t N, double clean_thr, double noise_thr, int32_t cap,
                   int32_t* top_host, int32_t* n_top, int32_t* bot_host, int32_t* n_bot)
{
    ARGCHK(e && top_host && n_top && bot_host && n_bot, "null");
    *n_top = *n_bot = 0;
    if (N == 0) return FM_OK;              // an empty pool selects nothing (sim_dev may be NULL then)
    ARGCHK(sim_dev, "null sim_dev");
    int counts[2];
    k_count_sign(sim_dev, N, e->sel_counts, e->st);
    HIPCHK(hipMemcpyAsync(counts, e->sel_counts, 8, hipMemcpyDeviceToHost, e->st));
    HIPCHK(hipStreamSynchronize(e->st));
    const int kt = (int)(1 * clean_thr * counts[0]);      // int() truncation as in :1069-1070
    const int kb = (int)(1 * noise_thr * counts[1]);
    ARGCHK(kt <= cap && kb <= cap, "selection exceeds cap");
    if (kt == 0 && kb == 0) return FM_OK;
    if (e->sel_cap < std::max(kt, kb)) {
        e->sel_cap = std::max(std::max(kt, kb), 1024);
        DALLOC(e->sel_top, e->sel_cap);
        DALLOC(e->sel_bot, e->sel_cap);
    }
    k_rank_select(sim_dev, N, kt, kb, e->sel_top, e->sel_bot, e->st);
    if (kt) HIPCHK(hipMemcpyAsync(top_host, e->sel_top, (size_t)kt * 4, hipMemcpyDeviceToHost, e->st));
    if (kb) HIPCHK(hipMemcpyAsync(bot_host, e->sel_bot, (size_t)kb * 4, hipMemcpyDeviceToHost, e->st));
    HIPCHK(hipStreamSynchronize(e->st));
    *n_top = kt; *n_bot = kb;
    return FM_OK;
}

int fm_select_topk_rows(fm_engine* e, const float* sim_dev, int64_t N, int32_t n_cls, const int32_t* pool_rows_host,
                        const int32_t* pool_n_host, int32_t stride, double clean_thr, double noise_thr, int32_t cap,
                        int32_t* top_host, int32_t* n_top, int32_t* bot_host, int32_t* n_bot)
{
    ARGCHK(e && pool_n_host && top_host && n_top && bot_host && n_bot, "null");
    ARGCHK(n_cls >= 0 && n_cls <= FM_MAX_CLASSES && cap >= 1, "n_cls / cap");
    for (int k = 0; k < n_cls; ++k) n_top[k] = n_bot[k] = 0;
    if (n_cls == 0 || N == 0) return FM_OK;
    ARGCHK(sim_dev, "null sim_dev");
    int maxn = 0;
    for (int k = 0; k < n_cls; ++k) {
        ARGCHK(pool_n_host[k] >= 0 && pool_n_host[k] <= (pool_rows_host ? stride : N), "pool size");
        maxn = std::max(maxn, pool_n_host[k]);
        // a row outside [0, N) would be an out-of-bounds read of sim in the counting / ranking kernels
        if (pool_rows_host)
            for (int i = 0; i < pool_n_host[k]; ++i) {
                const int32_t r = pool_rows_host[(size_t)k * stride + i];
                ARGCHK(r >= 0 && (int64_t)r < N, "pool row outside [0, N)");
            }
    }
    const size_t rows_ints = pool_rows_host ? (size_t)n_cls * stride : 0;
    const size_t need = (size_t)n_cls * (3 + 2 * (size_t)cap) + rows_ints;
    if (e->tag_buf_ints < need) {
        if (e->tag_buf) {               // the outgrown buffer goes back now, not at fm_destroy (the stream may still read it)
            HIPCHK(hipStreamSynchronize(e->st));
            auto it = std::find(e->allocs.begin(), e->allocs.end(), (void*)e->tag_buf);
            if (it != e->allocs.end()) e->allocs.erase(it);
            (void)hipFree(e->tag_buf);
            e->tag_buf = nullptr; e->tag_buf_ints = 0;
        }
        const size_t want = need + need / 2;
        DALLOC(e->tag_buf, want);
        e->tag_buf_ints = want;
    }
    int* d_pn = e->tag_buf;
    int* d_counts = d_pn + n_cls;
    int* d_top = d_counts + 2 * n_cls;
    int* d_bot = d_top + (size_t)n_cls * cap;
    int* d_rows = pool_rows_host ? d_bot + (size_t)n_cls * cap : nullptr;
    HIPCHK(hipMemcpyAsync(d_pn, pool_n_host, (size_t)n_cls * 4, hipMemcpyHostToDevice, e->st));
    if (d_rows) HIPCHK(hipMemcpyAsync(d_rows, pool_rows_host, rows_ints * 4, hipMemcpyHostToDevice, e->st));
    k_select_rows(sim_dev, N, n_cls, d_rows, d_pn, stride, maxn, clean_thr, noise_thr, cap, d_counts, d_top, d_bot, e->st);
    // ONE device-to-host read: the counts and both pick tables
    std::vector<int> h((size_t)n_cls * (2 + 2 * (size_t)cap));
    HIPCHK(hipMemcpyAsync(h.data(), d_counts, h.size() * 4, hipMemcpyDeviceToHost, e->st));
    HIPCHK(hipStreamSynchronize(e->st));
    for (int k = 0; k < n_cls; ++k) {
        const int kt = (int)(1 * clean_thr * h[2 * k]), kb = (int)(1 * noise_thr * h[2 * k + 1]);     // int() truncation as in :1069-1070
        ARGCHK(kt <= cap && kb <= cap, "selection exceeds cap");
        n_top[k] = kt; n_bot[k] = kb;
        memcpy(top_host + (size_t)k * cap, h.data() + 2 * n_cls + (size_t)k * cap, (size_t)kt * 4);
        memcpy(bot_host + (size_t)k * cap, h.data() + 2 * n_cls + (size_t)n_cls * cap + (size_t)k * cap, (size_t)kb * 4);
    }
    return FM_OK;
}

int fm_augment(fm_engine* e, const uint8_t* cache_dev, const int32_t* idx_dev, const int32_t* params_dev, int32_t B,
               const float* mean_host, const float* std_host, float* out_dev)
{
    ARGCHK(e && cache_dev && idx_dev && params_dev && mean_host && std_host && out_dev && B >= 1, "null");
    k_augment(cache_dev, idx_dev, params_dev, out_dev, B, e->H, e->W, mean_host[0], mean_host[1], mean_host[2],
              std_host[0], std_host[1], std_host[2], e->st);
    return FM_OK;
}

int fm_forward_train(fm_engine* e, const float* x1_dev, const float* x2_dev, int32_t B, float* feat_dev,
                     float* logits_dev)
{
    ARGCHK(e && x1_dev, "null");
    const int views = x2_dev ? 2 : 1;
    ARGCHK(B >= 1 && views * B <= e->maxB, "views*B exceeds max_images");
    const float* xs[2] = {x1_dev, x2_dev};
    to_nhwc4(e, xs, views, B);
    net_forward_train(e, views, B);
    if (feat_dev)
        HIPCHK(hipMemcpyAsync(feat_dev, e->feat, (size_t)views * B * e->D * 4, hipMemcpyDeviceToDevice, e->st));
    if (logits_dev)
        HIPCHK(hipMemcpyAsync(logits_dev, e->logits, (size_t)views * B * e->C * 4, hipMemcpyDeviceToDevice, e->st));
    e->pending_views = views; e->pending_B = B;
    STEP_DONE(e);
    return FM_OK;
}

int fm_backward_step(fm_engine* e, const float* dlogits_dev)
{
    ARGCHK(e && dlogits_dev, "null");
    ARGCHK(e->pending_views > 0, "fm_backward_step without a preceding fm_forward_train");
    const int views = e->pending_views, B = e->pending_B;
    HIPCHK(hipMemcpyAsync(e->dlogits, dlogits_dev, (size_t)views * B * e->C * 4, hipMemcpyDeviceToDevice, e->st));
    net_backward_and_step(e, views, B);
    e->pending_views = 0;
    STEP_DONE(e);
    return FM_OK;
}

int fm_teacher_axpby(fm_engine* e, float w_teacher, float w_student)
{
    ARGCHK(e, "null engine");
    k_axpby(e->tstate, e->state, w_teacher, w_student, (int64_t)e->NS, e->st);
    // int64 num_batches_tracked: the float result is truncated when it is loaded back (like FedAvg, Q7)
    for (size_t i = 0; i < e->counters.size(); ++i)
        e->tcounters[i] = (int64_t)(w_teacher * (float)e->tcounters[i] + w_student * (float)e->counters[i]);
    e->tev_dirty = true;
    e->twb_dirty = true;
    return FM_OK;
}

int fm_teacher_swap(fm_engine* e)
{
    ARGCHK(e, "null engine");
    std::swap(e->state, e->tstate);
    std::swap(e->ev_scale, e->tev_scale);
    std::swap(e->ev_shift, e->tev_shift);
    std::swap(e->ev_dirty, e->tev_dirty);
    std::swap(e->counters, e->tcounters);
    e->wpack_dirty = true;
    e->twb_dirty = true;
    return FM_OK;
}

int fm_set_stochastic(fm_engine* e, const float* drop_connect_dev, const float* dropout_dev)
{
    ARGCHK(e, "null engine");
    e->dc_dev = drop_connect_dev;
    e->drop_dev = dropout_dev;
    return FM_OK;
}

int fm_feature_dim(fm_engine* e) { return e ? e->D : 0; }

int fm_profile_enable(fm_engine* e, int32_t on)
{
    ARGCHK(e, "null engine");
    e->prof = on != 0;
    return FM_OK;
}

int fm_profile_read(fm_engine* e, int32_t family, int64_t* launches, double* ms, double* flops)
{
    ARGCHK(e && family >= 0 && family < FM_PROFILE_FAMILIES, "family");
    HIPCHK(hipStreamSynchronize(e->st));
    if (e->prof_fail) { e->prof_fail = false; g_err = "a profiling event could not be created/recorded"; return FM_ERR_HIP; }
    for (auto& p : e->evs) {
        float t = 0.f;
        if (hipEventElapsedTime(&t, p.a, p.b) == hipSuccess) {
            e->prof_ms[p.family] += t; e->prof_flops[p.family] += p.flops; e->prof_n[p.family] += 1;
        }
        e->ev_free.push_back(p.a); e->ev_free.push_back(p.b);
    }
    e->evs.clear();
    if (launches) *launches = e->prof_n[family];
    if (ms) *ms = e->prof_ms[family];
    if (flops) *flops = e->prof_flops[family];
    e->prof_n[family] = 0; e->prof_ms[family] = 0; e->prof_flops[family] = 0;
    return FM_OK;
}

int fm_debug_pw(fm_engine* e, int32_t op, int32_t conv, const void* x_dev, const void* dy_dev, void* out_dev,
                int32_t imgs, int32_t groups, const float* psc_dev, const float* psh_dev, const float* gate_dev,
                float* stats_dev)
{
    ARGCHK(e && out_dev && conv >= 0 && conv < (int)e->convs.size(), "conv index");
    ARGCHK(e->precision == 1 && e->convs[conv].k == 1, "bf16 engine and a 1x1 convolution");
    ARGCHK(imgs >= 1 && imgs <= e->maxB && groups >= 1 && imgs % groups == 0, "imgs/groups");
    Conv& c = e->convs[conv];
    ensure_packed(e);
    const Prologue pro{psc_dev, psh_dev, gate_dev};
    if (op == 0) {
        ARGCHK(x_dev, "x");
        conv_fwd(e, conv, e->state, reinterpret_cast<const float*>(x_dev), reinterpret_cast<float*>(out_dev), imgs, groups,
                 nullptr, nullptr, nullptr, 0, stats_dev ? e->ws_stats : nullptr, gate_dev ? &pro : nullptr);
        if (stats_dev) {
            const int tiles = stats_tiles(e, conv, imgs / groups, groups);
            std::vector<float> h((size_t)groups * tiles * 2 * c.cout_p), o((size_t)groups * 2 * c.cout_p, 0.f);
            HIPCHK(hipMemcpyAsync(h.data(), e->ws_stats, h.size() * 4, hipMemcpyDeviceToHost, e->st));
            HIPCHK(hipStreamSynchronize(e->st));
            for (int g = 0; g < groups; ++g)
                for (int k = 0; k < 2 * c.cout_p; ++k) {
                    double sum = 0;
                    for (int t = 0; t < tiles; ++t) sum += h[((size_t)g * tiles + t) * 2 * c.cout_p + k];
                    o[(size_t)g * 2 * c.cout_p + k] = (float)sum;
                }
            HIPCHK(hipMemcpy(stats_dev, o.data(), o.size() * 4, hipMemcpyHostToDevice));
        }
    } else if (op == 1) {
        ARGCHK(dy_dev, "dy");
        conv_dgrad(e, conv, e->state, reinterpret_cast<const float*>(dy_dev), reinterpret_cast<float*>(out_dev), imgs,
                   reinterpret_cast<const float*>(x_dev), false);      // x_dev = optional residual [npix][cin_p] bf16
    } else if (op == 2) {
        ARGCHK(x_dev && dy_dev, "x/dy");
        conv_wgrad(e, conv, reinterpret_cast<const float*>(x_dev), reinterpret_cast<const float*>(dy_dev), imgs,
                   gate_dev ? &pro : nullptr, (imgs / groups) * c.hout * c.wout);
        HIPCHK(hipMemcpyAsync(out_dev, e->grad + c.w_off, c.w_numel * 4, hipMemcpyDeviceToDevice, e->st));
    } else {
        ARGCHK(false, "op");
    }
    HIPCHK(hipGetLastError());
    return FM_OK;
}

int fm_debug_proj_bwd(fm_engine* e, int32_t conv, int32_t phase, const void* dyp_dev, const void* yd_dev, const float* bn_dev,
                      const float* gate_dev, const float* ds_dev, int32_t imgs, int32_t groups, void* out_dev, float* pool5_dev)
{
    ARGCHK(e && out_dev && conv >= 0 && conv < (int)e->convs.size(), "conv index");
    ARGCHK(e->model == 1 && e->convs[conv].k == 1, "an EfficientNet engine and a 1x1 convolution");
    ARGCHK(imgs >= 1 && imgs <= e->maxB && groups >= 1 && imgs % groups == 0, "imgs/groups");
    ARGCHK(dyp_dev && yd_dev && bn_dev && gate_dev && (phase == 0 ? pool5_dev != nullptr : ds_dev != nullptr), "operands");
    Conv& c = e->convs[conv];
    ensure_packed(e);
    const int L = c.cin_p, HW = c.hout * c.wout;
    const int nch = e->precision ? pw_proj_bwd_nch(L, c.cout_p, imgs, HW) : pw_proj_bwd_f32_nch(L, c.cout_p, imgs, HW);
    ARGCHK(nch > 0, "shape not handled by the fused project backward");
    const size_t gl = (size_t)groups * L;
    int sk;
    if (e->precision) {
        PwProjBwdParams q{};
        q.dYp = reinterpret_cast<const bf16*>(dyp_dev); q.Yd = reinterpret_cast<const bf16*>(yd_dev);
        q.Wt = e->wb + c.wbt_off; q.dYd = reinterpret_cast<bf16*>(out_dev);
        q.slab = e->ws_slab; q.pool5 = e->se_pool;
        q.sc = bn_dev; q.sh = bn_dev + gl; q.mean = bn_dev + 2 * gl; q.istd = bn_dev + 3 * gl;
        q.ca = bn_dev + 4 * gl; q.cb = bn_dev + 5 * gl; q.cc = bn_dev + 6 * gl;
        q.gate = gate_dev; q.ds = ds_dev;
        q.L = L; q.S = c.cout_p; q.imgs = imgs; q.HW = HW; q.ipg = imgs / groups; q.nch = nch;
        sk = launch_pw_proj_bwd(q, phase, e->slab_floats, e->st);
    } else {
        PwProjBwdF32Params q{};
        q.dYp = reinterpret_cast<const float*>(dyp_dev); q.Yd = reinterpret_cast<const float*>(yd_dev);
        q.W = e->state + c.w_off; q.dYd = reinterpret_cast<float*>(out_dev);
        q.slab = e->ws_slab; q.pool5 = e->se_pool;
        q.sc = bn_dev; q.sh = bn_dev + gl; q.mean = bn_dev + 2 * gl; q.istd = bn_dev + 3 * gl;
        q.ca = bn_dev + 4 * gl; q.cb = bn_dev + 5 * gl; q.cc = bn_dev + 6 * gl;
        q.gate = gate_dev; q.ds = ds_dev;
        q.L = L; q.S = c.cout_p; q.imgs = imgs; q.HW = HW; q.ipg = imgs / groups; q.nch = nch;
        sk = launch_pw_proj_bwd_f32(q, phase, e->slab_floats, e->st);
    }
    ARGCHK(sk > 0, "launch refused");
    if (phase == 0) {
        k_reduce_slabs(e->ws_slab, reinterpret_cast<float*>(out_dev), sk, (int64_t)c.w_numel, e->st);
        std::vector<float> h((size_t)imgs * nch * 5 * L), o((size_t)imgs * 5 * L);
        HIPCHK(hipMemcpyAsync(h.data(), e->se_pool, h.size() * 4, hipMemcpyDeviceToHost, e->st));
        HIPCHK(hipStreamSynchronize(e->st));
        for (int i = 0; i < imgs; ++i)
            for (int k = 0; k < 5 * L; ++k) {
                double sum = 0;
                for (int t = 0; t < nch; ++t) sum += h[((size_t)i * nch + t) * 5 * L + k];
                o[(size_t)i * 5 * L + k] = (float)sum;
            }
        HIPCHK(hipMemcpy(pool5_dev, o.data(), o.size() * 4, hipMemcpyHostToDevice));
    }
    HIPCHK(hipGetLastError());
    return FM_OK;
}

int fm_debug_exp_bwd(fm_engine* e, int32_t conv, const void* da_dev, const void* ye_dev, const void* x_dev, const void* res_dev,
                     const float* bn_dev, int32_t imgs, int32_t groups, void* dx_dev, float* dw_dev)
{
    ARGCHK(e && da_dev && ye_dev && x_dev && bn_dev && dx_dev && dw_dev && conv >= 0 && conv < (int)e->convs.size(), "operands");
    ARGCHK(e->model == 1 && e->convs[conv].k == 1, "an EfficientNet engine and a 1x1 convolution");
    ARGCHK(imgs >= 1 && imgs <= e->maxB && groups >= 1 && groups <= 2 && imgs % groups == 0, "imgs/groups");
    Conv& c = e->convs[conv];
    ensure_packed(e);
    const int L = c.cout_p, S = c.cin_p, HW = c.hout * c.wout;
    const size_t gl = (size_t)groups * L;
    int sk;
    if (e->precision) {
        PwExpBwdParams q{};
        q.dA = reinterpret_cast<const bf16*>(da_dev); q.Ye = reinterpret_cast<const bf16*>(ye_dev);
        q.X = reinterpret_cast<const bf16*>(x_dev); q.Wt = e->wb + c.wbt_off; q.res = reinterpret_cast<const bf16*>(res_dev);
        q.dX = reinterpret_cast<bf16*>(dx_dev); q.slab = e->ws_slab;
        q.ca = bn_dev; q.cb = bn_dev + gl; q.cc = bn_dev + 2 * gl; q.sc = bn_dev + 3 * gl; q.sh = bn_dev + 4 * gl;
        q.L = L; q.S = S; q.npix = imgs * HW; q.pix_per_group = (imgs / groups) * HW; q.groups = groups;
        sk = launch_pw_exp_bwd(q, e->slab_floats, e->st);
    } else {
        PwExpBwdF32Params q{};
        q.dA = reinterpret_cast<const float*>(da_dev); q.Ye = reinterpret_cast<const float*>(ye_dev);
        q.X = reinterpret_cast<const float*>(x_dev); q.W = e->state + c.w_off; q.res = reinterpret_cast<const float*>(res_dev);
        q.dX = reinterpret_cast<float*>(dx_dev); q.slab = e->ws_slab;
        q.ca = bn_dev; q.cb = bn_dev + gl; q.cc = bn_dev + 2 * gl; q.sc = bn_dev + 3 * gl; q.sh = bn_dev + 4 * gl;
        q.L = L; q.S = S; q.npix = imgs * HW; q.pix_per_group = (imgs / groups) * HW; q.groups = groups;
        sk = launch_pw_exp_bwd_f32(q, e->slab_floats, e->st);
    }
    ARGCHK(sk > 0, "shape not handled by the fused expand backward");
    k_reduce_slabs(e->ws_slab, dw_dev, sk, (int64_t)c.w_numel, e->st);
    HIPCHK(hipGetLastError());
    return FM_OK;
}

int fm_debug_get_grads(fm_engine* e, float* host_f32)
{
    ARGCHK(e && host_f32, "null");
    size_t off = 0;
    for (auto& en : e->entries) {
        if (en.kind == 0) {
            k_ohwi_to_oihw(e->grad + en.eng_off, e->stage_sd + off, en.O, en.I, en.KH, en.KW, en.Wpad, en.Ipad, e->st, en.Ostride);
            off += en.n;
        } else if (en.kind == 1) {
            if (en.eng_off < e->NP)
                HIPCHK(hipMemcpyAsync(e->stage_sd + off, e->grad + en.eng_off, en.n * 4, hipMemcpyDeviceToDevice, e->st));
            else
                HIPCHK(hipMemsetAsync(e->stage_sd + off, 0, en.n * 4, e->st));
            off += en.n;
        }
    }
    HIPCHK(hipMemcpyAsync(host_f32, e->stage_sd, (size_t)e->nf_sd * 4, hipMemcpyDeviceToHost, e->st));
    HIPCHK(hipStreamSynchronize(e->st));
    return FM_OK;
}

int fm_profile_ops(fm_engine* e, int32_t enable, char* buf, int32_t cap)
{
    ARGCHK(e, "null engine");
    HIPCHK(hipStreamSynchronize(e->st));
    for (auto& p : e->opevs) {
        float t = 0.f;
        if (hipEventElapsedTime(&t, p.a, p.b) == hipSuccess) { e->op_ms[p.id] += t; e->op_n[p.id] += 1; }
        e->ev_free.push_back(p.a); e->ev_free.push_back(p.b);
    }
    e->opevs.clear();
    if (buf && cap > 0) {
        std::string out;
        char line[160];
        for (size_t i = 0; i < e->op_names.size(); ++i) {
            snprintf(line, sizeof line, "%s\t%lld\t%.6f\n", e->op_names[i].c_str(), (long long)e->op_n[i], e->op_ms[i]);
            out += line;
        }
        snprintf(buf, (size_t)cap, "%s", out.c_str());
        e->op_names.clear(); e->op_ms.clear(); e->op_n.clear();
    }
    e->oprof = enable != 0;
    return FM_OK;
}

int fm_debug_num_convs(fm_engine* e) { return e ? (int)e->convs.size() : 0; }

int fm_debug_activation(fm_engine* e, int32_t kind, int32_t block, int32_t imgs, float* host_nhwc, int32_t* dims4)
{
    ARGCHK(e && dims4 && e->model == 0, "ResNet-18 engine only");
    ARGCHK(block >= 0 && block < (int)e->blocks.size() && (kind == 0 || kind == 1), "kind/block");
    ARGCHK(imgs >= 1 && imgs <= e->maxB, "imgs");
    const Block& b = e->blocks[block];
    const Conv& c = e->convs[b.c1];
    dims4[0] = imgs; dims4[1] = c.hout; dims4[2] = c.wout; dims4[3] = c.cout;
    if (host_nhwc) {
        // planes mode: z1 / out (all blocks but the last) live only as planes: re-form the fp32 values in the idle fp32 buffer
        if (e->planes && (kind == 0 || block + 1 < (int)e->blocks.size()))
            k_planes_to_f32(kind == 0 ? b.z1p : b.outp, kind == 0 ? b.z1 : b.out, (long long)imgs * c.hout * c.wout, c.cout, e->st);
        HIPCHK(hipMemcpyAsync(host_nhwc, kind == 0 ? b.z1 : b.out, (size_t)imgs * c.hout * c.wout * c.cout * 4,
                              hipMemcpyDeviceToHost, e->st));
        HIPCHK(hipStreamSynchronize(e->st));
    }
    return FM_OK;
}

int fm_debug_lose_part(int32_t on)
{
    pconv_debug_lose_part(on);
    return FM_OK;
}

int fm_debug_stem_masks(fm_engine* e, int32_t imgs, int32_t groups, uint8_t* relu_bits_host, uint8_t* argmax_host)
{
    ARGCHK(e && e->model == 0 && !e->precision, "ResNet-18 engine only");
    ARGCHK(imgs >= 1 && imgs <= e->maxB && groups >= 1 && imgs % groups == 0, "imgs / groups");
    const Conv& c0 = e->convs[0];
    const int64_t pix = (int64_t)imgs * c0.hout * c0.wout;
    if (relu_bits_host) {
        uint8_t* bits = nullptr;
        HIPCHK(hipMalloc(&bits, (size_t)pix * 8));
        k_stem_relu_bits(c0.y, e->bns[0].scale, e->bns[0].shift, bits, groups, pix / groups, 64, e->st);
        hipError_t rc = hipMemcpyAsync(relu_bits_host, bits, (size_t)pix * 8, hipMemcpyDeviceToHost, e->st);
        if (rc == hipSuccess) rc = hipStreamSynchronize(e->st);
        (void)hipFree(bits);
        HIPCHK(rc);
    }
    if (argmax_host) {
        HIPCHK(hipMemcpyAsync(argmax_host, e->idx0, (size_t)imgs * (c0.hout / 2) * (c0.wout / 2) * 64, hipMemcpyDeviceToHost, e->st));
        HIPCHK(hipStreamSynchronize(e->st));
    }
    return FM_OK;
}

int fm_debug_conv_info(fm_engine* e, int32_t conv, int32_t* info16)
{
    ARGCHK(e && info16 && conv >= 0 && conv < (int)e->convs.size(), "conv index");
    const Conv& c = e->convs[conv];
    const int v[16] = {c.cin, c.cout, c.k, c.stride, c.pad, c.hin, c.win, c.hout, c.wout, c.cin_p, c.Kw, c.kw_p,
                       c.cout_p, 0, 0, 0};
    memcpy(info16, v, sizeof v);
    return FM_OK;
}

int fm_debug_conv(fm_engine* e, int32_t op, int32_t conv, const float* x_dev, const float* dy_dev, float* out_dev,
                  int32_t imgs, int32_t groups, float* stats_dev)
{
    ARGCHK(e && out_dev && conv >= 0 && conv < (int)e->convs.size(), "conv index");
    ARGCHK(!e->precision, "fm_debug_conv works on fp32 tensors: create the engine with precision 0");
    ARGCHK(imgs >= 1 && imgs <= e->maxB && groups >= 1 && imgs % groups == 0, "imgs/groups");
    Conv& c = e->convs[conv];
    if (c.stem3 && x_dev)       // the packed stem reads the framed copy of its [imgs][H][W][3] input
        k_frame_nhwc3(x_dev, e->x3, imgs, c.hin, c.win, c.Hp, c.Wp, 3, 3, 1, e->st, e->stem_rows ? e->x3p : nullptr, e->x3p_plane_elems);
    ensure_packed(e);           // the forward reads the weight planes, the data gradient the transposed packs
    if (op == 0) {
        ARGCHK(x_dev, "x");
        conv_fwd(e, conv, e->state, x_dev, out_dev, imgs, groups, nullptr, nullptr, nullptr, 0,
                 stats_dev ? e->ws_stats : nullptr);
        if (stats_dev) {
            // fold the per-tile partials with the finalize kernel's own reduction order: sum/sumsq only
            const int tiles = stats_tiles(e, conv, imgs / groups, groups);
            std::vector<float> h((size_t)groups * tiles * 2 * c.cout_p), o((size_t)groups * 2 * c.cout_p, 0.f);
            HIPCHK(hipMemcpyAsync(h.data(), e->ws_stats, h.size() * 4, hipMemcpyDeviceToHost, e->st));
            HIPCHK(hipStreamSynchronize(e->st));
            for (int g = 0; g < groups; ++g)
                for (int k = 0; k < 2 * c.cout_p; ++k) {
                    double s = 0;
                    for (int t = 0; t < tiles; ++t) s += h[((size_t)g * tiles + t) * 2 * c.cout_p + k];
                    o[(size_t)g * 2 * c.cout_p + k] = (float)s;
                }
            HIPCHK(hipMemcpy(stats_dev, o.data(), o.size() * 4, hipMemcpyHostToDevice));
        }
    } else if (op == 1) {
        ARGCHK(dy_dev && c.ncls > 0, "dgrad unavailable for this conv");
        ensure_packed(e);
        conv_dgrad(e, conv, e->state, dy_dev, out_dev, imgs, nullptr, false);
    } else if (op == 2) {
        ARGCHK(x_dev && dy_dev, "x/dy");
        conv_wgrad(e, conv, x_dev, dy_dev, imgs);
        HIPCHK(hipMemcpyAsync(out_dev, e->grad + c.w_off, c.w_numel * 4, hipMemcpyDeviceToDevice, e->st));
    } else {
        ARGCHK(false, "op");
    }
    return FM_OK;
}

}  // extern "C"
