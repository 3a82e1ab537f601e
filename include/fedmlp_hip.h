/*
 * fedmlp_hip.h -- C ABI of the MI355X-native FedMLP per-client training engine.
 *
 * The reference (szbonaldo/FedMLP) has no FFI layer: its boundary for this path
 * is the Python call surface build_model() / LocalUpdate / FedAvg* (SURVEY.md
 * 8b).  This library is what a ctypes binding of that surface calls; the
 * reference-side stub is shown in INTEGRATION.md.  Plain pointers and sizes
 * only: no torch types, no C++ types, no exceptions across the boundary.
 *
 * Conventions
 *   - One opaque engine per GPU/process; not thread-safe per handle
 *     (the reference drives the model from a single thread, main.py:135).
 *   - Every function returns 0 on success, a negative code on failure;
 *     fm_last_error() returns a human-readable message for the calling thread.
 *   - "dev" pointers are HIP device pointers owned by the caller (the Python
 *     shim passes torch tensor data_ptr()s); "host" pointers are host memory.
 *   - Images are fp32 NCHW [B,3,H,W] exactly as the reference's DataLoader
 *     yields them (dataset/all_dataset.py:73-83); labels/masks fp32 [B,C].
 *   - All work is enqueued on the stream given at fm_create (0 = the null
 *     stream, which is also torch's default stream); nothing synchronises
 *     except fm_sync, fm_get_state, fm_set_state and the *_host getters.
 *   - The model state crosses the boundary in the reference's state_dict
 *     order (torchvision key order, conv weights OIHW): one flat fp32 buffer
 *     + one int64 buffer of num_batches_tracked counters (fedmlp_amd/spec.py).
 */
#ifndef FEDMLP_HIP_H
#define FEDMLP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FM_MAX_CLASSES 32

#define FM_OK 0
#define FM_ERR_ARG (-1)     /* bad argument / unsupported configuration */
#define FM_ERR_HIP (-2)     /* a HIP runtime call failed                */
#define FM_ERR_STATE (-3)   /* call sequence error                      */

typedef struct fm_engine fm_engine;

typedef struct fm_config {
    int32_t model;        /* 0 = ResNet-18 (model/all_models.py:53-54, 117-120);
                             1 = EfficientNet-B0 (model/all_models.py:73-75, 121-124) */
    int32_t n_classes;    /* args.n_classes, <= FM_MAX_CLASSES                       */
    int32_t in_h, in_w;   /* input spatial size (224 in the reference, >= 32)        */
    int32_t max_images;   /* max images in ONE forward call (views x batch; eval     */
                          /* passes use 4*batch_size, utils/local_training.py:977)   */
    int32_t reserved[3];  /* reserved[0] = activation precision: 0 fp32 (the reference's arithmetic,
                             utils/local_training.py:14 imports autocast and never uses it);
                             1 bf16 activation storage + bf16 MFMA for the 1x1 convolutions, fp32
                             accumulation / BN statistics / master weights / Adam -- EfficientNet-B0
                             only (BASELINE configs[4]).
                             reserved[1] = stream mode: 0 (default) the frozen teacher's forward and the
                             backward's weight gradients are enqueued on an engine-owned side stream that is
                             forked from and joined to `stream` inside every call (same bits as one stream);
                             1 = everything on `stream` (use for per-kernel profiling); 2 = teacher forward on
                             the side stream only.
                             reserved[2] = product form of the fp32 convolution GEMMs, fixed for the handle's life:
                             0 (default) the library default = every fp32 product as six exact bf16 partial products
                             on the bf16 matrix pipe, fp32 accumulation (csrc/split3.h); 1 = products on the fp32
                             matrix pipe; 2 = all nine partial products; 3 = exactly six (what 0 resolves to in the
                             shipped library, but not overridden by the test-only FM_MFMA_SPLIT).  fm_products() reports it. */
    void*   stream;       /* hipStream_t; NULL = null stream                         */
} fm_config;

/* Adam hyper-parameters of utils/local_training.py:636-637 (torch.optim.Adam,
 * coupled L2 weight decay). */
typedef struct fm_adam {
    float lr, beta1, beta2, eps, weight_decay;
} fm_adam;

const char* fm_last_error(void);
const char* fm_version(void);

/* ---- lifetime ---------------------------------------------------------- */
int fm_create(const fm_config* cfg, fm_engine** out);
int fm_destroy(fm_engine* e);
int fm_sync(fm_engine* e);

/* ---- state: net.state_dict() / load_state_dict()  (main.py:196, 222) ----- */
/* Sizes of the state_dict-order buffers (fp32 elements, int64 counters). */
int fm_state_sizes(fm_engine* e, int64_t* n_f32, int64_t* n_i64);
int fm_set_state(fm_engine* e, const float* host_f32, const int64_t* host_i64);
int fm_get_state(fm_engine* e, float* host_f32, int64_t* host_i64);
/* The engine-layout device buffer holding every fp32 state entry (parameters
 * then BN running statistics).  FedAvg (utils/FedAvg.py:7-14) is element-wise,
 * so the RCCL all-reduce of main.py:218's aggregation runs on this buffer in
 * place; *numel is its length.  Valid until fm_destroy. */
int fm_state_device(fm_engine* e, float** dev_ptr, int64_t* numel);
/* num_batches_tracked counters (state_dict order): read (set == 0) or overwrite.
 * They never enter the arithmetic (momentum is fixed at 0.1) but FedAvg
 * averages them like every other entry (utils/FedAvg.py:9-13). */
int fm_counters(fm_engine* e, int64_t* host_i64, int32_t set);
/* state *= w  (the n_i / sum(n) pre-scale before the all-reduce SUM). */
int fm_state_scale(fm_engine* e, float w);
/* FedAvg (utils/FedAvg.py:7-14) of K client states that share ONE GPU (main.py:135-218 trains its clients in turn and
 * aggregates their state_dicts in one process): out[j] = ((s_0[j]*n_0 + s_1[j]*n_1) + ...) / sum(n) in the reference's
 * left-to-right order with its roundings (separate fp32 product and sum, IEEE division), so fp32 entries are bit-identical
 * to utils/FedAvg.py run on CPU tensors (what the KAT pins; on CUDA tensors torch divides by multiplying with the reciprocal,
 * which can differ in the last bit).  states_dev: HOST array of K device pointers to engine-layout states (fm_state_device()'s layout and
 * length, e.g. copies of it taken after each client's round); n_host: the K sample counts (dict_len); K <= 16; out_dev may
 * be fm_state_device()'s buffer itself or any of the inputs.  The num_batches_tracked counters stay with the caller
 * (fm_counters). */
int fm_fedavg_fold(fm_engine* e, const float* const* states_dev, const float* n_host, int32_t K, float* out_dev);
/* ---- FedAvg across ranks as RCCL calls inside the library (one client per GPU) --------------
 * utils/FedAvg.py:7-14 (FedAvg), :51-70 (FedAvg_tao), :72-93 (FedAvg_proto) walk a Python list of
 * client results in one process.  With one process per GPU the same weighted sums are
 * ncclAllReduce(SUM) calls over xGMI, enqueued on the engine's stream.  RCCL is resolved at run
 * time (dlopen); the communicator is the library's own, so a C caller needs no Python:
 *   rank 0: fm_comm_unique_id(id)  -> ship the 128 bytes to every rank (any transport)
 *   all   : fm_comm_init(e, id, rank, world)
 * fm_comm_size returns the rank count (0 = no communicator).  Without a communicator (or with a
 * world of 1) the fm_fedavg_* calls compute the single-client result locally. */
#define FM_COMM_ID_BYTES 128
/* Local, non-collective: load librccl and resolve its entry points.  fm_comm_init is a collective (ncclCommInitRank
 * blocks until every rank has entered it), so call this on every rank first and agree on the result over the
 * transport that ships the id: a rank whose RCCL cannot be loaded must not leave its peers waiting inside the init. */
int fm_comm_preflight(void);
int fm_comm_unique_id(uint8_t* id128);
int fm_comm_init(fm_engine* e, const uint8_t* id128, int32_t rank, int32_t world);
int fm_comm_destroy(fm_engine* e);
int fm_comm_size(fm_engine* e);
/* FedAvg: state <- sum_ranks w_rank * state_rank, w_rank = n_rank / sum(n): the pre-scale kernel,
 * ONE in-place all-reduce of the whole fp32 state arena (44.8 MB for ResNet-18) and the float64
 * weighted mean of the num_batches_tracked counters (truncated on load like utils/FedAvg.py:13). */
int fm_fedavg_allreduce(fm_engine* e, float w);
/* FedAvg_tao over ranks: out[c] = sum_r t_r[c] n_r m_r[c] / sum_r n_r m_r[c] with m_r = this
 * rank's negative_mask (1 = class c is missing on this client; a rank that stands for several folded
 * clients passes n_i = 1 and their summed sample counts as the mask); 1.0 where no rank has it (:66-67). */
int fm_fedavg_tao(fm_engine* e, const double* t_host, double n_i, const float* negative_mask_host,
                  double* out_host);
/* FedAvg_proto over ranks: rows 2c, 2c+1 averaged over the ranks whose active_mask[c] = 1,
 * NaN rows where no rank annotates c (0/0 like :85-86).  proto/out: host [2C*D] fp32. */
int fm_fedavg_proto(fm_engine* e, const float* proto_host, double n_i, const float* active_mask_host,
                    float* out_host);

/* glob_model = deepcopy(net) at round start (utils/local_training.py:909, 1017):
 * snapshot the current state as the frozen eval-mode teacher. */
int fm_teacher_snapshot(fm_engine* e);
/* A fresh torch.optim.Adam every round (utils/local_training.py:912-913):
 * zero both moments and the step count. */
int fm_adam_reset(fm_engine* e, const fm_adam* hp);

/* ---- net(x) in eval mode: -> (feature[B,D], logits[B,C]) ----------------- */
/* utils/local_training.py:983, 1030, 1227 (student) / :944-947 (teacher).
 * use_teacher != 0 runs the snapshot taken by fm_teacher_snapshot. */
int fm_forward_eval(fm_engine* e, const float* x_dev, int32_t B, int32_t use_teacher,
                    float* feat_dev, float* logits_dev);

/* ---- training steps: forward + loss + backward + Adam.step --------------- */
/* Every step writes its scalar loss to *loss_dev (a device float the caller
 * reads back once per round instead of loss.item() every step) and bumps the
 * BN counters like a train-mode forward does.  bs_norm is args.batch_size: the
 * reference normalises by it, not by the actual batch length B. */

/* LocalUpdate.train (utils/local_training.py:628-703):
 * sum(BCEWithLogits(pos_weight)(z, y)) / (bs_norm * C). pos_weight: host[C]. */
int fm_step_bce(fm_engine* e, const float* x_dev, const float* y_dev, int32_t B,
                const float* pos_weight_host, int32_t bs_norm, float* loss_dev);

/* train_FedMLP stage 1 (utils/local_training.py:920-967): two views, frozen
 * teacher, BCE on the active classes + MSE-to-teacher on the missing classes.
 * active_mask: host[C] of 0/1 (1 = class annotated by this client). */
int fm_step_stage1(fm_engine* e, const float* x1_dev, const float* x2_dev, const float* y_dev,
                   int32_t B, const float* active_mask_host, int32_t annotation_num,
                   int32_t bs_norm, float* loss_dev);

/* train_FedMLP stage 2 (utils/local_training.py:1171-1192): masked BCE,
 * sup_cls = 1 - distill_cls (device [B,C]).  The reference's teacher forward
 * there is dead code and is not executed. */
int fm_step_stage2(fm_engine* e, const float* x_dev, const float* y_dev,
                   const float* distill_dev, int32_t B, float* loss_dev);

/* train_FixMatch (utils/local_training.py:789-818): weak/strong consistency.
 * pos_weight / pos_weight_unknown: host[C]. */
int fm_step_fixmatch(fm_engine* e, const float* xw_dev, const float* xs_dev, const float* y_dev,
                     int32_t B, const float* pos_weight_host, const float* pos_weight_unk_host,
                     const float* active_mask_host, int32_t annotation_num, int32_t bs_norm,
                     float* loss_dev);

/* ---- prototype + t pass (utils/local_training.py:971-1002, 1208-1250) ---- */
/* Accumulators live in the engine: proto sums [2C,D], counts [2C], t counts [C]. */
int fm_proto_reset(fm_engine* e);
/* One eval batch: feature[B,D], logits[B,C], labels[B,C] on device.
 * active_mask / negative_mask: host[C] 0/1 (classes for prototypes / for t). */
int fm_proto_accumulate(fm_engine* e, const float* feat_dev, const float* logits_dev,
                        const float* labels_dev, int32_t B, const float* active_mask_host,
                        const float* negative_mask_host, float L, float U);
/* proto rows /= counts (zero_guard: leave a row with count 0 as is, the stage-2
 * variant :1240-1248; otherwise 0/0 = NaN like :997-999), t = counts / n_local.
 * Outputs on host: proto[2C*D] fp32, t[C] f64. */
int fm_proto_finalize(fm_engine* e, int32_t zero_guard, int64_t n_local,
                      const float* active_mask_host, float* proto_host, double* t_host);

/* ---- cosine tagging (CosineSimilarityFast, utils/local_training.py:1417-1435,
 *      call site :1052-1057) ------------------------------------------------ */
/* sim[k][n] = cos(f[n], P[2c_k]) - cos(f[n], P[2c_k+1]) for the n_cls classes
 * listed in classes_host.  feat_dev [N,D], proto_dev [2C,D], sim_dev [n_cls,N]. */
int fm_cos_tag(fm_engine* e, const float* feat_dev, int64_t N, const float* proto_dev,
               const int32_t* classes_host, int32_t n_cls, float* sim_dev);
/* Stable top-/bottom-k positions of one similarity row (utils/utils.py:24-35 via
 * utils/local_training.py:1061-1075): n_clean = #(sim>=0), n_noise = #(sim<0),
 * k_top = (int)(clean_thr*n_clean), k_bot = (int)(noise_thr*n_noise); writes the
 * positions of the k_top largest (descending, first position wins ties) and the
 * k_bot smallest (ascending) values to host arrays of capacity cap. */
int fm_select_topk(fm_engine* e, const float* sim_dev, int64_t N, double clean_thr,
                   double noise_thr, int32_t cap, int32_t* top_host, int32_t* n_top,
                   int32_t* bot_host, int32_t* n_bot);
/* The same selection for EVERY missing class of a round (the loop of utils/local_training.py:1052-1112) in one launch pair
 * and one device-to-host read.  sim_dev [n_cls][N] (fm_cos_tag's output); class k's pool = the pool_n_host[k] positions
 * pool_rows_host[k*stride ..] of its similarity row, in pool order (find_indices_in_a, :901-902; ties go to the earlier pool
 * position), or all N rows when pool_rows_host is NULL.  Writes pool POSITIONS: top_host / bot_host [n_cls][cap],
 * n_top / n_bot [n_cls]. */
int fm_select_topk_rows(fm_engine* e, const float* sim_dev, int64_t N, int32_t n_cls, const int32_t* pool_rows_host,
                        const int32_t* pool_n_host, int32_t stride, double clean_thr, double noise_thr, int32_t cap,
                        int32_t* top_host, int32_t* n_top, int32_t* bot_host, int32_t* n_bot);

/* ---- HBM-resident input pipeline --------------------------- */
/* Train transform of dataset/dataset.py:40-53 on a uint8 cache of already-resized images
 * [N,3,H,W] kept in HBM: RandomAffine(10 deg, translate 2 %) with NEAREST sampling and fill 0,
 * RandomHorizontalFlip, ToTensor (/255), Normalize(mean, std).  The random draws are the
 * caller's.  params_dev[b] = int32 {c0, c1, c2, c3, c4, c5, flip (0/1), unused}: the inverse affine
 * matrix (PIL AFFINE convention: source = M * (x, y, 1)) in the 16.16 fixed point Pillow's
 * nearest-neighbour affine uses (libImaging/Geometry.c affine_fixed):
 *   c0, c1, c3, c4 = FIX(a0, a1, a3, a4), c2 = FIX(a2 + a0/2 + a1/2), c5 = FIX(a5 + a3/2 + a4/2),
 *   FIX(v) = floor(v * 65536 + 0.5); source pixel = ((c2 + c0 x + c1 y) >> 16, (c5 + c3 x + c4 y) >> 16).
 * Integer arithmetic: the result is bit-exact with the PIL pipeline (tests/golden/augment_pil.npz).
 * idx_dev[b] = sample index in the cache.  out_dev: fp32 NCHW [B,3,H,W], i.e. what fm_step_* consume.
 * mean/std: host[3]. */
int fm_augment(fm_engine* e, const uint8_t* cache_dev, const int32_t* idx_dev, const int32_t* params_dev,
               int32_t B, const float* mean_host, const float* std_host, float* out_dev);

/* ---- generic train step (the other baselines of main.py's --exp switch) -----
 * train_RSCFed (utils/local_training.py:705-769), train_FedNoRo (:115-234) and train_CBAFed
 * (:236-342) differ from the steps above only in the loss head on the [B,C] logits.  The split
 * step lets the host mirror compute that head: fm_forward_train runs the train-mode forward of one
 * view (x2_dev NULL) or two views (BN statistics per view, logits view-major [views*B][C]);
 * fm_backward_step takes d(loss)/d(logits) of the same shape, runs the backward pass and
 * optimizer.step().  The pair must be called back to back. */
int fm_forward_train(fm_engine* e, const float* x1_dev, const float* x2_dev, int32_t B, float* feat_dev,
                     float* logits_dev);
int fm_backward_step(fm_engine* e, const float* dlogits_dev);
/* teacher <- w_teacher*teacher + w_student*student over every state entry (train_RSCFed's EMA,
 * utils/local_training.py:751-759, weights 0.999 / 0.001). */
int fm_teacher_axpby(fm_engine* e, float w_teacher, float w_student);
/* exchange the student and teacher slots (so fm_set_state / fm_get_state reach the teacher) */
int fm_teacher_swap(fm_engine* e);

/* ---- EfficientNet-B0 training-time randomness -----------------------------------
 * The reference's model draws drop-connect (MBConvBlock, p = 0.2*idx/16 per block, per sample)
 * and dropout (p = 0.2 before `_fc`) inside net(images) (utils/local_training.py:657, 937-947,
 * 1178 through efficientnet_pytorch 0.7.1).  Here the draws are the caller's: device arrays of
 * MULTIPLIERS, kept by pointer until replaced (NULL = identity, i.e. no drop).
 *   drop_connect_dev [16][imgs]   imgs = images in the next train step (2B for two-view steps,
 *                                 view-major), value floor(keep+U)/keep
 *   dropout_dev      [imgs][1280] value 0 or 1/(1-p)
 * Ignored by model 0. */
int fm_set_stochastic(fm_engine* e, const float* drop_connect_dev, const float* dropout_dev);
/* feature width of the model: 512 (ResNet-18) or 1280 (EfficientNet-B0) */
int fm_feature_dim(fm_engine* e);
/* The stream mode that was actually set up (fm_config.reserved[1] asks; fm_create falls back to one stream when the
 * second activation / workspace set would not fit in free device memory): 0 teacher + weight gradients on the side
 * stream, 2 teacher only, 1 one stream. */
int fm_stream_mode(fm_engine* e);
/* How the fp32 convolution GEMMs (ResNet-18's 3x3 / 1x1 convs, EfficientNet's wide 1x1 convs) form their products right
 * now: 0 = v_mfma_f32_16x16x4_f32; 9 / 6 = each fp32 product as 9 / 6 exact bf16 partial products on
 * v_mfma_f32_16x16x32_bf16, accumulated in fp32 (csrc/split3.h).  fm_products(e): the form of this handle, set at
 * fm_create from fm_config.reserved[2].  fm_mfma_products(): what reserved[2] = 0 resolves to (6 in the shipped library;
 * the test-only environment variable FM_MFMA_SPLIT = 0 | 6 | 9 overrides it and is read once per fm_create, never per
 * launch).  The reference has one arithmetic (utils/local_training.py:14 imports autocast and never uses it); cuDNN picks
 * its own algorithm behind nn.Conv2d (model/all_models.py:53-54). */
int fm_mfma_products(void);
int fm_products(fm_engine* e);
/* 1 when the handle's conv GEMMs read their operands as bf16 planes written by the producing kernels (ResNet-18 in a split
 * product form: csrc/pconv.hip, pwgrad.hip, planes_ew.hip; activations between convs then exist only as planes), 0 when they
 * read fp32 tensors (csrc/igemm.hip, wgrad.hip: EfficientNet-B0, the fp32 matrix pipe, FM_PLANES=0).  Same arithmetic either way. */
int fm_planes_mode(fm_engine* e);

/* ---- measurement hooks (bench.py roofline leg) ---------------------------- */
/* When enabled, HIP events bracket every convolution GEMM launch on the
 * engine's stream; fm_profile_read drains them (synchronises) and returns, per
 * kernel (family 0 = igemm_kernel<128,128,2,0,...>, 1 = igemm_kernel<64,192,4,0,...>,
 * 2 = igemm_kernel<64,256,4,2,...> (7x7 stem): conv forward + data gradient; 3 = wgrad_kernel<128,128,2,4>,
 * 4 = wgrad_kernel<64,192,4,3> (64-channel 3x3 layers) and the skinny 1x1 wgrad, 5 = the 7x7 stem's weight
 * gradient), the launch count, total milliseconds and algorithmic FLOPs since the last read.  Planes mode (fm_planes_mode):
 * 0 / 1 = pconv_kernel<4 | 2, 4, 2, SP, true> (3x3 stride-1 convs and their data gradients, M >= 128 | M = 64), 6 / 7 =
 * pconv_kernel<4 | 2, 4, 2, SP, false> (stride-2 / 1x1 convs, parity classes), 3 / 4 = pwgrad_kernel<4 | 2, ...>,
 * 8 = pwgrad_ring_kernel<SP> (weight gradients of the 3x3 stride-1 convs on the 56 x 56 and 28 x 28 maps), 2 = stem_rows_kernel<SP>
 * (the 7x7 stem forward at 112-pixel output rows; other input sizes: igemm_kernel<64,256,4,2,...>). */
#define FM_PROFILE_FAMILIES 9
int fm_profile_enable(fm_engine* e, int32_t on);
int fm_profile_read(fm_engine* e, int32_t family, int64_t* launches, double* ms, double* flops);

/* Per-op timing of the EfficientNet-B0 graph (tools/op_profile.py): while enabled, HIP events bracket every op
 * of the step; a call drains them and, when buf != NULL, writes "op@block<TAB>calls<TAB>total ms" lines and
 * resets the table.  Blocks: -1 stem, 0..15 train forward, 100 head, 201..216 / 300 eval forward,
 * 399..415 / 500 backward. */
int fm_profile_ops(fm_engine* e, int32_t enable, char* buf, int32_t cap);

/* Kernel-level test hooks (fm_debug_*: single-kernel entry points, saved activations, last gradients) are declared in
 * fedmlp_hip_debug.h; they are not part of the drop-in surface. */

#ifdef __cplusplus
}
#endif
#endif /* FEDMLP_HIP_H */
