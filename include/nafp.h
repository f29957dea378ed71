/*
 * nafp.h -- C ABI of libnafp.so: the MI355X (gfx950) hot path of neural audio
 * fingerprinting: 1-s segment -> log-mel -> conv encoder -> 128-d fingerprint
 * (generate), the contrastive train step around it (batch assembly + time-domain
 * augmentation, spec-augment, NT-Xent / online-triplet loss, the encoder's backward
 * pass, Adam / LAMB), and the exact search that consumes the fingerprints.
 *
 * The reference (mimbres/neural-audio-fp) is pure Python on TensorFlow; it has no
 * FFI of its own.  Each entry point below replaces the device work behind one
 * Python-level operator of the reference, cited as file:line of the reference
 * checkout.  INTEGRATION.md shows the ctypes stub a maintainer of the reference
 * would add to route those operators here.
 *
 * Conventions
 *   - every function returns an int status (NAFP_OK == 0); no C++ exception
 *     crosses this boundary;
 *   - all data pointers are CALLER-OWNED DEVICE pointers unless the name ends in
 *     `_host`; the library allocates only what hangs off an opaque handle
 *     (tables, packed weights) and frees it in the matching *_destroy;
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream);
 *     all work is enqueued asynchronously on it;
 *   - a handle is re-entrant across handles, not thread-safe per handle (the
 *     reference drives the device from one host thread: generate.py:176-181).
 */
#ifndef NAFP_H
#define NAFP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NAFP_ABI_VERSION 1

enum {
    NAFP_OK = 0,
    NAFP_ERR_INVALID_ARG = 1,
    NAFP_ERR_UNSUPPORTED = 2,   /* geometry the kernels are not built for */
    NAFP_ERR_HIP = 3,           /* a HIP runtime call failed; see nafp_last_hip_error */
    NAFP_ERR_WORKSPACE = 4,     /* caller workspace too small */
    NAFP_ERR_NO_WEIGHTS = 5     /* encoder forward before set_weights */
};

typedef struct nafp_melspec nafp_melspec;
typedef struct nafp_encoder nafp_encoder;

int nafp_abi_version(void);
const char* nafp_status_string(int status);
/* hipError_t (as int) of the most recent failing HIP call on this thread. */
int nafp_last_hip_error(void);
/* CRC-32C (Castagnoli) of n host bytes, continuing from `crc` (0 to start): the checksum TensorFlow's TensorBundle
 * stores per tensor and per index block; used by the reader of the reference's checkpoints (model/generate.py:26-52). */
uint32_t nafp_crc32c_host(const void* data_host, int64_t n, uint32_t crc);

/* ------------------------------------------------------------------------
 * Front end: Melspec_layer (model/fp/melspec/melspectrogram.py:10-112)
 * ---------------------------------------------------------------------- */

/* Host-side Slaney mel filter bank, float32, row-major (n_mels, n_fft/2+1).
 * Replaces librosa.filters.mel reached via kapre ApplyFilterbank
 * (melspectrogram.py:93-98).  No GPU needed. */
int nafp_mel_filterbank_host(int fs, int n_fft, int n_mels, float f_min, float f_max,
                             float* out_host);

/* Plan for get_melspec_layer(cfg) (melspectrogram.py:115-141).  seg_len = FS*DUR.
 * Supported: n_fft == 1024, hop == 256, n_mels % 64 == 0 && n_mels <= 256,
 * every mel filter <= 8 taps wide. */
int nafp_melspec_create(nafp_melspec** plan, int fs, int seg_len, int n_fft, int hop,
                        int n_mels, float f_min, float f_max);
int nafp_melspec_destroy(nafp_melspec* plan);
int nafp_melspec_n_frames(const nafp_melspec* plan);   /* 1 + seg_len/hop   (32)  */
int nafp_melspec_n_mels(const nafp_melspec* plan);

/* Melspec_layer.call (melspectrogram.py:102-112).
 *   audio      (n_seg, seg_len) float32 or int16 PCM (int16 is scaled by 2^-15 as
 *              load_audio does, model/utils/audio_utils.py:245-246)
 *   group_size the reference subtracts the max over the WHOLE device batch
 *              (melspectrogram.py:108); consecutive runs of group_size segments
 *              (last ragged) form one such batch.  <= 0 means one group.
 *   segment_norm  bit 0: FEAT == 'melspec_maxnorm' (melspectrogram.py:110-111);
 *              bit 1 (NAFP_MELSPEC_DEFER): stop before melspectrogram.py:108 -- feat is the raw log10 mel and
 *              group_stat the (max, min) per group; nafp_encoder_forward_raw finishes the layer inside its first
 *              conv (one kernel and one round trip of the feature tensor through HBM fewer; same result)
 *   feat       out, (n_seg, n_mels, n_frames) float32 == the reference's
 *              (B, n_mels, n_frames, 1)
 *   group_stat scratch, 2*ceil(n_seg/group_size) floats (raw max / min per group)
 */
#define NAFP_MELSPEC_DEFER 2
int nafp_melspec_forward_f32(nafp_melspec* plan, const float* audio, int64_t n_seg,
                             int group_size, int segment_norm, float* feat,
                             float* group_stat, void* stream);
int nafp_melspec_forward_i16(nafp_melspec* plan, const int16_t* audio, int64_t n_seg,
                             int group_size, int segment_norm, float* feat,
                             float* group_stat, void* stream);

/* Finish a deferred call in place: feat <- max(feat - group max, -80) [, segment normalisation]
 * (melspectrogram.py:108-111), for consumers other than nafp_encoder_forward_raw. */
int nafp_melspec_finish(nafp_melspec* plan, float* feat, const float* group_stat, int64_t n_seg, int group_size,
                        int segment_norm, void* stream);

/* The same layer fed by WINDOWS of one int16 PCM arena instead of a materialised (n_seg, seg_len)
 * array: segment i starts at sample seg_offset[i] of `pcm` and has seg_valid[i] (<= seg_len) real
 * samples, the rest being the zero tail.  Replaces the segment assembly of the reference's loader
 * (get_fns_seg_list + load_audio per segment, model/utils/audio_utils.py:140-264; np.vstack per
 * batch, model/utils/dataloader_keras.py:389-397): whole files are uploaded once and the 50 %
 * overlap between consecutive segments (HOP = 0.5 s) is never duplicated.  All device pointers. */
int nafp_melspec_forward_windows_i16(nafp_melspec* plan, const int16_t* pcm, const int64_t* seg_offset,
                                     const int32_t* seg_valid, int64_t n_seg, int group_size,
                                     int segment_norm, float* feat, float* group_stat, void* stream);

/* ------------------------------------------------------------------------
 * Encoder: FingerPrinter (model/fp/nnfp.py:159-231)
 * ---------------------------------------------------------------------- */

/* get_fingerprinter(cfg) (nnfp.py:234-258): 8 ConvLayer blocks with the channel
 * and stride tables of nnfp.py:193-197 on an (in_f, in_t, 1) input, DivEncLayer
 * with q = emb_sz slices, unit_dim [32,1].  (256,32,1) is the deployed shape. */
int nafp_encoder_create(nafp_encoder** enc, int in_f, int in_t, int emb_sz);
int nafp_encoder_destroy(nafp_encoder* enc);

/* The same with the normalisation of the ConvLayers chosen (config MODEL.BN -> ConvLayer(norm=...), nnfp.py:63-71, 250):
 *   NAFP_NORM_LAYER2D  'layer_norm2d' (the default and what nafp_encoder_create builds): LayerNormalization(axis=(1,2,3)),
 *                      folded into the GEMM epilogues;
 *   NAFP_NORM_LAYER1D  'layer_norm1d': LayerNormalization(axis=-1), statistics over the channels of one position; one extra
 *                      elementwise pass per layer and direction;
 *   NAFP_NORM_BATCH    'batch_norm' (or any other string): BatchNormalization(axis=-1) AS THE REFERENCE CALLS IT -- `m_fp(feat)`
 *                      without a `training` argument in train_step, val_step and generate (trainer.py:44, 60; generate.py:88),
 *                      i.e. always in inference mode: the per-channel affine map of the moving statistics (never updated:
 *                      keras initialises them to 0 / 1; a checkpoint may hold others), trainable gamma / beta.  Folded exactly
 *                      into the same epilogues.
 * With the alternates the tensors 4j+2 / 4j+3 are gamma / beta of shape (C); NAFP_NORM_BATCH appends, behind tensor 67, the
 * NON-trainable moving statistics: 68+2j moving_mean (C), 69+2j moving_variance (C), j = 0..15.  The gradient list of
 * nafp_encoder_backward covers the trainable tensors 0..67 only (nafp_encoder_n_trainable).  NAFP_OPT_FUSE_CONV0,
 * and NAFP_OPT_FUSED_LN_BWD are ignored by the alternates. */
#define NAFP_NORM_LAYER2D 0
#define NAFP_NORM_LAYER1D 1
#define NAFP_NORM_BATCH 2
int nafp_encoder_create_ex(nafp_encoder** enc, int in_f, int in_t, int emb_sz, int norm);
int nafp_encoder_norm(const nafp_encoder* enc);            /* NAFP_NORM_* of the handle */
int nafp_encoder_n_trainable(const nafp_encoder* enc);     /* 68: the tensors nafp_encoder_backward writes gradients for */

/* Parameter tensors, in this fixed order (n = nafp_encoder_n_tensors):
 *   for j in 0..15 (even = conv 1x3, odd = conv 3x1; nnfp.py:48-79):
 *     4j+0 kernel (kh,kw,Cin,Cout)   4j+1 bias (Cout)
 *     4j+2 LN gamma (F,T,C)          4j+3 LN beta (F,T,C)      [(C) each with NAFP_NORM_LAYER1D / NAFP_NORM_BATCH]
 *   64 div.w1 (Q,S,32)   65 div.b1 (Q,32)   66 div.w2 (Q,32,1)   67 div.b2 (Q,1)
 * i.e. the keras variable shapes, C-order float32. */
int nafp_encoder_n_tensors(const nafp_encoder* enc);
int64_t nafp_encoder_tensor_numel(const nafp_encoder* enc, int index);
/* writes up to 4 dims, returns rank (or -1) */
int nafp_encoder_tensor_shape(const nafp_encoder* enc, int index, int64_t dims_out[4]);
int64_t nafp_encoder_flat_dim(const nafp_encoder* enc);    /* F'*T'*C of front_conv (1024) */

/* Copy/pack weights from caller device tensors (array of n device pointers, the
 * array itself in host memory).  May be called again after every optimizer step.
 * Ordering: the call may be given a stream of its own.  Every later pass of this handle on a
 * DIFFERENT stream (forward*, div_enc, forward_train, backward) waits on the device for what it
 * reads of this call's output -- the training forward starts its first conv as soon as the plain
 * copies are done, its second behind layer 1's share (formed first) and only its third behind the
 * whole re-pack --, so the caller does not order them.  What the caller still orders: this call behind the passes enqueued EARLIER on
 * other streams (they read the blob it overwrites), and behind whatever produced `tensors`. */
int nafp_encoder_set_weights(nafp_encoder* enc, const float* const* tensors_host_array,
                             void* stream);

int64_t nafp_encoder_workspace_bytes(const nafp_encoder* enc, int64_t n_seg);

/* m_fp(feat) (nnfp.py:223-231).  feat (n_seg, in_f, in_t) float32.
 *   out_flat  optional (n_seg, flat_dim): front_conv output (nnfp.py:225), may be NULL
 *   out_emb   optional (n_seg, emb_sz): l2_normalize(div_enc(.)) if l2norm != 0,
 *             else div_enc(.) (use_L2layer, nnfp.py:228-231), may be NULL
 * LAUNCH-SIZE INDEPENDENCE AND WHAT IT COSTS SMALL LAUNCHES.  Every launch is planned -- tile shape, split-K factor, which finish runs --
 * as if it held 640 segments (environment NAFP_PLAN_B; the generate / bench launch size), whatever n_seg is, and only the grids follow
 * n_seg: the bytes of a segment's fingerprint do not depend on what shares its launch, at any n_seg (a launch with more split-K tiles
 * than arrival counters, n_seg > ~4096, runs those layers as several sample ranges).  The price is paid by very small launches
 * (n_seg = 1 .. 8, query-time latency): they no longer get the deeper split-K a plan for their own size would choose and leave most
 * of the chip idle in the mid layers.  NAFP_PLAN_B=0 plans per launch again (fastest single-segment latency, fingerprints then differ
 * in their last bits between launch sizes); a deployment that serves single queries next to a database built at 640 per launch
 * should keep the default.   */
int nafp_encoder_forward(nafp_encoder* enc, const float* feat, int64_t n_seg,
                         void* workspace, int64_t workspace_bytes,
                         float* out_flat, float* out_emb, int l2norm, void* stream);

/* m_fp(m_pre(X)) with the tail of the log-mel layer (x - reduce_max(x), clamp at -80, optional segment
 * normalisation: melspectrogram.py:108-111) applied by the first conv as it loads: raw_feat / group_stat are what
 * nafp_melspec_forward_* leave with NAFP_MELSPEC_DEFER, group_size the same grouping.  Bit-identical to
 * nafp_encoder_forward on the finished features. */
int nafp_encoder_forward_raw(nafp_encoder* enc, const float* raw_feat, const float* group_stat, int group_size,
                             int segment_norm, int64_t n_seg, void* workspace, int64_t workspace_bytes,
                             float* out_flat, float* out_emb, int l2norm, void* stream);

/* Per-kernel timing of nafp_encoder_forward with HIP events recorded on the
 * caller's stream (bench.py's roofline leg).  enable(max_forwards > 0) allocates a
 * ring of event sets, one per forward call; enable(0) turns it off.  read() waits
 * for slot `slot` (0 = the first forward after enable) and writes 17 durations in
 * milliseconds: [0] conv0, [1..15] the implicit-GEMM convs, [16] the tail. */
#define NAFP_ENCODER_PROFILE_KERNELS 17
int nafp_encoder_profile_enable(nafp_encoder* enc, int max_forwards);
int nafp_encoder_profile_count(const nafp_encoder* enc);   /* forwards recorded so far */
/* Stamp granularity of the forwards that follow.
 *   0 (default): conv0, every GEMM conv, tail.  The GEMM convs are stamped on their own dispatch packets -- start of the
 *      first kernel, end of the last (a split-K finish kernel included): hipExtLaunchKernel, nothing is put into the queue --,
 *      conv0 and the tail between recorded events.  (Stamping every launch is still not free: the per-conv figures sum to
 *      about 10 % more than the span mode 2 measures for the same launches.)
 *   1: conv0 | the 15 GEMM convs as one span | tail, between recorded events (an event recorded between two kernels costs
 *      the GPU about 20 us of un-overlapped dispatch set-up).
 *   2: only the span of the 15 GEMM convs, from the dispatch packets of conv1 and of the last GEMM-conv kernel: no queue
 *      entry at all -- what a timed region uses.  profile_read returns the span in slot 1, zeros elsewhere. */
int nafp_encoder_profile_coarse(nafp_encoder* enc, int coarse);
int nafp_encoder_profile_read(nafp_encoder* enc, int slot, float* ms_out_host);
/* Diagnostic, process-wide: while `dev_buf` is non-null every forward GEMM-conv launch (full mode) whose conv has
 * (cin, cout, output positions per segment) as given stamps the shader clock at the phase boundaries of each tile:
 * 8 u64 per wave, 8 wave slots per workgroup, workgroups in launch order -- [hw_id | xcc_id << 32, entry, geometry done,
 * first operands landed, K-loop done, after the barrier behind it, epilogue stores issued, end].  `capacity_u64` bounds
 * the buffer (launches needing more are not recorded).  nafp_conv_timeline_grid returns {grid x, y, z, tile rows, tile
 * columns} of the last recorded launch.  tools/conv_timeline.py turns the stamps into per-phase times and per-CU overlap. */
int nafp_conv_timeline(void* dev_buf, int64_t capacity_u64, int cin, int cout, int positions);
int nafp_conv_timeline_grid(int* out5_host);

/* Training (model/trainer.py:41-47: emb = m_fp(feat) under tf.GradientTape, then
 * tape.gradient(loss, m_fp.trainable_variables)).  forward_train is nafp_encoder_forward that
 * keeps every activation in `workspace`; backward consumes the SAME workspace (untouched in
 * between) and d_emb = dL/d(out_emb), and writes the 68 parameter gradients in the keras shapes
 * and order of the parameter tensors (grads_host_array: host array of device pointers; the
 * gradients are overwritten, not accumulated).  `feat` is the same (n_seg, in_f, in_t) input. */
int64_t nafp_encoder_train_workspace_bytes(const nafp_encoder* enc, int64_t n_seg);
int nafp_encoder_forward_train(nafp_encoder* enc, const float* feat, int64_t n_seg, void* workspace,
                               int64_t workspace_bytes, float* out_emb, int l2norm, void* stream);
int nafp_encoder_backward(nafp_encoder* enc, const float* feat, const float* d_emb, int64_t n_seg,
                          void* workspace, int64_t workspace_bytes, float* const* grads_host_array,
                          int l2norm, void* stream);

/* Overlap of the data-parallel gradient reduction with the backward pass.  The backward pass finishes the
 * parameter gradients last layer first; it records one event per GROUP of tensors, in completion order:
 *   group 0 = tensors 48..67 (convs 12-15 and the divide-and-encode tensors: 65 % of all parameters),
 *   group 1 = 32..47, group 2 = 16..31, group 3 = 0..15.
 * nafp_encoder_grad_group_wait makes `stream` (the communication stream) wait, on the device, until group
 * `group` of the most recent nafp_encoder_backward call is complete; the host does not block. */
#define NAFP_GRAD_GROUPS 4
int nafp_encoder_grad_group_range(const nafp_encoder* enc, int group, int* first_tensor, int* last_tensor);
int nafp_encoder_grad_group_wait(nafp_encoder* enc, int group, void* stream);

/* Execution options of an encoder handle (results are identical either way, except NAFP_OPT_BF16X3).
 *   NAFP_OPT_FUSE_CONV0  0 (default): b0.conv1x3 writes its activation, b0.conv3x1 reads it back.
 *                        1: only conv0's LayerNorm statistics are computed up front and conv1
 *                           re-generates its input tiles in-kernel from the log-mel features.
 *   NAFP_OPT_FUSED_LN_BWD  training: the LayerNorm + ELU backward of a layer runs in the epilogue of the transposed conv
 *                        that produces its gradient (one read of the stored pre-activation, one write, instead of a
 *                        separate pass that re-reads and re-writes the gradient).  0 (default): never -- measured slower
 *                        than the separate pass; 1: for the layers of >= 32768 elements per sample at
 *                        batches >= 64; 2: wherever the geometry permits.
 *   NAFP_OPT_BWD_OVERLAP  training: weight gradients of nafp_encoder_backward on a second stream owned by the handle; `stream` is
 *                        made to wait for them before the last event of the call, so callers see the same ordering as with 0.
 *                        2 (default): those of the SMALL layers (fewer than 16 output positions), next to the LayerNorm backward
 *                        and transposed conv of the layer below; 1: every layer's (measured slower); 0: everything on
 *                        `stream`.
 *   (5, NAFP_OPT_SMALLNET: retired in round 6 -- the small layers as one persistent launch measured slower than the per-layer launches
 *                        at every setting, docs/DESIGN_HISTORY.md; the number stays reserved, setting it is accepted and ignored.) */
#define NAFP_OPT_FUSE_CONV0 1
#define NAFP_OPT_FUSED_LN_BWD 2
#define NAFP_OPT_BWD_OVERLAP 4
#define NAFP_OPT_SMALLNET 5           /* retired, ignored */
/* (6 is a test hook of tests/test_gpu_train_dp.py -- it delays the handle's weight-gradient stream -- and is refused unless the process
 * runs with NAFP_TEST_HOOKS=1 in its environment; not part of the interface.) */
/* EXPERIMENTAL, changes the arithmetic (the only option that does): the unsplit GEMM convs of nafp_encoder_forward form their
 * products on the bf16 matrix pipe from f32 operands split into hi + lo bf16 halves (hi*hi + hi*lo + lo*hi, f32
 * accumulation).  Fingerprints move at the 1e-6 level against the f32 path.  Off by default; bench.py reports it as a
 * separate object with its measured error, never as the headline value.
 * value 2: the EXACT 3-way split x = h + m + l (three bf16 terms hold a float32's 24 significant bits) with the six products of
 * relative weight >= 2^-16 (hh, hm, mh, hl, mm, lh; the dropped ml + lm + ll < 2^-25 of |a||b|): float32-equivalent arithmetic on
 * the bf16 matrix pipe -- its error against the float64 oracle equals the f32 path's own (tests/test_gpu_parity_forward.py).
 * Value 2 also covers the TRAIN step: nafp_encoder_forward_train, the transposed convs and the weight gradients of
 * nafp_encoder_backward run on the same arithmetic (the flipped kernels are split alongside in nafp_encoder_set_weights; LayerNorm
 * backward, loss and optimizer stay f32): gradients within the f32 path's tolerances
 * (tests/test_gpu_backward.py, tests/test_gpu_configs.py), the step 84 -> 69 ms at BSZ 5120, 11.9 -> 9.7 ms at 640 (bench.py
 * `train_x6_experimental`).
 * CAUTION for kernels of OTHER libraries (values 1 and 2): on gfx950 a packed-f32 vector instruction that carries an op_sel modifier
 * (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 ... op_sel:[..]; compilers emit them for complex arithmetic -- rocFFT's kernels hold
 * them) returns wrong values in a wave that shares a compute unit with waves issuing the 128-bit-operand matrix instructions
 * (v_mfma_f32_32x32x16_bf16, what this option runs on) next to vector work.  Found in round 6; tools/probes/pk_opsel_hazard_probe.hip
 * reproduces it without this library, profiles/r06_experiments.md section 5 has the matrix of instructions.  EVERY kernel of this
 * library is compiled without packed-f32 instructions (build.py NO_PACKED_F32; tests/test_abi.py disassembles the library and holds
 * it), so the library's own kernels -- front end, other forwards, training -- overlap such a forward freely and stay run-to-run
 * bit-identical (tests/test_gpu_generate.py, tools/x6_determinism_check*.py).  A foreign kernel holding such instructions must not
 * run on the device while a forward with this option is in flight.  The fp32 path (f32 matrix instructions) does not disturb anything. */
#define NAFP_OPT_BF16X3 3
/* Options are host-side state of the handle: set them while no pass of the handle is being enqueued from another thread; passes already
 * enqueued keep what they were launched with.  (NAFP_OPT_BF16X3 = 2 allocates its split weights, 1.5 x the packed conv kernels, on first use.) */
int nafp_encoder_set_option(nafp_encoder* enc, int option, int value);

/* m_fp.div_enc(x) alone (nnfp.py:141-156; called separately at trainer.py:73-76). */
int nafp_encoder_div_enc(nafp_encoder* enc, const float* flat, int64_t n_seg,
                         float* out_emb, int l2norm, void* stream);

/* tf.math.l2_normalize(x, axis=1) of a (n_rows, dim) array: x * rsqrt(max(sum x^2, 1e-12)), as test_step
 * applies it to front_conv's and div_enc's outputs (trainer.py:74, 76).  out may alias x. */
int nafp_l2_normalize_rows(const float* x, int64_t n_rows, int dim, float* out, void* stream);

/* Send buffer of the embedding-gradient reduce-scatter (the backward of the all-gather at model/fp/NTxent_loss_tpu.py:57-87,
 * neural-audio-fp_amd/model/trainer.py scatter_embedding_gradients), packed in ONE launch:
 *   send[r, 0 : n_anchors*d]            = d_a_all[r*n_anchors : (r+1)*n_anchors, :]      r = destination rank
 *   send[r, n_anchors*d : 2*n_anchors*d] = d_b_all[r*n_anchors : (r+1)*n_anchors, :]
 *   send[r, 2*n_anchors*d : +4]          = loss_sum[0] * loss_scale                      (the loss rides in a 4-float tail)
 * d_a_all, d_b_all (world*n_anchors, d); send (world, 2*n_anchors*d + 4); all device pointers. */
int nafp_pack_embedding_grads(const float* d_a_all, const float* d_b_all, const float* loss_sum, float loss_scale,
                              int64_t world, int64_t n_anchors, int dim, float* send, void* stream);

/* ------------------------------------------------------------------------
 * NT-Xent loss (model/fp/NTxent_loss_single_gpu.py:52-82; sharded form
 * model/fp/NTxent_loss_tpu.py:90-137)
 * ---------------------------------------------------------------------- */

int64_t nafp_ntxent_workspace_bytes(int64_t n_local, int64_t n_global);

/* Rows of this rank: emb_org_local/emb_rep_local (n_local, d); columns: the
 * (all-gathered) emb_org_all/emb_rep_all (n_global, d); this rank's rows sit at
 * [rank_offset, rank_offset+n_local) of the global arrays.  Single device:
 * local == all, rank_offset = 0.
 *   loss_sum  out, 1 float: sum over local rows of (CE_a + CE_b); the caller
 *             divides by n_global for the reference's mean (and all-reduces)
 *   sim_mtx   optional (n_local, 2*n_global-1): [ab | aa without its diagonal]
 *             as compute_loss returns it; NULL to skip
 *   d_org_all/d_rep_all optional (n_global, d): gradient of (loss_sum/n_global)
 *             w.r.t. the global arrays contributed by this rank's rows AND
 *             columns restricted to local rows (sum over ranks = full gradient);
 *             NULL to skip.  d (MODEL.EMB_SZ, nnfp.py:250) is 64, 128 or 256. */
int nafp_ntxent_forward(const float* emb_org_local, const float* emb_rep_local,
                        const float* emb_org_all, const float* emb_rep_all,
                        int64_t n_local, int64_t n_global, int64_t rank_offset, int d,
                        float tau, float* loss_sum, float* sim_mtx,
                        float* d_org_all, float* d_rep_all,
                        void* workspace, int64_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------
 * Spec-augment (train step only): SpecNCutout with uniform_mask=True
 * (model/fp/specaug_chain/layers/ncutout_tarray.py:214-268)
 * ---------------------------------------------------------------------- */

/* A hole: frequency rows [f0, f1] x frames [t0, t1], INCLUSIVE on both ends
 * (generate_single_mask, ncutout_tarray.py:117-128). */
typedef struct { int f0, f1, t0, t1; } nafp_rect;

/* In place on feat (n_seg, F, T): elements inside any of the <= 8 holes become fill_value for the
 * samples whose `active` flag is set (active == NULL: every sample).  T % 4 == 0. */
int nafp_specaug_apply(float* feat, int64_t n_seg, int F, int T, const nafp_rect* rects_host,
                       int n_rects, const unsigned char* active, float fill_value, void* stream);

/* The same with the fill value read from device memory (hole_fill = 'min' fills with reduce_mean(x),
 * ncutout_tarray.py:203-204: nafp_specaug_mean leaves that mean on the device, no host round trip). */
int nafp_specaug_apply_fill_dev(float* feat, int64_t n_seg, int F, int T, const nafp_rect* rects_host,
                                int n_rects, const unsigned char* active, const float* fill_value_dev,
                                void* stream);
/* mean_out[0] = mean of the n floats of feat (16-byte aligned), summed in double in a fixed order. */
int64_t nafp_specaug_mean_workspace_bytes(void);
int nafp_specaug_mean(const float* feat, int64_t n, float* mean_out, void* workspace, int64_t workspace_bytes,
                      void* stream);

/* The general form of the masking: SpecNCutout with uniform_mask=False (a rectangle set and an activation draw per sample and
 * hole, ncutout_tarray.py:131-186, 270-276) and / or a filler TENSOR (hole_fill = 'random' | [min_mag, max_mag], :106-115,
 * 200-211: the layer's noise tensor, drawn once, scaled to the batch's value range on every call).
 *   rects_dev   (n_sets, n_rects, 4) int32 in device memory: f0, f1, t0, t1, inclusive; n_sets == 1 (one set for the batch) or
 *               n_seg (one per sample); n_rects <= 32
 *   active_dev  (n_seg, n_rects) bytes or NULL (every hole of every sample)
 *   filler      (n_fill_seg, F, T) floats, 16-byte aligned, or NULL (reads as 1); sample b uses row b % n_fill_seg
 *   scale_offset_dev  2 floats in device memory: a hole element becomes filler * scale + offset
 * nafp_specaug_range leaves {max - min, min} of feat on the device (the 'random' filler's scale and offset, :207-208); it takes
 * the workspace of nafp_specaug_mean. */
int nafp_specaug_apply_ex(float* feat, int64_t n_seg, int F, int T, const int32_t* rects_dev, int n_rects, int64_t n_sets,
                          const unsigned char* active_dev, const float* filler, int64_t n_fill_seg,
                          const float* scale_offset_dev, void* stream);
int nafp_specaug_range(const float* feat, int64_t n, float* scale_offset_out, void* workspace, int64_t workspace_bytes,
                       void* stream);

/* ------------------------------------------------------------------------
 * Optimizer steps of the train step (model/trainer.py:47-48, 119-140)
 * ---------------------------------------------------------------------- */

/* One parameter tensor with its gradient and moment slots (all device pointers, float32).
 * var_len = length of ONE keras variable inside the tensor: numel for the conv / LN tensors;
 * the stacked divide-and-encode tensors hold 128 variables each (var_len = numel / 128), and
 * LAMB takes its trust ratio per variable, as the reference does over its 576 variables. */
typedef struct {
    float* param; const float* grad; float* m; float* v;
    int64_t numel; int64_t var_len;
} nafp_opt_tensor;

/* tf.keras.experimental.CosineDecay(lr0, decay_steps, alpha) at `step` (trainer.py:119-124). */
float nafp_cosine_decay_lr_host(float lr0, int64_t step, int64_t decay_steps, float alpha);
/* tf.keras.experimental.CosineDecayRestarts(lr0, first_decay_steps, t_mul, m_mul, alpha) at `step`
 * (LR_SCHEDULE 'COS-RESTART', trainer.py:125-131). */
float nafp_cosine_decay_restarts_lr_host(float lr0, int64_t step, int64_t first_decay_steps, float t_mul,
                                         float m_mul, float alpha);

/* tf.keras.optimizers.Adam.apply_gradients (trainer.py:138): step is 1-based (iterations + 1);
 * keras defaults beta1 0.9, beta2 0.999, eps 1e-7. */
int nafp_adam_step(const nafp_opt_tensor* tensors_host, int n, float lr, float beta1, float beta2,
                   float eps, int64_t step, void* stream);

/* LAMB._resource_apply_dense (model/fp/lamb_optimizer.py:123-158): defaults beta1 0.9, beta2 0.999,
 * eps 1e-6, weight_decay 1e-6, decay and layer adaptation on every variable. */
int64_t nafp_lamb_workspace_bytes(const nafp_opt_tensor* tensors_host, int n);
int nafp_lamb_step(const nafp_opt_tensor* tensors_host, int n, float lr, float beta1, float beta2,
                   float eps, float weight_decay, int64_t step, void* workspace,
                   int64_t workspace_bytes, void* stream);

/* Online triplet loss of the now-playing baseline: OnlineTripletLoss.compute_loss with use_anc_as_pos = True
 * (model/fp/online_triplet_loss.py:199-239; masks :98-121; distances :185-196).  mode 0 = 'semi-hard'
 * (training, trainer.py:160-164), 1 = 'all' (validation, :165-169), 2 = 'all-balanced' (:215-222), 3 = 'hardest'
 * (:223-227, with the reference's min over the MASKED matrix, i.e. hardest negative = 0).  n_pos = n_anchor * n_pos_per_anchor,
 * replicas in anchor order.  loss_out (1); pairwise_dist (n_anchor, n_pos + n_anchor) or NULL; d_anchor /
 * d_pos: gradients of the loss (both or neither).  All device pointers. */
int64_t nafp_triplet_workspace_bytes(int64_t n_anchor, int64_t n_pos);
int nafp_triplet_forward(const float* emb_anchor, const float* emb_pos, int64_t n_anchor, int64_t n_pos, int dim,
                         int mode, float margin, float* loss_out, float* pairwise_dist, float* d_anchor,
                         float* d_pos, void* workspace, int64_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Training batch assembly + time-domain augmentation on the device.  Replaces, per output row, the host
 * work of genUnbalSequence.__getitem__ (model/utils/dataloader_keras.py:223-311): load_audio
 * (model/utils/audio_utils.py:221-264), bg_mix_batch (:82-117, background_mix :28-72) and
 * ir_aug_batch (:120-137).  All windows index ONE int16 PCM arena; the random draws (offsets, SNR,
 * amplitude ratio) are made by the caller, as the reference makes them on the host.
 * ------------------------------------------------------------------------------------------- */
typedef struct nafp_aug_row {
    int64_t ev_off;      /* event window: first sample in the arena                                   */
    int64_t nz_off;      /* background window (same length), or -1                                    */
    int64_t nz2_off;     /* second noise window added to the first (speech), or -1                    */
    int64_t ir_off;      /* impulse response, or -1                                                   */
    int32_t ev_valid;    /* real samples of the event window (<= seg_len; the tail is zero)           */
    int32_t nz_valid, nz2_valid;
    int32_t ir_len;      /* taps used (<= 600 = MAX_IR_LENGTH, dataloader_keras.py:8)                 */
    float snr_db;        /* bg_mix_batch's uniform draw in snr_range (audio_utils.py:92-95)           */
    float amp;           /* its log-uniform amplitude ratio in (0.1, 1) (audio_utils.py:98-99)        */
    int32_t mix;         /* 1: run bg_mix_batch's arithmetic for this row (replicas with BG/speech)   */
    int32_t reserved;
} nafp_aug_row;

/* out (n_rows, seg_len) float32; rows is a DEVICE array.  A row with mix = 0 and ir_off = -1 is the
 * plain window / 2^15 (the anchors).  seg_len % 4 == 0. */
int nafp_augment_rows(const int16_t* pcm, const nafp_aug_row* rows, int64_t n_rows, int seg_len, float* out,
                      void* stream);

/* ---------------------------------------------------------------------------------------------
 * Search / evaluation over resident fingerprints (consumer of the generate path's output).
 * Replaces faiss.IndexFlatL2 as eval/eval_faiss.py uses it with index_type 'L2'
 * (eval/utils/get_index_faiss.py:57-62; index.add at eval_faiss.py:145-146; index.search at :209)
 * and the sequence-score loop of eval_faiss.py:221-230.  The index is the caller's device array
 * [dummy_db ; db] (n_index x dim float32, dim 64 or 128); nothing is trained or quantised.
 * ------------------------------------------------------------------------------------------- */

/* floats of the auxiliary array (|x|^2/2 per row, padded) nafp_search_index_prepare fills */
int64_t nafp_search_index_aux_floats(int64_t n_index);
int nafp_search_index_prepare(const float* index, int64_t n_index, int dim, float* aux, void* stream);

int64_t nafp_search_workspace_bytes(int64_t n_query, int64_t n_index, int k);
/* out_dist / out_ids (n_query, k): the k smallest squared L2 distances and their row ids, nearest
 * first; equal distances: smaller id first; fewer than k rows: id -1, distance +inf.  k <= 32. */
int nafp_search_topk_l2(const float* query, int64_t n_query, const float* index, const float* aux,
                        int64_t n_index, int dim, int k, float* out_dist, int32_t* out_ids,
                        void* workspace, int64_t workspace_bytes, void* stream);

/* out_scores[t, s] = mean_{i < min(task_len[t], n_index - c)} query[task_q0[t] + i] . index[c + i]
 * for the candidate sequence start c = cand[t, s] (>= 0; -1 -> -inf): np.mean(np.diag(np.dot(q,
 * index[c:c+l].T))) of eval_faiss.py:224-230.  All device pointers. */
int nafp_search_seq_scores(const float* query, const float* index, int64_t n_index, int dim,
                           const int32_t* task_q0, const int32_t* task_len, int64_t n_tasks,
                           const int32_t* cand, int n_slots, float* out_scores, void* stream);

/* In-training mini search test (model/utils/mini_search_subroutines.py): pairwise_distances_for_eval (:28-93;
 * mode 0 = squared L2 clipped at 0 for 'argmin', 1 = dot product for 'argmax'; any dim) into out_scores
 * (n_query, n_db); then, per sequence length `scope`, the rank of the ground-truth start id
 * (t + gt_id_offset) among all candidate starts under the eye(scope) diagonal sum (conv_eye_func :96-119 and
 * the argsort / np.where of mini_search_eval :180-205): out_rank (n_query - scope + 1) int32. */
int nafp_minisearch_scores(const float* query, const float* db, int64_t n_query, int64_t n_db, int dim, int mode,
                           float* out_scores, void* stream);
int nafp_minisearch_ranks(const float* scores, int64_t n_query, int64_t n_db, int scope, int mode, int gt_id_offset,
                          int32_t* out_rank, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NAFP_H */
