/*
 * isbfsar.h -- C ABI of libisbfsar_hip.so (MI355X / gfx950).
 *
 * One flat, torch-free boundary for the reference's hot path
 *     frame -> 3D pose (modules/hpe) -> skeleton-window embedding + few-shot / open-set match (modules/ar).
 * Each entry point names the reference interface it replaces (paths relative to the
 * reference tree, steb6/ISBFSAR).  The reference seam is `Runner.__call__`
 * (utils/tensorrt_runner.py:64-77: numpy in -> H2D -> engine -> D2H -> sync) for the pose
 * stages and the eager `TRXOS.forward` call (modules/ar/ar.py:69) for the AR stage.
 *
 * Conventions
 *   - every function returns 0 on success, <0 on error; isb_last_error() gives the message
 *     (thread local).  No C++ exception crosses this boundary.
 *   - pointers named d_* are DEVICE pointers (HBM of the handle's device); h_* are host
 *     pointers.  The caller owns every buffer it passes; the library never returns internal
 *     pointers and keeps no reference to host memory after a call returns.
 *   - `stream` is a hipStream_t passed as void* (NULL = the HIP null stream, which is what
 *     PyTorch's default stream is).  Calls are asynchronous on that stream unless documented
 *     otherwise; the *_host variants run on a private stream and synchronise before returning.
 *   - a handle is bound to one device and is not thread safe: one handle per process / rank.
 *   - ONE STREAM PER HANDLE AT A TIME: a handle owns one set of activation / tuple workspaces, ordered only by
 *     the stream a call is issued on. Calls on the same handle from two streams race on those buffers; to move a
 *     handle to another stream, make the new stream wait on the old one's work first (event or synchronise).
 *     Different handles are independent and may run on different streams concurrently. isb_hpe_forward forks
 *     internally onto private lane streams and joins back into `stream` before it returns to the caller's order.
 */
#ifndef ISBFSAR_H
#define ISBFSAR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ISB_OK 0
#define ISB_ERR_INVALID (-1)   /* bad argument / shape */
#define ISB_ERR_STATE (-2)     /* call order (e.g. infer before weights/support) */
#define ISB_ERR_HIP (-3)       /* a HIP runtime call failed */
#define ISB_ERR_WEIGHTS (-4)   /* malformed blob or missing tensor */
#define ISB_ERR_NOMEM (-5)

/* precision of the two tuple-attention contractions (K10). MLP/projection/discriminator
 * GEMMs always run on the exact f32 MFMA path.
 * ABI version 2 (isb_version() == 2): 0 is "the library's default" and the default is fp16 operands, as for
 * isb_hpe_cfg.precision; bf16 operands moved from 0 to 3. (Version 1: 0 = bf16. A zero-initialised isb_ar_cfg therefore
 * gets fp16 attention operands now -- logits 2-13x closer to the fp32 reference at the same matrix rate)
 * isb_ar_precision() returns the resolved value of a handle. */
#define ISB_AR_PREC_DEFAULT 0  /* = ISB_AR_PREC_F16 */
#define ISB_AR_PREC_BF16X3 1   /* split-bf16 (hi+lo, 3 MFMA / product), ~2^-16 rel.  */
#define ISB_AR_PREC_F16 2      /* IEEE fp16 operands (11 significant bits, 8x bf16's resolution) at bf16's rate: K, V^T and the
                                  attention weights of the all-classes pass; the arg-max class's pass stays split-bf16 */
#define ISB_AR_PREC_BF16 3     /* bf16 operands, f32 accumulate (round 3's default)  */

const char* isb_last_error(void);
int isb_version(void);
/* 1 if the build verified, in the disassembly of this very library, that the weights-stationary expand kernels' literally
 * named staging registers are touched by nothing else (isbfsar_amd/build.py); 0 = unverified: those kernels are then never
 * selected by default and the tile kernels run instead (same results, ~4 % slower pose stage). */
int isb_wsreg_verified(void);
/* number of visible HIP devices (does not initialise a context) */
int isb_device_count(void);
/* Hardware queues the HIP runtime multiplexes this process's streams onto (GPU_MAX_HW_QUEUES; the runtime reads it once, at its
 * first call, default 4). Engines kept in flight on their own streams (isb_hpe_create_shared) need a queue each plus the caller's: with
 * four, a third pose stream serialises behind another stream of the process (19.3 vs 17.55 ms per 256-frame pipeline step). The
 * library asks for 8 when it is LOADED unless the caller's environment already holds a value; that takes effect iff no HIP call has
 * been made in the process before (import / dlopen the library before the first HIP call, or export the variable yourself).
 * Returns the value the runtime reads from the environment; *source (may be NULL): 0 = unset (default 4), 1 = set by the caller,
 * 2 = set by this library at load time. No HIP call is made. */
int isb_hw_queues(int32_t* source);

/* ------------------------------------------------------------------------------------------
 * Action recognition: TRXOS skeleton branch + Discriminator
 *   replaces  ActionRecognizer.__init__  modules/ar/ar.py:11-28   (model build + weight load)
 *             TRXOS.forward              modules/ar/utils/model.py:291-328
 * ---------------------------------------------------------------------------------------- */
typedef struct isb_ar isb_ar;

typedef struct isb_ar_cfg {
    int32_t seq_len;     /* L  : TRXConfig.seq_len   (utils/params.py:8,95)  */
    int32_t n_joints;    /* J  : TRXConfig.n_joints  (utils/params.py:63)    */
    int32_t way_max;     /* TRXConfig.way (utils/params.py:53): capacity of the support set */
    int32_t device;      /* HIP device ordinal */
    int32_t precision;   /* ISB_AR_PREC_* */
    int32_t max_batch;   /* windows processed per internal chunk (workspace size); 0 = 1024 */
} isb_ar_cfg;

int isb_ar_create(const isb_ar_cfg* cfg, isb_ar** out);
void isb_ar_destroy(isb_ar* h);
/* the handle's attention precision with the default resolved: ISB_AR_PREC_BF16X3 / _F16 / _BF16 (never 0); < 0 on a null handle */
int isb_ar_precision(const isb_ar* h);

/* ISBW blob (see isbfsar_amd/weights.py) holding the TRXOS state-dict tensors
 * features_extractor.sk.fc{1,2}, transformers.0.{k_linear,v_linear,norm_k},
 * discriminator.{dimensionality_reduction,fc1,fc2,fc3}  (model.py:269,41-46,283-285).
 * Replaces torch.load + load_state_dict at modules/ar/ar.py:17-19. The blob may be freed
 * after return. Synchronous. */
int isb_ar_load_weights(isb_ar* h, const void* h_blob, size_t nbytes);

/* Install the support set (replaces ActionRecognizer.train, ar.py:94-96, plus the lazy
 * feature computation at ar.py:62-74 and the per-call support K/V at model.py:75-84, which
 * the reference recomputes every call).  Exactly one of h_poses [n,L,3J] / h_features
 * [n,L,256] (the cached MLP features, ar.py:56-61) is non-NULL.  Caches features, K
 * (LayerNorm'ed) and V of every support tuple on the device.  n in [1, way_max].
 * Synchronous. */
int isb_ar_set_support(isb_ar* h, const float* h_poses, const float* h_features, int32_t n);
/* copy the cached support MLP features [n,L,256] back (outputs['support_features'], ar.py:72-74) */
int isb_ar_get_support_features(isb_ar* h, float* h_features);

/* TRXOS.forward on B query windows that share the installed support set.
 *   d_windows [B,L,3J] f32                      query_data['sk']          (model.py:301-303)
 *   d_logits  [B,n]    f32                      out['logits']             (model.py:137-145)
 *   d_is_true [B]      f32                      out['is_true']            (model.py:325)
 *   d_embed   [B,L,256] f32 or NULL             features_extractor output (model.py:302)
 * Any B >= 1 (internally chunked by max_batch). Asynchronous on `stream`. */
int isb_ar_infer(isb_ar* h, const float* d_windows, int32_t B, float* d_logits, float* d_is_true,
                 float* d_embed, void* stream);
/* same with host buffers (H2D + kernels + D2H + synchronise): the reference's per-call
 * `.cuda()` / `.cpu()` pattern, ar.py:41,77-78 */
int isb_ar_infer_host(isb_ar* h, const float* h_windows, int32_t B, float* h_logits,
                      float* h_is_true, float* h_embed);

/* Input type (TRXConfig.input_type, utils/params.py:15,81). ISB_AR_INPUT_SKELETON (default): per-frame features = MLP(pose).
 * ISB_AR_INPUT_HYBRID: features = [PostResNet(ResNet-50 trunk features) | MLP(pose)] = 512 wide (model.py:207-216, 270-277,
 * 296-316: RGB first); the transformer then takes 512-wide rows (k/v_linear [128,1024], 512-wide positional encoding) and
 * the blob must also hold post_resnet.l1.{weight,bias}. Call BEFORE isb_ar_load_weights (it invalidates weights and support
 * set). The trunk features [.., L, 2048] come from isb_rgb_forward (the ResNet-50 engine below) or from the caller. In hybrid
 * mode isb_ar_set_support takes cached features only ([n, L, 512]); embeddings / support features are 512 wide. */
#define ISB_AR_INPUT_SKELETON 0
#define ISB_AR_INPUT_HYBRID 1
int isb_ar_set_input_type(isb_ar* h, int32_t type);
/* h_poses [n, L, 3J], h_trunk [n, L, 2048] (host): the support set of ar.py:62-67 with both "imgs" and "poses" */
int isb_ar_set_support_hybrid(isb_ar* h, const float* h_poses, const float* h_trunk, int32_t n);
/* isb_ar_infer with d_trunk [B, L, 2048] f32 (device) beside the pose windows; d_embed (optional) [B, L, 512] */
int isb_ar_infer_hybrid(isb_ar* h, const float* d_windows, const float* d_trunk, int32_t B, float* d_logits, float* d_is_true,
                        float* d_embed, void* stream);

/* test hook: argmax class per window of the last isb_ar_infer chunk sequence, int32 [B] (model.py:323) */
int isb_ar_last_chosen(isb_ar* h, int32_t* h_chosen, int32_t B);

/* device time of the dominant kernel inside isb_ar_infer (ar_proto_kernel, all-classes mode:
 * S^T tiles, A^T = exp2(S^T - lse2), P^T = V^T A^T, distance), measured with HIP events on the
 * launch stream: enable, run, then read the accumulated milliseconds and launch count.
 * Used by bench.py's roofline object. */
int isb_ar_profile(isb_ar* h, int32_t enable);
int isb_ar_profile_read(isb_ar* h, double* ms_total, int64_t* launches);

/* ------------------------------------------------------------------------------------------
 * Human pose estimation: crop homography -> warp -> EfficientNetV2-L -> pose head -> soft-argmax
 * decode -> absolute reconstruction -> joint expansion/selection
 *   replaces  HumanPoseEstimator.__init__  modules/hpe/hpe.py:15-46  (K, joint assets, 4 engines)
 *             HumanPoseEstimator.estimate  modules/hpe/hpe.py:76-173 (everything after the detector)
 *   and the four Runner(...) calls it makes (utils/tensorrt_runner.py:64-77; hpe.py:97,103,106).
 * The person box (x1,x2,y1,y2 pixel ints, the order estimate() returns them in, hpe.py:173) is an input: it comes from
 * isb_det_forward + isb_hpe_select_person (the reference's detector + post-processing, hpe.py:51-79) or from the caller.
 * ---------------------------------------------------------------------------------------- */
typedef struct isb_hpe isb_hpe;

typedef struct isb_hpe_cfg {
    float fx, fy, ppx, ppy;   /* RealSenseIntrinsics (utils/params.py:40-47) -> K, hpe.py:28-33 */
    int32_t width, height;    /* frame size (640 x 480) */
    int32_t device;           /* HIP device ordinal */
    int32_t max_batch;        /* frames per internal micro-batch (activation workspace); 0 = 64; clamped to 1024: the
                               * convolution kernels address a tensor with 32-bit byte offsets and the largest activation
                               * is 2 MiB per frame. Any B is accepted by isb_hpe_forward (it micro-batches). */
    int32_t n_out_joints;     /* informational: joints per pose after selection (30 / 122) */
    int32_t precision;        /* 16-bit storage type of the backbone (weights AND activations; f32 accumulate everywhere):
                               * 0 = default = 2. (CHANGED in round 4 with no version marker: until then 0 meant layout 3 below;
                               *     a caller that zero-initialises the struct gets different output bits, ~2 % lower rate and
                               *     the better parity. isb_version() >= 2 implies this meaning.)
                               * 2 = IEEE fp16 in every stage -- the precision the reference runs its backbone at (TensorRT
                               *     engines built with fp16=True, 7_create_engines.py:10). Same MFMA rate and bytes as bf16,
                               *     3 more mantissa bits; conversions saturate at +-65504. Closest 16-bit layout to the fp32
                               *     definition on both weight profiles.
                               * 1 = bf16 everywhere (the round-2 layout).
                               * 3 = bf16, with the two 8x8 stages (32 of the 79 blocks) and the 640 -> 1280 convolution in fp16
                               *     (round 3's default). */
} isb_hpe_cfg;

int isb_hpe_create(const isb_hpe_cfg* cfg, isb_hpe** out);
void isb_hpe_destroy(isb_hpe* h);
/* One more engine on the parent's device weights (no counterpart in the reference, whose worker owns one set of TensorRT engines and
 * runs one frame at a time, utils/tensorrt_runner.py:64-77; main.py:336-342): the child READS the parent's model -- folded and packed
 * weights, joint map; 240 MB for EfficientNetV2-L -- and OWNS what a pass writes: its streams, events and activation workspaces. A
 * caller that keeps K batches in flight (one engine per batch in flight, each isb_hpe_forward on its own stream; with
 * isb_hpe_set_lanes(h, 1) every launch covers the whole batch) pays one model + K workspaces instead of K of each, and gets the same
 * bits as one engine run batch after batch. The child copies the parent's configuration (device, precision, max_batch, lanes at the
 * time of the call); isb_hpe_load_weights / isb_hpe_set_joint_map are refused on a child (ISB_ERR_STATE) and, on the parent, wait for
 * the device and update the model every engine of the family reads. Either may be destroyed first. A family is ONE host thread's (the
 * handles share host-side state of the model as well as its device memory). Needs a hardware queue per stream in flight: isb_hw_queues(). */
int isb_hpe_create_shared(isb_hpe* parent, isb_hpe** out);
/* device memory behind a handle: bytes of the model it reads (shared by *engines_on_model handles), bytes of the workspaces it owns
 * (allocated lazily by the first passes). Any output may be NULL. */
int isb_hpe_memory(isb_hpe* h, uint64_t* model_bytes, uint64_t* workspace_bytes, int32_t* engines_on_model);
/* ISBW blob with bbone.* (EfficientNetV2-L, folded BN) and head.{weight,bias} (Linear(1280,288),
 * modules/hpe/setup/4_create_heads_onnx.py:10,22-25). Replaces the bbone1/heads1 engine
 * deserialisation (hpe.py:45-46). Synchronous. */
int isb_hpe_load_weights(isb_hpe* h, const void* h_blob, size_t nbytes);
/* assets/32_to_122.npy [32,122] and skeleton_types[...]['indices'] (hpe.py:37-39,162-164);
 * h_indices NULL = keep all 122 joints (skeleton=None) */
int isb_hpe_set_joint_map(isb_hpe* h, const float* h_expand, const int32_t* h_indices, int32_t n_out);

/* estimate() for B frames:
 *   d_frames [B,height,width,3] u8 BGR    the frame main.py:74 puts on the HPE queue
 *   d_bbox   [B,4] i32  x1,x2,y1,y2       detector result (hpe.py:76-79)
 *   d_joints [B,n_out,3] f32              result["pose"] (un-centred; hpe.py:169,171)
 *   d_valid  [B] u8                       0 where estimate() would return None (hpe.py:152-153)
 * Asynchronous on `stream`; internally micro-batched by max_batch. */
int isb_hpe_forward(isb_hpe* h, const uint8_t* d_frames, const int32_t* d_bbox, int32_t B, float* d_joints,
                    uint8_t* d_valid, void* stream);
int isb_hpe_forward_host(isb_hpe* h, const uint8_t* h_frames, const int32_t* h_bbox, int32_t B, float* h_joints,
                         uint8_t* h_valid);
/* How a micro-batch of >= 64 frames is run: n_lanes = 2 (default; ISB_HPE_LANES overrides it at creation) splits it into two
 * halves on two streams that fill each other's launch gaps; n_lanes = 1 keeps whole-batch launches on the caller's stream -- for a
 * caller that keeps SEVERAL batches in flight itself (one engine per batch in flight, each on its own stream: two 256-frame
 * batches side by side run 4 % faster per frame than one 256-frame batch split in halves, bench_workloads.py). Every frame is
 * independent, so the choice changes no result bit. 1 <= n_lanes <= 4; synchronous (waits for the engine's work). */
int isb_hpe_set_lanes(isb_hpe* h, int32_t n_lanes);
/* The same call split in two for a caller that has batch k + 1 in hand while batch k computes (no counterpart in the
 * reference, whose Runner copies and waits inside one call, utils/tensorrt_runner.py:64-77):
 *   isb_hpe_submit_host  enqueues everything -- frames H2D on the copy engine (the handle's copy stream), the pose pass behind
 *                        it, results D2H into pinned staging -- and returns; h_frames (PINNED memory, else the call blocks for
 *                        the copy), h_joints and h_valid must stay valid until the matching wait;
 *   isb_hpe_wait_host    blocks until the OLDEST outstanding submission is finished and writes its h_joints / h_valid.
 * Up to two submissions may be in flight (a third submit first finishes the oldest, writing its results then). Submissions run
 * in order on the handle's stream: batch k + 1's transfer hides behind batch k's kernels, and the lanes go from batch to batch
 * without draining. Results are the bits isb_hpe_forward_host gives. ISB_ERR_STATE from wait when nothing is outstanding, and
 * from load_weights / set_joint_map / set_augmentations while a submission is. */
int isb_hpe_submit_host(isb_hpe* h, const uint8_t* h_frames, const int32_t* h_bbox, int32_t B, float* h_joints, uint8_t* h_valid);
int isb_hpe_wait_host(isb_hpe* h);

/* detector post-processing, hpe.py:59-79 + misc.py:64-107: YOLOv4 export tensors
 *   d_boxes [B,4032,1,4] f32 (x1,y1,x2,y2 normalised), d_confs [B,4032,80] f32
 * -> d_bbox [B,4] i32 (x1,x2,y1,y2) of the most confident anchor whose arg-max class is 0 (person)
 * and whose confidence exceeds conf_thresh (MetrabsTRTConfig.yolo_thresh, utils/params.py:34), or
 * (-1,-1,-1,-1); d_found [B] u8 (may be NULL). The NMS of the reference cannot change this winner.
 * A (-1,...) box makes isb_hpe_forward report valid = 0 for that frame (estimate() -> None). */
int isb_hpe_select_person(isb_hpe* h, const float* d_boxes, const float* d_confs, int32_t B, float conf_thresh,
                          int32_t* d_bbox, uint8_t* d_found, void* stream);
int isb_hpe_select_person_host(isb_hpe* h, const float* h_boxes, const float* h_confs, int32_t B, float conf_thresh,
                               int32_t* h_bbox, uint8_t* h_found);

/* ------------------------------------------------------------------------------------------
 * YOLOv4 person detector
 *   replaces  Runner(model_config.yolo_engine_path)        modules/hpe/hpe.py:42   (engine load)
 *             the pre-processing + self.yolo(yolo_in) call modules/hpe/hpe.py:51-60
 *   contract of the engine: modules/hpe/setup/1_extract_yolo_onnx.py:21-25,44-60 (Yolov4(n_classes=80, inference=True) of the
 *   un-vendored Tianxiaomo/pytorch-YOLOv4 at 256 x 256; neither definition nor weights are in the reference tree: the network
 *   here is the public architecture, "parity unpinned").
 * Input: the BGR frames main.py:74 hands to estimate(); they are area-resized to 256 x 256 (cv2 INTER_AREA, hpe.py:51), turned
 * to RGB and divided by 255 (hpe.py:53-56) on the device. Output: exactly what hpe.py:60 reshapes the engine outputs to --
 *   d_boxes [B,4032,1,4] f32 (x1,y1,x2,y2 normalised), d_confs [B,4032,80] f32 -- ready for isb_hpe_select_person.
 * ---------------------------------------------------------------------------------------- */
typedef struct isb_det isb_det;

typedef struct isb_det_cfg {
    int32_t width, height;    /* frame size (640 x 480) */
    int32_t device;           /* HIP device ordinal */
    int32_t max_batch;        /* frames per internal micro-batch; 0 = 16; clamped to 128 */
} isb_det_cfg;

int isb_det_create(const isb_det_cfg* cfg, isb_det** out);
void isb_det_destroy(isb_det* h);
/* the 110 convolutions in module order: name (e.g. "down3.resblock.module_list.4.1"; the blob tensors are
 * yolo.<name>.{w,scale,shift}), dims = {cin, cout, k, stride, act (0 none, 2 Mish, 3 LeakyReLU 0.1), has_batchnorm} */
int isb_det_n_convs(void);
int isb_det_describe(int32_t idx, char* name, int32_t name_cap, int32_t* dims);
/* ISBW blob with yolo.* (isbfsar_amd/yolov4.py: folded BatchNorm, [cout,k,k,cin] kernels). Synchronous. */
int isb_det_load_weights(isb_det* h, const void* h_blob, size_t nbytes);
int isb_det_forward(isb_det* h, const uint8_t* d_frames, int32_t B, float* d_boxes, float* d_confs, void* stream);
int isb_det_forward_host(isb_det* h, const uint8_t* h_frames, int32_t B, float* h_boxes, float* h_confs);
/* test hook (B <= max_batch): the pre-processed image f32 [B,256,256,3] RGB and the three raw detection maps
 * f32 [B,32,32,256], [B,16,16,256], [B,8,8,256] (255 channels used); any output may be NULL */
int isb_det_debug_host(isb_det* h, const uint8_t* h_frames, int32_t B, float* h_image, float* h_map8, float* h_map16, float* h_map32);

/* stage-level entry points (host buffers, synchronous, B <= max_batch): each stage can be pinned
 * against the oracle on its own.
 *   crop_params : misc.homography + hpe.py:96       -> H f32 [B,9], new_K f64 [B,9], R f64 [B,9]
 *   warp        : hpe.py:97-100                      -> crops f32 [B,256,256,3] in [0,1], BGR
 *   backbone    : hpe.py:103,106                     crops -> features f32 [B,8,8,1280] (may be NULL),
 *                                                              head logits f32 [B,8,8,288] (may be NULL)
 *   post        : hpe.py:109-169                     head logits + bbox -> joints, valid,
 *                                                    optional pred f64 [B,32,5] = pred2d(x,y), pred3d(x,y,z) */
/* Test-time augmentation as far as the reference executes it (hpe.py:88-100, MetrabsTRTConfig.num_aug, params.py:36):
 * each box yields n_aug parameter sets, new_K[k][:2,:2] *= scales[k] and homo_inv[k] = rotflip[k] @ homo_inv, and
 * n_aug crops. The tables are what misc.py:312-327 (get_augmentations) returns: rotflip f64 [n_aug,3,3], scales f64
 * [n_aug]; n_aug = 0 switches it off. With augmentation on, isb_hpe_crop_params_host / isb_hpe_warp_host return
 * B x n_aug items ([B][n_aug] order) and the full forward is refused (ISB_ERR_STATE): the reference's decode reshapes
 * the head output to ONE sample (hpe.py:108), so nothing downstream of the crops is defined there. */
int isb_hpe_set_augmentations(isb_hpe* h, int32_t n_aug, const double* h_rotflip, const double* h_scales);
int isb_hpe_crop_params_host(isb_hpe* h, const int32_t* h_bbox, int32_t B, float* h_H, double* h_newK, double* h_R);
int isb_hpe_warp_host(isb_hpe* h, const uint8_t* h_frames, const int32_t* h_bbox, int32_t B, float* h_crops);
int isb_hpe_backbone_host(isb_hpe* h, const float* h_crops, int32_t B, float* h_features, float* h_logits);
int isb_hpe_post_host(isb_hpe* h, const float* h_logits, const int32_t* h_bbox, int32_t B, float* h_joints,
                      uint8_t* h_valid, double* h_pred);

/* device time of the dominant kernel family (conv_igemm_kernel, every launch of a forward pass),
 * HIP events on the launch stream; used by bench.py's roofline object */
int isb_hpe_profile(isb_hpe* h, int32_t enable);
int isb_hpe_profile_read(isb_hpe* h, double* ms_total, int64_t* launches);
/* the same for the STAND-ALONE depthwise launches of the profiled passes (the fused MBConv fronts carry their depthwise work inside
 * the family's launches): family + depthwise time is what stays comparable when a fusion moves work from one list to the other */
int isb_hpe_profile_read_dw(isb_hpe* h, double* ms_total, int64_t* launches);

/* test / tuning hook: ONE backbone convolution (the conv_igemm kernel family) on host tensors.
 *   h_x bf16 [B,H,W,Cin], h_w f32 [Cout,k,k,Cin], folded BN h_scale/h_shift [Cout], optional residual
 *   h_res bf16 [B,OH,OW,Cout] and squeeze-excite gate h_gate f32 [B,Cin] (1x1 only), act 1 = SiLU,
 *   variant 0 = automatic tile choice; variant = 1000 * splits + v (splits >= 2, v a gemm1x1 variant 131-148) runs
 *   the split-K GEMM + reduction pair; variant = 900000 + v (v = 131-148, 181) runs the kernel with in-kernel s_memtime
 *   stamps and prints the phase clocks of its first workgroups to stderr (tuning probe);
 *   out bf16 [B,OH,OW,Cout]; ms_per_iter = HIP-event time of one launch. 
 *   stride | 0x100 (with stride 2, k 3): PyTorch's symmetric padding 1 instead of TF-SAME (bottom / right) -- the detector's and
 *   the ResNet trunk's down-sampling layers;
 *   act | 0x100: fp16 operands (ConvArgs.f16): h_x / h_res / h_out hold fp16 bits and the weights are rounded to fp16;
 *   implemented by the variants the 8x8 stages select (131, 132, 138, 141, 144, 146, 147, 185, 186; 0 = automatic). */
int isb_debug_conv(int32_t device, const uint16_t* h_x, const float* h_w, const float* h_scale, const float* h_shift,
                   const uint16_t* h_res, const float* h_gate, int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout,
                   int32_t k, int32_t stride, int32_t act, int32_t variant, int32_t iters, uint16_t* h_out,
                   float* ms_per_iter);

/* tuning probe of the fused 8 x 8 chain (conv_mb8.hip): enable != 0 arms in-kernel s_memtime stamps (the following forward passes
 * write them), enable == 0 copies them to host_out ([32 workgroups][2 blocks][32 marks] uint64_t: marks 0-9 of wave 0, 16-25 of the
 * last wave; block 0 = a 384 -> 2304 -> 384 block, block 1 = a 640 -> 3840 -> 640 block) and disarms. tools/exp_mb8.py */
int isb_debug_hpe_mb8_stamps(isb_hpe* h, int32_t enable, uint64_t* host_out);

/* tuning probe of the all-classes attention pass (ar_kernels.hip ar_proto_kernel): enable != 0 arms in-kernel s_memtime stamps (the
 * following isb_ar_forward calls write them), enable == 0 copies them to host_out ([64 workgroups][8 waves][4] uint64_t: cycles of
 * the prologue, of the tile loop, of the epilogue, and the number of tiles the wave walked) and disarms. tools/exp_ar_stamps.py */
int isb_debug_ar_stamps(isb_ar* h, int32_t enable, uint64_t* host_out);

/* test / tuning hook: a whole Fused-MBConv block (3x3 expand + BN + SiLU -> 1x1 project + BN [+ residual]) in ONE
 * launch on host tensors. h_x bf16 [B,H,H,Cin], h_w1 f32 [Cexp,3,3,Cin] (Cexp = 128, 256 or 384), h_w2 f32 [Cout2,Cexp]
 * (Cout2 <= 128), optional residual h_res bf16 [B,H/stride,H/stride,Cout2]; out bf16 of that shape. */
int isb_debug_fused_mb(int32_t device, const uint16_t* h_x, const float* h_w1, const float* h_scale1, const float* h_shift1,
                       const float* h_w2, const float* h_scale2, const float* h_shift2, const uint16_t* h_res, int32_t B,
                       int32_t H, int32_t Cin, int32_t Cexp, int32_t Cout2, int32_t stride, int32_t iters, uint16_t* h_out,
                       float* ms_per_iter);

/* test / tuning hook: the exact-f32 Linear kernel (gemm_f32.hip) on host tensors:
 *   C[m,n] = act( bias[n] + sum_k a_act( sum_{s < a_parts} A[s][m][k] + a_bias[k] ) (+ a_add[m % add_period][k]) * W[n][k] )
 *   A f32 [a_parts,M,K], W f32 [N,K] (torch Linear layout), bias [N] / a_bias [K] / a_add [add_period,K] optional (NULL);
 *   act / a_act: 0 none, 1 ReLU, 2 SiLU, 3 sigmoid; splits > 1 = split-K with the fixed-order reduction; a_offset (0..3):
 *   A is placed that many floats past a 16-byte boundary on the device (the 8-byte and 4-byte staging paths) */
int isb_debug_gemm_f32(int32_t device, const float* h_A, const float* h_W, const float* h_bias, const float* h_a_bias,
                       const float* h_a_add, int32_t M, int32_t N, int32_t K, int32_t a_parts, int32_t a_act,
                       int32_t add_period, int32_t act, int32_t splits, int32_t a_offset, int32_t iters, float* h_C,
                       float* ms_per_iter);

/* test / tuning hook: the depthwise 3x3 (+ folded BN + SiLU) + squeeze-excite mean kernel on host tensors.
 *   h_x bf16 [B,H,H,C], h_w f32 [C,3,3] (taps are rounded to bf16 after the BN scale is folded in, like every
 *   conv weight), stride 1 (pad 1) or 2 (TF SAME: pad bottom/right); out bf16 [B,H/stride,H/stride,C], pooled f32 [B,C].
 *   stride | 0x100: h_x and the taps are fp16; stride | 0x200: out is fp16 (forms of the fp16 stages: stride 1 with both,
 *   stride 2 with 0x200 alone); stride | 0x400: the general kernel also on 8 x 8 maps (which otherwise take the LDS-staged
 *   dwconv3x3_map8_kernel) */
int isb_debug_dwconv(int32_t device, const uint16_t* h_x, const float* h_w, const float* h_scale, const float* h_shift,
                     int32_t B, int32_t H, int32_t C, int32_t stride, int32_t iters, uint16_t* h_out, float* h_pooled,
                     float* ms_per_iter);
/* the same launch with the squeeze-excite FC1 riding in it (DwArgs.se_w1): h_se_w1 f32 [cse,C], cse <= 160;
 *   h_se_part f32 [*n_parts][B][cse] receives each channel slab's partial sums (room for 32 slabs), *n_parts their count */
int isb_debug_dwconv_fc1(int32_t device, const uint16_t* h_x, const float* h_w, const float* h_scale, const float* h_shift,
                         int32_t B, int32_t H, int32_t C, int32_t stride, int32_t iters, uint16_t* h_out, float* h_pooled,
                         float* ms_per_iter, const float* h_se_w1, int32_t cse, float* h_se_part, int32_t* n_parts);

/* test / tuning hook: the FRONT half of a stride-1 MBConv block -- 1x1 expand + folded BN + SiLU -> depthwise 3x3 + folded BN + SiLU ->
 * D [B,hw,hw,cexp] (what the gated projection reads) + the squeeze-excite pool [B,cexp] -- on host tensors, three ways that must give the
 * same bits: form 0 = two launches (expand GEMM, matrix-pipe depthwise kernel), 1 = the fused fronts of rounds 4 / 5 (every wave does
 * everything: two / three waves per SIMD), 2 = the fused fronts with producer / consumer waves (round 6: three / four waves per SIMD).
 *   hw = 16: h_x 16-bit [B,16,16,cin] with cin 192 / 224, cexp a multiple of 32; hw = 8: cin 384, cexp 2304.
 *   h_w1 f32 [cexp,cin], h_dww f32 [cexp,3,3]; f16: the 16-bit type is IEEE fp16, else bf16.
 *   form | 0x100 (hw 16, form 2, cin 224, fp16): in-kernel clocks of the tick loops to stderr. */
int isb_debug_mbfront(int32_t device, int32_t hw, const uint16_t* h_x, const float* h_w1, const float* h_scale1, const float* h_shift1,
                      const float* h_dww, const float* h_dwscale, const float* h_dwshift, int32_t B, int32_t cin, int32_t cexp,
                      int32_t f16, int32_t form, int32_t iters, uint16_t* h_d, float* h_pooled, float* ms_per_iter);

/* test hook: the squeeze-excite FCs of a batch on host tensors -- gate[b][c] = sigmoid(b2[c] + sum_j silu(b1[j] + sum_c' pooled[b][c']
 * w1[j][c']) w2t[j][c]) with the fixed summation orders of the pose path (256-channel chunks, four hidden-unit quarters; a sample's gate
 * does not depend on the batch it arrives in).
 *   h_pooled f32 [B,C], h_w1 f32 [cse,C], h_b1 [cse], h_w2t f32 [cse,C], h_b2 [C] -> h_gate f32 [B,C]; C a multiple of 4, at most 3840; cse <= 160. */
int isb_debug_se_fcs(int32_t device, const float* h_pooled, const float* h_w1, const float* h_b1, const float* h_w2t, const float* h_b2,
                     int32_t B, int32_t C, int32_t cse, int32_t iters, float* h_gate, float* ms_per_iter);

/* ------------------------------------------------------------------------------------------
 * Glue between the two stages (main.py:102-105 + ar.py:42-50): root-centre every pose on joint 0,
 * flatten to 3J and cut sliding windows of L consecutive frames per camera.
 *   d_joints  [n_cam, n_frames, J, 3] f32
 *   d_windows [n_cam * (n_frames - L + 1), L, 3J] f32   (camera-major, oldest frame first)
 * No handle: the launch goes to the device that owns d_joints (hipPointerGetAttributes).
 * ---------------------------------------------------------------------------------------- */
int isb_pose_windows(const float* d_joints, int32_t n_cam, int32_t n_frames, int32_t J, int32_t L,
                     float* d_windows, void* stream);

/* The frame's "distance" element (main.py:102): d_distance[i] = ||d_joints[i, 0, :]|| * 2.5 (camera-frame root joint,
 * evaluated in float64 like the reference's numpy expression on the float64 pose).
 *   d_joints [n, J, 3] f32 (absolute joints as isb_hpe_forward writes them)    d_distance [n] f32 */
int isb_pose_distance(const float* d_joints, int32_t n, int32_t J, float* d_distance, void* stream);

/* ------------------------------------------------------------------------------------------------
 * ResNet-50 trunk of the RGB / hybrid input types: the reference's
 * nn.Sequential(*list(resnet50(pretrained=True).children())[:-1]) (modules/ar/utils/model.py:270-277), run per frame on the
 * 224 x 224 person crop main.py:85-92 prepares. Images -> "trunk features" [N, 2048] f32 = the input of PostResNet, which
 * lives inside isb_ar_infer_hybrid / isb_ar_set_support_hybrid. Public torchvision architecture, BatchNorm folded, bf16
 * storage / f32 accumulate; neither torchvision nor its weights are in the reference tree: parity unpinned.
 *   d_images f32 [N,3,224,224] (nchw = 1: the layout main.py:91 produces) or [N,224,224,3] (nchw = 0)
 *   blob: rgb.conv1.{w,scale,shift}, rgb.layer{1..4}.{i}.conv{1,2,3}.*, rgb.layer{l}.0.downsample.* (isbfsar_amd/resnet50.py) */
typedef struct isb_rgb isb_rgb;
typedef struct isb_rgb_cfg {
    int32_t device;
    int32_t max_batch;        /* images per internal micro-batch; 0 = 64, clamped to 512 */
} isb_rgb_cfg;
int isb_rgb_create(const isb_rgb_cfg* cfg, isb_rgb** out);
void isb_rgb_destroy(isb_rgb* r);
int isb_rgb_load_weights(isb_rgb* r, const void* h_blob, size_t nbytes);
int isb_rgb_forward(isb_rgb* r, const float* d_images, int32_t N, int32_t nchw, float* d_trunk, void* stream);
int isb_rgb_forward_host(isb_rgb* r, const float* h_images, int32_t N, int32_t nchw, float* h_trunk);

/* ------------------------------------------------------------------------------------------------
 * Multi-GPU: one process per GPU, frames / windows sharded across ranks with no data-path collective
 * (the reference has no multi-GPU path; its units are independent: SURVEY.md 8e). The ONE exchange is an
 * all-gather of the packed per-window records [logits(n) | is_true(1) | embed(L*256, optional)] so that
 * every rank holds every window's result. The library owns the RCCL communicator (SURVEY.md 8b
 * ownership); the collective is issued on the caller's stream and may be captured in a hipGraph together
 * with the step that produces the records (BASELINE configs[4]).
 *   rank 0: isb_dist_unique_id(id); ship the 128 bytes to the other ranks by any channel (the drop-in uses
 *   the launcher's torch.distributed group); every rank: isb_dist_create(id, rank, world, device, &h).
 * isb_dist_create is collective (ncclCommInitRank). RCCL is bound with dlopen when first needed: without
 * librccl.so these entry points return ISB_ERR_STATE and nothing else in the library is affected. */
typedef struct isb_dist isb_dist;
#define ISB_DIST_ID_BYTES 128
int isb_dist_unique_id(void* h_id_out /* ISB_DIST_ID_BYTES */);
int isb_dist_create(const void* h_unique_id, int32_t rank, int32_t world, int32_t device, isb_dist** out);
void isb_dist_destroy(isb_dist* d);
/* ranks in the communicator as RCCL reports them (ncclCommCount) -- cited by bench.py's config.rccl_ranks */
int isb_dist_comm_count(const isb_dist* d, int32_t* n_ranks);
int isb_dist_info(const isb_dist* d, int32_t* rank, int32_t* world);
/* d_recv[r * bytes_per_rank .. ] = rank r's d_send[0 .. bytes_per_rank) for every r; asynchronous on `stream`
 * (ncclAllGather over xGMI). Equal block sizes on every rank (ragged shards: pad to the largest, as the
 * host-side helper isbfsar_amd/dist.py does). d_send may alias its own slot of d_recv. */
int isb_dist_all_gather(isb_dist* d, const void* d_send, void* d_recv, size_t bytes_per_rank, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ISBFSAR_H */
