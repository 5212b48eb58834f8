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
 *   - `stream` is a hipStream_t passed as void* (NULL = the handle's own stream).  Calls are
 *     asynchronous on that stream unless documented otherwise (the *_host variants synchronise).
 *   - a handle is bound to one device and is not thread safe: one handle per process / rank.
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
 * GEMMs always run on the exact f32 MFMA path. */
#define ISB_AR_PREC_BF16 0     /* bf16 operands, f32 accumulate                      */
#define ISB_AR_PREC_BF16X3 1   /* split-bf16 (hi+lo, 3 MFMA / product), ~2^-16 rel.  */

const char* isb_last_error(void);
int isb_version(void);
/* number of visible HIP devices (does not initialise a context) */
int isb_device_count(void);

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
/* test hook: argmax class per window of the last isb_ar_infer chunk sequence, int32 [B] (model.py:323) */
int isb_ar_last_chosen(isb_ar* h, int32_t* h_chosen, int32_t B);

/* device time of the dominant kernel inside isb_ar_infer (ar_proto_kernel, all-classes mode:
 * S^T tiles, A^T = exp2(S^T - lse2), P^T = V^T A^T, distance), measured with HIP events on the
 * launch stream: enable, run, then read the accumulated milliseconds and launch count.
 * Used by bench.py's roofline object. */
int isb_ar_profile(isb_ar* h, int32_t enable);
int isb_ar_profile_read(isb_ar* h, double* ms_total, int64_t* launches);

#ifdef __cplusplus
}
#endif
#endif /* ISBFSAR_H */
