/*
 * pngpd.h - C ABI of the MI355X (gfx950) grasp-scoring library.
 *
 * The reference (Hymwgk/PointNetGPD) has no FFI of its own: the hot path sits behind three
 * PyTorch modules (model/pointnet.py) and a numpy crop loop (apps/kinect2grasp.py).  This header
 * is what a binding for that path calls instead; every entry point cites the reference code whose
 * arithmetic it replaces.
 *
 * Conventions
 *  - Every pointer named *_dev is caller-owned DEVICE memory on the device that was current when
 *    the model was created.  The library never frees or keeps caller buffers.
 *  - `stream` is a hipStream_t passed as void*; all work is stream-ordered on it, nothing
 *    synchronises the host unless stated.
 *  - Return value: PNGPD_OK (0) or an error code; pngpd_last_error() gives the message of the
 *    last failure on the calling thread.  No C++ exception crosses this boundary.
 *  - Shapes follow the reference: P points per candidate must equal the model's num_points
 *    (model/pointnet.py:15,132 MaxPool1d(num_points) fails otherwise) and input_chann is 3
 *    (model/pointnet.py:139, 3x3 bmm).
 */
#ifndef PNGPD_H
#define PNGPD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
    PNGPD_OK = 0,
    PNGPD_ERR_INVALID = 1,   /* null pointer / bad enum / bad size                  */
    PNGPD_ERR_SHAPE = 2,     /* P != num_points, B < 1, k out of range ...           */
    PNGPD_ERR_HIP = 3,       /* a HIP runtime call failed (message has the reason)   */
    PNGPD_ERR_STATE = 4,     /* weights not loaded, model part missing               */
    PNGPD_ERR_WORKSPACE = 5  /* workspace smaller than pngpd_*_workspace_bytes()     */
};

/* arithmetic of the two fused conv stacks:
 *   PNGPD_FP32    exact-f32 MFMA (every product an f32 fma chain);
 *   PNGPD_BF16    bf16 operands, f32 accumulation: the error of a log-probability is about 0.4 % of the class
 *                 scores' magnitude;
 *   PNGPD_BF16X3  split-bf16 operands (hi + lo pieces, three bf16 MFMAs per product, f32 accumulation): f32-grade
 *                 results (operands exact to 2^-18) at a third of the bf16 rate.
 * The FC layers give f32-grade results in every mode (f32 MFMA in FP32, split-bf16 products otherwise). */
enum { PNGPD_FP32 = 0, PNGPD_BF16 = 1, PNGPD_BF16X3 = 2 };

/* memory layout of the candidate points */
enum {
    PNGPD_LAYOUT_C3P = 0, /* [B,3,P]  what the reference modules take (model/pointnet.py:185) */
    PNGPD_LAYOUT_P3 = 1   /* [B,P,3]  what the crop emits (apps/kinect2grasp.py:423,768)       */
};

/* which sub-networks a model owns */
enum {
    PNGPD_PART_STN = 1,   /* STN3d        model/pointnet.py:8-45     */
    PNGPD_PART_TRUNK = 2, /* PointNetfeat model/pointnet.py:122-152  */
    PNGPD_PART_HEAD = 4   /* PointNetCls  model/pointnet.py:174-190  */
};

/*
 * Weight table passed to pngpd_model_load(): 12 layer units x 6 fields, host float32 pointers.
 * slot = unit * PNGPD_FIELDS + field.  Units without BatchNorm (the two fc3) leave the four bn
 * fields NULL.  Conv weights are [Cout,Cin,1] and linear weights [out,in], both contiguous, i.e.
 * exactly the tensors of the reference state_dict (model/pointnet.py:12-25,126-131,179-183).
 */
enum {
    PNGPD_U_STN_CONV1 = 0, PNGPD_U_STN_CONV2, PNGPD_U_STN_CONV3,
    PNGPD_U_STN_FC1, PNGPD_U_STN_FC2, PNGPD_U_STN_FC3,
    PNGPD_U_FEAT_CONV1, PNGPD_U_FEAT_CONV2, PNGPD_U_FEAT_CONV3,
    PNGPD_U_FC1, PNGPD_U_FC2, PNGPD_U_FC3,
    PNGPD_UNITS
};
enum {
    PNGPD_F_WEIGHT = 0, PNGPD_F_BIAS, PNGPD_F_BN_WEIGHT, PNGPD_F_BN_BIAS, PNGPD_F_BN_MEAN,
    PNGPD_F_BN_VAR, PNGPD_FIELDS
};
#define PNGPD_SLOTS (PNGPD_UNITS * PNGPD_FIELDS)

typedef struct pngpd_model pngpd_model;

/* Message of the last error on this thread ("" if none). */
const char* pngpd_last_error(void);

/* ABI version of this header; bumped on any incompatible change. */
int pngpd_abi_version(void);

/*
 * Create / destroy a model for (num_points, k) on the CURRENT device.
 * Replaces the constructors STN3d / PointNetfeat / PointNetCls (model/pointnet.py:9,123,175);
 * `parts` is a mask of PNGPD_PART_*.
 */
int pngpd_model_create(pngpd_model** out, int num_points, int k, int parts);
int pngpd_model_destroy(pngpd_model* m);

/*
 * Fold eval-mode BatchNorm (eps 1e-5, running statistics) into the preceding conv/linear layer in
 * float64, permute into MFMA fragment order (fp32 and bf16 copies) and upload.  Replaces what
 * load_state_dict + .eval() establish for model/pointnet.py:30-38,142-145,187-189.
 * host_tensors: PNGPD_SLOTS host pointers (see above).  Synchronises `stream` before returning
 * (the host staging buffers are released).
 */
int pngpd_model_load(pngpd_model* m, const float* const* host_tensors, int n_slots, void* stream);

/*
 * The same for parameters that already live in device memory (a network that was just trained on this GPU;
 * model.eval() after the training loop of main_1v.py:262-274): device_tensors is a HOST array of PNGPD_SLOTS
 * DEVICE pointers laid out like the table above.  Folding and the fragment images are computed by kernels queued on
 * `stream` - the same float64 operations in the same order as pngpd_model_load, so both loaders give bit-identical
 * models - and the call returns without waiting: the tensors must stay valid and unchanged until the stream has
 * passed this point.  Every pointer is looked up before anything is launched (a device allocation on the model's
 * GPU that holds the tensor's extent, else PNGPD_ERR_INVALID).
 */
int pngpd_model_load_device(pngpd_model* m, const float* const* device_tensors, int n_slots, void* stream);

/* Bytes of device workspace pngpd_forward_infer() needs for a batch of B candidates. */
size_t pngpd_forward_workspace_bytes(const pngpd_model* m, int B);

/*
 * Eval-mode forward of the loaded parts for B candidates.
 *   PART_STN only : trans_dev                       = STN3d.forward        (model/pointnet.py:28-45)
 *   STN|TRUNK     : feat_dev, trans_dev             = PointNetfeat.forward (model/pointnet.py:135-149)
 *   all three     : logp_dev, trans_dev (, feat_dev)= PointNetCls.forward  (model/pointnet.py:185-190)
 * x_dev: float32, layout PNGPD_LAYOUT_*; P must equal num_points.
 * logp_dev [B,k] log-probabilities, trans_dev [B,3,3], feat_dev [B,1024]; any of them may be NULL
 * when the caller does not want it.  `precision` is one of PNGPD_FP32, PNGPD_BF16, PNGPD_BF16X3.
 */
int pngpd_forward_infer(const pngpd_model* m, const float* x_dev, int layout, int B, int P,
                        int precision, float* logp_dev, float* trans_dev, float* feat_dev,
                        void* workspace_dev, size_t workspace_bytes, void* stream);

/*
 * pngpd_forward_infer without a workspace argument and without any allocation.  It has no counterpart in the
 * reference; it is for callers that cannot allocate device memory: code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc.  The per-batch intermediates (11.3 KB per candidate, plus 256 KB for the
 * partial maxima of batches of at most 8 candidates, whose conv stacks are also cut over points) live in a device
 * array inside the library, PNGPD_NOALLOC_MAX_B * 11.3 KB + 256 KB = 3.2 MB, which every GPU the library runs on holds
 * until the process ends; its address is looked up once per device in pngpd_model_create.  The call itself
 * allocates and frees nothing.  Same arguments, kernels and results (bit for bit) as pngpd_forward_infer;
 * B > PNGPD_NOALLOC_MAX_B returns PNGPD_ERR_SHAPE.  All calls on one GPU share the one array, so two forwards
 * that overlap on a GPU would race; calls on one stream are ordered and safe, and a call on ANOTHER stream (or from
 * another thread) while the previous forward has not finished on the GPU is refused with PNGPD_ERR_STATE (checked
 * with an event created in pngpd_model_create and recorded behind every forward; the check is skipped while the
 * stream is being captured into a graph, where the caller orders the launches).
 */
#define PNGPD_NOALLOC_MAX_B 256
int pngpd_forward_infer_noalloc(const pngpd_model* m, const float* x_dev, int layout, int B, int P,
                                int precision, float* logp_dev, float* trans_dev, float* feat_dev,
                                void* stream);

/*
 * softmax over the log-probabilities and first-index arg-max, i.e. the tail of
 * test_network() (main_test.py:80-83) for a whole batch.
 * probs_dev [B,k] float32, pred_dev [B] int64.
 */
int pngpd_score(const float* logp_dev, int B, int k, float* probs_dev, int64_t* pred_dev,
                void* stream);

/* ------------------------------------------------------------------------------------------
 * In-gripper crop: collect_pc / check_collision_square (apps/kinect2grasp.py:377-438, 299-344,
 * way="p_open").  float64 arithmetic in the reference's operation order; the index set is
 * bit-exact and ascending (np.where order, :340).
 *
 *   scene_dev   float32 [S,3]   scene cloud (apps/kinect2grasp.py:159)
 *   grasps_dev  float64 [G,5,3] the array collect_pc reshapes (:389): row 4 bottom centre,
 *                               rows 1,2,3 approach / binormal / minor axis (not yet unit length)
 *   limits      6 host doubles  ylo, yhi, zlo, zhi, xlo, xhi = p[1].y, p[2].y, p[4].z, p[1].z,
 *                               p[8].x, p[4].x of the gripper corner table (:323, 332-337)
 *
 * Call order: pngpd_crop_count -> (read offsets_dev[G] = total, size idx_dev) -> pngpd_crop_emit
 * -> pngpd_crop_gather.  The three calls share `workspace_dev` (normalised frames + the scene binned into
 * a uniform grid); it must stay untouched between them.  pngpd_crop_emit takes the same `limits` and the
 * counts pngpd_crop_count wrote.
 * ------------------------------------------------------------------------------------------ */
size_t pngpd_crop_workspace_bytes(int S, int G);

/* cnt_dev int32 [G] = points inside each gripper; offsets_dev int64 [G+1] = exclusive prefix. */
int pngpd_crop_count(const float* scene_dev, int S, const double* grasps_dev, int G,
                     const double* limits, void* workspace_dev, size_t workspace_bytes,
                     int32_t* cnt_dev, int64_t* offsets_dev, void* stream);

/*
 * Same crop for frames the caller has already built: the training-side collect_pc_
 * (model/dataset.py:767-924) derives its frame from an 8-float grasp record and a 4x4 view
 * transform and applies it WITHOUT re-normalising.  frames_dev float64 [G,12] = origin (bottom_,
 * :825) followed by the rows of `matrix` (:835: approach_, binormal_, minor_normal_);
 * pc_c2g = matrix (pc - bottom_) (:843) and the six plane tests (:870-880) are those of the
 * online crop.  Same call order, workspace and outputs as pngpd_crop_count.
 */
int pngpd_crop_count_frames(const float* scene_dev, int S, const double* frames_dev, int G,
                            const double* limits, void* workspace_dev, size_t workspace_bytes,
                            int32_t* cnt_dev, int64_t* offsets_dev, void* stream);

/*
 * The frames of the training-data crop, built on the device from the dataset's grasp records: the frame
 * construction at the head of collect_pc_ (model/dataset.py:776-835).  records_dev float64 [G, record_stride]
 * with fields 0-2 contact centre, 3-5 closing axis, 6 width, 7 angle (record_stride >= 8; the .npy records of the
 * dataset carry two score fields behind them); transform: 16 host doubles, the view's 4x4 matrix row-major (:825-833);
 * hand_depth = ags.gripper.hand_depth (:815).  frames_dev float64 [G,12] as pngpd_crop_count_frames takes them.
 * sin / cos of the angle are the device library's: frames agree with a host float64 evaluation to a few units in
 * the last place, not bit for bit.
 */
int pngpd_dataset_frames(const double* records_dev, int G, int record_stride, const double* transform,
                         double hand_depth, double* frames_dev, void* stream);

/* idx_dev int32 [idx_capacity]: for grasp g the ascending scene indices sit at
 * offsets[g] .. offsets[g+1]-1 (in_ind_, :413); entries at or beyond idx_capacity are dropped.
 * pts_g_dev (optional) float64 [idx_capacity,3]: the same points in the gripper frame
 * (in_ind_points_, :423). */
int pngpd_crop_emit(const float* scene_dev, int S, int G, const double* limits, const void* workspace_dev,
                    size_t workspace_bytes, const int32_t* cnt_dev, const int64_t* offsets_dev,
                    int32_t* idx_dev, int64_t idx_capacity, double* pts_g_dev, void* stream);

/*
 * Subsample / pad every grasp to P points, `repeat` times (apps/kinect2grasp.py:746-774), and
 * write the network input float32 [G*repeat, P, 3] (gripper frame, rounded from float64 like
 * torch.FloatTensor(local_pc), main_test.py:76).  Grasps with fewer than min_points (20, :746)
 * points get valid_dev[g] = 0 and zeros.
 * Selection: sel_dev int32 [G*repeat, P] gives, per output point, the position inside the
 * grasp's index list (what np.random.choice returned in the reference); NULL = draw on the
 * device with the library's counter-based generator keyed by (seed, first_grasp_id + grasp,
 * repeat, i): without replacement when cnt >= P, with replacement otherwise (first_grasp_id lets
 * a rank that holds a slice of the grasps draw what a single GPU would).  sel_out_dev (optional)
 * receives the positions used.
 */
int pngpd_crop_gather(const float* scene_dev, int S, int G, const void* workspace_dev,
                      size_t workspace_bytes, const int32_t* idx_dev, const int64_t* offsets_dev,
                      const int32_t* cnt_dev, int repeat, int P, int min_points,
                      const int32_t* sel_dev, uint64_t seed, int first_grasp_id,
                      float* pts_out_dev, int32_t* sel_out_dev, uint8_t* valid_dev, void* stream);

/*
 * pngpd_crop_count + pngpd_crop_emit + pngpd_crop_gather in one pass for callers that want the network input
 * only (the online loop, apps/kinect2grasp.py:712-817, never looks at the index lists again): no index list in
 * memory and no host read of its total in between.  pts_out_dev, valid_dev as pngpd_crop_gather writes them for
 * the same seed (device-drawn selection); cnt_dev int32 [G] = points inside the gripper.  A grasp holding more
 * than 1536 points does not fit the kernel's on-chip list: *overflow_dev (int32) is then set to 1, that grasp's
 * rows of pts_out_dev are left unwritten and the caller falls back on the three calls above (same workspace).
 */
int pngpd_crop_subsample(const float* scene_dev, int S, const double* grasps_dev, int G, const double* limits,
                         void* workspace_dev, size_t workspace_bytes, int repeat, int P, int min_points,
                         uint64_t seed, int first_grasp_id, float* pts_out_dev, uint8_t* valid_dev,
                         int32_t* cnt_dev, int32_t* overflow_dev, void* stream);

/*
 * The dataset's form of the same step (PointGraspOneViewDataset.__getitem__,
 * model/dataset.py:1098-1111): one draw per grasp, output float32 [G,3,P] (the `.T` of :1106),
 * without replacement only when cnt > P (:1104; cnt == P is drawn WITH replacement, unlike the
 * online rule), grasps with fewer than min_points (min_point_limit = 50, :918) get
 * valid_dev[g] = 0 and zeros (the dataset returns None for them).
 */
int pngpd_crop_gather_dataset(const float* scene_dev, int S, int G, const void* workspace_dev,
                              size_t workspace_bytes, const int32_t* idx_dev,
                              const int64_t* offsets_dev, const int32_t* cnt_dev, int P,
                              int min_points, const int32_t* sel_dev, uint64_t seed,
                              int first_grasp_id, float* pts_out_dev, int32_t* sel_out_dev,
                              uint8_t* valid_dev, void* stream);

/*
 * Repeat vote (apps/kinect2grasp.py:799-817): pred_dev int64 [G*repeat], probs_dev [G*repeat,k]
 * -> label_dev int32 [G] (mode, smallest label on ties), score_dev float32 [G] (mean probability
 * of class k-1 over the repeats that agree with the vote).  valid_dev[g] == 0 gives (0, 0.0).
 */
int pngpd_vote(const int64_t* pred_dev, const float* probs_dev, const uint8_t* valid_dev, int G,
               int repeat, int k, int32_t* label_dev, float* score_dev, void* stream);

/* ------------------------------------------------------------------------------------------
 * Voxel down-sampling of the scene, the step in front of the crop:
 *   get_voxel_fun(points_, n)             apps/kinect2grasp.py:145-150
 *   VoxelGrid(points, n, n, n).compute()  apps/voxelgrid.py:115-160
 * points_dev [S,3] float32 -> the unique float32 centres of the occupied voxels of an n x n x n
 * grid over the point cloud's bounding box made cubic, in np.unique(axis=0) order (lexicographic),
 * written to out_dev [capacity,3]; *count_dev = number of unique centres (rows beyond `capacity`
 * are counted but not written).  Centres are float32.  `edges` selects the type np.linspace forms the bin edges
 * in, which is numpy's choice and not the reference's (the reference does not pin numpy): PNGPD_EDGES_F64 is what
 * numpy 1.x does with two float32 scalars (the releases the reference was written for), PNGPD_EDGES_F32 what
 * numpy >= 2 does; a point within an ulp of an edge, and the last bit of a centre, can differ between the two.
 * n in [1, 1600].  Workspace: pngpd_voxel_workspace_bytes(n) bytes (n^3 / 8 + small).
 * ------------------------------------------------------------------------------------------ */
enum { PNGPD_EDGES_F64 = 0, PNGPD_EDGES_F32 = 1 };
size_t pngpd_voxel_workspace_bytes(int n);
int pngpd_voxel_unique(const float* points_dev, int S, int n, int edges, void* workspace_dev, size_t workspace_bytes,
                       float* out_dev, int64_t capacity, int32_t* count_dev, void* stream);

/* ------------------------------------------------------------------------------------------
 * Training step: main_1v.py:262-274 (model.train(); output,_ = model(data); nll_loss; backward;
 * Adam).  fp32 throughout, train-mode BatchNorm1d with torch defaults (batch statistics, biased
 * variance for normalisation, running stats updated with momentum 0.1 and the unbiased variance,
 * eps 1e-5), max-pool gradient to the first arg-max.
 *
 * params_dev / grads_dev: PNGPD_SLOTS DEVICE pointers laid out like the weight table of
 * pngpd_model_load (weight, bias, bn_weight, bn_bias, bn_mean = running_mean, bn_var =
 * running_var); all 12 units are required.  Forward updates the running statistics in place;
 * backward writes gradients for weight / bias / bn_weight / bn_bias (the bn_mean / bn_var slots of
 * grads_dev are ignored).  The bias of a layer followed by BatchNorm gets an exact 0.
 * The workspace carries the saved activations from forward to backward: same B, P, k, same
 * buffer, nothing else may write it in between.  B >= 2 (main_1v.py:258 skips smaller batches).
 * ------------------------------------------------------------------------------------------ */
size_t pngpd_train_workspace_bytes(int B, int P, int k);

/* x_dev float32 [B,3,P] -> logp_dev [B,k], trans_dev [B,3,3] (optional), feat_dev [B,1024] (optional). */
int pngpd_train_forward(int B, int P, int k, const float* x_dev, float* const* params_dev,
                        int n_slots, float momentum, float* logp_dev, float* trans_dev,
                        float* feat_dev, void* workspace_dev, size_t workspace_bytes, void* stream);

/* The same forward for a model that owns only some parts (PNGPD_PART_*; STN, STN|TRUNK or all three): the train-mode
 * forward of a standalone STN3d (model/pointnet.py:28-45 -> trans_dev) or PointNetfeat (:135-149 -> feat_dev,
 * trans_dev).  Units of params_dev outside `parts` may be NULL; logp_dev is needed only with the head. */
int pngpd_train_forward_parts(int parts, int B, int P, int k, const float* x_dev, float* const* params_dev,
                              int n_slots, float momentum, float* logp_dev, float* trans_dev, float* feat_dev,
                              void* workspace_dev, size_t workspace_bytes, void* stream);

/* The point index that holds each channel's maximum in the two MaxPool1d(num_points) of the last
 * pngpd_train_forward on this workspace (model/pointnet.py:33 in the STN, :147 in the trunk; the first
 * maximum, as max_pool1d reports it): int32 [B,1024] each, either pointer may be NULL.  These are the
 * rows the backward pass routes the pooled gradient to. */
int pngpd_train_argmax(int B, int P, int k, const void* workspace_dev, size_t workspace_bytes,
                       int* stn_arg_dev, int* trunk_arg_dev, void* stream);

/* Read-only view of an activation that the last pngpd_train_forward saved in this workspace for the backward
 * pass (row-major float32; per-point layers as [B*P, C] rows ordered (candidate, point)):
 *   *_H1 [B*P,64], *_H2 [B*P,128]  relu(bn(conv)) of the first two layers (model/pointnet.py:30-31, :143-145)
 *   *_POOLED [B,1024]              output of the max-pool (:33-34, :147)
 *   *_A1 [B,512], *_A2 [B,256]     relu(bn(fc)) of the regressor / classifier (:36-37, :187-188)
 * Their signs are the ReLU masks the backward pass applies. */
enum {
    PNGPD_SAVED_STN_H1 = 0, PNGPD_SAVED_STN_H2 = 1, PNGPD_SAVED_STN_POOLED = 2, PNGPD_SAVED_STN_A1 = 3,
    PNGPD_SAVED_STN_A2 = 4, PNGPD_SAVED_TRUNK_H1 = 5, PNGPD_SAVED_TRUNK_H2 = 6, PNGPD_SAVED_TRUNK_POOLED = 7,
    PNGPD_SAVED_HEAD_A1 = 8, PNGPD_SAVED_HEAD_A2 = 9
};
int pngpd_train_saved(int B, int P, int k, const void* workspace_dev, size_t workspace_bytes, int which,
                      const float** data_dev_out, size_t* count_out);

/* dlogp_dev [B,k]: gradient of the loss w.r.t. the log-probabilities; dtrans_dev [B,3,3] optional
 * gradient w.r.t. the returned transform (the reference discards it, main_1v.py:268). */
int pngpd_train_backward(int B, int P, int k, const float* x_dev, const float* const* params_dev,
                         float* const* grads_dev, int n_slots, const float* logp_dev,
                         const float* dlogp_dev, const float* dtrans_dev, void* workspace_dev,
                         size_t workspace_bytes, void* stream);

/*
 * The same backward pass in up to three calls, for a data-parallel caller that reduces the gradients of one part
 * over the ranks while the next part is still being computed (the capability nn.DataParallel provides at
 * main_1v.py:224, reduce_add of the replicas' gradients).  `parts` is a mask of PNGPD_PART_*; the parts of one
 * mask run in the order HEAD, TRUNK, STN, and successive calls must keep that order because each part consumes the
 * input gradient the one before left in the workspace.  After a call returns, the gradients of its parts (HEAD:
 * units FC1..FC3; TRUNK: FEAT_CONV1..3; STN: STN_CONV1..STN_FC3) are final in stream order.  HEAD | TRUNK | STN in
 * one call is pngpd_train_backward.
 * The same entry point is the backward pass of a model without a head (pngpd_train_forward_parts): parts =
 * TRUNK | STN with dfeat_dev [B,1024], the gradient w.r.t. the global feature (logp_dev / dlogp_dev NULL), or parts
 * = STN with dtrans_dev alone.  dfeat_dev must be NULL when the head is among the parts; with TRUNK but neither
 * HEAD nor dfeat_dev the trunk takes the feature gradient an earlier HEAD call left in the workspace.
 */
int pngpd_train_backward_part(int B, int P, int k, const float* x_dev, const float* const* params_dev,
                              float* const* grads_dev, int n_slots, const float* logp_dev,
                              const float* dlogp_dev, const float* dtrans_dev, const float* dfeat_dev,
                              void* workspace_dev, size_t workspace_bytes, int parts, void* stream);

/* F.nll_loss(logp, target) with mean reduction (main_1v.py:270): loss_dev[1] and its gradient
 * dlogp_dev [B,k].  target_dev int64 [B].  A label outside [0,k), on which F.nll_loss raises, is not used as an
 * index: the loss comes out NaN and that sample contributes no gradient. */
int pngpd_nll_loss(const float* logp_dev, const int64_t* target_dev, int B, int k, float* loss_dev,
                   float* dlogp_dev, void* stream);

/* One torch.optim.Adam step (no weight decay, no amsgrad; main_1v.py:226) over flat buffers of n
 * floats; `step` counts from 1; gradients are multiplied by grad_scale first (1/world_size after
 * a sum all-reduce). */
int pngpd_adam_step(float* params_dev, const float* grads_dev, float* exp_avg_dev,
                    float* exp_avg_sq_dev, int64_t n, float lr, float beta1, float beta2, float eps,
                    int step, float grad_scale, void* stream);

/*
 * Diagnostic: pngpd_forward_infer with the two conv-stack launches of the step bracketed by HIP
 * events recorded on `stream`; waits for the trunk launch to finish and returns the two kernel
 * durations in milliseconds (conv_ms_out[0] = STN stack, [1] = trunk stack).  The model needs a
 * trunk.  Used by bench.py for the roofline figure; no counterpart in the reference.
 */
int pngpd_forward_infer_timed(const pngpd_model* m, const float* x_dev, int layout, int B, int P,
                              int precision, float* logp_dev, float* trans_dev, float* feat_dev,
                              void* workspace_dev, size_t workspace_bytes, float* conv_ms_out,
                              void* stream);

/*
 * Diagnostic: pngpd_train_forward with the fused conv3 product / max / arg-max kernel of each stack bracketed
 * by HIP events on `stream`; waits for the second one and returns the two kernel durations in milliseconds
 * (conv3_ms_out[0] = STN, [1] = trunk).  Used by bench.py for the roofline figure of the training step; no
 * counterpart in the reference.
 */
int pngpd_train_forward_timed(int B, int P, int k, const float* x_dev, float* const* params_dev,
                              int n_slots, float momentum, float* logp_dev, float* trans_dev,
                              float* feat_dev, void* workspace_dev, size_t workspace_bytes,
                              float* conv3_ms_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PNGPD_H */
