/*
 * gca_hip.h -- C ABI of libgca_hip.so: the MI355X (gfx950) kernels behind the GCA
 * contrastive pre-training hot path (encoder -> graph block -> head -> MoCo queue / InfoNCE).
 *
 * The reference (ACMMM2021-Anonymous/video-graph-ssl) is pure Python: it has no FFI layer,
 * every "kernel" is an ATen/cuDNN call made through torch.nn.  Each entry point below
 * therefore cites the reference call site (file:line under /root/reference) whose
 * arithmetic it replaces.  Conventions for every function:
 *
 *   - plain device pointers + sizes, fp32 data, NCDHW contiguous activations;
 *   - `stream` is a hipStream_t passed as void* (0 = default stream); nothing synchronises;
 *   - no hidden allocations: workspaces are caller-owned, sizes come from the *_ws_bytes() queries;
 *   - return 0 on success, negative on invalid arguments (GCA_EINVAL) or launch failure
 *     (GCA_ELAUNCH); no exceptions cross the ABI.
 *
 * The host side (Python, ctypes) lives in video-graph-ssl_amd/_hip.py; INTEGRATION.md shows
 * the stub a maintainer of the reference would add.
 */
#ifndef GCA_HIP_H
#define GCA_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GCA_OK 0
#define GCA_EINVAL (-1)
#define GCA_ELAUNCH (-2)

int gca_version(void);

/* Diagnostic: workgroups of the LDS-halo conv kernel with `tm` x `tn` tiles per wave (rows 32 tm, columns 128 tn) in
 * arithmetic `math` that the runtime co-schedules on one CU at `lds_bytes` of dynamic LDS; -1 for a shape the entry does not
 * know.  (DESIGN.md quotes it for the dominant kernel; nothing on the product path calls it.) */
int gca_conv_halo_occupancy(int tm, int tn, int math, int64_t lds_bytes);

/* Arithmetic of the convolution kernels (forward, dgrad, wgrad; tensors stay fp32 in HBM in every mode):
 *   0  f32:     v_mfma_f32_32x32x2_f32, bitwise an fmaf chain (157 TFLOP/s peak)
 *   1  bf16x3:  every fp32 operand is split in the kernel into hi = bf16(x), lo = bf16(x - hi); a product becomes
 *               hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_bf16 with fp32 accumulation: ~2^-17 relative error per
 *               product (16+ of fp32's 24 significand bits; TF32 -- the reference's default on its own GPUs,
 *               torch.backends.cudnn.allow_tf32 -- keeps 11), 5.3x less matrix-pipe time than mode 0.
 *   2  bf16x6:  three parts (x = hi + mid + lo to 2^-27) and the six products hh, hm, mh, hl, lh, mm: the dropped terms
 *               are <= 2^-25 relative, below fp32's own rounding -- fp32-grade results (same parity bars as mode 0 in
 *               tests/) at 2.7x less matrix-pipe time.
 * The default comes from the environment (GCA_CONV_MATH=f32|bf16x3|bf16x6, unset = bf16x6).  GCA_EINVAL for another mode. */
int gca_set_conv_math(int mode);
int gca_get_conv_math(void);

/* ---------------------------------------------------------------------------------------
 * 3D convolution as an MFMA (v_mfma_f32_32x32x2_f32) implicit GEMM, NCDHW, no bias.
 * Replaces every nn.Conv3d on the path: resnet2p1d.py:13-36,162-174; s3d_1.py:40,53,57;
 * resnet.py:14-22,77-78,120-126; temporal_graph.py:46,119-122 (1x1x1); nn.Linear of the heads
 * (project_head.py:23-28,39-50) runs through the same kernels as a 1x1x1 conv on (b,C,1,1,1).
 * ------------------------------------------------------------------------------------- */
typedef struct {
  int32_t N, C, D, H, W;       /* input  (N,C,D,H,W)            */
  int32_t K;                   /* output channels               */
  int32_t kd, kh, kw;          /* kernel                        */
  int32_t sd, sh, sw;          /* stride                        */
  int32_t pd, ph, pw;          /* zero padding                  */
  int32_t OD, OH, OW;          /* output spatial (caller-computed, checked).  Each tensor must hold
                                  < 2^30 elements (32-bit byte offsets inside the kernels). */
  int64_t x_batch_stride;      /* elements between consecutive clips of x; 0 = C*D*H*W (contiguous).
                                  Lets the two views of a (b,6,T,H,W) batch be read in place
                                  (tools/train_video_contrast_dis.py:404 torch.chunk on dim 1).
                                  Honoured by gca_conv_fwd and gca_conv_wgrad; dgrad needs 0. */
  /* Launch tuning (0 = built-in heuristic).  The host measures a few candidates per geometry once
   * -- the reference does the same through cudnn.benchmark = True, tools/train_video_contrast_dis.py:50 --
   * and pins the winner here.  Results are deterministic for a fixed setting. */
  int32_t tune_fwd_bm, tune_fwd_splits;       /* tile rows 32..160 (x128 columns), | 1024 = x256 columns with float4
                                                 gathers (pointwise-in-space convs only), | 2048 = LDS-halo kernel (tile
                                                 box in tune_*_box), 4096 | 64 = stem kernel (forward, C <= 4, stride 2
                                                 and <= 8 taps along W; bf16x3 / bf16x6 / fp16 storage), 8192 | 128 =
                                                 pointwise fp16 GEMM kernel (1x1x1, unit stride, act_f16); split-K factor */
  int32_t tune_dgrad_bm, tune_dgrad_splits;
  int32_t tune_wgrad_splits, tune_wgrad_tile;  /* split-K factor; tile: 0 = heuristic, 1..10 = shapes of the gather kernel (see
                                                * gca_conv_wgrad_cfg), 11 / 12 = streaming temporal kernel (32 / 64 output channels per
                                                * wave), 13 = streaming (1,3,3) kernel, 14 = stem kernel (<= 4 input channels, stride
                                                * (1,2,2)); a tile the geometry / arithmetic does not admit falls back to the heuristic
                                                * (gca_conv_wgrad_cfg reports what will run) */
  int32_t tune_fwd_tail, tune_dgrad_tail;      /* two-phase launch: (short tile rows / 32) | (column tiles run with the tall
                                                  tile << 8); 0 = single launch.  Used only with 128-column tiles, split 1 */
  int32_t tune_fwd_math, tune_dgrad_math, tune_wgrad_math;  /* 0 = the arithmetic set by gca_set_conv_math; 1 + m = run this
                                                  pass with arithmetic m, honoured only when m is at least as accurate as
                                                  the mode in force (f32 > bf16x6 > bf16x3): the mode is a floor on
                                                  accuracy and the host pins whichever admissible kernel is fastest */
  int32_t tune_fwd_box, tune_dgrad_box;        /* LDS-halo kernel (tune_*_bm & 2048): a tile is a box of bd x bh x bw output
                                                  positions inside one clip, coded bd | bh << 8 | bw << 16 (powers of two,
                                                  product 128 or 256); its input window is staged once per 16 channels in
                                                  LDS and every tap is formed from there.  0 with the flag clear */
  int32_t act_f16;                             /* 1: the activation tensors of this convolution (x, y, dy, dx) are IEEE fp16 in
                                                  HBM -- the fp16-storage path (the reference's apex-amp option,
                                                  tools/train_video_contrast_dis.py:134-141,415-417).  Weights stay fp32
                                                  masters (their packed copies are fp16 where the kernel multiplies in fp16),
                                                  accumulation, BatchNorm statistics and dw are fp32.  The LDS-halo kernels
                                                  multiply on v_mfma_f32_32x32x16_f16; the gather and weight-gradient kernels
                                                  widen each fp16 operand to its exact bf16 hi + lo pair (three bf16 MFMAs).
                                                  The conv arithmetic selector (gca_set_conv_math) does not apply. */
} gca_conv_geom;

/* Weight re-layout for the GEMM A operand (k-major, zero padded).  which: 0 = forward
 * ([C*taps -> pad16][K -> pad64]), 1 = dgrad ([K*taps -> pad16][C -> pad64]).
 * gca_conv_pack_elems() gives the float count of the packed buffer. */
int64_t gca_conv_pack_elems(const gca_conv_geom* g, int which);
int gca_conv_pack(const gca_conv_geom* g, int which, const float* w, float* packed, void* stream);

/* Batched form of gca_conv_pack for a whole encoder (one launch instead of one per layer and problem class;
 * the reference re-reads nn.Conv3d.weight in place, so this is part of what replaces cuDNN's per-call filter
 * transform).  gca_conv_pack_jobs_host writes one GCA_PACK_JOB_BYTES record per problem class of (g, which)
 * for weights `w` (device) -> `packed` (device) into HOST memory `jobs_out` and returns the number of records
 * (pass jobs_out = NULL to only count).  The caller concatenates the records of all layers, calls
 * gca_conv_pack_jobs_finalize_host on the concatenation (assigns each job its block range; returns the grid
 * size), copies the records to the device once, and replays gca_conv_pack_batched every step; the weight and
 * packed pointers must stay valid. */
#define GCA_PACK_JOB_BYTES 128
int64_t gca_conv_pack_jobs_host(const gca_conv_geom* g, int which, const float* w, float* packed, void* jobs_out);
int64_t gca_conv_pack_jobs_finalize_host(void* jobs, int64_t njobs);
int gca_conv_pack_batched(const void* jobs_dev, int64_t njobs, int64_t total_blocks, void* stream);

/* Gather table (one int2 per packed k-row: element offset + packed tap deltas); built on the
 * host once per geometry.  which as above, 2 = wgrad (same rows as forward). */
int64_t gca_conv_table_rows(const gca_conv_geom* g, int which);
int gca_conv_table_build_host(const gca_conv_geom* g, int which, int32_t* table_host /* 2*rows ints */);

/* y = conv(x, w) [+ bias[k]].  `wpack` from gca_conv_pack(which=0), `table` device copy of
 * the which=0 table.  If stat_sum/stat_sq are non-NULL the epilogue also writes per-channel
 * partial sums / sums of squares of y (training-mode BatchNorm statistics, fused):
 * layout [K][P], P = gca_conv_fwd_stat_parts(g). */
int64_t gca_conv_fwd_stat_parts(const gca_conv_geom* g);
/* Tooling: the launch configuration in force for which = 0 (fwd) / 1 (dgrad), first non-empty class:
 * out4 = {tile rows, tile columns, split-K factor, classes | tap-mask kind (0 none, 1: <=31 taps, 2: <=62)<<8 | float4-gather<<10 |
 *         arithmetic (0 f32, 1 bf16x3, 2 bf16x6, 3 fp16 MFMA)<<12 | LDS-halo kernel<<14 | fp16 storage<<15 | stem kernel<<16 | pointwise fp16 GEMM kernel<<17}. */
int gca_conv_kernel_cfg(const gca_conv_geom* g, int which, int32_t* out4);
/* Signature of the packed-weight layout that pass `which` (0 fwd, 1 dgrad) reads under the launch configuration in force:
 * one hex digit per problem class (0 = k-major fp32 rows of the gather kernels, 4 + arithmetic = the LDS-halo layout,
 * 1..3 = the stem layout in that arithmetic, 8 = fp16 [rows][channels] of the pointwise GEMM kernel).
 * A buffer packed by gca_conv_pack is valid exactly as long as this value does not change (the host re-packs when a
 * tune_* field moves it). */
int64_t gca_conv_pack_layout(const gca_conv_geom* g, int which);
/* Layers whose output grid cannot fill the 256 CUs split the reduction over workgroups; the fp32
 * partial slabs live in `ws` (gca_conv_fwd_ws_bytes / gca_conv_dgrad_ws_bytes; 0 = not needed, ws may
 * then be NULL) and are summed in a fixed order (deterministic). */
int64_t gca_conv_fwd_ws_bytes(const gca_conv_geom* g);
/* x / y / dy / dx: fp32, or IEEE fp16 when g->act_f16 = 1 (weights, bias, statistics and slabs are fp32 either way). */
int gca_conv_fwd(const gca_conv_geom* g, const void* x, const float* wpack, const int32_t* table,
                 const float* bias, void* y, float* stat_sum, float* stat_sq, void* ws, void* stream);

/* dx (+)= conv_transpose(dy, w).  `wpack`/`table` from which=1.  accumulate != 0 adds into dx. */
int64_t gca_conv_dgrad_ws_bytes(const gca_conv_geom* g);
int gca_conv_dgrad(const gca_conv_geom* g, const void* dy, const float* wpack, const int32_t* table,
                   void* dx, int accumulate, void* ws, void* stream);

/* dw (+)= sum_{n,o} dy[n,k,o] * x[n,c,o*s-p+tap].  `table` from which=2.  Split-K partial slabs go
 * to `ws` (gca_conv_wgrad_ws_bytes) and are reduced deterministically. */
int64_t gca_conv_wgrad_ws_bytes(const gca_conv_geom* g);
int gca_conv_wgrad(const gca_conv_geom* g, const void* x, const void* dy, const int32_t* table,
                   float* dw, int accumulate, void* ws, void* stream);
/* The same in two halves, so that the fixed-order reductions of MANY layers run as ONE launch at the end of the backward
 * pass (the reference gets a weight gradient per cuDNN call; here 39 per-layer reduce launches of 8-12 us are pure
 * latency): gca_conv_wgrad_partial leaves the `*out_splits` fp32 slabs in `slabs` (gca_conv_wgrad_ws_bytes, caller-owned,
 * must stay untouched until reduced); gca_splitk_reduce_batched does dw (+)= sum of slabs for every job, bit-identical
 * to gca_conv_wgrad's own reduction.  Jobs: fill slabs / dw / n (= K*C*taps) / splits / accumulate on the host, call
 * gca_reduce_jobs_finalize_host (assigns block ranges, returns the grid size), copy the records to the device once.
 * Two jobs of one launch must not write the same dw. */
typedef struct {
  const float* slabs;
  float* dw;
  int64_t n;
  int32_t splits, accumulate;
  int32_t first_block, nblocks;     /* filled by gca_reduce_jobs_finalize_host */
} gca_reduce_job;
int gca_conv_wgrad_partial(const gca_conv_geom* g, const void* x, const float* in_scale, const float* in_shift, const void* dy,
                           const int32_t* table, void* slabs, int32_t* out_splits, void* stream);   /* in_scale / in_shift: see below; NULL = x as it is */

/* BatchNorm + ReLU of the PRODUCING layer applied where the consumer reads its input, so the normalised tensor
 * z = relu(y * scale[c] + shift[c]) of a conv -> BN -> ReLU -> conv link (resnet2p1d.py:66-85: bn1_s -> conv1_t, bn1_t -> conv2_s,
 * bn2_s -> conv2_t; :252-253 the stem) is never written or read: 2 of the 7 tensor passes of such a BatchNorm.
 * gca_conv_fwd_xf / gca_conv_wgrad_xf take the PRE-activation tensor y as `x` plus (scale, shift) of gca_bn_finalize
 * (arrays padded to a multiple of 16 floats + 16) and compute exactly what gca_conv_fwd / gca_conv_wgrad compute on z
 * (zero padding pads z).  Only launch configurations that pass their input through registers can do it:
 * gca_conv_xf_ok(g) = 1 when, under the tune_* fields in force, the forward runs on the LDS-halo kernels and the weight
 * gradient on the streaming temporal kernels (fp32 storage, split-product arithmetic); otherwise these entries return
 * GCA_EINVAL and the caller materialises z with gca_bn_apply. */
int gca_conv_xf_ok(const gca_conv_geom* g);
int gca_conv_fwd_xf(const gca_conv_geom* g, const void* x, const float* in_scale, const float* in_shift, const float* wpack,
                    const int32_t* table, const float* bias, void* y, float* stat_sum, float* stat_sq, void* ws, void* stream);
int gca_conv_wgrad_xf(const gca_conv_geom* g, const void* x, const float* in_scale, const float* in_shift, const void* dy,
                      const int32_t* table, float* dw, int accumulate, void* ws, void* stream);
int64_t gca_reduce_jobs_finalize_host(gca_reduce_job* jobs, int64_t njobs);
int gca_splitk_reduce_batched(const gca_reduce_job* jobs_dev, int64_t njobs, int64_t total_blocks, void* stream);
/* Launch shape the wgrad kernel will use for g: out4 = {tile rows (output channels), tile columns (C*taps),
 * split-K factor, shape index | float4 dY loads<<8 | tap-mask kind<<9 | float4 X gathers<<11 | arithmetic<<12}. */
int gca_conv_wgrad_cfg(const gca_conv_geom* g, int32_t* out4);

/* db[k] (+)= sum over (n, spatial) of dy[n,k,:]   (bias gradient of the nn.Linear layers) */
int gca_bias_grad(const float* dy, int64_t N, int64_t K, int64_t SP, float* db, int accumulate, void* stream);

/* ---------------------------------------------------------------------------------------
 * BatchNorm (training mode: batch statistics + running-stat update), fused with ReLU and the
 * residual add.  Replaces nn.BatchNorm3d/1d + ReLU + `out += residual`:
 * resnet2p1d.py:49-57,66-85; s3d_1.py:41-47,54-68; project_head.py:39-50,64-68.
 * Data layout (N, C, SP) with SP = D*H*W (1 for BatchNorm1d).
 * act_f16 = 1: the activation tensors (x, residual, z, dz_in, dx, dres -- the `void*` arguments) are IEEE fp16
 * (the fp16-storage path of BASELINE configs[4]); statistics, parameters and all arithmetic stay fp32/fp64.
 * ------------------------------------------------------------------------------------- */
/* From conv-epilogue partials [C][P] (or computed by gca_bn_stats): mean/invstd (saved for
 * backward), running stats update (momentum, unbiased var), and the folded affine
 * scale = gamma*invstd, shift = beta - mean*scale.  count = N*SP. */
int gca_bn_stats(const void* x, int64_t N, int64_t C, int64_t SP, float* stat_sum, float* stat_sq,
                 int64_t* parts_out, int act_f16, void* stream);          /* P is fixed: gca_bn_stats_parts() */
int64_t gca_bn_stats_parts(int64_t N, int64_t C, int64_t SP);
int gca_bn_finalize(const float* stat_sum, const float* stat_sq, int64_t P, int64_t C, double count,
                    const float* gamma, const float* beta, float eps, float momentum,
                    float* running_mean, float* running_var, int64_t* num_batches_tracked,
                    float* save_mean, float* save_invstd, float* scale, float* shift, void* stream);
/* A conv whose launch shape splits the reduction, followed by a small BatchNorm (N*SP <= 32768), in two launches instead of
 * three.  gca_conv_fwd_slabs runs the forward class and LEAVES the fp32 slabs [splits][K][N*SP] in `ws` (no bias; fp32
 * tensors only; GCA_EINVAL when the pinned launch shape does not split -- gca_conv_kernel_cfg tells beforehand);
 * gca_bn_train_fwd_slabs folds them in the finishing kernel's order (same bits for y), writes y (the backward pass reads
 * it), takes the batch statistics from those values in fp64 and normalises: finalize + apply as gca_bn_train_fwd.
 * Replaces the conv -> BatchNorm link of resnet2p1d.py:66-85 / s3d_1.py:43-47 on the deep, narrow layers. */
int gca_conv_fwd_slabs(const gca_conv_geom* g, const void* x, const float* wpack, const int32_t* table, void* ws,
                       int32_t* out_splits, void* stream);
int gca_bn_train_fwd_slabs(const float* slabs, int64_t splits, double count, const float* gamma, const float* beta, float eps,
                           float momentum, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                           float* save_mean, float* save_invstd, float* scale, float* shift, float* y, const float* residual,
                           int relu, int64_t N, int64_t C, int64_t SP, float* z, int64_t z_batch_stride, void* stream);
/* gca_bn_finalize followed by gca_bn_apply behind one call (one LAUNCH when N*SP is small: the deep layers, and
 * every layer of a small batch, are launch-latency bound).  count must equal N*SP. */
int gca_bn_train_fwd(const float* stat_sum, const float* stat_sq, int64_t P, int64_t C, double count,
                     const float* gamma, const float* beta, float eps, float momentum,
                     float* running_mean, float* running_var, int64_t* num_batches_tracked,
                     float* save_mean, float* save_invstd, float* scale, float* shift,
                     const void* x, const void* residual, int relu, int64_t N, int64_t SP, void* z,
                     int64_t z_batch_stride, int act_f16, void* stream);
/* Eval-mode fold (running stats): scale/shift only. */
int gca_bn_fold_eval(const float* gamma, const float* beta, const float* running_mean,
                     const float* running_var, float eps, int64_t C, float* scale, float* shift, void* stream);
/* z = [relu]( x*scale[c] + shift[c] [+ residual] ).  z_batch_stride (elements; 0 = C*SP) lets z be a
 * channel slice of a wider (N, Ctot, SP) buffer: the Inception branches write straight into their
 * slice of the concat output (s3d_1.py:96 torch.cat) instead of being copied there. */
int gca_bn_apply(const void* x, const float* scale, const float* shift, const void* residual,
                 int relu, int64_t N, int64_t C, int64_t SP, void* z, int64_t z_batch_stride, int act_f16, void* stream);
/* Backward of the fused op.  dz_in = gradient wrt z; x = saved conv output.  relu: 0 none; 1 mask from the saved
 * output z (z > 0); 2 mask recomputed from x with the forward's own scale/shift (x*scale+shift > 0) -- z may be
 * NULL and is not read: one tensor less to stream, valid when no residual was added before the ReLU.
 * Writes dx (gradient wrt x), accumulates dgamma/dbeta (+=) and, if dres != NULL, writes (dres_accumulate=0) or
 * adds (=1) the gradient wrt the residual input.  ws: gca_bn_bwd_ws_bytes(). */
int64_t gca_bn_bwd_ws_bytes(int64_t N, int64_t C, int64_t SP);
int gca_bn_bwd(const void* dz_in, const void* z, const void* x, const float* gamma,
               const float* save_mean, const float* save_invstd, int relu,
               int64_t N, int64_t C, int64_t SP, void* dx, float* dgamma, float* dbeta,
               void* dres, int dres_accumulate, int64_t z_batch_stride /* of dz_in and z */,
               const float* scale, const float* shift, void* ws, int act_f16, void* stream);

/* ---------------------------------------------------------------------------------------
 * Pooling.  MaxPool3d: resnet2p1d.py:178, s3d_1.py:10,13,16,22,87, temporal_graph.py:100
 * (first-max tie break in (d,h,w) scan order, as ATen).  Global/temporal-weighted average:
 * resnet2p1d.py:197,260 (AdaptiveAvgPool3d(1)); s3d_1.py:30-33 (avg_pool3d((2,H,W),1) then mean
 * over T' == frame-weighted global mean); resnet.py:137-140,187.
 * ------------------------------------------------------------------------------------- */
typedef struct {
  int32_t N, C, D, H, W;
  int32_t kd, kh, kw, sd, sh, sw, pd, ph, pw;
  int32_t OD, OH, OW;
} gca_pool_geom;
/* scale/shift (both or neither): pool relu(x*scale[c] + shift[c]) instead of x -- the BatchNorm+ReLU in front of
 * the pool (resnet2p1d.py:252-255 conv1_t -> bn1_t -> relu -> maxpool) evaluated on the fly. */
int gca_maxpool3d_fwd(const gca_pool_geom* g, const void* x, void* y, int32_t* argmax, const float* scale,
                      const float* shift, int act_f16, void* stream);
int gca_maxpool3d_bwd(const gca_pool_geom* g, const void* dy, const int32_t* argmax, void* dx,
                      int accumulate, int act_f16, void* stream);
/* nn.AvgPool3d(kernel_size=k) (stride = k, no padding, floor mode; fp32 maps): the `max_pool=False` option of
 * TemporalGraphAug (lib/ops/module_wrappers/temporal_graph.py:100).  Other strides / paddings: GCA_EINVAL. */
int gca_avgpool3d_fwd(const gca_pool_geom* g, const float* x, float* y, void* stream);
int gca_avgpool3d_bwd(const gca_pool_geom* g, const float* dy, float* dx, int accumulate, void* stream);
/* y[n,c] = sum_{d,h,w} wt[d] * x[n,c,d,h,w] * norm   (wt == NULL -> all ones) */
/* x_f16: the feature map (x, dx) is stored fp16; the pooled features y / dy are fp32 either way (the head is fp32). */
int gca_wavgpool_fwd(const void* x, const float* wt, float norm, int64_t NC, int64_t D, int64_t HW,
                     float* y, int x_f16, void* stream);
int gca_wavgpool_bwd(const float* dy, const float* wt, float norm, int64_t NC, int64_t D, int64_t HW,
                     void* dx, int x_f16, void* stream);

/* ---------------------------------------------------------------------------------------
 * Head pieces.  ReLU between the two Linear layers (project_head.py:24-27) and row L2
 * normalisation `Normalize` (project_head.py:4-10, F.normalize eps 1e-12).
 * ------------------------------------------------------------------------------------- */
int gca_relu_fwd(const float* x, int64_t n, float* y, void* stream);
int gca_relu_bwd(const float* dy, const float* y, int64_t n, float* dx, void* stream);
int gca_l2norm_fwd(const float* x, int64_t rows, int64_t dim, float eps, float* y, float* inv_norm, void* stream);
int gca_l2norm_bwd(const float* dy, const float* y, const float* inv_norm, int64_t rows, int64_t dim,
                   float* dx, void* stream);
/* Negative cosine similarity D(p, stopgrad(z)) of SimSiam (graph_wrappers.py:105-106):
 * loss[0] (+)= -scale * mean_i cos(p_i, z_i); dp = d loss / d p.  `loss` needs 1 + rows floats
 * (loss[1..rows] receives the per-row cosines). */
int gca_negcos_fwd_bwd(const float* p, const float* z, int64_t rows, int64_t dim, float scale,
                       float* loss, int accumulate, float* dp, void* stream);

/* ---------------------------------------------------------------------------------------
 * MoCo queue + InfoNCE.  RGBMoCo._compute_logit (lib/memory/mem_moco.py:29-49):
 * logits[i,0] = q_i.k_i / T, logits[i,1+j] = q_i.queue_j / T, as one clip x queue MFMA GEMM
 * streaming the queue once.  NCESoftmaxLoss (lib/memory/criterion.py:34-45) = mean_i
 * (logsumexp(logits_i) - logits_i0).  _update_memory/_update_pointer (mem_moco.py:14-27).
 * ------------------------------------------------------------------------------------- */
/* logits (b, K+1).  Optional fused row statistics (any may be NULL): row_lse (b), rank_ge (b) = number of negatives
 * with logit >= the positive's (top-1 hit <=> 0, top-5 <=> < 5; lib/evaluation/metric.py:44-67 with label 0), and
 * loss (1) = NCESoftmaxLoss of these logits (needs row_lse) -- the whole forward of mem_moco.py:60-88 +
 * criterion.py:34-45 behind one call.  ws: gca_infonce_ws_bytes(b, K).
 * sync_counter: one device uint32 that is ZERO before the first call (every call leaves it zero); with it, b <= 32 and
 * D <= 128 the whole forward is ONE launch -- persistent waves stream the queue, the last workgroup to arrive folds the
 * row statistics.  NULL selects the multi-launch path. */
int64_t gca_infonce_ws_bytes(int64_t b, int64_t K);
int gca_moco_logits_fwd(const float* q, const float* k, const float* queue, int64_t b, int64_t K,
                        int64_t D, float inv_T, float* logits, float* row_lse, int32_t* rank_ge, float* loss,
                        void* ws, uint32_t* sync_counter, void* stream);
/* loss = mean_i (lse_i - logits[i,0]); lse computed here if row_lse_in == NULL. */
int gca_nce_softmax_loss_fwd(const float* logits, int64_t b, int64_t ncol, const float* row_lse_in,
                             float* row_lse_out, float* loss, void* stream);
/* dlogits[i,j] = gscale * (softmax(logits_i)[j] - [j==0]) / b   (gscale: upstream d loss) */
int gca_nce_softmax_loss_bwd(const float* logits, const float* row_lse, int64_t b, int64_t ncol,
                             const float* gscale_dev, float gscale_host, float* dlogits, void* stream);
/* dq = dlogits[:,0:1]*k/T + dlogits[:,1:] @ queue / T.  Rows [ov_start, ov_start+ov_n) mod K of
 * `queue` are read from ov_rows instead (the rows the enqueue of this step has already
 * overwritten -- the reference computes the gradient against the pre-enqueue snapshot,
 * mem_moco.py:72).  If dlogits == NULL it is formed on the fly from logits/row_lse as in
 * gca_nce_softmax_loss_bwd (fused path; no dlogits tensor is materialised). */
int gca_moco_logits_bwd(const float* dlogits, const float* logits, const float* row_lse,
                        const float* gscale_dev, float gscale_host,
                        const float* k, const float* queue, int64_t b, int64_t K, int64_t D, float inv_T,
                        int64_t ov_start, const int64_t* ov_start_dev /* overrides ov_start if non-NULL */,
                        int64_t ov_n, const float* ov_rows, float* dq, void* ws, void* stream);
/* accuracy(output, target, topk) of lib/evaluation/metric.py:44-67 (called at tools/train_video_contrast_dis.py:428) without
 * the top-k sort or a host sync: rank_ge[i] = number of columns j != target[i] with output[i,j] >= output[i,target[i]];
 * the target is in the top k of row i iff rank_ge[i] < k.  target must lie in [0, ncol) (not checked on the device). */
int gca_rank_ge(const float* output, const int64_t* target, int64_t b, int64_t ncol, int32_t* rank_ge, void* stream);
/* queue[(ptr + i) % K] = keys[i], i < n; optionally saves the overwritten rows first.  The
 * reference keeps the pointer as a Python int (mem_moco.py:12); for hipGraph replay it can live in
 * device memory instead: ptr_dev (non-NULL) overrides ptr, gca_queue_advance moves it. */
int gca_queue_enqueue(float* queue, int64_t K, int64_t D, const float* keys, int64_t n, int64_t ptr,
                      const int64_t* ptr_dev, float* saved_rows, void* stream);
int gca_queue_advance(int64_t* ptr_dev, int64_t n, int64_t K, void* stream);

/* ---------------------------------------------------------------------------------------
 * Temporal-graph block (lib/ops/module_wrappers/temporal_graph.py).  The 1x1x1 convs and the
 * (1,2,2) max pool go through gca_conv_* / gca_maxpool3d_*; these are the graph-specific parts.
 * ------------------------------------------------------------------------------------- */
/* sim = softmax_j( sum_f gq[b,i,f]*gk[b,j,f] ), gq/gk stored (B, Ci, T, HW) as the pooled convs
 * leave them (temporal_graph.py:159-176); then adj_pre = sim * theta(hop(i,j)) within max_hop
 * else 0 (:204-210); then adj = sigmoid((logit(u)+logit(adj_pre))/temperature)  (:187-192,
 * RelaxedBernoulli.rsample with explicit uniforms u).  ws: gca_graph_gram_ws_bytes(B, Ci, T, HW) bytes of scratch
 * for the two-pass (T x F)(F x T) product (NULL = slower one-workgroup-per-entry fallback). */
int64_t gca_graph_gram_ws_bytes(int64_t B, int64_t C, int64_t T, int64_t HW);
int gca_graph_adj_fwd(const float* gq, const float* gk, int64_t B, int64_t Ci, int64_t T, int64_t HW,
                      int max_hop, float alpha, float temperature, const float* u,
                      float* sim, float* adj_pre, float* adj, void* ws, void* stream);
/* NOTE: dadj is in/out -- it is overwritten with the gradient wrt the pre-softmax similarity. */
int gca_graph_adj_bwd(const float* dadj, const float* gq, const float* gk, const float* sim,
                      const float* adj_pre, const float* adj, int64_t B, int64_t Ci, int64_t T, int64_t HW,
                      int max_hop, float alpha, float temperature, float* dgq, float* dgk, void* stream);
/* out[b,c,i,:] = sum_j adj[b,i,j]*s[b,c,j,:] + s[b,c,i,:]   (einsum 'bij,bcjhw->bcihw' + skip,
 * temporal_graph.py:59-62): wave-reduced dense neighbourhood GEMV, one pass over s. */
int gca_graph_gcn_fwd(const float* adj, const float* s, int64_t B, int64_t C, int64_t T, int64_t HW,
                      float* out, void* stream);
/* ds[b,c,j,:] = sum_i adj[b,i,j]*dout[b,c,i,:] + dout[b,c,j,:];  dadj[b,i,j] = sum_{c,hw} dout[b,c,i,:]*s[b,c,j,:] */
int64_t gca_graph_gcn_bwd_ws_bytes(int64_t B, int64_t C, int64_t T, int64_t HW);
int gca_graph_gcn_bwd(const float* adj, const float* s, const float* dout, int64_t B, int64_t C, int64_t T,
                      int64_t HW, float* ds, float* dadj, void* ws, void* stream);

/* ---------------------------------------------------------------------------------------
 * Device-side tail of the input pipeline.  Replaces, for uint8 frames already decoded / resized on the
 * host, the last stages of the reference's transform chain and its collation:
 *   VideoRandomHorizontalFlip + VideoNormalize + VideoToTensor  (lib/data/transform/consistency_transforms.py:11-65,
 *   composed in lib/data/transform/build.py:45-62), an integer crop window (albumentations random_crop coordinates as
 *   VideoRandomCrop / VideoCenterCrop use them), the channel concatenation of the two views
 *   (lib/data/datasets/video_contrast_dataset.py:196-203) and the H2D copy of the fp32 batch (tools/...dis.py:402).
 *   frames : (b, views, T, Hs, Ws, 3) uint8, HWC frames as cv2 delivers them (device memory)
 *   params : (b, views, 4) int32 on the device: {h0, w0, flip, 0} -- ONE crop origin and flip decision per (clip, view), shared
 *            by its T frames (the reference's transforms draw them once per clip); windows are clamped into the frame
 *   mean255, inv_std255 : HOST pointers to 3 floats each: f32(mean_c)*255 and 1/(f32(std_c)*255), computed in fp32 as numpy
 *            does in VideoNormalize.normalize (:53-65)
 *   out    : (b, 3*views, T, H, W) fp32, or fp16 when out_f16 (the fp16-storage path): out = (float(px) - m_c) * d_c,
 *            two fp32 roundings -- bit-identical to the reference's arithmetic
 * One pass over HBM, 3 B read + 12 (6) B written per pixel. */
int gca_clip_prepare(const uint8_t* frames, int64_t b, int64_t views, int64_t T, int64_t Hs, int64_t Ws,
                     const int32_t* params, const float* mean255, const float* inv_std255, int64_t H, int64_t W,
                     void* out, int out_f16, void* stream);

/* ---------------------------------------------------------------------------------------
 * Multi-tensor parameter updates over flat, 256-element-aligned parameter arenas.
 * _momentum_update (tools/train_video_contrast_dis.py:177-180) and torch.optim.SGD as
 * configured by make_optimizer (lib/solver/build.py:24-59: one group per parameter).
 * ------------------------------------------------------------------------------------- */
int gca_ema_update(float* p_ema, const float* p, int64_t n, float m, void* stream);
/* seg table: per 256-element chunk i: lr[i], wd[i] (device arrays of length n/256).
 * grad_clip: NULL, or the 4-float result of gca_grad_clip_coef / gca_grad_unscale_clip -- every gradient is multiplied by
 * grad_clip[1] on the fly (the grads.mul_(clip_coef) pass of clip_grad_norm_ without a pass over the arena), and when
 * grad_clip[2] != 0 (a non-finite gradient under loss scaling) the whole update is skipped: parameters and momentum stay
 * as they are, as apex amp skips optimizer.step() (tools/train_video_contrast_dis.py:413-419 under APEX.FLAG). */
int gca_sgd_step(float* p, const float* grad, float* mom_buf, int64_t n, const float* chunk_lr,
                 const float* chunk_wd, float lr_scale, float momentum, int nesterov, int first_step,
                 const float* grad_clip, void* stream);
/* clip_grad_norm_(parameters, max_norm) of tools/train_video_contrast_dis.py:420-423 over the flat gradient arena
 * (n % 4 == 0; padding elements are zero): out4[0] = total 2-norm, out4[1] = min(1, max_norm / (norm + 1e-6)),
 * out4[2] = out4[3] = 0.  fp64 partial sums folded in a fixed order (deterministic); `ws` = gca_grad_clip_ws_bytes() bytes. */
int64_t gca_grad_clip_ws_bytes(void);
int gca_grad_clip_coef(const float* grad, int64_t n, float max_norm, float* out4, void* ws, void* stream);
/* Dynamic loss scaling for the fp16-storage path -- what `amp.scale_loss(loss, optimizer)` + the patched optimizer.step()
 * do in the reference (tools/train_video_contrast_dis.py:134-141 amp.initialize, :413-418 scale_loss; apex's LossScaler:
 * 2x after 2000 clean steps, 0.5x and a skipped step on inf / nan), decided ON THE DEVICE (no host sync, hipGraph-safe):
 *   scale_state4 = {loss scale S, clean steps in a row, skipped steps, steps seen}; S is read by the loss-gradient kernels
 *   through their gscale_dev argument (gca_moco_logits_bwd, gca_nce_softmax_loss_bwd, gca_scale_dev).
 *   out4[0] = 2-norm of grad / S, out4[1] = (1/S) * clip coefficient (max_norm <= 0: no clipping) -- the factor the SGD
 *   kernel multiplies every gradient by, out4[2] = 1 when any element of grad is inf / nan (then out4[1] = 0, S is
 *   multiplied by `backoff` and gca_sgd_step leaves parameters and momentum untouched), out4[3] = the S these gradients
 *   were computed with.  After `growth_interval` clean steps S *= growth (capped at max_scale). `ws`: gca_grad_clip_ws_bytes(). */
int gca_grad_unscale_clip(const float* grad, int64_t n, float max_norm, float* scale_state4, float growth, float backoff,
                          int growth_interval, float max_scale, float* out4, void* ws, void* stream);
int gca_scale_dev(float* y, int64_t n, const float* a_dev, float a_host, void* stream);   /* y *= (*a_dev) * a_host */
int gca_fill(float* p, int64_t n, float v, void* stream);
int gca_axpy(float* y, const float* x, int64_t n, float a, void* stream);          /* y += a*x */
int gca_scale(float* y, int64_t n, float a, void* stream);
/* fp16-storage path: y += a*x on fp16 tensors (fp32 arithmetic, one rounding), and the fp32 -> fp16 cast of the input clips */
int gca_axpy_f16(void* y, const void* x, int64_t n, float a, void* stream);
/* rows x row_elems (row_elems % 4 == 0), source rows src_row_stride elements apart (a channel slice of a clip batch) */
int gca_cast_f16(const float* x, int64_t rows, int64_t row_elems, int64_t src_row_stride, void* y, void* stream);
/* dst[r,:] = src[idx[r]*src_row_stride : +row_elems]; rows <= 65535.  src rows may be strided (the key view of a
 * (b,6,T,H,W) batch is gathered in place for the ShuffleBN exchange, tools/...dis.py:404,213-217). */
int gca_gather_rows(const float* src, const int64_t* idx, int64_t rows, int64_t row_elems, int64_t src_row_stride,
                    float* dst, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GCA_HIP_H */
