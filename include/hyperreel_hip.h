/*
 * hyperreel_hip.h -- C ABI of libhyperreel_hip.so, the MI355X (gfx950) forward renderer
 * for HyperReel scenes.
 *
 * This is the drop-in boundary for ONE path of the reference: what
 * `render_fn(rays)` computes, i.e. RenderLightfield.forward
 * (nlf/rendering.py:72-77) -> LightfieldModel.forward (nlf/models/models.py:135-138)
 * -> RayPointEmbedding (nlf/embedding/embedding.py:100-117) -> TensorVMNoSample /
 * TensorVMKeyframeTime .forward (nlf/nets/tensorf_no_sample.py:128-280,
 * nlf/nets/tensorf_dynamic.py:645-839).  The reference has no FFI of its own (it is
 * pure PyTorch); the binding a maintainer adds is the ctypes stub shown in
 * INTEGRATION.md, selected with `experiment.model.render.type: lightfield_hip`
 * through the reference's own registry (nlf/rendering.py:95-97).
 *
 * Conventions: plain C types only; every pointer named *_dev is a device pointer
 * owned by the caller (e.g. torch tensor .data_ptr()); the library never frees
 * caller memory; `stream` is a hipStream_t passed as void*; all calls return 0 on
 * success and a negative HR_E_* code otherwise (hr_last_error() has the text; the
 * Python wrapper raises RuntimeError, the reference's only error channel).  Render
 * calls only enqueue work on `stream`: no allocation, no synchronisation, so they
 * can be captured in a hipGraph.
 */
#ifndef HYPERREEL_HIP_H
#define HYPERREEL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HR_ABI_VERSION 26

#define HR_MAX_Z 256         /* samples per ray (z_channels) supported by the sample kernel */
#define HR_MAX_P 64          /* per-sample head columns (preds_per_z) */
#define HR_MAX_GROUPS 4      /* ray-parameterisation groups feeding the MLP (`params:` in the YAML) */
#define HR_MAX_LAYERS 8      /* Linear layers of the sample-prediction MLP */
#define HR_MAX_FREQS 8       /* frequencies of a windowed positional encoding */
#define HR_MAX_MLP_IN 64     /* MLP input features after positional encoding */

/* error codes */
#define HR_OK 0
#define HR_E_INVALID (-1)    /* bad argument / unsupported configuration */
#define HR_E_STATE (-2)      /* call order: upload after finalize, render before finalize ... */
#define HR_E_HIP (-3)        /* a HIP runtime call failed */
#define HR_E_MISSING (-4)    /* finalize: a required tensor was never uploaded */
#define HR_E_RANGE (-5)      /* finalize / calibrate: a forced fp16 MLP arithmetic would overflow on this model's activations */

/* nlf/activations.py: Identity (:163-178), Sigmoid (:53-69), Tanh (:121-137).  y = act(x*inner+shift)*outer + add.
 * EaseValue (:462-496), w * act(x) + (1 - w) * start_value with the iteration-dependent weight w, is folded by the host
 * into outer (times w) and add ((1 - w) * start_value); once its window has passed w == 1 and add == 0. */
enum { HR_ACT_IDENTITY = 0, HR_ACT_SIGMOID = 1, HR_ACT_TANH = 2 };
typedef struct hr_act {
    int32_t type;
    float inner, shift, outer, add;
} hr_act;

/* nlf/param.py: identity (:20-24), PlueckerParam (:223-256), TwoPlaneParam (:63-118) */
enum { HR_PARAM_IDENTITY = 0, HR_PARAM_PLUECKER = 1, HR_PARAM_TWO_PLANE = 2 };
/* nlf/pe.py: IdentityPE, WindowedPE (:130-224, per-frequency weights in pe_weight), BasicPE (:32-71) */
enum { HR_PE_NONE = 0, HR_PE_WINDOWED = 1, HR_PE_BASIC = 2 };
typedef struct hr_param_group {
    int32_t start, end;          /* ray columns [start, end) (nlf/embedding/ray.py:319-321) */
    int32_t fn;                  /* HR_PARAM_* */
    float origin[3];
    float a, b;                  /* pluecker: direction/moment multiplier; two_plane: near/far plane z */
    int32_t pe_type;             /* HR_PE_* */
    int32_t pe_n_freqs;
    int32_t pe_exclude_identity;
    float pe_freq_mult, pe_base_mult;
    float pe_weight[HR_MAX_FREQS]; /* windowed: WindowedPE.weight(j) of frequency j (nlf/pe.py:186-208), 1 once its window has passed */
} hr_param_group;

/* One per-sample output of the MLP head (nlf/embedding/ray.py:333-337): `channels`
 * consecutive columns starting at `offset` inside the P = preds_per_z columns of a
 * sample; offset < 0: the field does not exist in this model. */
typedef struct hr_head_field {
    int32_t offset, channels;
    hr_act act;
} hr_head_field;

/* nlf/intersect: z.py:15-97 (z_plane), primitive.py:366-438 (sphere), :181-253 (cylinder), :441-545
 * (sphere_new), :256-363 (cylinder_new), :131-176 (euclidean_distance_unified), voxel.py:19-112
 * (voxel_grid), :115-215 (deformable_voxel_grid).  `plane` and `euclidean_distance` cannot run in the reference itself (their scalar
 * z_scale breaks Intersect.process_z_vals, base.py:129), so there is nothing to be compatible with. */
enum {
    HR_ISECT_Z_PLANE = 0, HR_ISECT_SPHERE = 1, HR_ISECT_CYLINDER = 2, HR_ISECT_SPHERE_NEW = 3,
    HR_ISECT_CYLINDER_NEW = 4, HR_ISECT_EUCLIDEAN_UNIFIED = 5, HR_ISECT_VOXEL_GRID = 6,
    HR_ISECT_DEFORMABLE_VOXEL_GRID = 7
};
/* nlf/contract.py: IdentityContract (:53-62), MIPNeRFContract (:113-192); BBoxContract (:65-87) and
 * ZDepthContract (:90-111) are both the affine map p -> (p - c_aff_min) / c_aff_size, d -> d / c_aff_fac */
enum { HR_CONTRACT_IDENTITY = 0, HR_CONTRACT_MIPNERF = 1, HR_CONTRACT_AFFINE = 2,
       /* DoNeRFContract (contract.py:195-240): p -> p / |p| * (|p| fac + 1e-8)^(1/power), d -> sign(d) (|d| + 1e-8)^power / fac */
       HR_CONTRACT_DONERF = 3 };
enum { HR_DENSITY_RELU = 0, HR_DENSITY_SOFTPLUS = 1, HR_DENSITY_RELU_ABS = 2 };
enum { HR_SHADING_RGB = 0, HR_SHADING_SH = 1 };
/* arithmetic of the MLP GEMMs: exact fp32 MFMA, or three 16-bit MFMA products of the hi/lo split operands with fp32
 * accumulation (needs mlp_hidden 256): bf16 halves (~2^-17 relative per product, fp32 range) or fp16 halves (~2^-22,
 * i.e. fp32-grade, but activations must stay below 65504); F16X2 additionally takes the weights as single halfs (two
 * products, 2^-12 relative weight rounding) */
enum { HR_MLP_FP32 = 0, HR_MLP_BF16X3 = 1, HR_MLP_F16X3 = 2, HR_MLP_F16X2 = 3,
       /* the library chooses: f16x3 when an activation-range calibration of the uploaded weights (hr_model_finalize: 4096 synthetic
        * rays; hr_model_calibrate: the caller's rays) keeps every input feature and hidden activation below 65504 / 8, bf16x3
        * otherwise (fp32 when mlp_hidden != 256).  The same test makes a FORCED f16x3 / f16x2 fail with HR_E_RANGE instead of
        * rendering infinities (the reference's BaseMLP is fp32, nlf/nets/mlp.py:127-172: any finite activation is legal there) */
       HR_MLP_AUTO = 4,
       /* fp16 halves like F16X3, but only the leading product x_hi*w_hi is an f16 MFMA: the two correction products (2^-11 of it) are ONE
        * fp8 (OCP e4m3) K=64 MFMA per 32 k with power-of-two block scales -- two thirds of F16X3's matrix-pipe time at ~2^-16 relative
        * per product (F16X3 2^-22, F16X2 2^-12).  Same range rule as F16X3, plus a per-layer exponent for the fp8 images taken from the
        * calibration (16x headroom); an activation beyond either range sets HR_OPT_MLP_OVERFLOW.  gfx950 (v_mfma_scale_f32_32x32x64_f8f6f4). */
       HR_MLP_F16F8 = 5,
       /* F16F8 with its discrete decisions VERIFIED (what HR_MLP_AUTO resolves to where it applies: a plain ray MLP of width 256, at most 64
        * samples per ray, activations inside the fp16 range).  The per-sample stage is continuous in the MLP's head except at a few
        * comparisons -- `dist <= near` / `>= far` (nlf/intersect/base.py:194), the quadratic's discriminant and root choice
        * (utils/intersect_utils.py:45-125), the bounding box (nlf/nets/tensorf_base.py:349-353), a positive weight threshold -- and F16F8's head
        * error (2e-5 of the head's range; distances move by < 1e-6 of the scene's extent) only shows when one of them falls the other way.  The
        * sample kernel therefore lists, on the device, every ray with a comparison inside a BAND of flipping (and the MLP kernel every tile that
        * raised a range bit), and hr_render / hr_render_frame end with a second, list-driven pass that renders exactly those rays (0.05 - 3 %
        * of a frame) again with F16X3's tiles.  The band is PER MODEL and PER SAMPLE: hr_model_finalize / hr_model_calibrate run the MLP in both
        * arithmetics on the calibration rays and measure how far the length fed to the intersection moves; the sample kernel pushes 4x that through
        * the derivatives of the inverse contraction and of the intersection of each sample (hr_verify_info below; csrc/hr_math.h, HrRisk).
        * Under HR_MLP_AUTO a model whose margins would list more than 5 % of the calibration rays, or whose calibration image differs from
        * F16X3's by more than 6e-5 anywhere, is rendered with F16X3 throughout instead; so are intersections the margins are not derived for
        * (the `_new` primitives, the deformable grid, learned sphere origins, DoNeRFContract).  Both passes are ordinary launches on the caller's stream: capturable.  hr_render_fields with diagnostics,
        * and models with an occupancy volume set (its cell test is a head-dependent decision the band does not cover), render everything with the
        * F16X3 tiles.  HR_OPT_MLP_PRECISION_ACTIVE reports HR_MLP_F16F8, HR_OPT_MLP_VERIFIED 1. */
       HR_MLP_F16F8V = 6 };
/* storage of the feature grids on the device: the reference's float32, or float16 texels (viewer
 * path, BASELINE config 5: half the gather bytes; values are rounded once at finalize, all arithmetic
 * stays fp32 -- results equal the fp32 path run on the rounded grids) */
enum { HR_GRID_FP32 = 0, HR_GRID_FP16 = 1 };
/* inputs of a point_prediction row: x['points'] of the coarse intersect, rays[..., 3:6], rays[..., 0:3], rays[..., -1:] */
enum { HR_PIN_POINTS = 0, HR_PIN_VIEWDIRS = 1, HR_PIN_ORIGINS = 2, HR_PIN_TIMES = 3 };

/* Everything the kernels need that the reference derives from the model YAML and the
 * five dataset scalars (near, far, depth_range, num_keyframes, num_frames).  Derived
 * floats are computed by the host in the reference's own precision and order
 * (hyperreel_amd/plan.py cites the lines). */
typedef struct hr_config {
    /* ---- rays + MLP (nlf/embedding/ray.py:213-347, nlf/nets/mlp.py:60-172) */
    int32_t ray_dim;                     /* 6 static [o,d]; 8 video [o,d,cam_id,t] */
    int32_t n_groups;
    hr_param_group groups[HR_MAX_GROUPS];
    int32_t mlp_in;                      /* input features (sum over groups after PE) */
    int32_t mlp_layers;                  /* number of Linear layers (D+2); 0 = ZeroMLP (nlf/nets/mlp.py:14-33) */
    int32_t mlp_hidden;                  /* W */
    int32_t mlp_skip_mask;               /* bit i set: layer i takes cat([input, x]) */
    float leaky_slope;                   /* 0.01 */
    int32_t z_channels;                  /* Z */
    int32_t preds_per_z;                 /* P */
    hr_head_field f_z_vals;
    hr_head_field f_isect_sigma;         /* x[intersect.in_density_field], intersect/base.py:152-158 */
    hr_head_field f_offset_sigma;        /* x[point_offset.in_density_field], embedding/point.py:378-381 */
    hr_head_field f_point_offset;
    hr_head_field f_color_scale;
    hr_head_field f_color_shift;
    hr_head_field f_spatial_flow;
    hr_head_field f_color_scale_global;  /* scale_shift_color_one, utils/tensorf_utils.py:275-281 (sample 0's values); with 9 channels: the head
                                          * `color_transform_global`, a row-major 3x3 for transform_color_one (:308-320) */
    hr_head_field f_color_shift_global;
    /* ---- intersect (nlf/intersect/base.py:142-259, z.py, primitive.py) */
    int32_t isect_type;                  /* HR_ISECT_* */
    float isect_origin[3];
    float near, far;                     /* mask: dist <= near | dist >= far */
    hr_act z_act;
    int32_t sort;
    float samples[HR_MAX_Z];             /* anchor samples (contracted space when contract_samples) */
    float z_scale;
    float origin_scale;                  /* sphere/cylinder: origins = z[:3]*origin_scale + origin_initial */
    float origin_initial[3];             /*   (*_new: origins = z[:3]*origin_scale, primitive.py:490-492) */
    float resize_scale;                  /* *_new: resize = z[3:6]*resize_scale + resize_initial (:494-496) */
    float resize_initial[3];
    float voxel_scale[3];                /* voxel_grid: per-axis z_scale; samples[] is (Z/3, 3) row-major */
    int32_t isect_outward;               /* voxel_grid: outward_facing (planes mirrored by sign(d)) */
    int32_t dvg_axes;                    /* deformable_voxel_grid: number of start normals (sample k uses normal k % axes) */
    float dvg_normals[9];                /*   start_normal rows */
    float dvg_normal_scale;              /*   normal = z[:3]*normal_scale_factor + start_normal, then normalised */
    int32_t isect_mask_off;              /* mask.stop_iters passed: no near/far masking (base.py:197-198) */
    /* ---- contraction (nlf/contract.py:113-192) */
    int32_t contract_type;               /* HR_CONTRACT_* */
    int32_t contract_samples;
    float c_r0, c_r_inv_end, c_r_scale;  /* points:   start radius,   r0/r1, 1/(1-r0/r1) */
    float c_d0, c_d_inv_end, c_d_scale;  /* distance: start distance, d0/d1, 1/(1-d0/d1) */
    float c_aff_min[3], c_aff_size[3];   /* affine: bbox_min, bbox_max - bbox_min (z_depth: 0, fac) */
    float c_aff_fac;                     /* affine: inverse_contract_distance(d) = d * fac */
    float c_pow_fac, c_pow_power, c_pow_inv_power;   /* donerf: fac, power, 1/power as the reference holds them (float32 of its python floats) */
    /* ---- advect (nlf/embedding/point.py:780-831, utils/flow_utils.py:10-35) */
    int32_t advect;
    int32_t use_spatial_flow;
    float flow_fac, flow_inv_fac;        /* K*(F-1)/F and its reciprocal */
    float flow_kmax;                     /* K-1 */
    hr_act flow_act;
    /* ---- point offset (nlf/embedding/point.py:371-396) */
    int32_t point_offset;
    hr_act offset_act;
    /* ---- colour net (nlf/nets/tensorf_no_sample.py, tensorf_dynamic.py, tensorf_base.py) */
    int32_t video;                       /* 0: tensor_vm_split_no_sample, 1: tensor_vm_split_time */
    float aabb[6];                       /* [min xyz, max xyz] */
    float inv_size[3];                   /* 2/(max-min) */
    int32_t grid[3];                     /* gridSize [Nx,Ny,Nz] */
    int32_t num_keyframes;               /* K (video) */
    int32_t n_den[3], n_app[3];          /* n_lamb_sigma, n_lamb_sh */
    int32_t app_dim;                     /* 3 (RGB) or 27 (SH degree 2) */
    int32_t shading;                     /* HR_SHADING_* */
    float distance_scale;
    float weight_thresh;                 /* rm_weight_mask_thre */
    int32_t density_act;                 /* HR_DENSITY_* */
    float density_shift;
    float time_scale, time_offset;       /* (F-1)/F and 0.5/K, tensorf_dynamic.py:58-59 */
    int32_t white_bg;
    int32_t mlp_precision;               /* HR_MLP_* */
    int32_t grid_dtype;                  /* HR_GRID_* */
    /* ---- per-camera colour correction (ColorTransformEmbedding, nlf/embedding/point.py:558-602):
     *      table row round(rays[..., -2]) = [3x3 transform | shift]; applied to the composited colour
     *      (transform_color_one, utils/tensorf_utils.py:308-320).  0 views: stage absent, or
     *      dataset.val_all false, in which case the reference's stage returns x unchanged. */
    int32_t color_table_views;
    hr_act color_table_t_act, color_table_s_act;
    /* ---- point_prediction cascade (PointPredictionEmbedding, nlf/embedding/point.py:39-218), set only in the
     *      FINE config of hr_model_create_cascade: the MLP of this config runs once per COARSE sample on the row
     *      [inputs...] (groups[] index the row's columns) and emits z_channels / casc_in_z samples per row. */
    int32_t casc_in_z;                   /* coarse samples per ray (in_z_channels); 0: not a cascade */
    int32_t casc_row_dim;                /* columns of an input row = sum of casc_input_dim */
    int32_t casc_n_inputs;
    int32_t casc_input_kind[4];          /* HR_PIN_* in `inputs` order (point.py:142-156) */
    int32_t casc_input_dim[4];           /* columns taken from each */
} hr_config;

/* Optional per-sample diagnostics of hr_render_fields (all device pointers, any may be
 * NULL).  They mirror what the reference exposes through render_kwargs `fields`
 * (tensorf_no_sample.py:254-278) and LightfieldModel.embed. */
typedef struct hr_fields {
    float* distances_dev;       /* (n, Z)    x['distances'] after sort + contraction */
    float* points_dev;          /* (n, Z, 3) x['points'] fed to the colour net */
    float* sigma_dev;           /* (n, Z)    density after feature2density */
    float* weights_dev;         /* (n, Z)    'render_weights' */
    float* head_dev;            /* (n, Z*P)  raw MLP output */
} hr_fields;

/* Pinhole camera of the viewer / offline-render path: what get_coords_from_camera
 * (datasets/base.py:485-518) consumes -- a 3x4 camera-to-world pose and intrinsics K. */
typedef struct hr_camera {
    float c2w[12];              /* row-major 3x4 [R | t], -z forward (utils/ray_utils.py:121-135) */
    float fx, fy, cx, cy;       /* K[0,0], K[1,1], K[0,2], K[1,2] */
    int32_t width, height;
    float cam_id, time;         /* columns 6 and 7 of 8-column rays (datasets/base.py:511-515) */
} hr_camera;

typedef struct hr_model hr_model;

int hr_abi_version(void);
int hr_sizeof_config(void);      /* sizeof(hr_config) as compiled, for binding self-checks */
const char* hr_last_error(void);

/* Builds a model for `cfg` on the current HIP device.  Replaces the constructors
 * LightfieldModel.__init__ (nlf/models/models.py:104-129) + RenderLightfield.__init__
 * (nlf/rendering.py:59-70). */
int hr_model_create(const hr_config* cfg, hr_model** out);

/* Two-level model for the reference's point_prediction cascades (the shipped ..._cascaded.yaml and
 * ..._feedback.yaml model groups): `coarse` describes ray_prediction + the first ray_intersect (its colour-net fields are
 * ignored), `fine` the point_prediction MLP (casc_* set), the second ray_intersect and everything after it.
 * Tensors of the coarse MLP are uploaded as mlp.<i>.*, those of the point MLP as mlp1.<i>.*. */
int hr_model_create_cascade(const hr_config* coarse, const hr_config* fine, hr_model** out);

/* Hands over one tensor of the reference state_dict, float32, in the reference's own
 * layout (nlf/__init__.py:433-479 are the reference's loader).  `name` is the key with
 * the module prefix removed:
 *   mlp.<i>.weight (out,in)  mlp.<i>.bias (out)                      i < mlp_layers
 *   density_plane.<j> (1,C,H,W)  density_line.<j> (1,C,N,1)
 *   app_plane.<j>  app_line.<j>                                      static net, j < 3
 *   density_plane_space.<j> density_plane_time.<j> (1,C,K,N)
 *   app_plane_space.<j> app_plane_time.<j>                           video net
 *   basis_mat.weight (app_dim, sum n_app)
 *   color_embedding (color_table_views, 12)                           when color_table_views > 0
 * `ptr` may be host or device memory; the data is copied before the call returns. */
int hr_model_upload(hr_model* m, const char* name, const void* ptr, size_t bytes);

/* Re-lays the uploaded tensors out for the kernels (channel-last interleaved planes,
 * MFMA-tiled MLP weights).  May be called again after further uploads.  For mlp_precision AUTO / F16X3 / F16X2 it first measures
 * the MLP's activation range on 4096 synthetic rays (origins uniform in the model's aabb, unit directions) and resolves / checks
 * the arithmetic (see HR_MLP_AUTO); HR_E_RANGE when a forced fp16 mode does not fit. */
int hr_model_finalize(hr_model* m);

/* The same decision on the caller's own rays (device memory, n_rays x ray_dim): measures the activation range of the MLP
 * (BaseMLP.forward, nlf/nets/mlp.py:159-172, evaluated in plain fp32) on them, re-resolves HR_MLP_AUTO and re-packs the MLP
 * weights if the choice changes.  Synchronises `stream`.  `act_max` (may be NULL) receives mlp_layers floats: the largest
 * |input feature|, then the largest |pre-activation| of each hidden Linear.  HR_E_RANGE as for hr_model_finalize. */
int hr_model_calibrate(hr_model* m, const float* rays_dev, int64_t n_rays, float* act_max, void* stream);

/* Replaces the model's configuration by one that differs only in schedule-dependent constants -- the `outer` / `add`
 * of the activations (EaseValue), `pe_weight` (WindowedPE) and `isect_mask_off` (mask.stop_iters) -- as INRSystem.set_train_iter does for the reference
 * modules every training step (nlf/__init__.py:608-614).  No weights are re-packed.  Any other difference is refused
 * with HR_E_INVALID.  Waits for `stream` before the device copies are replaced. */
int hr_model_update_config(hr_model* m, const hr_config* cfg, void* stream);

/* Sizes the per-launch workspace (rays processed per internal chunk).  Optional. */
int hr_model_reserve(hr_model* m, int64_t rays_per_chunk);

/* Execution plan of hr_render.  The arithmetic is the same under every setting (bit-identical images); the options choose
 * how it is laid out on the device.
 *   HR_OPT_FRAME_KERNEL   0 (default): two kernels per chunk of rays -- the MLP writes the head to an HBM workspace, the sample kernel
 *                         reads it back.  Level in time with the frame kernel (round 5, interleaved per-frame events: 1.96 vs 1.99 ms
 *                         on the DoNeRF frame) and with the tighter tail: the hardware dispatches its blocks, nothing is dealt statically.
 *                         1: models whose head tile fits the CU's LDS are rendered by ONE persistent kernel in which
 *                         MLP wavefronts hand the (B, Z*P) head that the reference materialises between RayPredictionEmbedding and
 *                         Intersect (nlf/embedding/ray.py:332-337 -> nlf/intersect/base.py:142-259) to sample wavefronts of the
 *                         same workgroup through LDS -- no workspace traffic (static nets, 64-ray tiles).  Models that do not
 *                         fit (wider heads, cascades, the exact-fp32 MLP, other plane decompositions), and every hr_render_fields
 *                         call with a non-NULL `fields`, take the two-kernel path.
 *                         2: additionally the keyframe families (480-column heads, 960 at 64 samples per ray;
 *                         nlf/nets/tensorf_dynamic.py:645-839) on 32-ray tiles, two head buffers where they fit: same images, no
 *                         head workspace traffic, measured as fast as or slower than two kernels (hence not part of 1).
 *   HR_OPT_SAMPLE_WAVES   sample wavefronts per workgroup of the frame kernel: 4 or 8 (0: the plan's default, 8).
 *   HR_OPT_TRAIN_DETERMINISTIC  1: hr_train_backward accumulates every gradient that many samples add to -- texel gradients, basis_mat's,
 *                         the colour table's -- as 64-bit fixed point with integer atomics instead of fp32 atomics (the unit is a power of two
 *                         chosen per step: the step's largest |dL/d rgb| = 2^32 units; a non-finite contribution, or one beyond 2^62 units,
 *                         turns the step's totals into NaN, as the fp32 path would hold inf / NaN there): the
 *                         result does not depend on the order of the adds, so two runs of the same step agree bit for bit (the
 *                         reference's loop, nlf/__init__.py:634-709, is deterministic for a given thread count).  The values agree
 *                         with the default mode's to fp32 rounding of the sums; slower (no LDS staging of the contended lines).
 *                         0 (default): fp32 atomics.  The MLP's GEMM gradients are reduced in a fixed order in both modes.
 * Read-only: HR_OPT_FRAME_KERNEL_ACTIVE whether hr_render currently takes the frame kernel; HR_OPT_MLP_PRECISION_ACTIVE the HR_MLP_*
 * arithmetic the MLP kernels run (HR_MLP_AUTO resolved); HR_OPT_MLP_CALIBRATED 0 / 1 (finalize's synthetic rays) / 2 (hr_model_calibrate);
 * HR_OPT_MLP_OVERFLOW the sticky bit the fp16-split kernels set when an input feature or hidden activation of a RENDERED ray reached
 * the IEEE-half range (reading it synchronises the device; cleared by hr_model_finalize / hr_model_calibrate); HR_OPT_MLP_F8_SATURATED the
 * sticky bit HR_MLP_F16F8's kernels set when a hidden activation of a rendered ray was beyond the range of its fp8 image (16x the calibration's
 * largest activation of that layer): the image saturates, the ray's correction products lose accuracy (towards HR_MLP_F16X2's), nothing
 * overflows -- hr_model_calibrate on such rays moves the exponents and clears the bit.
 * HR_OPT_MLP_VERIFIED 1 when hr_render runs the verified fast path (HR_MLP_F16F8V); HR_OPT_REDO_COUNT the number of rays the last hr_render listed
 * for its second pass (reading it synchronises the device); HR_OPT_REDO_OVERFLOW the sticky bit raised when a call listed more rays than the
 * list holds -- max(32 768, n_rays / 16) per call, i.e. more than 6.25 % of a large call's rays at risk, which the calibration's 5 % rule
 * makes a property of rays unlike the calibration's: the excess rays keep their first-pass pixels.  A caller that renders without
 * hyperreel_amd's host guard polls this bit (hr_model_calibrate on such rays re-decides; HR_MLP_F16X3 never lists).
 * HR_OPT_WIDE_COUNT: rays the last hr_render passed on to the THIRD pass (bf16x3 tiles: halves with the fp32 exponent range) because an activation of
 * theirs left the IEEE-half range in the second -- the device-side form of the overflow fallback, inside a captured graph too.  In the verified
 * mode HR_OPT_MLP_OVERFLOW / HR_OPT_MLP_F8_SATURATED are raised only for rays that could not be listed (a full list).
 * HR_OPT_CHUNK_RAYS: rays per launch of the head workspace (hr_model_reserve's, or hr_model_finalize's default): what hr_stage_* accept. */
enum { HR_OPT_FRAME_KERNEL = 0, HR_OPT_SAMPLE_WAVES = 1, HR_OPT_FRAME_KERNEL_ACTIVE = 2, HR_OPT_MLP_PRECISION_ACTIVE = 3,
       HR_OPT_MLP_OVERFLOW = 4, HR_OPT_MLP_CALIBRATED = 5, HR_OPT_TRAIN_DETERMINISTIC = 6, HR_OPT_MLP_F8_SATURATED = 7,
       HR_OPT_MLP_VERIFIED = 8, HR_OPT_REDO_COUNT = 9, HR_OPT_REDO_OVERFLOW = 10, HR_OPT_WIDE_COUNT = 11, HR_OPT_CHUNK_RAYS = 12 };
int hr_model_set_option(hr_model* m, int32_t option, int32_t value);
int hr_model_get_option(hr_model* m, int32_t option, int32_t* value);

/* What the verified fast path (HR_MLP_F16F8V) rests on for THIS model: the margins inside which a comparison counts as "at risk" and the
 * measurement they were derived from (hr_model_finalize on 4096 synthetic rays, hr_model_calibrate on the caller's, again after
 * hr_model_update_config).  The reference needs none of this: its MLP is fp32 (nlf/nets/mlp.py:159-172) and its decisions are exact
 * comparisons (nlf/intersect/base.py:194, utils/intersect_utils.py:45-150).
 * The margins are per SAMPLE (csrc/hr_math.h, HrRisk): with zc = z * scale + anchor (process_z_vals, nlf/intersect/base.py:128-140, before
 * the inverse contraction), dlen = |d length / d zc| of the inverse contraction and amp = |d distance / d length| of the intersection
 * (1 / |d_axis| for a plane, 2 |r| / sqrt(discriminant) for sphere and cylinder), a length is at risk within band * dlen, a distance within
 * band * dlen * amp, a point coordinate within band_q * (largest amp of the ray) + band_off. */
typedef struct hr_verify_info {
    int32_t verified;            /* 1: hr_render runs the verified fast path */
    int32_t fallback;            /* HR_MLP_AUTO gave the fast path up for this model (it renders F16X3): 1 = `listed_frac` exceeded 0.05,
                                  * 2 = `max_d_rgb` exceeded 6e-5 (the cheap arithmetic's own error is too large on these weights) */
    float band;                  /* margin of zc: max(band_floor, 4 * max(max_d_zc, max_d_dist_n)) */
    float band_q;                /* margin of a point coordinate per unit of amplification: max(band_floor, 4 * max_d_geo_n) */
    float band_off;              /* margin of the point-offset / flow heads: 4 * max_d_off */
    float band_floor;            /* 1e-6: four float32 ulps of the largest |zc| (2) */
    float max_d_zc;              /* largest |zc(f16f8) - zc(f16x3)| over all calibration samples */
    float max_d_dist_n;          /* largest |distance(f16f8) - distance(f16x3)| / (dlen * amp) over the samples alive under both: the margins' model, checked */
    float max_d_geo_n;           /* largest point difference caused by the distance difference, / amp */
    float max_d_off;             /* largest point difference caused by the offset / flow heads */
    float max_d_dist;            /* largest |distance difference| as it is (scene units; reported) */
    float max_d_head;            /* largest difference of a raw head column (reported) */
    float listed_frac;           /* fraction of the well-conditioned calibration rays the first pass lists with these margins (hr_model_calibrate: of
                                  * all the caller's rays, the ill-conditioned ones -- always listed -- included) */
    float max_d_rgb;             /* largest |rgb(verified path) - rgb(F16X3)| over the well-conditioned calibration rays: f16f8's continuous error on THIS model */
    int64_t n_rays;              /* calibration rays */
    int64_t n_rays_used;         /* ... of which well-conditioned (no live sample with amp > 2): every figure above is taken over these; at render time a ray
                                  * with a live sample beyond that is listed by that alone.  Fewer than 64: listed_frac and max_d_rgb are not measured */
    int64_t n_samples;           /* samples alive under both arithmetics */
    int64_t n_flipped;           /* samples left out because their normalised distance moved by more than 1e-3 (a decision fell the other way) */
    int64_t n_shaky;             /* samples left out because a radius, root or discriminant was exactly 0 */
} hr_verify_info;
int hr_model_verify_info(hr_model* m, hr_verify_info* out);

/* Opt-in occupancy early-reject of the render path.  The reference builds an AlphaGridMask while training
 * (TensorBase.updateAlphaMask) and carries the test in its forward -- `alphas = self.alphaMask.sample_alpha(xyz_sampled[ray_valid]);
 * ray_valid &= alphas > 0` (nlf/nets/tensorf_no_sample.py:171-177, utils/tensorf_utils.py:459-484) -- but ships it disabled
 * (`... and False`).  With a volume set, hr_render / hr_render_fields apply exactly that test to every valid sample before the
 * feature gather: rejected samples are not fetched and composite with sigma = 0, i.e. the image is the reference's with its
 * `and False` removed (pinned by a fixture rendered that way).  volume_dev: (n[2], n[1], n[0]) floats as AlphaGridMask.alpha_volume
 * holds them (x fastest), copied; aabb: its box [min xyz, max xyz].  NULL volume: back to the shipped behaviour.  Waits for `stream`. */
int hr_model_set_occupancy(hr_model* m, const float* volume_dev, const int32_t n[3], const float aabb[6], void* stream);

/* rgb_dev[n,3] = render_fn(rays_dev[n,ray_dim])['rgb']  (eval mode: clamped to [0,1]). */
int hr_render(hr_model* m, const float* rays_dev, int64_t n_rays, float* rgb_dev, void* stream);
/* hr_render for ONE FRAME of a keyframe net: the caller states that the last column of every ray is `time` (what
 * get_coords_from_camera builds for a frame, datasets/base.py:485-518; the viewer and validation_video render frame by frame,
 * nlf/__init__.py:754-893).  Every sample of the frame then blends the SAME two keyframe rows of each time plane with the same
 * weights (TensorVMKeyframeTime's grid_sample over (position, time), nlf/nets/tensorf_dynamic.py:287-371): the library folds them
 * into one line per time plane on `stream` first and the gather reads lines, as it does for a static net (2 taps instead of 4 per
 * plane pair and sample).  The blend is re-associated -- (b00 wt0 + b10 wt1) wx0 + ... instead of b00 (wx0 wt0) + ... -- so images
 * agree with hr_render's to ~1e-6, not bit for bit.  Static nets, cascades and float16 texels take hr_render's path unchanged.
 * Rays whose time differs from `time` are rendered with `time`'s rows (undefined with respect to the reference).  The lines are
 * per-model scratch: frames of one model at different times must not be in flight on different streams at once. */
int hr_render_frame(hr_model* m, const float* rays_dev, int64_t n_rays, float time, float* rgb_dev, void* stream);
int hr_render_fields(hr_model* m, const float* rays_dev, int64_t n_rays, float* rgb_dev,
                     const hr_fields* fields, void* stream);

/* Image-parallel frames: every rank renders a contiguous pixel range of the frame into `tile_dev` and the tiles are
 * assembled on every rank by ONE all-gather over RCCL / xGMI: full_dev[r * floats_per_rank ..] = rank r's tile_dev[0 .. floats_per_rank).
 * `nccl_comm` is the caller's ncclComm_t (one process per GPU; the communicator is the integrator's: torch.distributed's, MPI's, ...),
 * passed as void* so that this header stays free of RCCL types; the collective is enqueued on `stream` (hipGraph-capturable like
 * hr_render; no host synchronisation).  The library resolves ncclAllGather from the RCCL already loaded into the process (or
 * librccl.so) on first use -- it has no link-time dependency on it; HR_E_HIP when no RCCL can be found.  The reference has no
 * counterpart: it shards whole validation images over DDP ranks and never gathers pixels (nlf/__init__.py:896). */
int hr_allgather_tiles(void* nccl_comm, const float* tile_dev, float* full_dev, int64_t floats_per_rank, void* stream);
/* pixel range [first, first + count) of rank `rank` of `world` for an image of n_pixels (even contiguous split, the first
 * n_pixels % world ranks take one pixel more); floats_per_rank for hr_allgather_tiles is 3 * the count of rank 0 (the largest). */
int hr_shard_range(int64_t n_pixels, int32_t rank, int32_t world, int64_t* first, int64_t* count);

/* rays_dev[n_pixels, ray_dim] for pixels [first_pixel, first_pixel + n_pixels) of the image in
 * row-major order: get_ray_directions_K(centered_pixels=True) + get_rays(normalize=True)
 * (utils/ray_utils.py:98-135) [+ cam_id, time when ray_dim == 8].  Generating the rays on the
 * device from 80 bytes of camera replaces the 15-20 MB host->device copy of a frame's ray list;
 * with image-parallel rendering every rank generates only its own pixel range. */
int hr_generate_rays(const hr_camera* cam, int32_t ray_dim, int64_t first_pixel, int64_t n_pixels, float* rays_dev, void* stream);

/* Grid management: F.interpolate(plane, size=(h2, w2), mode='bilinear', align_corners=True) of one
 * (1, C, H, W) float32 plane or line, as TensorVMSplit.up_sampling_VM / TensorVMKeyframeTime.up_sampling_VM apply it
 * when the reference grows its grids (nlf/nets/tensorf_base.py:1152-1176, tensorf_dynamic.py:395-427).  Both buffers
 * are device memory in the reference layout; no model is involved. */
int hr_upsample_plane(const float* src_dev, int32_t channels, int32_t h, int32_t w, float* dst_dev, int32_t h2, int32_t w2, void* stream);

/* ---- training path --------------------------------------------------------------------------
 * What torch.autograd does for the reference in INRSystem.training_step (nlf/__init__.py:634-709), for the stage
 * after the MLP: forward without the eval-mode clamp (nlf/nets/tensorf_no_sample.py:246) and the reverse-mode
 * derivative of Intersect.forward (nlf/intersect/base.py:142-259), the point embeddings (nlf/embedding/point.py:371-396,
 * 780-831), TensorVMNoSample / TensorVMKeyframeTime.forward (tensorf_no_sample.py:128-280, tensorf_dynamic.py:645-839)
 * and raw2alpha (utils/tensorf_utils.py:242-253).  The MLP itself (plain GEMMs) stays with the caller's autograd:
 * hr_train_features gives its input, `head_dev` is its raw output (n_rays, z_channels * preds_per_z) row-major in the
 * caller's column order, and hr_train_backward returns dL/d head for it.
 * All tensors are device memory, float32, in the reference's parameter layouts (same shapes as the hr_model_upload
 * names): a[j] = density_plane.j | density_plane_space.j, b[j] = density_line.j | density_plane_time.j, likewise app_*.
 * Supported: every model hr_model_create / hr_model_create_cascade accepts, with float32 grids (float16 grids return
 * HR_E_INVALID naming the feature).
 * Activation / encoding / mask schedules are those of the model's current configuration (hr_model_update_config). */
typedef struct hr_train_tensors {
    float* density_a[3];
    float* density_b[3];
    float* app_a[3];
    float* app_b[3];
    float* basis;                        /* basis_mat.weight (app_dim, sum n_app) */
    float* color_table;                  /* color_embedding (color_table_views, 12); NULL / ignored when the model has none */
} hr_train_tensors;

/* point_prediction cascades (nlf/embedding/point.py:137-203): the same three calls serve their fine level -- head_dev is then
 * the point MLP's raw output, one row of z_channels / casc_in_z samples per coarse point, which is (n_rays, z_channels *
 * preds_per_z) in memory -- and hr_train_features gives the input of the cascade's RAY MLP.  Between the two MLPs:
 *   hr_train_rows_forward   ray MLP head (n, Zc * Pc) -> rows_dev (n * Zc, casc_row_dim): per coarse sample the point after
 *                           the first intersect next to the ray constants the YAML lists, i.e. the point MLP's input row
 *                           (before its own positional encoding, which stays with the caller's autograd);
 *   hr_train_rows_backward  d_rows_dev -> d_head_dev (n, Zc * Pc); rows_scratch_dev: (n * Zc, casc_row_dim) workspace. */
int hr_train_rows_forward(hr_model* m, const float* rays_dev, const float* head_dev, int64_t n_rays, float* rows_dev, void* stream);
int hr_train_rows_backward(hr_model* m, const float* rays_dev, const float* head_dev, const float* d_rows_dev, int64_t n_rays,
                           float* rows_scratch_dev, float* d_head_dev, void* stream);

/* rays (n, ray_dim) -> feats_dev (n, mlp_in): ray parameterisation + positional encoding (nlf/param.py, nlf/pe.py) */
int hr_train_features(hr_model* m, const float* rays_dev, int64_t n_rays, float* feats_dev, void* stream);

/* rgb_dev (n, 3), not clamped.  `params`: current parameter values, re-packed into the model's texel layout on `stream`
 * before the launch (NULL: keep what the model holds).  white_bg: this step's background decision
 * (`white_bg or (training and rand() < 0.5)`, tensorf_no_sample.py:236). */
int hr_train_forward(hr_model* m, const hr_train_tensors* params, const float* rays_dev, const float* head_dev, int64_t n_rays,
                     int32_t white_bg, float* rgb_dev, void* stream);
/* hr_train_forward that also writes per-sample values of ITS OWN forward -- fields->distances_dev (n, Z), points_dev (n, Z, 3),
 * weights_dev (n, Z), by sorted rank, the definitions of hr_render_fields; NULL members are skipped, sigma_dev / head_dev must be NULL --
 * so that a training step whose regularisers read such fields (INRSystem.training_step passes their names on the main forward,
 * nlf/__init__.py:658-690) needs no second, inference pass over re-uploaded weights.  Up to 64 samples per ray. */
int hr_train_forward_fields(hr_model* m, const hr_train_tensors* params, const float* rays_dev, const float* head_dev, int64_t n_rays,
                            int32_t white_bg, float* rgb_dev, const hr_fields* fields, void* stream);

/* Given d_rgb_dev (n, 3) writes d_head_dev (n, z_channels * preds_per_z) and every non-NULL tensor of `grads` (overwritten,
 * not accumulated), for the parameter values of the last hr_train_forward / hr_model_finalize. */
int hr_train_backward(hr_model* m, const float* rays_dev, const float* head_dev, const float* d_rgb_dev, int64_t n_rays,
                      int32_t white_bg, float* d_head_dev, const hr_train_tensors* grads, void* stream);

/* The Linear layers of the sample-prediction MLP in training (BaseMLP.forward under autograd, nlf/nets/mlp.py:159-172;
 * INRSystem.training_step, nlf/__init__.py:634-709), no model involved; all tensors device memory, float32, row-major with
 * the given leading dimensions (so that the `cat([input, x])` of a skip layer can be a view into a wider buffer).  Every
 * fp32 GEMM runs on the matrix cores as bf16 MFMA products of split operands with fp32 accumulation (bf16 parts because
 * gradients span the fp32 exponent range): the backward GEMMs as three products of hi/lo halves (the bf16x3 arithmetic of
 * the render path), the forward as six products of a three-way split (24 mantissa bits: its output feeds the sample stage's
 * threshold decisions, which must fall as in the fp32 reference).
 *   forward:   y (rows, out) = act(x (rows, in) W^T + b),  W (out, in) as torch stores nn.Linear.weight;
 *              act = LeakyReLU(leaky_slope) when leaky_slope >= 0, none when < 0 (the last Linear)
 *   backward:  with dy (rows, out) the gradient of y and, for an activated layer, y itself (its sign is LeakyReLU's mask):
 *              dy' = dy * (y > 0 ? 1 : slope);  dx (rows, in) = dy' W (skipped when dx_dev is NULL);  dw (out, in) = dy'^T x;
 *              db (out) = column sums of dy'.  The batch dimension of dw / db is reduced in a fixed order (no atomics).
 *              workspace_dev: hr_linear_workspace(rows, in, out) bytes. */
size_t hr_linear_workspace(int64_t rows, int32_t in, int32_t out);
/* The ray MLP's FORWARD of a training step in ONE launch (BaseMLP.forward, nlf/nets/mlp.py:159-172, as INRSystem.training_step runs it,
 * nlf/__init__.py:658-690): weights_dev[l] / biases_dev[l] are the CURRENT values of the reference's parameters (torch layout (out, in) /
 * (out), device memory) -- the library splits them into bf16 hi / lo tiles on the device first (three v_mfma_f32_32x32x16_bf16 products
 * per fp32 GEMM, fp32 accumulation: head within 7e-6 of max |head| of the fp32 chain; bf16 halves keep the fp32 exponent range
 * whatever the weights become).  rays_dev (n, ray_dim) -> head_dev (n, Z * P) in the caller's column order (columns no stage reads: 0),
 * and the output of hidden Linear l after its LeakyReLU -> acts_dev[l] + ray * act_ld[l] + act_off[l] + feature, fp32 (l < layers - 1;
 * NULL: not kept): what hr_linear_backward needs as `x` of layer l + 1 and `y` of layer l (a skip layer's input starts at column mlp_in
 * of a wider row).  Hidden width 256, no cascades; the activation / encoding schedules are those of the last hr_model_update_config. */
int hr_mlp_train_forward(hr_model* m, const float* const* weights_dev, const float* const* biases_dev, const float* rays_dev, int64_t n_rays,
                         float* const* acts_dev, const int64_t* act_ld, const int32_t* act_off, float* head_dev, void* stream);

int hr_linear_forward(const float* x_dev, int64_t ldx, int64_t rows, int32_t in, const float* w_dev, const float* b_dev, int32_t out,
                      float leaky_slope, float* y_dev, int64_t ldy, void* stream);
int hr_linear_backward(const float* x_dev, int64_t ldx, const float* w_dev, const float* y_dev, int64_t ldy, const float* dy_dev, int64_t ld_dy,
                       int64_t rows, int32_t in, int32_t out, float leaky_slope, float* dx_dev, int64_t ld_dx, float* dw_dev, float* db_dev,
                       float* workspace_dev, void* stream);

/* Occupancy of the feature grids: TensorBase.getDenseAlpha (nlf/nets/tensorf_base.py:381-401) and its keyframe
 * override (nlf/nets/tensorf_dynamic.py:499-536) -- alpha = 1 - exp(-sigma * length) at the n[0] x n[1] x n[2] lattice points
 * aabb0 * (1 - s) + aabb1 * s, s = linspace(0, 1, n) of the model's box, written to alpha_dev (n[0], n[1], n[2]) (x slowest);
 * keyframe nets take the maximum over the `num_frames` frames of the sequence.  prev_volume_dev (D = prev_n[2], H = prev_n[1],
 * W = prev_n[0]; may be NULL) is the previous mask with its box prev_aabb[6]: points where its trilinear sample
 * (AlphaGridMask.sample_alpha, utils/tensorf_utils.py:459-484) is not positive get alpha 0, as in compute_alpha
 * (tensorf_base.py:489-507).  The max-pool / threshold / crop that follow are host-side tensor ops. */
int hr_dense_alpha(hr_model* m, const int32_t n[3], float length, int32_t num_frames, const float* prev_volume_dev, const int32_t prev_n[3],
                   const float prev_aabb[6], float* alpha_dev, void* stream);

/* Viewer hand-over: rgb_dev (h * w, 3) float32 as hr_render wrote it -> out_dev, the buffer NeRFGUI.test_step
 * builds on the host (utils/gui_utils.py:174-205): transposed to (w, h) if `transpose`, then flipped vertically if `flip`;
 * rgba8 != 0: 4 bytes per pixel, to8b(x) = (uint8)(255 * clip(x, 0, 1)) (utils/__init__.py:47) and alpha 255;
 * rgba8 == 0: 3 floats per pixel, values unchanged. */
int hr_pack_display(const float* rgb_dev, int32_t h, int32_t w, int32_t transpose, int32_t flip, int32_t rgba8, void* out_dev, void* stream);

/* TensoRF regularisers of one (1, C, H, W) float32 plane (TVLoss, nlf/regularizers/tensorf.py:14-34; density_L1,
 * nlf/nets/tensorf_base.py:1024-1035), no model involved.  Forward ADDS to sums_dev[3] =
 *   { sum (x[:, 1:, :] - x[:, :-1, :])^2,  sum (x[:, :, 1:] - x[:, :, :-1])^2,  sum |x| };
 * backward writes grad_dev = coef_dev[0] * d sums[0]/dx + coef_dev[1] * d sums[1]/dx + coef_dev[2] * sign(x), with the
 * three upstream coefficients read from device memory (no host synchronisation inside a training step). */
int hr_plane_reg_forward(const float* plane_dev, int32_t channels, int32_t h, int32_t w, float* sums_dev, void* stream);
int hr_plane_reg_backward(const float* plane_dev, int32_t channels, int32_t h, int32_t w, const float* coef_dev, float* grad_dev, void* stream);

/* The optimizer of the training loop (utils/__init__.py:49-76: torch.optim.Adam(params, lr, eps=1e-8, weight_decay, betas=(0.9, 0.99)); stepped by
 * Lightning after INRSystem.training_step, nlf/__init__.py:634-709): one Adam step over any number of parameter tensors in ONE pass over memory
 * (28 bytes per parameter).  Tensor i is n[i] contiguous floats at param_dev[i] with its gradient, first and second moment (torch's state names
 * exp_avg / exp_avg_sq) in grad_dev[i], exp_avg_dev[i], exp_avg_sq_dev[i]; the pointer arrays and hp are HOST memory.  hp holds six doubles per
 * tensor: lr, beta1, beta2, eps, weight_decay, step (the 1-based step count this call performs -- bias corrections are 1 - beta^step; doubles
 * because torch derives 1 - beta and the corrections from Python floats).  The arithmetic is torch/optim/adam.py's single-tensor form
 * (amsgrad / maximize off), evaluated in fp32. */
int hr_adam_step(float* const* param_dev, const float* const* grad_dev, float* const* exp_avg_dev, float* const* exp_avg_sq_dev, const int64_t* n,
                 const double* hp, int32_t n_tensors, void* stream);

/* The two stages of hr_render on their own, for profiling: the sample-prediction MLP
 * (rays -> raw head in the workspace) and the per-sample stage (head -> rgb).  n_rays
 * must not exceed the reserved chunk size. */
int hr_stage_mlp(hr_model* m, const float* rays_dev, int64_t n_rays, void* stream);
int hr_stage_samples(hr_model* m, const float* rays_dev, int64_t n_rays, float* rgb_dev, void* stream);

/* Profiling aid: runs the MLP stage with a per-wave phase timeline.  trace_dev receives 64
 * s_memtime stamps per wave (4 waves per workgroup, workgroups of 64 or 128 rays); only the
 * split-precision kernel records stamps. */
int hr_debug_trace_mlp(hr_model* m, const float* rays_dev, int64_t n_rays, unsigned long long* trace_dev, void* stream);

/* bytes of device memory held by the model (packed grids + weights + workspace) */
int64_t hr_model_device_bytes(const hr_model* m);

void hr_model_destroy(hr_model* m);

#ifdef __cplusplus
}
#endif
#endif /* HYPERREEL_HIP_H */
