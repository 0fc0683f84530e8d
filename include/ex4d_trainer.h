/*
 * ex4d_trainer.h -- C ABI of the compiled host path of one training iteration.
 *
 * One call = the per-iteration device work of the reference's training loop for one view (train.py:124-153, :244-255), driven from C++
 * with a workspace that lives across iterations -- no Python, no autograd graph, no per-iteration allocation:
 *     ex4d_attributes_forward                  (getters of scene/c_gaussian_model.py:170-215, :330-375)
 *  -> ex4d_forward_split_sh                    (render: gaussian_renderer/__init__.py:19-124, the SH tensors as the model stores them)
 *  -> ex4d_l1_ssim_forward / _backward         (train.py:144-151, utils/loss_utils.py)
 *  -> ex4d_backward_split_sh
 *  -> ex4d_attributes_backward_sliced          (keyframe gradients as the 4 / 2 touched time slices)
 *  -> ex4d_radam_step + ex4d_radam_step_sliced (scene/c_gaussian_model.py:430-449, train.py:250)
 * all on the caller's stream.  The rasterizer's scratch buffers are capacity-bounded arenas of the trainer (grown geometrically when
 * a frame needs more), so the only host <-> device synchronisation of an iteration is the instance-count read-back the reference has
 * too (rasterizer_impl.cu:298-299), which overlaps the depth sort.
 *
 * SCOPE: this is the render + L1/SSIM + RAdam core of the iteration, not the reference's whole loop.  Not included (they live in the
 * reference's Python policy layer, out of SURVEY.md 8's scope): the regularisers static_reg / motion_reg / rot_reg (train.py:156-168;
 * motion_reg and rot_reg make the keyframe gradients dense over all K, which the sliced optimizer path here does not take), the
 * l1_accum error-map hook on the flow output (train.py:149-152: dL_dout_flow is NULL here, so viewspace_l1points stays zero),
 * densification / pruning and their statistics.  What the loop changes over time is settable: ex4d_trainer_set_lr (the position
 * learning-rate schedule, update_learning_rate) and ex4d_trainer_set_sh_degree (oneupSHdegree every 1000 iterations).
 * _opacity_duration_var's gradient is read through nan_to_num like train.py:244-247 does before optimizer.step().
 *
 * The 15 parameter tensors stay the caller's (device pointers, CGaussianModel order, float32 contiguous); they are updated in place.
 * Optimizer state (exp_avg, exp_avg_sq, step counts) and every intermediate belong to the trainer.
 */
#ifndef EX4D_TRAINER_H_INCLUDED
#define EX4D_TRAINER_H_INCLUDED

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EX4D_TRAINER_PARAMS 15      /* _xyz _xyz_disp _rotation _opacity _scaling _features_dc _features_rest _xyz_motion _rotation_motion
                                       _opacity_motion _opacity_duration_center _opacity_duration_var _scaling_motion _features_dc_motion
                                       _features_rest_motion   (ex4dgs_amd/attributes.py: PARAM_ORDER) */

typedef struct Ex4dTrainerConfig {
    int32_t Ns, Nd, K;                  /* static / dynamic Gaussians, keyframes per dynamic Gaussian */
    int32_t W, H;
    int32_t sh_degree;                  /* active SH degree (0..3) */
    float tanfovx, tanfovy, kernel_size;
    float min_depth, max_depth;
    double duration, interval, time_shift, var_pad;     /* the Python numbers of c_gaussian_model.py:184-186, :364 */
    float lambda_dssim;
    float window[11];                   /* normalised 1-D SSIM taps (utils/loss_utils.py:32-34), float32 as the reference computes them */
    double lr[EX4D_TRAINER_PARAMS];
    double beta1, beta2, eps;
    int32_t optimizer;                  /* 0: gradients only (ex4d_trainer_grad reads them), 1: RAdam step */
} Ex4dTrainerConfig;

typedef struct Ex4dTrainer Ex4dTrainer;

const char *ex4d_trainer_last_error(void);

/* params: HOST array of the 15 device pointers.  Returns NULL on failure (ex4d_trainer_last_error). */
Ex4dTrainer *ex4d_trainer_create(const Ex4dTrainerConfig *cfg, float *const *params);
void ex4d_trainer_destroy(Ex4dTrainer *t);

/* The host-side scalars of one timestamp exactly as the Python reference computes them (float floor division / modulo / ** in double,
 * then float32) -- what ex4d_trainer_step passes to ex4d_attributes_forward; exported so that the arithmetic can be pinned without a GPU. */
struct Ex4dAttrParams;
void ex4d_trainer_time_scalars(const Ex4dTrainerConfig *cfg, double timestamp, struct Ex4dAttrParams *out);

/* One iteration at timestamp t for the camera (viewmatrix [16], projmatrix [16], campos [3]: device), background [3] (device) against
 * gt_image [3,H,W] (device).  Asynchronous on `stream` apart from the instance-count read-back.  Returns EX4D_OK or an error code of
 * ex4d_rasterizer.h.  *num_rendered (host, may be NULL) receives the instance count. */
int ex4d_trainer_step(Ex4dTrainer *t, double timestamp, const float *viewmatrix, const float *projmatrix, const float *campos,
                      const float *background, const float *gt_image, void *stream, int32_t *num_rendered);

/* The per-group learning rates (15, PARAM order) / the active SH degree used from the next step on: update_learning_rate
 * (c_gaussian_model.py:451-470) and oneupSHdegree (train.py:113-114) of the reference change them during training. */
int ex4d_trainer_set_lr(Ex4dTrainer *t, const double *lr15);
int ex4d_trainer_set_sh_degree(Ex4dTrainer *t, int32_t degree);
/* on != 0: the rasterizer forward runs ASYNCHRONOUSLY (ex4d_rasterizer.h: Ex4dParams.instance_capacity) -- no instance-count read-back
 * in the middle of the frame; the binning arena is sized for 1.25 x the largest instance count seen (the first frame runs synchronously
 * and seeds it), the frame's status is looked at once, right before the optimizer step, and a frame that overflowed its capacity is run
 * again before anything is applied: same parameters as the synchronous path.  ex4d_trainer_replays counts such re-runs. */
int ex4d_trainer_set_async(Ex4dTrainer *t, int32_t on);
int64_t ex4d_trainer_replays(const Ex4dTrainer *t);

/* Device pointers into the trainer's workspace, valid until the next step / destroy:
 * what = 0 loss [1], 1 render [3,H,W], 2 radii int32 [P], 3 dL_dmeans2D [P,3] (viewspace gradient, densification statistics),
 *        4 depth [1,H,W], 5 acc [1,H,W]. */
const void *ex4d_trainer_output(const Ex4dTrainer *t, int32_t what);
/* Gradient of parameter i of the last step: dense [shape of the parameter] except i = 7 (_xyz_motion: [Nd,4,3]) and i = 8
 * (_rotation_motion: [Nd,2,4]), the slices of ex4d_attributes_backward_sliced; slices4 (host int32[4], may be NULL) receives
 * {xyz first, 4, rotation first, 2}. */
const float *ex4d_trainer_grad(const Ex4dTrainer *t, int32_t i, int32_t *slices4);
/* Asynchronous device-to-device copy of one of the buffers above into caller memory (bindings that cannot wrap a raw pointer):
 * what = 0..5 as in ex4d_trainer_output, 100 + i = gradient of parameter i.  `bytes` must not exceed the buffer's size. */
int ex4d_trainer_read(const Ex4dTrainer *t, int32_t what, void *dst, size_t bytes, void *stream);
/* bytes of device memory the trainer holds (workspace + optimizer state + arenas) */
size_t ex4d_trainer_bytes(const Ex4dTrainer *t);

#ifdef __cplusplus
}
#endif
#endif
