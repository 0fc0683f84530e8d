/*
 * ex4d_attributes.h -- C ABI of the fused per-frame attribute evaluation of the static + keyframe-interpolated
 * dynamic Gaussians, the producer of the five tensors the rasterizer boundary consumes.
 *
 * Replaces, as one forward and one backward call, the reference's Python getters and their autograd graph:
 *   CGaussianModel.get_xyz_at_t / get_rotation_at_t        scene/c_gaussian_model.py:170-215
 *   get_scaling / get_features / get_opacity_at_t          scene/c_gaussian_model.py:330-375
 *   cube_interpolate / quat_slerp_interp_uniiterval / time_bigaussian   utils/interpolations.py:81-93, :33-52, :55-61
 * ('cube' position interpolation, 'slerp' rotation interpolation: the configuration of configs/N3V and configs/techni).
 *
 * All pointers are device pointers (float32, contiguous, shapes as in CGaussianModel); `stream` is a hipStream_t.
 * Outputs are [Ns+Nd, ...] with the static rows first (c_gaussian_model.py:193, :215, :374).
 */
#ifndef EX4D_ATTRIBUTES_H_INCLUDED
#define EX4D_ATTRIBUTES_H_INCLUDED

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Host-evaluated scalars of one timestamp t (Python numbers in the reference, c_gaussian_model.py:184-186, :364):
 * t' = t + time_shift; k = t' // interval; delta = (t' % interval) / interval; Hermite basis at delta
 * (utils/interpolations.py:83-86, evaluated in double, used as float32); tau = t' / interval; var_min = var_pad / interval. */
typedef struct Ex4dAttrParams {
    int32_t Ns, Nd;            /* static / dynamic Gaussian counts */
    int32_t K;                 /* keyframes per dynamic Gaussian (_xyz_motion.shape[1]) */
    int32_t k;                 /* keyframe index of this timestamp; slices k-1 .. k+2 are read */
    float t, duration;         /* static positions: _xyz + _xyz_disp * t / duration */
    float delta;               /* slerp parameter in [0,1) */
    float h00, h10, h01, h11;  /* Hermite basis at delta */
    float tau;                 /* time in keyframe units for the bi-Gaussian opacity window */
    float var_min;             /* var_pad / interval */
} Ex4dAttrParams;

const char *ex4d_attributes_last_error(void);

/* Forward.  Parameters in CGaussianModel order; outputs means3D[N,3], rotations[N,4], opacities[N,1], scales[N,3], shs[N,16,3].
 * shs may be NULL (then the four feature tensors are not read): the rasterizer can take them as they are, see Ex4dSplitSH in
 * ex4d_rasterizer.h; likewise g_shs may be NULL in the backward (the four feature gradients are then not written). */
int ex4d_attributes_forward(const Ex4dAttrParams *a,
    const float *xyz /*[Ns,3]*/, const float *xyz_disp /*[Ns,3]*/, const float *rotation /*[Ns,4]*/, const float *opacity /*[Ns,1]*/,
    const float *scaling /*[Ns,3]*/, const float *features_dc /*[Ns,1,3]*/, const float *features_rest /*[Ns,15,3]*/,
    const float *xyz_motion /*[Nd,K,3]*/, const float *rotation_motion /*[Nd,K,4]*/, const float *opacity_motion /*[Nd,1]*/,
    const float *opacity_duration_center /*[Nd,2,1]*/, const float *opacity_duration_var /*[Nd,2,1]*/,
    const float *scaling_motion /*[Nd,3]*/, const float *features_dc_motion /*[Nd,1,3]*/, const float *features_rest_motion /*[Nd,15,3]*/,
    float *means3D, float *rotations, float *opacities, float *scales, float *shs, void *stream);

/* Backward: every gradient tensor is fully written (the dense keyframe gradients are zero-filled, then the 4 position / 2
 * rotation slices of this timestamp are set), so callers may pass uninitialised memory. */
int ex4d_attributes_backward(const Ex4dAttrParams *a,
    const float *opacity, const float *scaling, const float *rotation_motion, const float *opacity_motion,
    const float *opacity_duration_center, const float *opacity_duration_var, const float *scaling_motion,
    const float *g_means3D, const float *g_rotations, const float *g_opacities, const float *g_scales, const float *g_shs,
    float *g_xyz, float *g_xyz_disp, float *g_rotation, float *g_opacity, float *g_scaling, float *g_features_dc, float *g_features_rest,
    float *g_xyz_motion, float *g_rotation_motion, float *g_opacity_motion, float *g_opacity_duration_center,
    float *g_opacity_duration_var, float *g_scaling_motion, float *g_features_dc_motion, float *g_features_rest_motion, void *stream);

/* The same backward with the keyframe gradients as SLICES instead of dense tensors: g_xyz_motion_slices[Nd,4,3] holds the gradients of
 * keyframes slices[0] .. slices[0]+3, g_rotation_motion_slices[Nd,2,4] those of keyframes slices[2], slices[2]+1 -- every other time
 * slice of the dense gradient is zero for this timestamp.  slices = int32[4] {xyz first, xyz count (4), rotation first, rotation count (2)},
 * written on the host (the slice hint an optimizer / gradient exchange needs).  No zero fill of 7 K floats per dynamic Gaussian, and
 * ex4d_radam_step_sliced (ex4d_optim.h) consumes the slices directly. */
int ex4d_attributes_backward_sliced(const Ex4dAttrParams *a,
    const float *opacity, const float *scaling, const float *rotation_motion, const float *opacity_motion,
    const float *opacity_duration_center, const float *opacity_duration_var, const float *scaling_motion,
    const float *g_means3D, const float *g_rotations, const float *g_opacities, const float *g_scales, const float *g_shs,
    float *g_xyz, float *g_xyz_disp, float *g_rotation, float *g_opacity, float *g_scaling, float *g_features_dc, float *g_features_rest,
    float *g_xyz_motion_slices, float *g_rotation_motion_slices, float *g_opacity_motion, float *g_opacity_duration_center,
    float *g_opacity_duration_var, float *g_scaling_motion, float *g_features_dc_motion, float *g_features_rest_motion,
    int32_t *slices, void *stream);

#ifdef __cplusplus
}
#endif
#endif
