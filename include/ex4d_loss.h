/*
 * ex4d_loss.h -- C ABI of the fused L1 + SSIM training loss, forward and backward.
 *
 * Replaces, as one forward and one backward call, what train.py:144-151 of the reference builds out of
 * utils/loss_utils.py:22-25 (l1_loss) and :47-81 (ssim/_ssim: five depthwise 11x11 Gaussian-window conv2d, sigma 1.5,
 * zero padding 5) plus its autograd graph:
 *     loss        = (1 - lambda) * mean|img - gt| + lambda * (1 - mean(ssim_map))
 *     l1_errors   = mean_c |img - gt|          [H,W]      (train.py:149, hook tensor of the flow channel)
 *     ssim_errors = mean_c ssim_map            [H,W]      (train.py:150)
 * All pointers are device pointers (float32, [C,H,W] contiguous) except `window`, a HOST array of the 11 normalised 1-D
 * Gaussian taps (loss_utils.py:32-34; the 2-D window is their outer product, :38-39).  `stream` is a hipStream_t.
 */
#ifndef EX4D_LOSS_H_INCLUDED
#define EX4D_LOSS_H_INCLUDED

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EX4D_SSIM_WINDOW 11

const char *ex4d_loss_last_error(void);

/* floats of scratch the forward needs besides its outputs (per-workgroup partial sums) */
size_t ex4d_l1_ssim_scratch_floats(int32_t H, int32_t W);

/* Forward.  loss[1]; l1_errors / ssim_errors [H,W] (either may be NULL); dmaps[3][C][H][W] receives the three per-pixel
 * partial derivatives dS/dmu1, dS/dE[x^2], dS/dE[xy] of the SSIM map that the backward convolves (kept for backward). */
int ex4d_l1_ssim_forward(int32_t C, int32_t H, int32_t W, const float *img, const float *gt, float lambda_dssim,
                         const float *window /* host [11] */, float *loss, float *l1_errors, float *ssim_errors,
                         float *dmaps, float *scratch, void *stream);

/* Backward: grad_img[C,H,W] = grad_loss[0] * dloss/dimg (fully written). */
int ex4d_l1_ssim_backward(int32_t C, int32_t H, int32_t W, const float *img, const float *gt, float lambda_dssim,
                          const float *window /* host [11] */, const float *dmaps, const float *grad_loss /* device [1] */,
                          float *grad_img, void *stream);

#ifdef __cplusplus
}
#endif
#endif
