/*
 * ex4d_optim.h -- C ABI of the fused multi-tensor RAdam step.
 *
 * Replaces `gaussians.optimizer.step()` (train.py:250 of the reference) for the optimizer the reference builds at
 * scene/c_gaussian_model.py:449, `torch.optim.RAdam(l, lr=0.001)` over 15 parameter groups with per-group learning rates
 * (:430-447; defaults betas=(0.9, 0.999), eps=1e-8, weight_decay=0).  torch.optim is a third-party dependency of the
 * reference (environment.yml:10 pins pytorch=2.1.2); the algorithm restated here is its documented one
 * (Liu et al., "On the Variance of the Adaptive Learning Rate and Beyond", as implemented by torch's _single_tensor_radam):
 *     m <- m + (1-b1)(g - m);  v <- b2 v + (1-b2) g g;  mhat = m / (1 - b1^t)
 *     rho_t = rho_inf - 2 t b2^t / (1 - b2^t),  rho_inf = 2/(1-b2) - 1
 *     rho_t > 5:  p <- p - mhat * lr * sqrt(1 - b2^t)/(sqrt(v) + eps) * rect(rho_t)        else   p <- p - mhat * lr
 * One launch updates every tensor: 28 bytes of HBM traffic per element (p, m, v read+write, g read), nothing else.
 * All scalar coefficients are computed on the host in double precision exactly like the Python scalars of torch.
 */
#ifndef EX4D_OPTIM_H_INCLUDED
#define EX4D_OPTIM_H_INCLUDED

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EX4D_RADAM_MAX_TENSORS 32

typedef struct Ex4dRadamTensor {
    float *param;            /* device, updated in place */
    const float *grad;       /* device */
    float *exp_avg;          /* device, updated in place */
    float *exp_avg_sq;       /* device, updated in place */
    int64_t numel;
    double lr;               /* the group's learning rate for this step */
    int64_t step;            /* t: the tensor's step count AFTER this step's increment (>= 1) */
    int32_t nan_to_num;      /* != 0: the gradient is read through torch.nan_to_num (NaN -> 0, +-inf -> +-FLT_MAX) -- what train.py:244-247
                                applies to _opacity_duration_var.grad before optimizer.step(); 0 for every other tensor */
    int32_t reserved;
} Ex4dRadamTensor;

const char *ex4d_optim_last_error(void);

/* tensors: HOST array of `count` descriptors (count <= EX4D_RADAM_MAX_TENSORS per call).  stream: hipStream_t. */
int ex4d_radam_step(const Ex4dRadamTensor *tensors, int32_t count, double beta1, double beta2, double eps, void *stream);

/* The same step for a keyframe tensor param[rows, K, C] (C = 3 or 4) whose gradient is known to be zero outside a few time slices:
 * gradient(row, k, c) = sum over the windows w with first[w] <= k < first[w] + count[w] of grad[w][row, k - first[w], c]
 * (ex4d_attributes_backward_sliced writes such windows; several windows = several frames / ranks accumulated).  Every element is
 * still updated -- RAdam moves zero-gradient elements by their momentum -- but the dense gradient is neither zero-filled nor read:
 * 24 instead of 28 bytes per element, and no 7 K-float memset per dynamic Gaussian in the backward.  Bit-identical to ex4d_radam_step
 * on the equivalent dense gradient (windows summed in index order). */
#define EX4D_RADAM_MAX_WINDOWS 8
#define EX4D_RADAM_MAX_SLICED 4
typedef struct Ex4dRadamSlicedTensor {
    float *param;            /* device [rows, K, C], updated in place (a row range of a larger tensor is fine: pass offset pointers) */
    float *exp_avg;
    float *exp_avg_sq;
    int64_t rows;
    int32_t K, C;
    double lr;
    int64_t step;
    int32_t n_windows;       /* 0 .. EX4D_RADAM_MAX_WINDOWS */
    int32_t first[EX4D_RADAM_MAX_WINDOWS], count[EX4D_RADAM_MAX_WINDOWS];
    const float *grad[EX4D_RADAM_MAX_WINDOWS];      /* device [rows, count[w], C] */
    const int32_t *first_dev;                       /* optional DEVICE array of n_windows first-keyframe indices: when non-NULL the kernel
                                                       reads the window positions from it and first[] is ignored (windows gathered from
                                                       other ranks: no device -> host round trip before the launch; a position outside
                                                       [0, K - count] simply matches no element) */
} Ex4dRadamSlicedTensor;

int ex4d_radam_step_sliced(const Ex4dRadamSlicedTensor *tensors, int32_t count, double beta1, double beta2, double eps, void *stream);

#ifdef __cplusplus
}
#endif
#endif
