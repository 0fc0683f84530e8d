/*
 * ex4d_knn.h -- C ABI of distCUDA2: mean squared distance of every point to its 3 nearest neighbours.
 *
 * Replaces simple_knn._C.distCUDA2 of the reference (submodules/simple-knn/spatial.cu:15-27 -> SimpleKNN::knn,
 * simple_knn.cu:185-221), called once at initialisation to set the Gaussian scales (scene/c_gaussian_model.py:395).
 * Result definition (simple_knn.cu:129-183): for point i, the three smallest values of
 *     (x_j - x_i)^2 + (y_j - y_i)^2 + (z_j - z_i)^2        over j != i  (by index: coincident points count, distance 0)
 * averaged as (b0 + b1 + b2) / 3 with b0 <= b1 <= b2; slots never filled stay FLT_MAX (P < 4 gives inf).
 * The spatial structure used to find them (Morton order + bounding boxes) only prunes; the result is the exact 3-NN.
 */
#ifndef EX4D_KNN_H_INCLUDED
#define EX4D_KNN_H_INCLUDED

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

const char *ex4d_knn_last_error(void);

/* bytes of device scratch ex4d_dist2 needs for P points */
size_t ex4d_dist2_scratch_bytes(int32_t P);

/* points: device float [P,3]; mean_dist2: device float [P] (fully written); scratch: device, >= ex4d_dist2_scratch_bytes(P),
 * 256-byte aligned; stream: hipStream_t.  No host synchronisation. */
int ex4d_dist2(int32_t P, const float *points, float *mean_dist2, void *scratch, void *stream);

#ifdef __cplusplus
}
#endif
#endif
