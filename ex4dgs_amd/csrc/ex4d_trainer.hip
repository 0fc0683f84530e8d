// Compiled host path of one training iteration (include/ex4d_trainer.h): host code only -- it sequences the C-ABI entry points of
// this library on the caller's stream with a persistent workspace.  Nothing here runs on the CPU in place of a kernel.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>

#include "../../include/ex4d_attributes.h"
#include "../../include/ex4d_loss.h"
#include "../../include/ex4d_optim.h"
#include "../../include/ex4d_rasterizer.h"
#include "../../include/ex4d_trainer.h"

namespace {

thread_local char t_err[512] = "";

int tfail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(t_err, sizeof(t_err), fmt, ap);
    va_end(ap);
    return code;
}

// a device buffer that only ever grows (the rasterizer's opaque scratch buffers: the binning buffer depends on the frame's instance count)
struct Arena {
    void *ptr = nullptr;
    size_t cap = 0;
    size_t *total = nullptr;
    void *get(size_t bytes)
    {
        if (bytes <= cap) return ptr;
        // growth is rare (first frames, then never or when a view sees ~25 % more instances than any before it); hipFree waits for the
        // kernels still using the old buffer
        const size_t want = bytes + bytes / 4 + 256;
        void *p = nullptr;
        if (hipMalloc(&p, want) != hipSuccess) return nullptr;
        if (ptr) (void)hipFree(ptr);
        if (total) *total += want - cap;
        ptr = p; cap = want;
        return ptr;
    }
};
void *arena_alloc(void *user, size_t bytes) { return static_cast<Arena *>(user)->get(bytes); }

}  // namespace

struct Ex4dTrainer {
    Ex4dTrainerConfig cfg;
    int P = 0;
    size_t HW = 0;
    size_t bytes = 0;
    int64_t step = 0;
    float *param[EX4D_TRAINER_PARAMS] = {};
    int64_t numel[EX4D_TRAINER_PARAMS] = {};
    int64_t grad_numel[EX4D_TRAINER_PARAMS] = {};
    float *grad[EX4D_TRAINER_PARAMS] = {}, *m[EX4D_TRAINER_PARAMS] = {}, *v[EX4D_TRAINER_PARAMS] = {};
    int32_t slices[4] = { 0, 0, 0, 0 };
    // per-frame tensors
    float *means3D = nullptr, *rotations = nullptr, *opacities = nullptr, *scales = nullptr;
    float *color = nullptr, *depth = nullptr, *acc = nullptr, *flow = nullptr, *loss = nullptr, *dmaps = nullptr, *loss_scratch = nullptr;
    float *grad_img = nullptr, *grad_loss = nullptr;
    int32_t *idx = nullptr, *radii = nullptr;
    float *g_means2D = nullptr, *g_opacity = nullptr, *g_means3D = nullptr, *g_scales = nullptr,
          *g_rotations = nullptr, *g_dir = nullptr;
    void *bwd_scratch = nullptr;
    Arena geom, binning, img;
    // asynchronous rasterizer forward (ex4d_trainer_set_async): capacity of the binning buffer in tile instances (0 = not known yet: the
    // next frame runs synchronously and seeds it), the frame status in pinned host memory, the event behind the forward
    bool async = false;
    uint32_t capacity = 0;
    Ex4dFrameStatus *status = nullptr;
    hipEvent_t status_ev = nullptr;
    int64_t replays = 0;
    void *owned[96] = {};
    int n_owned = 0;

    template <typename T> bool take(T *&p, size_t count, bool zero = false)
    {
        p = nullptr;
        if (count == 0) return true;
        void *q = nullptr;
        if (hipMalloc(&q, count * sizeof(T)) != hipSuccess) return false;
        if (zero && hipMemset(q, 0, count * sizeof(T)) != hipSuccess) { (void)hipFree(q); return false; }
        owned[n_owned++] = q;
        bytes += count * sizeof(T);
        p = static_cast<T *>(q);
        return true;
    }
};

extern "C" {

const char *ex4d_trainer_last_error(void) { return t_err; }

void ex4d_trainer_destroy(Ex4dTrainer *t)
{
    if (!t) return;
    (void)hipDeviceSynchronize();
    for (int i = 0; i < t->n_owned; i++) (void)hipFree(t->owned[i]);
    if (t->geom.ptr) (void)hipFree(t->geom.ptr);
    if (t->binning.ptr) (void)hipFree(t->binning.ptr);
    if (t->img.ptr) (void)hipFree(t->img.ptr);
    if (t->status) (void)hipHostFree(t->status);
    if (t->status_ev) (void)hipEventDestroy(t->status_ev);
    delete t;
}

Ex4dTrainer *ex4d_trainer_create(const Ex4dTrainerConfig *cfg, float *const *params)
{
    t_err[0] = 0;
    if (!cfg || !params) { tfail(1, "null config / parameter list"); return nullptr; }
    if (cfg->Ns < 0 || cfg->Nd < 0 || cfg->Ns + cfg->Nd <= 0 || cfg->W <= 0 || cfg->H <= 0 || (cfg->Nd > 0 && cfg->K < 4)
        || cfg->sh_degree < 0 || cfg->sh_degree > 3 || !(cfg->interval > 0.0)) {
        tfail(1, "trainer config out of range (Ns %d, Nd %d, K %d, %dx%d, SH degree %d)", cfg->Ns, cfg->Nd, cfg->K, cfg->W, cfg->H, cfg->sh_degree);
        return nullptr;
    }
    Ex4dTrainer *t = new (std::nothrow) Ex4dTrainer();
    if (!t) { tfail(3, "out of host memory"); return nullptr; }
    t->cfg = *cfg;
    t->P = cfg->Ns + cfg->Nd;
    t->HW = (size_t)cfg->W * cfg->H;
    t->geom.total = t->binning.total = t->img.total = &t->bytes;
    const int64_t Ns = cfg->Ns, Nd = cfg->Nd, K = cfg->K;
    const int64_t n[EX4D_TRAINER_PARAMS] = { Ns * 3, Ns * 3, Ns * 4, Ns, Ns * 3, Ns * 3, Ns * 45, Nd * K * 3, Nd * K * 4, Nd, Nd * 2, Nd * 2, Nd * 3, Nd * 3, Nd * 45 };
    bool ok = true;
    for (int i = 0; i < EX4D_TRAINER_PARAMS && ok; i++) {
        t->param[i] = params[i];
        t->numel[i] = n[i];
        t->grad_numel[i] = i == 7 ? Nd * 4 * 3 : (i == 8 ? Nd * 2 * 4 : n[i]);
        if (n[i] > 0 && !params[i]) { tfail(1, "parameter %d is NULL but has %lld elements", i, (long long)n[i]); ok = false; break; }
        ok = ok && t->take(t->grad[i], (size_t)t->grad_numel[i]);
        if (cfg->optimizer) ok = ok && t->take(t->m[i], (size_t)n[i], true) && t->take(t->v[i], (size_t)n[i], true);
    }
    const size_t P = (size_t)t->P, HW = t->HW;
    ok = ok && t->take(t->means3D, 3 * P) && t->take(t->rotations, 4 * P) && t->take(t->opacities, P) && t->take(t->scales, 3 * P)
            && t->take(t->color, 3 * HW) && t->take(t->depth, HW) && t->take(t->acc, HW) && t->take(t->flow, 3 * HW) && t->take(t->idx, HW)
            && t->take(t->radii, P) && t->take(t->loss, 1) && t->take(t->dmaps, 9 * HW)
            && t->take(t->loss_scratch, ex4d_l1_ssim_scratch_floats(cfg->H, cfg->W)) && t->take(t->grad_img, 3 * HW) && t->take(t->grad_loss, 1)
            && t->take(t->g_means2D, 3 * P) && t->take(t->g_opacity, P) && t->take(t->g_means3D, 3 * P)
            && t->take(t->g_scales, 3 * P) && t->take(t->g_rotations, 4 * P) && t->take(t->g_dir, 3 * P);
    if (ok) {
        void *s = nullptr;
        ok = t->take(reinterpret_cast<unsigned char *&>(s), ex4d_backward_scratch_bytes(t->P));
        t->bwd_scratch = s;
    }
    if (ok) {
        const float one = 1.0f;
        ok = hipMemcpy(t->grad_loss, &one, sizeof(float), hipMemcpyHostToDevice) == hipSuccess;
    }
    if (!ok) {
        if (!t_err[0]) tfail(3, "device allocation failed (%s)", hipGetErrorString(hipGetLastError()));
        ex4d_trainer_destroy(t);
        return nullptr;
    }
    return t;
}

// The Python-number arithmetic of c_gaussian_model.py:184-186, :364 and interpolations.py:83-86 (double precision, then float32):
// Python's `//` and `%` on floats (floor division; the remainder takes the sign of the divisor, and the quotient is the floor of the
// exact division except where fmod says otherwise -- float_divmod of CPython), `**` = pow.
void ex4d_trainer_time_scalars(const Ex4dTrainerConfig *cfg, double timestamp, Ex4dAttrParams *out)
{
    const Ex4dTrainerConfig &c = *cfg;
    Ex4dAttrParams &a = *out;
    const double tp = timestamp + c.time_shift;
    // CPython float_divmod: mod = fmod(vx, wx); div = (vx - mod) / wx; adjust when the signs differ; floordiv = floor(div) rounded
    double mod = std::fmod(tp, c.interval);
    double div = (tp - mod) / c.interval;
    if (mod != 0.0) {
        if ((c.interval < 0.0) != (mod < 0.0)) { mod += c.interval; div -= 1.0; }
    } else {
        mod = std::copysign(0.0, c.interval);
    }
    double floordiv;
    if (div != 0.0) { floordiv = std::floor(div); if (div - floordiv > 0.5) floordiv += 1.0; }
    else floordiv = std::copysign(0.0, tp / c.interval);
    const double d = mod / c.interval;
    a.Ns = c.Ns; a.Nd = c.Nd; a.K = c.K; a.k = (int32_t)floordiv;
    a.t = (float)timestamp; a.duration = (float)(c.duration > 1.0 ? c.duration : 1.0);
    a.delta = (float)d;
    a.h00 = (float)(2 * std::pow(d, 3) - 3 * std::pow(d, 2) + 1);
    a.h10 = (float)(std::pow(d, 3) - 2 * std::pow(d, 2) + d);
    a.h01 = (float)(-2 * std::pow(d, 3) + 3 * std::pow(d, 2));
    a.h11 = (float)(std::pow(d, 3) - std::pow(d, 2));
    a.tau = (float)(tp / c.interval); a.var_min = (float)(c.var_pad / c.interval);
}

int ex4d_trainer_step(Ex4dTrainer *t, double timestamp, const float *viewmatrix, const float *projmatrix, const float *campos,
                      const float *background, const float *gt_image, void *stream, int32_t *num_rendered)
{
    t_err[0] = 0;
    if (!t || !viewmatrix || !projmatrix || !campos || !background || !gt_image) return tfail(EX4D_ERR_ARG, "null argument");
    const Ex4dTrainerConfig &c = t->cfg;
    Ex4dAttrParams a;
    ex4d_trainer_time_scalars(&c, timestamp, &a);
    float *const *p = t->param;
    if (ex4d_attributes_forward(&a, p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7], p[8], p[9], p[10], p[11], p[12], p[13], p[14],
                                t->means3D, t->rotations, t->opacities, t->scales, nullptr, stream))
        return tfail(EX4D_ERR_HIP, "attributes forward: %s", ex4d_attributes_last_error());

    Ex4dParams prm;
    prm.P = t->P; prm.D = c.sh_degree; prm.M = 16; prm.W = c.W; prm.H = c.H; prm.tanfovx = c.tanfovx; prm.tanfovy = c.tanfovy;
    prm.kernel_size = c.kernel_size; prm.scale_modifier = 1.0f; prm.min_depth = c.min_depth; prm.max_depth = c.max_depth;
    prm.prefiltered = 0; prm.debug = 0; prm.prepare_backward = 1; prm.instance_capacity = 0; prm.assume_no_flow = 0; prm.reserved = 0;
    Ex4dSplitSH sh;
    sh.dc[0] = p[5]; sh.rest[0] = p[6]; sh.dc[1] = p[13]; sh.rest[1] = p[14]; sh.n_static = c.Ns;
    Ex4dSplitSHGrad gsh;
    gsh.dc[0] = t->grad[5]; gsh.rest[0] = t->grad[6]; gsh.dc[1] = t->grad[13]; gsh.rest[1] = t->grad[14]; gsh.n_static = c.Ns;
    float *const *g = t->grad;
    // Asynchronous mode: the forward does not wait for the instance count (the reference's one host synchronisation per frame,
    // rasterizer_impl.cu:298-299); the frame's status is looked at once, right before the optimizer step -- by then every kernel of the
    // frame is enqueued, the GPU has work for the rest of the iteration, and the status copy (it sits behind the tile scan, early in the
    // frame) has usually landed.  A frame whose instance count exceeded the capacity is simply run again with a larger one before anything
    // is applied: the parameters see exactly the gradients of the synchronous path.
    for (int attempt = 0; ; attempt++) {
        const bool async_frame = t->async && t->capacity > 0;
        prm.instance_capacity = async_frame ? (int32_t)t->capacity : 0;
        prm.assume_no_flow = async_frame ? 1 : 0;            // (dir3D is NULL: no Gaussian carries a flow vector)
        int32_t R = 0;
        int rc = ex4d_forward_split_sh(&prm, background, t->means3D, nullptr, &sh, t->opacities, t->scales, t->rotations, nullptr,
                                       viewmatrix, projmatrix, campos, nullptr, arena_alloc, &t->geom, arena_alloc, &t->binning, arena_alloc, &t->img,
                                       t->color, t->radii, t->depth, t->acc, t->flow, t->idx, stream,
                                       async_frame ? reinterpret_cast<int32_t *>(t->status) : &R);
        if (rc) return tfail(rc, "rasterizer forward: %s", ex4d_last_error());
        if (async_frame) {
            if (hipEventRecord(t->status_ev, (hipStream_t)stream) != hipSuccess) return tfail(EX4D_ERR_HIP, "event record failed");
            R = (int32_t)t->capacity;                        // the backward lays the buffers out for the capacity
        }

        if (ex4d_l1_ssim_forward(3, c.H, c.W, t->color, gt_image, c.lambda_dssim, c.window, t->loss, nullptr, nullptr, t->dmaps, t->loss_scratch, stream))
            return tfail(EX4D_ERR_HIP, "loss forward: %s", ex4d_loss_last_error());
        if (ex4d_l1_ssim_backward(3, c.H, c.W, t->color, gt_image, c.lambda_dssim, c.window, t->dmaps, t->grad_loss, t->grad_img, stream))
            return tfail(EX4D_ERR_HIP, "loss backward: %s", ex4d_loss_last_error());

        rc = ex4d_backward_split_sh(&prm, R, background, t->means3D, t->radii, &sh, t->scales, t->rotations, nullptr, viewmatrix, projmatrix, campos,
                                    nullptr, t->depth, t->acc, t->geom.ptr, t->binning.ptr, t->img.ptr, t->grad_img, nullptr, nullptr, nullptr,
                                    t->g_means2D, nullptr, t->g_opacity, t->g_means3D, nullptr, &gsh, t->g_scales, t->g_rotations, t->g_dir,
                                    t->bwd_scratch, stream);
        if (rc) return tfail(rc, "rasterizer backward: %s", ex4d_last_error());

        if (ex4d_attributes_backward_sliced(&a, p[3], p[4], p[8], p[9], p[10], p[11], p[12],
                                            t->g_means3D, t->g_rotations, t->g_opacity, t->g_scales, nullptr,
                                            g[0], g[1], g[2], g[3], g[4], nullptr, nullptr, g[7], g[8], g[9], g[10], g[11], g[12], nullptr, nullptr,
                                            t->slices, stream))
            return tfail(EX4D_ERR_HIP, "attributes backward: %s", ex4d_attributes_last_error());

        uint32_t count = (uint32_t)R;
        if (async_frame) {
            if (hipEventSynchronize(t->status_ev) != hipSuccess) return tfail(EX4D_ERR_HIP, "event wait failed");
            count = t->status->num_rendered;
        }
        if (num_rendered) *num_rendered = (int32_t)count;
        const uint32_t want = count + count / 4 + 4096;      // 25 % headroom over the largest frame seen
        const bool overflow = async_frame && count > t->capacity;
        if (t->async && want > t->capacity) t->capacity = want < 0x7FFFFFFFu ? want : 0x7FFFFFFFu;
        if (!overflow) break;
        if (attempt >= 2) return tfail(EX4D_ERR_HIP, "internal: the frame overflowed its instance capacity three times");
        t->replays++;
    }

    if (c.optimizer) {
        t->step += 1;
        Ex4dRadamTensor dense[EX4D_TRAINER_PARAMS];
        Ex4dRadamSlicedTensor sl[2];
        int nd = 0, ns = 0;
        for (int i = 0; i < EX4D_TRAINER_PARAMS; i++) {
            if (t->numel[i] == 0) continue;
            if (i == 7 || i == 8) {
                Ex4dRadamSlicedTensor &s = sl[ns++];
                memset(&s, 0, sizeof(s));
                s.param = p[i]; s.exp_avg = t->m[i]; s.exp_avg_sq = t->v[i]; s.rows = c.Nd; s.K = c.K; s.C = i == 7 ? 3 : 4;
                s.lr = c.lr[i]; s.step = t->step; s.n_windows = 1;
                s.first[0] = t->slices[i == 7 ? 0 : 2]; s.count[0] = t->slices[i == 7 ? 1 : 3]; s.grad[0] = g[i];
            } else {
                Ex4dRadamTensor &q = dense[nd++];
                memset(&q, 0, sizeof(q));
                q.nan_to_num = i == 11;            // train.py:244-247: _opacity_duration_var.grad.nan_to_num() before optimizer.step()
                q.param = p[i]; q.grad = g[i]; q.exp_avg = t->m[i]; q.exp_avg_sq = t->v[i]; q.numel = t->numel[i]; q.lr = c.lr[i]; q.step = t->step;
            }
        }
        if (nd && ex4d_radam_step(dense, nd, c.beta1, c.beta2, c.eps, stream)) return tfail(EX4D_ERR_HIP, "RAdam: %s", ex4d_optim_last_error());
        if (ns && ex4d_radam_step_sliced(sl, ns, c.beta1, c.beta2, c.eps, stream)) return tfail(EX4D_ERR_HIP, "RAdam (sliced): %s", ex4d_optim_last_error());
    }
    return EX4D_OK;
}

int ex4d_trainer_set_async(Ex4dTrainer *t, int32_t on)
{
    t_err[0] = 0;
    if (!t) return tfail(EX4D_ERR_ARG, "null argument");
    if (on && !t->status) {
        if (hipHostMalloc((void **)&t->status, sizeof(Ex4dFrameStatus), hipHostMallocDefault) != hipSuccess) { t->status = nullptr; return tfail(EX4D_ERR_HIP, "pinned status allocation failed"); }
        if (hipEventCreateWithFlags(&t->status_ev, hipEventDisableTiming) != hipSuccess) { t->status_ev = nullptr; return tfail(EX4D_ERR_HIP, "event creation failed"); }
    }
    t->async = on != 0;
    return EX4D_OK;
}

int64_t ex4d_trainer_replays(const Ex4dTrainer *t) { return t ? t->replays : 0; }

int ex4d_trainer_set_lr(Ex4dTrainer *t, const double *lr15)
{
    t_err[0] = 0;
    if (!t || !lr15) return tfail(EX4D_ERR_ARG, "null argument");
    for (int i = 0; i < EX4D_TRAINER_PARAMS; i++) {
        if (!(lr15[i] >= 0.0)) return tfail(EX4D_ERR_ARG, "learning rate %d is negative or NaN", i);
        t->cfg.lr[i] = lr15[i];
    }
    return EX4D_OK;
}

int ex4d_trainer_set_sh_degree(Ex4dTrainer *t, int32_t degree)
{
    t_err[0] = 0;
    if (!t || degree < 0 || degree > 3) return tfail(EX4D_ERR_ARG, "SH degree outside [0, 3]");
    t->cfg.sh_degree = degree;
    return EX4D_OK;
}

const void *ex4d_trainer_output(const Ex4dTrainer *t, int32_t what)
{
    if (!t) return nullptr;
    switch (what) {
    case 0: return t->loss;
    case 1: return t->color;
    case 2: return t->radii;
    case 3: return t->g_means2D;
    case 4: return t->depth;
    case 5: return t->acc;
    default: return nullptr;
    }
}

const float *ex4d_trainer_grad(const Ex4dTrainer *t, int32_t i, int32_t *slices4)
{
    if (!t || i < 0 || i >= EX4D_TRAINER_PARAMS) return nullptr;
    if (slices4) memcpy(slices4, t->slices, sizeof(t->slices));
    return t->grad[i];
}

int ex4d_trainer_read(const Ex4dTrainer *t, int32_t what, void *dst, size_t bytes, void *stream)
{
    t_err[0] = 0;
    if (!t || !dst) return tfail(EX4D_ERR_ARG, "null argument");
    const void *src = nullptr;
    size_t have = 0;
    const size_t P = (size_t)t->P, HW = t->HW;
    if (what >= 100 && what < 100 + EX4D_TRAINER_PARAMS) { src = t->grad[what - 100]; have = (size_t)t->grad_numel[what - 100] * sizeof(float); }
    else {
        src = ex4d_trainer_output(t, what);
        const size_t sizes[6] = { sizeof(float), 3 * HW * sizeof(float), P * sizeof(int32_t), 3 * P * sizeof(float), HW * sizeof(float), HW * sizeof(float) };
        if (what >= 0 && what < 6) have = sizes[what];
    }
    if (bytes > have || (bytes > 0 && !src)) return tfail(EX4D_ERR_ARG, "buffer %d holds %zu bytes, %zu requested", what, have, bytes);
    if (bytes && hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess)
        return tfail(EX4D_ERR_HIP, "device copy failed: %s", hipGetErrorString(hipGetLastError()));
    return EX4D_OK;
}

size_t ex4d_trainer_bytes(const Ex4dTrainer *t) { return t ? t->bytes : 0; }

}  // extern "C"
