"""CPU oracle of the RAdam step -- TEST INFRASTRUCTURE ONLY (tests/, smoke, bench cpu legs).

The reference optimises with `torch.optim.RAdam(l, lr=0.001)` (scene/c_gaussian_model.py:449; stepped at train.py:250);
torch is a third-party dependency (environment.yml:10 pins pytorch=2.1.2) whose source is not under /root/reference, so
this restates its documented algorithm (torch.optim.RAdam docs; _single_tensor_radam op order) in numpy float32 with the
scalar coefficients in Python doubles.  Pinned by tests/golden/radam.npz = parameter trajectories of torch.optim.RAdam
itself (the torch installed in the build container, CPU) over 12 steps that cross the rho_t > 5 switch at step 6.
"""
import numpy as np

f32 = np.float32


def radam_step(p, g, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-8):
    """One step, in place on float32 arrays p, m, v; `step` is the count after the increment."""
    m += f32(1 - beta1) * (g - m)
    v *= f32(beta2)
    v += (f32(1 - beta2) * g) * g
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    mhat = m / f32(bc1)
    rho_inf = 2 / (1 - beta2) - 1
    rho_t = rho_inf - 2 * step * (beta2 ** step) / bc2
    if rho_t > 5.0:
        rect = ((rho_t - 4) * (rho_t - 2) * rho_inf / ((rho_inf - 4) * (rho_inf - 2) * rho_t)) ** 0.5
        adaptive = f32(bc2 ** 0.5) / (np.sqrt(v) + f32(eps))
        p -= ((mhat * f32(lr)) * adaptive) * f32(rect)
    else:
        p -= mhat * f32(lr)
    return rho_t > 5.0
