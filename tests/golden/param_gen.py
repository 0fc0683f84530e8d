"""Seeded parameter sets shared by tests/golden/make_golden.py (which feeds them to the imported reference model) and the tests
(which feed them to this repository's model): large cases are regenerated from the seed instead of being stored."""
import torch


def seeded_params(Ns, Nd, K, seed):
    g = torch.Generator().manual_seed(seed)
    R = lambda *s: torch.randn(*s, generator=g)
    P = dict(
        _xyz=R(Ns, 3), _xyz_disp=0.1 * R(Ns, 3), _rotation=R(Ns, 4), _opacity=R(Ns, 1), _scaling=0.3 * R(Ns, 3) - 2,
        _features_dc=R(Ns, 1, 3), _features_rest=0.2 * R(Ns, 15, 3),
        _xyz_motion=torch.cumsum(0.2 * R(Nd, K, 3), 1), _rotation_motion=R(Nd, K, 4), _opacity_motion=R(Nd, 1),
        _opacity_duration_center=torch.sort(2 + torch.rand(Nd, 2, 1, generator=g) * (K - 5), dim=1)[0],
        _opacity_duration_var=R(Nd, 2, 1), _scaling_motion=0.3 * R(Nd, 3) - 2,
        _features_dc_motion=R(Nd, 1, 3), _features_rest_motion=0.2 * R(Nd, 15, 3))
    N = Ns + Nd
    W = dict(xyz=R(N, 3), rot=R(N, 4), opa=R(N, 1), scl=R(N, 3))
    return P, W


def checksum(P):
    """Order-independent fingerprint of a parameter set: detects a torch RNG that no longer reproduces the stored case."""
    return [float(sum(v.double().sum() for v in P.values())), float(sum((v.double() ** 2).sum() for v in P.values()))]
