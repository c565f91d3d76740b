"""CPU checks of the two algebraic identities the interior-point form's fused passes rest on (csrc/dsp_ipm.hip, round 6):

  1. k_ipm_dir (mode 0) + k_ipm_steps: mu after the affine step is a bilinear form in the two step lengths,
         sum_l (z + ad dz)(w + ap dv) + sum_u (f + ad df)(t - ap dv) = mu nb + ap q1 + ad q2 + ap ad q3,
     so its three sums can be taken before the step lengths exist (the pass that evaluated the left-hand side is gone);
  2. k_ipm_dir (mode 0 writes, mode 1 / k_ipm_rhs read): the corrector's rt = rd - cz + cf with the targets
         cz = (sigma mu - corl) / w - z,  cf = (sigma mu - coru) / t - f
     is linear in sigma mu: rt = ta + sigma mu tb, ta = (predictor's rt) + corl / w - coru / t, tb = -1 / w + 1 / t,
     and the corrector's dv = th (A' dy - rt) = th A' dy - (th ta + sigma mu th tb).

Formulas are written as the kernels write them; bounds that do not exist drop their terms (hl / hu)."""
import numpy as np


def _state(seed, n=4000):
    g = np.random.default_rng(seed)
    hl = g.random(n) < 0.8
    hu = g.random(n) < 0.5
    w = g.uniform(1e-3, 50.0, n)          # v - l
    t = g.uniform(1e-3, 50.0, n)          # u - v
    z = np.where(hl, g.uniform(1e-6, 5.0, n), 0.0)
    f = np.where(hu, g.uniform(1e-6, 5.0, n), 0.0)
    both = hl | hu
    th = np.where(both, 1.0 / np.maximum(np.where(hl, z / w, 0.0) + np.where(hu, f / t, 0.0), 1e-300), 1e20)
    rd = g.normal(size=n)
    aty = g.normal(size=n)                # A' dy of the predictor
    return g, hl, hu, w, t, z, f, th, rd, aty


def test_mu_after_the_affine_step_is_bilinear_in_the_step_lengths():
    for seed in range(5):
        g, hl, hu, w, t, z, f, th, rd, aty = _state(seed)
        # predictor: targets cz = -z, cf = -f; rt = rd - cz + cf; dv = th (A' dy - rt); dz = cz - z / w dv; df = cf + f / t dv
        rt = rd + np.where(hl, z, 0.0) - np.where(hu, f, 0.0)
        dv = th * (aty - rt)
        dz = np.where(hl, -z - z / w * dv, 0.0)
        df = np.where(hu, -f + f / t * dv, 0.0)
        nb = hl.sum() + hu.sum()
        mu = (np.sum(z * w * hl) + np.sum(f * t * hu)) / nb
        q1 = np.sum(np.where(hl, z * dv, 0.0)) - np.sum(np.where(hu, f * dv, 0.0))
        q2 = np.sum(np.where(hl, w * dz, 0.0)) + np.sum(np.where(hu, t * df, 0.0))
        q3 = np.sum(np.where(hl, dz * dv, 0.0)) - np.sum(np.where(hu, df * dv, 0.0))
        for ap, ad in ((1.0, 1.0), (0.37, 0.91), (0.999, 0.02), (0.0, 0.5)):
            direct = np.sum(np.where(hl, (z + ad * dz) * (w + ap * dv), 0.0)) + np.sum(np.where(hu, (f + ad * df) * (t - ap * dv), 0.0))
            form = mu * nb + ap * q1 + ad * q2 + ap * ad * q3
            assert abs(direct - form) <= 1e-9 * max(1.0, abs(mu * nb)), (seed, ap, ad, direct, form)


def test_the_correctors_rt_is_linear_in_sigma_mu():
    for seed in range(5):
        g, hl, hu, w, t, z, f, th, rd, aty = _state(100 + seed)
        rt0 = rd + np.where(hl, z, 0.0) - np.where(hu, f, 0.0)                # the predictor's rt (k_ipm_resid)
        dv0 = th * (aty - rt0)
        dz0 = np.where(hl, -z - z / w * dv0, 0.0)
        df0 = np.where(hu, -f + f / t * dv0, 0.0)
        corl, coru = dv0 * dz0, -dv0 * df0
        ta = rt0 + np.where(hl, corl / w, 0.0) - np.where(hu, coru / t, 0.0)   # = rd + (corl / w + z) - (coru / t + f)
        tb = np.where(hl, -1.0 / w, 0.0) + np.where(hu, 1.0 / t, 0.0)
        aty1 = g.normal(size=len(w))                                          # A' dy of the corrector
        for sm in (0.0, 1e-8, 0.3, 17.0):
            cz = np.where(hl, (sm - corl) / w - z, 0.0)
            cf = np.where(hu, (sm - coru) / t - f, 0.0)
            rt1 = rd - cz + cf
            np.testing.assert_allclose(ta + sm * tb, rt1, rtol=1e-12, atol=1e-12 * np.abs(rt1).max())
            dv_direct = th * (aty1 - rt1)
            dv_parts = th * aty1 - (th * ta + sm * (th * tb))
            np.testing.assert_allclose(dv_parts, dv_direct, rtol=1e-9, atol=1e-12 * np.abs(dv_direct).max())
