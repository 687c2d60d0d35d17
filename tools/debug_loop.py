import sys, os, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "diff-reg_amd")); sys.path.insert(0, ROOT)
from diffreg_hip import synth
from oracle import diffreg_oracle as orc
from tests.helpers import T, weights, pair, masks
from tests.test_loop_gpu import engine, LOOPS
DEV = "cuda:0"
for (variant, N, M, nv, mv, steps, mc, seed, tag) in LOOPS:
    g = np.load(os.path.join(ROOT, "tests/golden/%s_loop_%s.npz" % (variant, tag)))
    eng = engine(variant, steps, mc)
    _, p = pair(variant, N, M, seed)
    ms, mt = masks(N, M, nv, mv)
    noise = T(synth.step_noise(N, M, seed, steps))[:, None]
    masked = variant == "4dmatch"
    out = eng.run(p["f_s"].to(DEV), p["f_t"].to(DEV), p["p_s"].to(DEV), p["p_t"].to(DEV), p["x_T"].to(DEV),
                  ms.to(DEV) if masked else None, mt.to(DEV) if masked else None, noise=noise.to(DEV) if masked else None, trace=True)
    x0 = out["x0"][:, 0].cpu()
    # fp64 evaluation of the same maths
    W64 = {k: v.double() for k, v in weights(variant).items()}
    v = synth.VARIANTS[variant]
    tr64 = []
    o64 = orc.denoise_loop(W64, v, p["f_s"].double(), p["f_t"].double(), p["p_s"], p["p_t"], ms, mt,
                           p["x_T"].double(), steps, mc, variant=variant, noise=noise.double(), trace=tr64)
    tr32 = []
    o32 = orc.denoise_loop(weights(variant), v, p["f_s"], p["f_t"], p["p_s"], p["p_t"], ms, mt, p["x_T"], steps, mc, variant=variant, noise=noise, trace=tr32)
    print("==", tag)
    for k in range(steps):
        t64 = tr64[k]["x0"][0].double(); t32 = tr32[k]["x0"][0].double(); h = x0[k].double()
        e_h = (h - t64).abs(); e_r = (t32 - t64).abs(); d = (h - t32).abs()
        if k in (0, 1, steps // 2, steps - 1):
            print(" step %2d  |hip-ref32| max %.2e n>1e-4 %4d | |hip-f64| max %.2e | |ref32-f64| max %.2e n>1e-4 %4d | R: hip-ref %.1e ref-f64 %.1e" % (
                k, d.max(), int((d > 1e-4).sum()), e_h.max(), e_r.max(), int((e_r > 1e-4).sum()),
                (out["R_forwd"][k, 0].cpu() - tr32[k]["R_forwd"][0]).abs().max(), (tr32[k]["R_forwd"][0].double() - tr64[k]["R_forwd"][0].double()).abs().max()))
    c = out["conf_matrix_pred"][0].cpu(); c32 = o32["conf_matrix_pred"][0]; c64 = o64["conf_matrix_pred"][0]
    print(" conf: |hip-ref32| %.2e  |hip-f64| %.2e  |ref32-f64| %.2e ; rel hip-ref32 %.2e ref32-f64 %.2e" % (
        (c - c32).abs().max(), (c - c64).abs().max(), (c32 - c64).abs().max(),
        ((c - c32).abs() / c32.abs().clamp_min(1e-9)).max(), ((c32 - c64).abs() / c64.abs().clamp_min(1e-9)).max()))
