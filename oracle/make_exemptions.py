"""Per-fixture exemption lists of the loop parity tests (test infrastructure; runs on CPU, no reference import needed).

The golden loop fixtures (tests/golden/*_loop_*.npz) are outputs of the reference's own float32 run.  On a few sharp,
ill-conditioned entries that run is itself 1e-4 .. 1e-2 away from a float64 evaluation of the same mathematics (the
oracle run in float64), so no float32 implementation can be held to 1e-4 against it there.  This script lists those entries
as DATA: for every fixture the flat indices where |reference - float64| exceeds TAU, with the reference's deviation and the
float64 value.  The GPU tests then hold every OTHER entry to a plain 1e-4 against the reference, and the listed ones to
|hip - f64| <= max(1e-4, 2 |ref - f64|).          python oracle/make_exemptions.py  ->  tests/golden/loop_exemptions.json
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "diff-reg_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)
from diffreg_hip import synth  # noqa: E402
from oracle import diffreg_oracle as orc  # noqa: E402
from tests.helpers import T, weights, pair, masks  # noqa: E402

TAU = 2e-5
LOOPS = [("3dmatch", 128, 128, 128, 128, 1, 200, 11, "n128_s1_mc200"),
         ("3dmatch", 128, 128, 128, 128, 20, 0, 11, "n128_s20_mc0"),
         ("3dmatch", 96, 80, 96, 80, 5, 200, 12, "n96x80_s5_mc200"),
         ("3dmatch", 256, 256, 256, 256, 20, 200, 13, "n256_s20_mc200"),
         ("4dmatch", 128, 128, 112, 100, 5, 40, 21, "n128_s5_mc40_masked"),
         ("4dmatch", 64, 96, 64, 96, 20, 40, 22, "n64x96_s20_mc40"),
         ("4dmatch", 512, 512, 470, 391, 20, 40, 62, "n512_s20_mc40_masked"),
         # the "soft" family (HEAD_GAIN_SOFT): expected -- and asserted below -- to need NO exemption
         ("3dmatch", 128, 128, 128, 128, 1, 200, 11, "soft_n128_s1_mc200"),
         ("3dmatch", 256, 256, 256, 256, 20, 200, 13, "soft_n256_s20_mc200"),
         ("4dmatch", 512, 512, 470, 391, 20, 40, 62, "soft_n512_s20_mc40_masked")]


def f64_eval(variant, N, M, nv, mv, steps, mc, seed, family="main"):
    v = synth.VARIANTS[variant]
    W64 = {k: t.double() for k, t in weights(variant, family).items()}
    _, p = pair(variant, N, M, seed)
    ms, mt = masks(N, M, nv, mv)
    noise = T(synth.step_noise(N, M, seed, steps))[:, None].double()
    tr = []
    o = orc.denoise_loop(W64, v, p["f_s"].double(), p["f_t"].double(), p["p_s"], p["p_t"], ms, mt, p["x_T"].double(),
                         steps, mc, variant=variant, noise=noise, trace=tr)
    return tr[-1]["x0"][0].double().numpy(), o["conf_matrix_pred"][0].double().numpy()


LOOPS_2D3D = [(96, 160, 90, 150, 141, 3, 200, 31, "n96x160_s3_masked"), (128, 192, 128, 192, 192, 10, 0, 32, "n128x192_s10_mc0")]


def f64_eval_2d3d(N, M, nv, mv, mv_da, steps, mc, seed):
    """2D-3D loop (oracle) with float64 weights, features and state; point / pixel coordinates stay float32 inputs"""
    Wn = synth.make_weights_2d3d(seed=9, head_gain=16.0)
    W64 = {k: T(a).double() for k, a in Wn.items()}
    pr = synth.make_pair_2d3d(N, M, seed, weights=Wn)
    q = lambda k: T(pr[k])[None]
    ms, mt = masks(N, M, nv, mv)
    mt_da = torch.arange(M)[None] < mv_da
    tr = []
    o = orc.denoise_loop_2d3d(W64, synth.VARIANTS["2d3d"], q("img_feats").double(), q("img_dino").double(), q("img_pixels"), q("pcd_feats").double(),
                              q("s_pcd"), q("t_pcd_da"), ms, mt, mt_da, q("x_T").double(), steps, mc, trace=tr)
    return tr[-1]["x0"][0].double().numpy(), o["conf_matrix_pred"][0].double().numpy()


def main():
    torch.set_num_threads(8)
    out = {"tau": TAU, "rule": "entries listed: |hip - f64| <= max(1e-4, 2 |ref - f64|); all others: |hip - ref| <= 1e-4",
           "fixtures": {}}
    for variant, N, M, nv, mv, steps, mc, seed, tag in LOOPS:
        name = "%s_loop_%s" % (variant, tag)
        g = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
        soft = tag.startswith("soft_")
        x0_f64, conf_f64 = f64_eval(variant, N, M, nv, mv, steps, mc, seed, "soft" if soft else "main")
        ent = {}
        for key, ref, f64 in (("x0_last", g["x0_last"], x0_f64), ("conf", g["conf"], conf_f64)):
            with np.errstate(invalid="ignore"):
                dev = np.abs(ref.astype(np.float64) - f64)
            dev = np.where(np.isfinite(dev), dev, 0.0)          # (masked entries: -inf / nan in both)
            idx = np.nonzero(dev.ravel() > TAU)[0]
            assert not (soft and idx.size), (name, key, "the soft family must not need exemptions", idx.size, dev.max())
            ent[key] = {"shape": list(ref.shape), "n_exempt": int(idx.size), "fraction": float(idx.size / ref.size),
                        "max_ref_minus_f64": float(dev.max()),
                        "index": idx.tolist(), "ref_minus_f64": [float(x) for x in dev.ravel()[idx]],
                        "f64": [float(x) for x in f64.ravel()[idx]]}
            print(name, key, "exempt", idx.size, "of", ref.size, "max |ref - f64|", dev.max())
        out["fixtures"][name] = ent
    for N, M, nv, mv, mv_da, steps, mc, seed, tag in LOOPS_2D3D:
        name = "2d3d_loop_" + tag
        g = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
        x0_f64, conf_f64 = f64_eval_2d3d(N, M, nv, mv, mv_da, steps, mc, seed)
        ent = {}
        for key, ref, f64 in (("x0_last", g["x0_last"], x0_f64), ("conf", g["conf"], conf_f64)):
            dev = np.abs(ref.astype(np.float64) - f64)
            idx = np.nonzero(dev.ravel() > TAU)[0]
            ent[key] = {"shape": list(ref.shape), "n_exempt": int(idx.size), "fraction": float(idx.size / ref.size),
                        "max_ref_minus_f64": float(dev.max()),
                        "index": idx.tolist(), "ref_minus_f64": [float(x) for x in dev.ravel()[idx]],
                        "f64": [float(x) for x in f64.ravel()[idx]]}
            print(name, key, "exempt", idx.size, "of", ref.size, "max |ref - f64|", dev.max())
        out["fixtures"][name] = ent
    json.dump(out, open(os.path.join(ROOT, "tests", "golden", "loop_exemptions.json"), "w"))


if __name__ == "__main__":
    main()
