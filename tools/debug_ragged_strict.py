import os, sys
sys.path.insert(0, "/root/repo/diff-reg_amd"); sys.path.insert(0, "/root/repo")
import torch
from diffreg_hip import synth
from diffreg_hip.engine import DenoiseEngine
from tests.helpers import weights, pair
DEV="cuda:0"; variant="3dmatch"; steps, mc = 4, 200
v = synth.VARIANTS[variant]
for strict in (False, True):
    eng = DenoiseEngine(weights(variant), variant=variant, C=v["C"], H=v["H"], voxel=v["voxel"], origin=v["origin"], steps=steps, sk_iters=v["skh_iters"],
                        sample_rate=v["sample_rate"], max_condition_num=mc, n_layers=v["n_layers"], device=DEV, strict_f64=strict)
    sizes = [(96, 80), (128, 128), (57, 121), (128, 40)]
    ps = [pair(variant, n, m, 61 + i)[1] for i, (n, m) in enumerate(sizes)]
    items = [dict(src_feats=q["f_s"][0].to(DEV), tgt_feats=q["f_t"][0].to(DEV), s_pcd=q["p_s"][0].to(DEV), t_pcd=q["p_t"][0].to(DEV), x_T=q["x_T"][0].to(DEV)) for q in ps]
    got = eng.run_ragged(items)
    for i, q in enumerate(ps):
        one = eng.run(q["f_s"].to(DEV), q["f_t"].to(DEV), q["p_s"].to(DEV), q["p_t"].to(DEV), q["x_T"].to(DEV))
        dd = (got[i]["conf_matrix_pred"] - one["conf_matrix_pred"][0]).abs()
        print("strict", strict, i, sizes[i], "conf max diff %.3e" % dd.max().item(), "R diff %.2e" % (got[i]["R_final"] - one["R_final"][0]).abs().max().item(), "nan", int(torch.isnan(got[i]["conf_matrix_pred"]).sum()))
