"""Mint golden vectors for the patch-correspondence block behind the 2D-3D loop (row f4) with the REFERENCE's own ops
(build container only; needs /root/reference):    python oracle/make_golden_fine2d3d.py   ->  tests/golden/fine2d3d.npz

`vision3d.ops.index_select`, `pairwise_cosine_similarity` and `batch_mutual_topk_select` are imported from where they lie and run inside
the block of EXP/model.py:699-774 as restated by oracle/fine2d3d_oracle.py (that block is inline code of MATR2D3D.forward: not importable).
Shims: no-op Tensor.cuda (the ops hard-code .cuda()).  Inputs from tests/helpers.py::fine2d3d_case; only outputs are stored."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "diff-reg_amd"))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference/Diff-Reg-2d3d")


def main():
    torch.Tensor.cuda = lambda self, *a, **k: self
    import importlib.util

    def load(name, rel):
        spec = importlib.util.spec_from_file_location(name, os.path.join("/root/reference/Diff-Reg-2d3d/vision3d/ops", rel))
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
        return m
    ops = dict(index_select=load("ref_index_select", "index_select.py").index_select,
               pairwise_cosine_similarity=load("ref_cosine", "cosine_similarity.py").pairwise_cosine_similarity,
               batch_mutual_topk_select=load("ref_topk", "mutual_topk_select.py").batch_mutual_topk_select)
    from oracle import fine2d3d_oracle as fo
    from tests.helpers import fine2d3d_case
    torch.set_num_threads(8)
    c = fine2d3d_case()
    tr = []
    o = fo.extract_patch_correspondences(ops=ops, trace=tr, **c)
    out = {k: v.numpy() for k, v in o.items()}
    for t in tr:
        out["sim_level%d" % t["level"]] = t["similarity"].numpy()
        out["sel_level%d" % t["level"]] = torch.stack([t["batch"], t["row"], t["col"]], 1).numpy()
    p = os.path.join(ROOT, "tests", "golden", "fine2d3d.npz")
    np.savez_compressed(p, **out)
    print("wrote", p, os.path.getsize(p), "bytes;", len(out["img_corr_indices"]), "correspondences;", {k: v.shape for k, v in out.items() if k.startswith("sel")})


if __name__ == "__main__":
    main()
