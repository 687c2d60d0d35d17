"""The two-plane fp16 split GEMM (default) beside the three-plane bf16 split (DR_GEMM_F16X2=0): error against float64 and time
per launch at the loop's shapes (stand-alone op: the fp16 kernel sweeps its rows for their maxima itself here)."""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "diff-reg_amd"))
import torch
from diffreg_hip import lib
lib.ensure_init()
res = {}
def timed(fn, reps=20):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
for rows, ncols, K in ((32768, 432, 432), (32768, 864, 864), (32768, 432, 864)):
    g = torch.Generator(device="cpu").manual_seed(rows + K)
    x = torch.randn(rows, K, generator=g).cuda()
    W = ((torch.rand(ncols, K, generator=g) * 2 - 1) / K ** 0.5).cuda()
    ref = x[:2048].double() @ W.double().T
    den = x[:2048].abs().double() @ W.abs().double().T
    row = {}
    for mode in (0, 1):
        lib.raw().dr_debug_gemm_f16x2(mode)
        Wp = lib.pack_weight(W)
        y = lib.linear_packed(x, W, Wp)
        e = ((y[:2048].double() - ref).abs() / den)
        row["f16x2" if mode else "bf16x3"] = {"mean_err": float(e.mean()), "max_err": float(e.max()),
                                               "us": timed(lambda: lib.linear_packed(x, W, Wp)),
                                               "tflops": 2.0 * rows * ncols * K / timed(lambda: lib.linear_packed(x, W, Wp)) / 1e6}
    e = ((x[:2048] @ W.T).double() - ref).abs() / den
    row["torch_fp32_matmul"] = {"mean_err": float(e.mean()), "max_err": float(e.max())}
    res["%dx%dx%d" % (rows, ncols, K)] = row
# operands spanning 12 orders of magnitude by row / column, tiny and huge activations: error relative to each output's own scale
g = torch.Generator(device="cpu").manual_seed(5)
rows, ncols, K = 2048, 432, 432
for name, sx, sw in (("rows and columns x 10^[-3,3]", 10.0 ** (torch.rand(rows, 1, generator=g) * 6 - 3), 10.0 ** (torch.rand(ncols, 1, generator=g) * 6 - 3)),
                     ("activations x 1e-6", torch.full((rows, 1), 1e-6), torch.ones(ncols, 1)),
                     ("activations x 1e+8, weights x 1e-5", torch.full((rows, 1), 1e8), torch.full((ncols, 1), 1e-5))):
    x = (torch.randn(rows, K, generator=g) * sx).cuda()
    W = (torch.randn(ncols, K, generator=g) / K ** 0.5 * sw).cuda()
    ref = x.double() @ W.double().T
    scale = (sx.double() * sw.double().T).cuda()
    row = {}
    for mode in (0, 1):
        lib.raw().dr_debug_gemm_f16x2(mode)
        Wp = lib.pack_weight(W)
        lib.raw().dr_debug_gemm_config(70 if mode else 50)     # 2048 rows are below the automatic threshold of the packed kernels
        y = lib.linear_packed(x, W, Wp)
        lib.raw().dr_debug_gemm_config(-1)
        row["f16x2" if mode else "bf16x3"] = float(((y.double() - ref).abs() / scale).max())
    row["torch_fp32_matmul"] = float((((x @ W.T).double() - ref).abs() / scale).max())
    res[name] = row
lib.raw().dr_debug_gemm_f16x2(-1)
print(json.dumps(res, indent=1))
