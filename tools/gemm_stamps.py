"""Phase stamps (100 MHz wall clock) of workgroup 0 of the wide split GEMM: per k-chunk compute / stage / barrier."""
import os, sys, ctypes, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "diff-reg_amd"))
from diffreg_hip import lib
lib.ensure_init()
rows = int(os.environ.get("ROWS", "8192"))
x = torch.randn(rows, 432, device="cuda"); W = torch.randn(432, 432, device="cuda") / 20
Wp = lib.pack_weight(W)
lib.raw().dr_debug_gemm_config(59)
for rep in range(3):
    lib.linear_packed(x, W, Wp); torch.cuda.synchronize()
    st = (ctypes.c_longlong * 256)()
    lib.check(lib.raw().dr_debug_gemm_stamps(st))
    t0 = st[0]
    print("prologue %.2f us | loop end %.2f | epilogue %.2f" % ((st[8] - t0) / 100, (st[1] - t0) / 100, (st[2] - st[1]) / 100))
    if rep < 2: continue
    line = ""
    for ch in range(27):
        b = 8 + 8 * ch
        d = [(st[b + i + 1] - st[b + i]) / 100 for i in range(6)]
        line += "[%d: issue %.2f wait0 %.2f t0-3 %.2f cvt %.2f t4-6 %.2f barrier %.2f]\n" % (ch, d[0], d[1], d[2], d[3], d[4], d[5])
    print(line)
