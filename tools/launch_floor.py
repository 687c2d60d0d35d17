"""What a DEPENDENT launch costs on this GPU whatever it does: chains of empty kernels on one stream, eager and replayed from a HIP
graph, timed with events.  This is the floor under the single-pair (B = 1) latency: ~1 430 dependent launches per pair."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "diff-reg_amd"))
import torch
from diffreg_hip import lib

lib.ensure_init()
out = {"what": "dependent launches of an empty kernel on one stream (us per launch, best of 5 chains of 2000)"}
st = torch.cuda.current_stream()
for wg, thr in ((1, 64), (64, 256), (64, 512), (384, 512)):
    def chain(n=2000):
        lib.check(lib._lib.dr_debug_launch_chain(n, wg, thr, st.cuda_stream))
    chain(200); torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); chain(); b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) * 1e3 / 2000)
    g = torch.cuda.CUDAGraph()
    s2 = torch.cuda.Stream()
    with torch.cuda.stream(s2):
        with torch.cuda.graph(g, stream=s2):
            lib.check(lib._lib.dr_debug_launch_chain(2000, wg, thr, s2.cuda_stream))
    g.replay(); torch.cuda.synchronize()
    bg = 1e9
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); g.replay(); b.record(); torch.cuda.synchronize()
        bg = min(bg, a.elapsed_time(b) * 1e3 / 2000)
    out["%d workgroups x %d threads" % (wg, thr)] = {"eager_us": round(best, 3), "graph_us": round(bg, 3)}
print(json.dumps(out, indent=1))
