"""throughput of k concurrent batches (one graph each) on k HIP streams vs one big batch"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "diff-reg_amd")); sys.path.insert(0, ROOT)
import bench
dev = torch.device("cuda:0")
W, eng = bench.make_engine("3dmatch", 20, 200.0, dev)
for per, groups, nstreams in [(32, 1, 1), (16, 2, 2), (8, 4, 4), (16, 4, 2), (32, 2, 2), (16, 4, 4), (32, 4, 4), (4, 8, 8)]:
    gs = []
    for g in range(groups):
        _, inp = bench.make_inputs("3dmatch", per, 256, 256, seed0=100 * g, device=dev)
        gs.append(dict(src_feats=inp["f_s"], tgt_feats=inp["f_t"], s_pcd=inp["p_s"], t_pcd=inp["p_t"], x_T=inp["x_T"]))
    for _ in range(2): eng.run_streams(gs, nstreams)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps): eng.run_streams(gs, nstreams)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print("pairs/batch %3d  batches %d  streams %d : %7.1f ms  %6.1f pairs/s" % (per, groups, nstreams, dt * 1e3, per * groups / dt))
    eng._graphs.clear()
