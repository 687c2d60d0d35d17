"""host-side profile (cProfile) of bench.py's whole-model training step: where the ~12 ms that are not kernel time go"""
import cProfile, io, os, pstats, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, os.path.join(ROOT, "diff-reg_amd")); sys.path.insert(0, ROOT)
import torch
import bench
orig = bench._time_calls
seen = []
def spy(fn, warm=2, reps=5):
    if not seen:
        seen.append(fn)
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        pr = cProfile.Profile()
        t0 = time.perf_counter()
        pr.enable()
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        pr.disable()
        print("profiled: %.2f ms per step" % ((time.perf_counter() - t0) / 5 * 1e3))
        for key in ("tottime", "cumulative"):
            s = io.StringIO()
            pstats.Stats(pr, stream=s).sort_stats(key).print_stats(45)
            print("\n".join(l[:170] for l in s.getvalue().splitlines()[:75]))
    return orig(fn, warm=warm, reps=reps)
bench._time_calls = spy
r = bench.bench_train_step(torch.device("cuda:0"))
print(r["ms_per_step"], r["backbone_forward_only_ms"])
