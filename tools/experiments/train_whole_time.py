"""the whole-model training step of bench.py (other_configs.train_step) alone, with the host time per phase"""
import os, sys, time, json
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, os.path.join(ROOT, "diff-reg_amd")); sys.path.insert(0, ROOT)
import torch
import bench
r = bench.bench_train_step(torch.device("cuda:0"))
print(json.dumps(r))
