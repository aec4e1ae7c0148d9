"""The whole bench loop on torch's default (null) stream against a stream of its own (torch.cuda.Stream: hipStreamNonBlocking)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench


def run(label):
    env, runner = bench.make_runner(4096, [512, 256, 128], "cuda:0", 0, 1)
    el, t_roll = bench.time_iterations(runner, 10, 3, 1)
    print(f"{label}: {el / 10 * 1e3:7.3f} ms per iteration (rollout {t_roll / 10 * 1e3:.3f})  stream handle {torch.cuda.current_stream().cuda_stream:#x}", flush=True)
    env.close(); runner.ppo.close()


for rep in range(2):
    run("default stream")
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        run("own stream    ")
    torch.cuda.synchronize()
