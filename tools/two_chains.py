#!/usr/bin/env python
"""Do two independent latency-bound launch chains overlap on this runtime?  (round 4)

The C3 step's forward + criterion + backward is a chain of ~140 dependent launches, most of them filling a third to seven eighths of the chip.  Two half-batch
chains (32 dialogues each) on two streams could fill each other's gaps - IF the runtime runs kernels of two streams side by side.  Measured here with two models
(separate buffers) whose B = 32 graphs are replayed
  seq   both on one stream, one after the other
  par   on two streams at the same time (each stream replays its own graph; the host enqueues both, then waits for both)
against the B = 64 graph of one model.  ms per pair of half-batch steps (= per 64 dialogues), forward + criterion + backward only."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from mer_amd import runtime  # noqa: E402
from mer_amd.model import M2FNet  # noqa: E402


def main():
    runtime.require_gpu()
    dev = torch.device("cuda:0")
    wl = bench.WORKLOADS["c3"]
    cfg, B, L = wl["cfg"], wl["B"], wl["L"]
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    torch.manual_seed(0)
    models = [M2FNet(cfg, precision="bf16", shape_buckets=False).to(dev).train() for _ in range(3)]
    s_main, s_a, s_b = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    plans = []
    for m, b, st in ((models[0], B, s_main), (models[1], B // 2, s_a), (models[2], B // 2, s_b)):
        text, audio, mask, emotion = bench.synthetic_batch(cfg, b, L, 0, dev)
        with torch.cuda.stream(st):
            eng = m.engine()
            pl = eng.plan(b, L, True, True, None)
            pl.set_inputs(text, audio, mask, emotion)
            eng.publish_grads()
            for _ in range(4):
                pl.step(0.1, False, False, True)
        plans.append(pl)
    torch.cuda.synchronize()
    full, ha, hb = plans

    def timed(fn):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        res = []
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(steps):
                fn()
            torch.cuda.synchronize()
            res.append((time.perf_counter() - t0) / steps * 1e3)
        return sorted(res)[1]

    def one_full():
        with torch.cuda.stream(s_main):
            full.step(0.1, False, False, True)

    def seq():
        with torch.cuda.stream(s_a):
            ha.step(0.1, False, False, True)
            hb.step(0.1, False, False, True)

    def par():
        with torch.cuda.stream(s_a):
            ha.step(0.1, False, False, True)
        with torch.cuda.stream(s_b):
            hb.step(0.1, False, False, True)

    def half():
        with torch.cuda.stream(s_a):
            ha.step(0.1, False, False, True)

    out = {"B64_one_graph_ms": timed(one_full), "B32_one_graph_ms": timed(half), "two_B32_graphs_one_stream_ms": timed(seq),
           "two_B32_graphs_two_streams_ms": timed(par), "B64_again_ms": timed(one_full)}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
