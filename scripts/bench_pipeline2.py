"""Experiment: N clips in flight.  The headline step replays one clip's hipGraph after the other on ONE stream; its encoder / flow
section (~110 latency-bound launches on 862 columns) leaves most CUs idle for ~1.9 ms.  Here N model instances (same weights, own
graphs and static buffers) replay on N streams, clip i + 1's encoder under clip i's decoder: throughput of independent B = 1 clips
(what Svc.slice_inference's chunk loop / a serving queue sees), not the latency of one.  usage: bench_pipeline2.py [steps] [N...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "so-vits-svc_amd"))
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    depths = [int(a) for a in sys.argv[2:]] or [1, 2, 3]
    dev = torch.device("cuda:0")
    nets = []
    for i in range(max(depths)):
        net, cfg, W = bench.build_model(dev)
        net.enable_graph(True)
        nets.append(net)
    c, f0, uv, sid = [t.to(dev) for t in W.make_inputs(cfg, 1, bench.T_FRAMES, seed=1234)]
    streams = [torch.cuda.Stream() for _ in nets]
    outs = [None] * len(nets)
    for i, net in enumerate(nets):                      # capture
        with torch.cuda.stream(streams[i]):
            outs[i] = net.infer(c, f0, uv, g=sid, noice_scale=0.4)[0]
    torch.cuda.synchronize()
    assert all(torch.equal(outs[0], o) for o in outs[1:]), "instances disagree"
    for n in depths:
        for _ in range(3):
            for i in range(n):
                with torch.cuda.stream(streams[i]):
                    nets[i].infer(c, f0, uv, g=sid, noice_scale=0.4)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(steps):
            i = k % n
            with torch.cuda.stream(streams[i]):
                nets[i].infer(c, f0, uv, g=sid, noice_scale=0.4)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        print(f"clips in flight {n}: {dt * 1e3:.3f} ms per clip, {bench.T_FRAMES * bench.HOP / dt / 1e6:.2f} M samples/s")


if __name__ == "__main__":
    main()
