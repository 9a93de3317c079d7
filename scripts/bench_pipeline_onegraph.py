"""Experiment: do two clips overlap when they are parallel BRANCHES of one hipGraph (fork / join on two streams inside one capture)?
Replays of different graphs on different streams do not (bench_pipeline2.py: 7.44 / 7.46 / 7.47 ms per clip for 1 / 2 / 3 in flight);
the three MRF chains of a decoder stage — branches of one graph — do.  usage: bench_pipeline_onegraph.py [steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "so-vits-svc_amd"))
import torch  # noqa: E402

import bench  # noqa: E402
import svc_hip as S  # noqa: E402


def main():
    import faulthandler
    faulthandler.enable()
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    dev = torch.device("cuda:0")
    nets = []
    for _ in range(2):
        net, cfg, W = bench.build_model(dev)
        nets.append(net)
    c, f0, uv, sid = [t.to(dev) for t in W.make_inputs(cfg, 1, bench.T_FRAMES, seed=1234)]
    T = bench.T_FRAMES
    noise = dict(enc_p=torch.randn(1, nets[0].inter_channels, T, device=dev), rand_ini=torch.rand(1, 9, device=dev),
                 sine=torch.randn(1, T * nets[0].dec.upp, 9, device=dev))
    g = [n._speaker(sid, c) for n in nets]

    def body(i):
        return nets[i]._infer_body(c, f0, uv, g[i], noise, 0.4, False, None)[0]
    with torch.no_grad():
        ref = body(0).clone()
        print('eager 0 ok', flush=True)
        body(1)
        print('eager 1 ok', flush=True)
        torch.cuda.synchronize()
        side = torch.cuda.Stream()
        for label, both in (("one clip per graph", False), ("two clips as parallel branches of one graph", True)):
            print('capturing', label, flush=True)
            gr = torch.cuda.CUDAGraph()
            with S.graph_capture(gr):
                main_s = torch.cuda.current_stream()
                o0 = None
                if both:
                    fork, done = torch.cuda.Event(), torch.cuda.Event()
                    fork.record(main_s)
                    with torch.cuda.stream(side):
                        side.wait_event(fork)
                        o1 = body(1)
                        done.record(side)
                o0 = body(0)
                if both:
                    main_s.wait_event(done)
            print('captured', flush=True)
            for _ in range(3):
                gr.replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                gr.replay()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / steps
            n = 2 if both else 1
            ok = torch.equal(o0, ref) and (not both or torch.equal(o1, ref))
            print(f"{label}: {dt * 1e3:.3f} ms per replay = {dt * 1e3 / n:.3f} ms per clip (outputs equal the eager clip: {ok})")


if __name__ == "__main__":
    main()
