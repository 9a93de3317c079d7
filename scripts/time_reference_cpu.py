"""Time the UNMODIFIED reference modules on CPU (build container only: /root/reference does not exist on the GPU box) on
bench.py's inference clip — BASELINE configs[1], full template, B = 1, T = 862 — next to the oracle ("port") on the same
cores, and write profiles/cpu_reference_build_container.json.  bench.py attaches that file to `cpu_baseline.reference`
(labelled with the machine it was measured on: it is NOT a measurement of the GPU box's host).
usage: python scripts/time_reference_cpu.py [threads]"""
import json
import os
import platform
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests", "golden")]
import torch  # noqa: E402

import make_golden as MG  # noqa: E402
import synthetic_data as W  # noqa: E402
from oracle import svc_oracle as O  # noqa: E402


def median_time(fn, runs):
    fn()                                                     # warm-up: oneDNN primitive cache
    ts = []
    for _ in range(runs):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2], ts


def main():
    threads = int(sys.argv[1]) if len(sys.argv) > 1 else os.cpu_count()
    torch.set_num_threads(threads)
    models, _ = MG.import_reference()
    cfg = W.full_config()
    sd = W.make_state_dict(cfg, 1234)
    net = MG.build_ref_model(models, cfg, sd).eval()
    T, HOP = 862, 512
    c, f0, uv, sid = W.make_inputs(cfg, 1, T, seed=1234)
    noise = W.make_noise(cfg, 1, T, seed=99)
    n = T * HOP

    def ref():
        with torch.no_grad():
            return net.infer(c, f0, uv, g=sid, noice_scale=0.4)[0]

    def port():
        with torch.no_grad():
            return O.synth_infer(sd, cfg, c, f0, uv, sid, noise, noice_scale=0.4)[0]

    assert tuple(ref().shape) == (1, 1, n)
    t_ref, all_ref = median_time(ref, 5)
    t_port, all_port = median_time(port, 5)
    cpu = ""
    try:
        cpu = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:      # noqa: BLE001
        pass
    out = dict(kind="reference", value=n / t_ref, unit="samples/s", cores=threads, seconds_per_clip=round(t_ref, 3),
               rtf=round(t_ref / (n / 44100.0), 4), runs=[round(t, 3) for t in all_ref],
               port_on_same_cores=dict(value=n / t_port, seconds_per_clip=round(t_port, 3), runs=[round(t, 3) for t in all_port]),
               sample=f"unmodified /root/reference models.SynthesizerTrn.infer (fp32, torch {torch.__version__} CPU) on the {n}-sample "
                      "bench clip, median of 5 after 1 warm-up",
               machine=f"build container: {cpu}, {threads} threads, {platform.platform()}",
               note="measured in the BUILD container (the GPU box has no /root/reference); compare with port_on_same_cores for the "
                    "reference / port ratio, not with the GPU box's cpu_baseline.value")
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    with open(os.path.join(ROOT, "profiles", "cpu_reference_build_container.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
