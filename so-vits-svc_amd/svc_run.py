#!/usr/bin/env python
"""Launcher: run an UNCHANGED reference entry point on the MI355X engine.

    cd /path/to/so-vits-svc                      # the reference checkout (configs/, logs/, raw/, pretrain/ live here)
    python /path/to/so-vits-svc_amd/svc_run.py inference_main.py -m logs/44k/G_0.pth -c configs/config.json -n x.wav -s spk
    python /path/to/so-vits-svc_amd/svc_run.py train.py -c configs/config.json -m 44k

Why a launcher: `python inference_main.py` puts the SCRIPT's directory at sys.path[0], ahead of PYTHONPATH, so `import models`
would find the reference's models.py no matter what PYTHONPATH says.  Here sys.path becomes
[this package, the script's directory, ...]: `models`, `utils`, `modules.*`, `vdecoder.*`, `vencoder.*`, `inference.*`,
`diffusion.*`, `data_utils` resolve to the engine, everything the engine does not mirror (`cluster`, `spkmix`,
`modules.F0Predictor.*`, ...) to the checkout (svc_overlay).  For `train.py` the engine's own train.py (same CLI, same
checkpoints / logs layout) is run instead of the reference's loop — see INTEGRATION.md.
"""
import os
import runpy
import sys

PKG = os.path.dirname(os.path.abspath(__file__))


def main():
    if len(sys.argv) < 2:
        print(__doc__)
        sys.exit(2)
    script = os.path.abspath(sys.argv[1])
    script_dir = os.path.dirname(script)
    sys.path[:] = [PKG, script_dir] + [p for p in sys.path if os.path.abspath(p or ".") not in (PKG, script_dir)]
    import svc_overlay
    svc_overlay.install()
    own = os.path.join(PKG, os.path.basename(script))
    if os.path.basename(script) in ("train.py", "train_diff.py") and os.path.exists(own) and os.environ.get("SVC_RUN_REFERENCE_LOOP") != "1":
        script = own      # the engine's training loop (fused optimizer, hipGraph step, RCCL reducer) behind the same CLI
    sys.argv = [script] + sys.argv[2:]
    runpy.run_path(script, run_name="__main__")


if __name__ == "__main__":
    main()
