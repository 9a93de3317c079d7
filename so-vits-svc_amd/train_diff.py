"""Entry point behind the reference's `train_diff.py` CLI (train_diff.py:14-76) on the MI355X engine:

    python so-vits-svc_amd/svc_run.py train_diff.py -c configs/diffusion.yaml
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 so-vits-svc_amd/svc_run.py train_diff.py -c ...

Same config file, same run directory (`env.expdir`: config.yaml, log_info.txt, model_<step>.pt with
`{'global_step', 'model'[, 'optimizer']}`), same resume rule (newest `model_<step>.pt`, lr decayed to the step it was saved at,
:55-60).  Below the CLI it is the engine's path: `Unit2Mel` on libsvc_hip.so, `optim.FusedAdamW` (one launch per step) where
the reference builds `torch.optim.AdamW`, `diffusion.solver.train` (hipGraph-replayed iteration; rank 0 logs / validates /
saves).  Under `torch.distributed.run` every rank takes its shard of the training list and gradients are averaged over RCCL
(BASELINE configs[4]; the reference's script is single-GPU).  `loguru` is not needed."""
import argparse
import os

import torch
from torch.optim import lr_scheduler

from diffusion.data_loaders import get_data_loaders
from diffusion.logger import utils
from diffusion.solver import train
from diffusion.unit2mel import Unit2Mel
from diffusion.vocoder import Vocoder
from optim import FusedAdamW


def parse_args(args=None, namespace=None):
    parser = argparse.ArgumentParser()
    parser.add_argument("-c", "--config", type=str, required=True, help="path to the config file")
    return parser.parse_args(args=args, namespace=namespace)


def main(argv=None):
    cmd = parse_args(argv)
    args = utils.load_config(cmd.config)
    if args.device != "cuda" or not torch.cuda.is_available():
        raise RuntimeError("train_diff.py: the MI355X engine trains on the GPU only (config `device: cuda`, a visible GPU)")
    rank, world = 0, 1
    if "RANK" in os.environ and "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
        local = int(os.environ.get("LOCAL_RANK", rank))
        torch.cuda.set_device(local)
        if not dist.is_initialized():
            dist.init_process_group(os.environ.get("SVC_DIST_BACKEND", "nccl"), init_method="env://", world_size=world, rank=rank,
                                    device_id=torch.device("cuda", local) if os.environ.get("SVC_DIST_BACKEND", "nccl") == "nccl" else None)
    else:
        torch.cuda.set_device(args.env.gpu_id or 0)
    say = print if rank == 0 else (lambda *a, **k: None)
    say(" > config:", cmd.config)
    say(" >    exp:", args.env.expdir)

    vocoder = Vocoder(args.vocoder.type, args.vocoder.ckpt, device=args.device)
    model = Unit2Mel(args.data.encoder_out_channels, args.model.n_spk, args.model.use_pitch_aug, vocoder.dimension,
                     args.model.n_layers, args.model.n_chans, args.model.n_hidden, args.model.timesteps, args.model.k_step_max)
    say(f" > Now model timesteps is {model.timesteps}, and k_step_max is {model.k_step_max}")
    model.to(args.device)                                  # before the optimizer: its arena takes over the parameters' storage
    optimizer = FusedAdamW(model.parameters())             # torch.optim.AdamW() defaults, as train_diff.py:52
    initial_global_step, model, optimizer = utils.load_model(args.env.expdir, model, optimizer, device=args.device)
    for group in optimizer.param_groups:                   # :55-59
        group["initial_lr"] = args.train.lr
        group["lr"] = args.train.lr * (args.train.gamma ** max((initial_global_step - 2) // args.train.decay_step, 0))
        group["weight_decay"] = args.train.weight_decay
    scheduler = lr_scheduler.StepLR(optimizer, step_size=args.train.decay_step, gamma=args.train.gamma,
                                    last_epoch=initial_global_step - 2)
    loader_train, loader_valid = get_data_loaders(args, whole_audio=False, rank=rank, world=world)
    train(args, initial_global_step, model, optimizer, scheduler, vocoder, loader_train, loader_valid)


if __name__ == "__main__":
    main()
