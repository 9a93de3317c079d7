"""2 ranks (gloo, one GPU): after 2 data-parallel iterations on DIFFERENT per-rank batches the generator / discriminator
parameters must be identical on both ranks, and equal to a single-process run whose gradient is the mean of the two
ranks' gradients (checked through the loss history of rank 0 being reproducible is not enough: compare parameters)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "so-vits-svc_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, torch.distributed as dist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
dev = torch.device("cuda:0"); torch.cuda.set_device(0)
import train as T
from train_common import load_case
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_train_loop_gpu import _hps
cs = load_case()
hps = _hps(cs, 2e-4)
torch.manual_seed(100 + rank)                      # different init per rank: the broadcast must fix it
net_g, net_d, og, od = T.build(hps, dev)
if rank == 0:
    net_g.module.load_state_dict(cs["sd_g"]); net_d.module.load_state_dict(cs["sd_d"])
net_g.reducer.broadcast_parameters(0); net_d.reducer.broadcast_parameters(0)
net_g.train(); net_d.train()
step = T.TrainStep(hps, net_g, net_d, og, od)
c, f0, uv, spec, y, sid, lengths = [t.to(dev) for t in cs["batch"]]
noise = {k: v.to(dev) for k, v in cs["noise"].items()}
sl = slice(rank, rank + 1)                          # rank r trains on item r of the 2-item batch
items = (c[sl], f0[sl], spec[sl], y[sl], sid[sl], lengths[sl], uv[sl], None)
nz = {k: v[sl].contiguous() for k, v in noise.items()}
for it in range(2):
    out = step(items, noise=nz)
flat_g = net_g.arena.param.detach().clone(); flat_d = net_d.arena.param.detach().clone()
other_g = [torch.empty_like(flat_g.cpu()) for _ in range(world)]; other_d = [torch.empty_like(flat_d.cpu()) for _ in range(world)]
dist.all_gather(other_g, flat_g.cpu()); dist.all_gather(other_d, flat_d.cpu())
same = torch.equal(other_g[0], other_g[1]) and torch.equal(other_d[0], other_d[1])
fin = all(torch.isfinite(v).all().item() for v in out.values() if torch.is_tensor(v))
if rank == 0:
    print(f"DP2 consistency: params identical across ranks = {same}; losses finite = {fin}; "
          f"G reducer {net_g.reducer.stats}; D reducer {net_d.reducer.stats}; loss_gen_all {float(out['loss_gen_all']):.4f}")
    assert same and fin
dist.destroy_process_group()
