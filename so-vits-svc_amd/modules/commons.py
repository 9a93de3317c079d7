"""Host-side helpers with the names train.py / models.py import from modules.commons in the reference
(modules/commons.py).  These are index/mask/bookkeeping utilities on the host control path, not arithmetic of the
hot path; the fused gate (commons.fused_add_tanh_sigmoid_multiply, :129-136) lives inside the conv epilogue
(SVC_EPI_GATE) and is exposed here for API completeness through the same kernel."""
import torch

import svc_hip as S


def get_padding(kernel_size, dilation=1):
    """'same' padding of a dilated odd kernel (reference modules/commons.py:33-34)."""
    return int((kernel_size * dilation - dilation) / 2)


def init_weights(m, mean=0.0, std=0.01):
    """Reference modules/commons.py:25-31: normal init of every *Conv* module's (effective) weight."""
    if hasattr(m, "init_normal_"):
        m.init_normal_(mean, std)
    elif m.__class__.__name__.find("Conv") != -1 and hasattr(m, "weight"):
        m.weight.data.normal_(mean, std)


def sequence_mask(length, max_length=None):
    """[B] lengths -> [B, max_length] bool (reference modules/commons.py:144-148)."""
    if max_length is None:
        max_length = length.max()
    pos = torch.arange(max_length, dtype=length.dtype, device=length.device)
    return pos.unsqueeze(0) < length.unsqueeze(1)


def slice_segments(x, ids_str, segment_size=4):
    """Gather x[b, :, ids_str[b] : ids_str[b]+segment_size] for every b (reference :67-73) with one batched
    index op instead of the reference's per-item Python loop."""
    idx = ids_str.view(-1, 1, 1).to(torch.long) + torch.arange(segment_size, device=x.device).view(1, 1, -1)
    return torch.gather(x, 2, idx.expand(-1, x.size(1), -1))


def slice_pitch_segments(x, ids_str, segment_size=4):
    idx = ids_str.view(-1, 1).to(torch.long) + torch.arange(segment_size, device=x.device).view(1, -1)
    return torch.gather(x, 1, idx)


DEVICE_RNG = False   # True: draw the per-item randoms on the device (needed inside a hipGraph capture: no H2D copies)


def _rand(b, device):
    """Reference: `torch.rand([b]).to(device)` — a CPU draw copied over (:20); on-device when DEVICE_RNG is set."""
    return torch.rand([b], device=device) if DEVICE_RNG else torch.rand([b]).to(device=device)


def rand_slice_segments_with_pitch(x, pitch, x_lengths=None, segment_size=4):
    """Reference :15-23: one uniform draw per batch item, start = floor(u * (len - seg + 1))."""
    b, d, t = x.size()
    if x_lengths is None:
        x_lengths = t
    ids_str_max = x_lengths - segment_size + 1
    ids_str = (_rand(b, x.device) * ids_str_max).to(dtype=torch.long)
    return slice_segments(x, ids_str, segment_size), slice_pitch_segments(pitch, ids_str, segment_size), ids_str


def rand_slice_segments(x, x_lengths=None, segment_size=4):
    b, d, t = x.size()
    if x_lengths is None:
        x_lengths = t
    ids_str_max = x_lengths - segment_size + 1
    ids_str = (_rand(b, x.device) * ids_str_max).to(dtype=torch.long)
    return slice_segments(x, ids_str, segment_size), ids_str


def subsequent_mask(length):
    return torch.tril(torch.ones(length, length)).unsqueeze(0).unsqueeze(0)


def convert_pad_shape(pad_shape):
    return [item for sub in pad_shape[::-1] for item in sub]


def kl_divergence(m_p, logs_p, m_q, logs_q):
    kl = (logs_q - logs_p) - 0.5
    kl += 0.5 * (torch.exp(2. * logs_p) + ((m_p - m_q) ** 2)) * torch.exp(-2. * logs_q)
    return kl


def clip_grad_value_(parameters, clip_value, norm_type=2):
    """Reference :168-183 returns the total grad norm and optionally clamps.  One fused norm instead of a
    host sync per parameter tensor (862 .item() calls per step in the reference)."""
    if isinstance(parameters, torch.Tensor):
        parameters = [parameters]
    grads = [p.grad for p in parameters if p.grad is not None]
    norm_type = float(norm_type)
    if not grads:
        return 0.0
    total = torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(g.detach(), norm_type) for g in grads]),
                                     norm_type)
    if clip_value is not None:
        for g in grads:
            g.clamp_(min=-float(clip_value), max=float(clip_value))
    return total.item()
