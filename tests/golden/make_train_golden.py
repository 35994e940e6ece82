"""Generates tests/golden/train_golden.npz: inputs and expected outputs of the reference's photometric loss and
optimizer step, computed with PLAIN TORCH CPU OPS — the reference's loss code is exactly these calls
(utils/loss_utils.py:40-41, 46-92: gaussian(), create_window(), F.conv2d(padding=5, groups=C); train.py:167-173) and its
optimizer is torch.optim.Adam(lr=0.0, eps=1e-15) with per-group learning rates (scene/gaussian_model.py:196-209).
The reference's own `utils.loss_utils.l1_loss` / `ssim` ARE importable on CPU in the build container,
so when /root/reference is present they are imported and evaluated on the same float32 inputs; their values are stored
as `*_ref_l1`, `*_ref_ssim`, `*_ref_grad` (float32 evaluation, as the reference runs) next to the float64 ones.
Run in the build container:  python tests/golden/make_train_golden.py
"""
import sys
import os
from math import exp

import numpy as np
import torch
import torch.nn.functional as F


def window(channel, dtype):
    g = torch.Tensor([exp(-(x - 11 // 2) ** 2 / float(2 * 1.5 ** 2)) for x in range(11)])
    g = (g / g.sum()).unsqueeze(1)
    w2d = g.mm(g.t()).float().unsqueeze(0).unsqueeze(0)
    return w2d.expand(channel, 1, 11, 11).contiguous().to(dtype)


def ssim_map(img1, img2):
    C = img1.size(-3)
    w = window(C, img1.dtype)
    mu1, mu2 = F.conv2d(img1, w, padding=5, groups=C), F.conv2d(img2, w, padding=5, groups=C)
    mu1_sq, mu2_sq, mu1_mu2 = mu1.pow(2), mu2.pow(2), mu1 * mu2
    s1 = F.conv2d(img1 * img1, w, padding=5, groups=C) - mu1_sq
    s2 = F.conv2d(img2 * img2, w, padding=5, groups=C) - mu2_sq
    s12 = F.conv2d(img1 * img2, w, padding=5, groups=C) - mu1_mu2
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    return ((2 * mu1_mu2 + C1) * (2 * s12 + C2)) / ((mu1_sq + mu2_sq + C1) * (s1 + s2 + C2))


def reference_loss_utils():
    ref = "/root/reference"
    if not os.path.isdir(ref):
        return None
    sys.dont_write_bytecode = True
    sys.path.insert(0, ref)
    try:
        from utils import loss_utils     # the reference's module (imports torch only; its fusedssim import is guarded)
        return loss_utils
    finally:
        sys.path.remove(ref)


def main():
    torch.manual_seed(7)
    out = {}
    ref_lu = reference_loss_utils()
    for tag, (C, H, W) in dict(a=(3, 37, 53), b=(1, 16, 16), c=(3, 9, 70)).items():
        gt = torch.rand(C, H, W, dtype=torch.float64)
        # a blurred + noisy version of gt: realistic SSIM range instead of ~0
        img = (0.7 * gt + 0.3 * torch.rand(C, H, W, dtype=torch.float64)).clamp(0, 1)
        img[:, : H // 3] = gt[:, : H // 3]          # exact-equality region: sign(0) = 0 in the L1 gradient
        img.requires_grad_(True)
        m = ssim_map(img, gt)
        l1 = (img - gt).abs().mean()
        loss = 0.8 * l1 + 0.2 * (1.0 - m.mean())
        loss.backward()
        out[f"{tag}_img"], out[f"{tag}_gt"] = img.detach().numpy().astype(np.float32), gt.numpy().astype(np.float32)
        out[f"{tag}_ssim_map"] = m.detach().numpy()
        out[f"{tag}_l1"], out[f"{tag}_ssim"], out[f"{tag}_loss"] = l1.item(), m.mean().item(), loss.item()
        out[f"{tag}_grad"] = img.grad.numpy()
        if ref_lu is not None:
            x32 = torch.from_numpy(out[f"{tag}_img"]).requires_grad_(True)
            g32 = torch.from_numpy(out[f"{tag}_gt"])
            r_l1, r_ss = ref_lu.l1_loss(x32, g32), ref_lu.ssim(x32, g32)
            (0.8 * r_l1 + 0.2 * (1.0 - r_ss)).backward()
            out[f"{tag}_ref_l1"], out[f"{tag}_ref_ssim"], out[f"{tag}_ref_grad"] = r_l1.item(), r_ss.item(), x32.grad.numpy()
            assert abs(r_l1.item() - l1.item()) < 1e-6 and abs(r_ss.item() - m.mean().item()) < 1e-5
        # float32 inputs were rounded from the float64 ones; expected values are for the float64 inputs (difference ~1e-8)
    # Adam: 3 steps on 2 groups with different lrs, float32 like the reference's parameters
    p1, p2 = torch.randn(257, dtype=torch.float32), torch.randn(96, dtype=torch.float32)
    out["adam_p0"] = torch.cat([p1, p2]).numpy().copy()
    p1.requires_grad_(True); p2.requires_grad_(True)
    opt = torch.optim.Adam([{"params": [p1], "lr": 0.00016}, {"params": [p2], "lr": 0.0025}], lr=0.0, eps=1e-15)
    grads = []
    for step in range(3):
        g = torch.randn(257 + 96, dtype=torch.float32) * (10.0 ** (-step))
        g[5] = 0.0                                     # zero gradient entries exercise the eps path
        grads.append(g.numpy().copy())
        p1.grad, p2.grad = g[:257].clone(), g[257:].clone()
        opt.step()
        out[f"adam_p{step + 1}"] = torch.cat([p1.detach(), p2.detach()]).numpy().copy()
    out["adam_grads"] = np.stack(grads)
    out["adam_lr"] = np.concatenate([np.full(257, 0.00016, np.float32), np.full(96, 0.0025, np.float32)])
    st1, st2 = opt.state[p1], opt.state[p2]
    out["adam_m3"] = torch.cat([st1["exp_avg"], st2["exp_avg"]]).numpy().copy()
    out["adam_v3"] = torch.cat([st1["exp_avg_sq"], st2["exp_avg_sq"]]).numpy().copy()
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "train_golden.npz"), **out)
    print({k: (v.shape if hasattr(v, "shape") else v) for k, v in out.items()})


if __name__ == "__main__":
    main()
