"""CPU: the training-pass oracle (oracle/oracle_train.cpp) against the torch-generated golden vectors
(tests/golden/train_golden.npz, made by tests/golden/make_train_golden.py from the reference's own torch formulas)."""
import os
import numpy as np
import pytest

from oracle import oracle as orc

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "train_golden.npz"))


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_ssim_l1_oracle_matches_torch_golden(tag):
    img, gt = G[f"{tag}_img"], G[f"{tag}_gt"]
    n = img.size
    s_l1, s_ss, smap = orc.ssim_l1_forward(img, gt, dtype=np.float64)
    assert abs(s_l1 / n - float(G[f"{tag}_l1"])) < 1e-6
    assert abs(s_ss / n - float(G[f"{tag}_ssim"])) < 1e-6
    assert np.abs(smap - G[f"{tag}_ssim_map"]).max() < 2e-6
    loss = 0.8 * s_l1 / n + 0.2 * (1 - s_ss / n)
    assert abs(loss - float(G[f"{tag}_loss"])) < 1e-6
    # loss = 0.8 * mean|x-y| + 0.2 * (1 - mean ssim)  ->  weights of the two sums
    d = orc.ssim_l1_backward(img, gt, 0.8 / n, -0.2 / n, dtype=np.float64)
    ref = G[f"{tag}_grad"]
    assert np.abs(d - ref).max() <= 2e-5 * np.abs(ref).max()
    # values produced by the reference's own utils/loss_utils.py (imported when the fixture was generated), float32
    assert abs(s_l1 / n - float(G[f"{tag}_ref_l1"])) < 1e-6 and abs(s_ss / n - float(G[f"{tag}_ref_ssim"])) < 1e-5
    rg = G[f"{tag}_ref_grad"]
    assert np.abs(d - rg).max() <= 2e-4 * np.abs(rg).max()
    # float32 instantiation (what the GPU tests compare against) stays close to the float64 one
    d32 = orc.ssim_l1_backward(img, gt, 0.8 / n, -0.2 / n, dtype=np.float32)
    assert np.abs(d32 - ref).max() <= 2e-4 * np.abs(ref).max()


def test_ssim_identical_images_is_one_with_zero_gradient():
    rs = np.random.RandomState(3)
    img = rs.rand(3, 20, 31).astype(np.float32)
    s_l1, s_ss, smap = orc.ssim_l1_forward(img, img, dtype=np.float64)
    assert s_l1 == 0.0 and np.abs(smap - 1.0).max() < 1e-12
    d = orc.ssim_l1_backward(img, img, 1.0, 1.0, dtype=np.float64)
    assert np.abs(d).max() < 1e-9        # SSIM is maximal at x = y and sign(0) = 0


def test_adam_oracle_matches_torch_golden():
    p, m, v = G["adam_p0"].copy(), np.zeros(353, np.float32), np.zeros(353, np.float32)
    for step in range(3):
        p, m, v = orc.adam(p, G["adam_grads"][step], m, v, G["adam_lr"], step=step + 1, dtype=np.float32)
        ref = G[f"adam_p{step + 1}"]
        assert np.abs(p - ref).max() <= 1e-6 * np.abs(ref).max(), step
    assert np.allclose(m, G["adam_m3"], rtol=1e-5, atol=1e-12) and np.allclose(v, G["adam_v3"], rtol=1e-5, atol=1e-20)
