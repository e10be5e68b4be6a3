"""Input pipeline (twig/dataset/sod_train.py:31-54): the numpy restatement against vectors produced by real Pillow
and against the Pillow installed here (CPU), and the HIP kernels against the restatement, bit-exactly (GPU)."""
import os

import numpy as np
import pytest
import torch

from oracle import preprocess_cpu as pc
from oracle.make_golden import GOLDEN_DIR, PREPROCESS_CASES


@pytest.fixture(scope="module")
def G():
    return np.load(os.path.join(GOLDEN_DIR, "preprocess.npz"))


@pytest.mark.parametrize("case", PREPROCESS_CASES, ids=[c[0] for c in PREPROCESS_CASES])
def test_oracle_resize_matches_pillow_vectors(G, case):
    name, h, w, c, s = case
    x = G[name + ".in"]
    xi = x[:, :, 0] if c == 1 else x
    assert np.array_equal(pc.resize_bilinear_u8(xi, s), G[name + ".resized"])
    assert np.array_equal(pc.resize_bilinear_u8(np.ascontiguousarray(xi[:, ::-1]), s), G[name + ".resized_flip"])


def test_oracle_matches_installed_pillow_and_totensor_normalize():
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(7)
    for (h, w, c, s) in [(61, 45, 3, 32), (40, 40, 1, 64), (128, 96, 3, 96)]:
        x = rng.integers(0, 256, (h, w, c), dtype=np.uint8)
        xi = x[:, :, 0] if c == 1 else x
        ref = np.asarray(Image.fromarray(xi).resize((s, s), Image.BILINEAR))
        assert np.array_equal(pc.resize_bilinear_u8(xi, s), ref)
        t = torch.from_numpy(ref if c == 3 else ref[:, :, None]).permute(2, 0, 1).float().div(255)       # ToTensor
        if c == 3:
            t = (t - torch.tensor(pc.IMAGENET_MEAN).view(3, 1, 1)) / torch.tensor(pc.IMAGENET_STD).view(3, 1, 1)   # Normalize
        assert np.array_equal(pc.preprocess(xi, s, normalize=(c == 3)), t.numpy())


@pytest.mark.gpu
@pytest.mark.parametrize("h,w,c,s", [(97, 131, 3, 64), (20, 30, 3, 48), (50, 33, 1, 40), (90, 64, 3, 64), (480, 640, 3, 512), (300, 400, 1, 384)])
@pytest.mark.parametrize("flip", [False, True])
def test_hip_pipeline_bit_exact(h, w, c, s, flip):
    import dgtd
    rng = np.random.default_rng(h * 1000 + w)
    x = rng.integers(0, 256, (h, w, c), dtype=np.uint8)
    xi = x[:, :, 0] if c == 1 else x
    want = pc.preprocess(xi, s, normalize=(c == 3), flip=flip)
    got = dgtd.runner.device_preprocess(torch.from_numpy(xi.copy()).cuda(), s, normalize=(c == 3), flip=flip)
    assert got.shape == (c, s, s) and got.dtype == torch.float32
    assert np.array_equal(got.cpu().numpy(), want)


@pytest.mark.gpu
def test_device_sample_shares_the_flip():
    import dgtd
    rng = np.random.default_rng(3)
    rgb, gt, dep = (rng.integers(0, 256, sh, dtype=np.uint8) for sh in ((70, 90, 3), (70, 90), (70, 90)))
    out = dgtd.runner.device_sample(torch.from_numpy(rgb).cuda(), torch.from_numpy(gt).cuda(), torch.from_numpy(dep).cuda(), 64, flip=True,
                                    out_dtype=torch.bfloat16)
    assert out["input"].shape == (3, 64, 64) and out["label"].shape == (1, 64, 64) and out["depth"].dtype == torch.bfloat16
    want = torch.from_numpy(pc.preprocess(gt, 64, normalize=False, flip=True)).bfloat16()
    assert torch.equal(out["label"].cpu(), want)
