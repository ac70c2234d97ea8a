"""GPU (-m gpu): corrupted frames through the C-ABI answer what the reference's portable decoder loops answer (bytes or
refusal) on both decode pipelines — see tests/test_corrupt_frames.py for the contract and tests/golden/corrupt/ for the frames."""
import os

import pytest

from test_corrupt_frames import MANIFEST, frame, check

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu(zj):
    import torch
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    zj.batch.init(0)
    return zj


@pytest.mark.parametrize("split_min", ["1", "1000000000"])
def test_gpu_corrupt_frames_answer_like_the_reference(gpu, monkeypatch, split_min):
    monkeypatch.setenv("ZJNI_DSPLIT_MIN", split_min)      # three-stage pipeline / fused kernel
    names = sorted(MANIFEST)
    outs = gpu.decompress_batch([frame(n) for n in names] * 3, [MANIFEST[n]["capacity"] for n in names] * 3)
    for n, o in zip(names * 3, outs):
        check(n, -abs(o.getErrorCode()) if isinstance(o, Exception) else o)
