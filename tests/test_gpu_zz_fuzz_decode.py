"""GPU (-m gpu): bounded differential fuzz of the decoders through the C-ABI (tests/gpu_fuzz_decode.py): ~2 000 valid and
damaged frames, both pipelines, with and without dictionary, guard bytes behind every destination slot.  The wave-parallel
code paths (#if ZJ_ON_GPU: dependency rounds of the LZ77 execution, staged windows, 4-stream Huffman) are exercised here —
the lane-serial CPU build of tests/emu compiles them out."""
import pytest

import gpu_fuzz_decode as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu(zj):
    import torch
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    zj.batch.init(0)
    return zj


@pytest.mark.parametrize("split_min", [1, 1000000000])
def test_gpu_decoders_answer_like_the_reference_on_random_frames(gpu, oracle_ref, monkeypatch, split_min):
    monkeypatch.setenv("ZJNI_DSPLIT_MIN", str(split_min))      # three-stage pipeline / fused kernel
    cases = F.make_cases(gpu, oracle_ref, seed=20260924, count=800)
    bad = F.run_cases(gpu, cases)
    assert not bad, bad[:8]


@pytest.mark.parametrize("split_min", [1, 1000000000])
def test_gpu_decoders_with_dictionary_on_random_frames(gpu, oracle_ref, monkeypatch, split_min):
    import util
    monkeypatch.setenv("ZJNI_DSPLIT_MIN", str(split_min))
    recs = util.json_records(20000, seed=5)
    dic = oracle_ref.train_dict([b",".join(recs[i * 13:i * 13 + 200])[:4096] for i in range(1000)], 60000)
    cases = F.make_cases(gpu, oracle_ref, seed=77, count=300, dictionary=dic)
    with gpu.ZstdDictDecompress(dic) as dd:
        bad = F.run_cases(gpu, cases, dictionary_obj=dd)
    assert not bad, bad[:8]
