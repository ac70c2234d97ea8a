"""CPU (-m "not gpu"): corrupted frames.  The contract: for ANY input the decoder answers what the reference's portable
decoder loops answer — the same bytes, or a refusal (tests/golden/make_corrupt_manifest.py explains the two reference builds:
the stock x86-64 build's fast Huffman loops skip the end-of-stream check, so the reference itself is not of one mind there).
Checked here for the kernel bodies (lane-serial build) on both decode pipelines, for the C restatement in oracle/, and the
manifest itself against the reference builds when they are present."""
import hashlib
import json
import os

import pytest

from conftest import GOLDEN
from util import emu_lib, emu_decompress, emu_decompress_split

CORRUPT = os.path.join(GOLDEN, "corrupt")
MANIFEST = json.load(open(os.path.join(CORRUPT, "manifest.json")))
ERR_CODE = {"Data corruption detected": 20, "Src size is incorrect": 72, "Destination buffer is too small": 70}     # ZSTD_ErrorCode, N/zstd_errors.h


def frame(name):
    return open(os.path.join(CORRUPT, name), "rb").read()


def check(name, got):
    """got: bytes, or a negative int / exception-with-code for a refusal"""
    want = MANIFEST[name]["portable"]
    if "error" in want:
        assert isinstance(got, int), (name, "reference refuses, we returned bytes")
        assert -got == ERR_CODE[want["error"]], (name, got, want)
    else:
        assert not isinstance(got, int), (name, got, "reference accepts")
        assert len(got) == want["size"] and hashlib.sha256(got).hexdigest() == want["sha256"], name


@pytest.fixture(scope="module")
def emu():
    return emu_lib()


@pytest.mark.parametrize("name", sorted(MANIFEST))
def test_emu_corrupt_frame_answers_like_the_reference(emu, name):
    z, cap = frame(name), MANIFEST[name]["capacity"]
    check(name, emu_decompress(emu, z, cap))
    check(name, emu_decompress_split(emu, z, cap)[0])


@pytest.mark.parametrize("name", sorted(MANIFEST))
def test_oracle_port_corrupt_frame(oracle_port, name):
    z, cap = frame(name), MANIFEST[name]["capacity"]
    try:
        got = oracle_port.decompress(z, cap)
    except oracle_port.ZstdOracleError as e:
        got = -e.code
    check(name, got)


def test_manifest_is_what_the_reference_builds_answer(oracle_ref):
    if not os.path.exists(oracle_ref.PORTABLE_PATH):
        pytest.skip("oracle/_ref/libzstd_ref_portable.so not built")
    for name, m in MANIFEST.items():
        z, cap = frame(name), m["capacity"]
        for key, fn in (("portable", oracle_ref.decompress_portable), ("default", oracle_ref.decompress)):
            try:
                out = fn(z, cap)
                got = {"size": len(out), "sha256": hashlib.sha256(out).hexdigest()}
            except oracle_ref.ZstdRefError as e:
                got = {"error": str(e)}
            assert got == m[key], (name, key)
    # the point of the fixture set: the reference's two builds disagree on two of these frames
    assert "error" in MANIFEST["stream_not_exhausted.zst"]["portable"] and "size" in MANIFEST["stream_not_exhausted.zst"]["default"]
    assert MANIFEST["x2_last_cell_no_bits_left.zst"]["portable"]["sha256"] != MANIFEST["x2_last_cell_no_bits_left.zst"]["default"]["sha256"]


def test_random_bit_flips_answer_like_the_portable_reference(emu, oracle_ref):
    """a seeded slice of tools/fuzz_emu_decode.py: one flipped bit per frame, both pipelines against the portable build"""
    if not os.path.exists(oracle_ref.PORTABLE_PATH):
        pytest.skip("oracle/_ref/libzstd_ref_portable.so not built")
    import random
    from util import json_records
    rnd = random.Random(99)
    recs = json_records(3000, seed=9)
    accepted = 0
    for it in range(40):
        n = rnd.choice([rnd.randrange(100, 3000), rnd.randrange(3000, 40000)])
        k = rnd.randrange(3)
        d = (bytes(rnd.randrange(rnd.choice([3, 17, 200])) for _ in range(n)) if k == 0 else
             b",".join(recs[rnd.randrange(0, 2000):][:800])[:n] if k == 1 else
             bytes(min(255, int(rnd.expovariate(0.05))) for _ in range(n)))
        z = oracle_ref.compress(d, rnd.choice([1, 3, 5]))
        for _ in range(6):
            zb = bytearray(z); zb[rnd.randrange(4, len(zb))] ^= 1 << rnd.randrange(8); zb = bytes(zb)
            try:
                want = oracle_ref.decompress_portable(zb, len(d)); accepted += 1
            except oracle_ref.ZstdRefError:
                want = None
            for got in (emu_decompress(emu, zb, len(d)), emu_decompress_split(emu, zb, len(d))[0]):
                assert (None if isinstance(got, int) else got) == want, (it, len(d))
    assert accepted > 20
