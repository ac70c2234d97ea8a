"""CPU (-m "not gpu"): the oracle restatement is pinned against the reference's golden vectors and
against oracle/_ref (the reference's own libzstd 1.5.7 compiled from its sources)."""
import hashlib

import pytest

from conftest import golden, XML_SHA256_PREFIX
from util import edge_inputs

GOLDEN_XML = ["xml-1.zst", "xml-3.zst", "xml-6.zst", "xml-9.zst", "xml-advanced.zst", "xml-1-sized.zst"]


@pytest.mark.parametrize("name", GOLDEN_XML)
def test_port_decodes_reference_golden_frames(oracle_port, name):
    # T/scala/Zstd.scala:427-566: real `zstd -1/-3/-9/--ultra` CLI frames, multi-block, window > block
    out = oracle_port.decompress(golden(name), 6_000_000)
    assert len(out) == 5_345_280
    assert hashlib.sha256(out).hexdigest().startswith(XML_SHA256_PREFIX)


def test_port_decodes_concatenated_frames(oracle_port):
    # T/scala/Zstd.scala:568-622 (xml-sized-combined.zst = xmlsmall-sized.zst ‖ xml-1-sized.zst)
    out = oracle_port.decompress(golden("xml-sized-combined.zst"), 6_000_000)
    assert out[:102] == golden("xmlsmall")
    assert hashlib.sha256(out[102:]).hexdigest().startswith(XML_SHA256_PREFIX)


def test_port_decodes_doubled_frames(oracle_port):
    # xml-1x2.zst / xml-1-sizedx2.zst of the reference's resources are xml-1.zst / xml-1-sized.zst twice (checked against the reference tree where it exists)
    import os
    for name in ("xml-1x2.zst", "xml-1-sizedx2.zst"):
        z = golden(name)
        refp = os.path.join("/root/reference/src/test/resources", name)
        if os.path.exists(refp):
            assert open(refp, "rb").read() == z, name
        out = oracle_port.decompress(z, 11_000_000)
        assert len(out) == 2 * 5_345_280 and out[:5_345_280] == out[5_345_280:], name


def test_port_small_golden(oracle_port):
    z = golden("xmlsmall-sized.zst")
    assert oracle_port.decompress(z, 102) == golden("xmlsmall")
    assert oracle_port.frame_content_size(z) == 102
    assert oracle_port.find_frame_compressed_size(z) == len(z)


def test_ref_is_libzstd_157(oracle_ref):
    assert oracle_ref.version() == "1.5.7"


@pytest.mark.parametrize("level", [1, 3])
def test_port_decoder_matches_ref_on_edge_inputs(oracle_port, oracle_ref, level):
    for name, data in edge_inputs():
        z = oracle_ref.compress(data, level)
        assert oracle_port.decompress(z, len(data)) == data, name
        assert oracle_port.frame_content_size(z) == len(data), name
        assert oracle_port.find_frame_compressed_size(z) == len(z), name


def test_port_decoder_checksum(oracle_port, oracle_ref):
    data = b"checksummed " * 1000
    z = oracle_ref.compress(data, 3, checksum=True)
    assert oracle_port.decompress(z, len(data)) == data
    bad = bytearray(z); bad[-1] ^= 0xFF
    with pytest.raises(oracle_port.ZstdOracleError) as e:
        oracle_port.decompress(bytes(bad), len(data))
    assert e.value.code == 22


def test_port_decoder_errors(oracle_port, oracle_ref):
    data = b"hello hello hello hello " * 100
    z = oracle_ref.compress(data, 3)
    with pytest.raises(oracle_port.ZstdOracleError) as e:      # T/scala/Zstd.scala:186-221
        oracle_port.decompress(z, len(data) - 1)
    assert e.value.code == 70
    with pytest.raises(oracle_port.ZstdOracleError):
        oracle_port.decompress(z[:-3], len(data))
    # short garbage is srcSize_wrong, long garbage prefix_unknown, garbage after a frame srcSize_wrong (zstd_decompress.c:966-979, :1136)
    for junk, code, name in ((b"\x00\x01\x02\x03\x04\x05", 72, "Src size is incorrect"), (bytes(range(20)), 10, "Unknown frame descriptor"),
                             (z + bytes(range(20)), 72, "Src size is incorrect")):
        with pytest.raises(oracle_port.ZstdOracleError) as e:
            oracle_port.decompress(junk, len(data))
        assert e.value.code == code
        with pytest.raises(oracle_ref.ZstdRefError, match=name):
            oracle_ref.decompress(junk, len(data))


# ---------------------------------------------------------------- encoder restatement -------
def test_port_encoder_reproduces_reference_golden(oracle_port, oracle_ref):
    # one-shot ZSTD_compress2(xmlsmall, L3) == xmlsmall-sized.zst, byte for byte
    xs = golden("xmlsmall")
    assert oracle_port.compress(xs, 3) == golden("xmlsmall-sized.zst") == oracle_ref.compress(xs, 3)


@pytest.mark.parametrize("level", [1, 2, 3])
def test_port_encoder_byte_identical_on_edge_inputs(oracle_port, oracle_ref, level):
    for name, data in edge_inputs():
        assert oracle_port.compress(data, level) == oracle_ref.compress(data, level), name


def test_port_encoder_byte_identical_on_xml_and_synthetic(oracle_port, oracle_ref, zj):
    import random
    rnd = random.Random(7)
    xml = oracle_ref.decompress(golden("xml-1.zst"), 6_000_000)
    for size in (4096, 65536, 131072):
        for _ in range(3):
            off = rnd.randrange(0, len(xml) - size)
            d = xml[off:off + size]
            for level in (1, 2, 3):
                assert oracle_port.compress(d, level) == oracle_ref.compress(d, level), (size, off, level)
            # ZSTD_c_hashLog / ZSTD_c_chainLog overrides (ZstdCompressCtx.setHashLog/setChainLog):
            # (14, 13) is the LDS-sized level-3 variant the GPU path runs
            for hl, cl in ((14, 13), (13, 12)):
                assert oracle_port.compress(d, 3, False, hl, cl) == oracle_ref.compress(d, 3, False, hl, cl), (size, off, hl, cl)
    for _ in range(120):
        size = rnd.choice([rnd.randrange(0, 300), rnd.randrange(0, 5000), rnd.randrange(0, 70000), rnd.randrange(0, 131073), 65536, 4096, 131072])
        idx = rnd.randrange(0, 10000)
        d = zj.synth_host(size, idx, 1) if size else b""
        for level in (1, 2, 3):
            assert oracle_port.compress(d, level) == oracle_ref.compress(d, level), (size, idx, level)
        assert oracle_port.compress(d, 3, True, 14, 13) == oracle_ref.compress(d, 3, True, 14, 13), (size, idx)


def test_port_encoder_scope_and_errors(oracle_port):
    with pytest.raises(oracle_port.ZstdOracleError) as e:
        oracle_port.compress(b"x" * 131073, 3)
    assert e.value.code == 40
    assert oracle_port.compress(b"", 3) == bytes.fromhex("28b52ffd2000010000")


def test_entropy_cost_table_equals_the_references():
    """zj_invprob.h (generated by tools/gen_invprob.py with exact integer arithmetic: floor(256 * -log2(x / 256))) is the table
    ZSTD_entropyCost / ZSTD_crossEntropyCost index (N/compress/zstd_compress_sequences.c:21-44); compared when the reference tree is here"""
    import os, re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    gen = [int(x) for x in re.findall(r"\d+", open(os.path.join(root, "zstd-jni_amd", "csrc", "zj_invprob.h")).read().split("{")[1])]
    assert len(gen) == 256 and gen[0] == 0 and gen[1] == 2048 and gen[128] == 256 and gen[255] == 1
    assert all(gen[i] >= gen[i + 1] for i in range(1, 255))
    for x in (3, 5, 7, 100, 200, 251):                     # floor(2048 - 256 * log2 x), checked with integers: 2^(2048 - v) >= x^256 > 2^(2047 - v)
        v = gen[x]
        assert (1 << (2048 - v)) >= x ** 256 > (1 << (2047 - v))
    src = "/root/reference/src/main/native/compress/zstd_compress_sequences.c"
    if os.path.exists(src):
        m = re.search(r"kInverseProbabilityLog256\[256\] = \{(.*?)\};", open(src).read(), re.S)
        assert [int(x) for x in re.findall(r"\d+", m.group(1))] == gen
