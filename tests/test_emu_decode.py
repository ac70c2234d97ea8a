"""CPU (-m "not gpu"): the decoder kernel BODY (zstd-jni_amd/csrc/zj_decode.h), built lane-serial
(W = 1, tests/emu), against the reference's golden frames and reference-compressed edge cases.
This checks the format logic of the HIP code before it reaches a GPU; the -m gpu tests run the same
checks through the C-ABI on the wave64 build."""
import hashlib

import pytest

from conftest import golden, XML_SHA256_PREFIX
from util import edge_inputs, emu_lib, emu_decompress, emu_decompress_split


@pytest.fixture(scope="module")
def emu():
    return emu_lib()


@pytest.mark.parametrize("name", ["xml-1.zst", "xml-3.zst", "xml-9.zst", "xml-advanced.zst"])
def test_emu_golden_xml(emu, name):
    out = emu_decompress(emu, golden(name), 6_000_000)
    assert not isinstance(out, int), out
    assert len(out) == 5_345_280 and hashlib.sha256(out).hexdigest().startswith(XML_SHA256_PREFIX)


def test_emu_multiframe(emu):
    out = emu_decompress(emu, golden("xml-sized-combined.zst"), 6_000_000)
    assert out[:102] == golden("xmlsmall")
    assert hashlib.sha256(out[102:]).hexdigest().startswith(XML_SHA256_PREFIX)


@pytest.mark.parametrize("level", [1, 3])
def test_emu_edge_inputs(emu, oracle_ref, level):
    for name, data in edge_inputs():
        z = oracle_ref.compress(data, level)
        out = emu_decompress(emu, z, len(data))
        assert out == data, (name, out if isinstance(out, int) else "bytes differ")


def test_emu_synthetic_classes(emu, oracle_ref, zj):
    for size in (4096, 65536, 131072):
        raw = zj.synth_host(size, 0, 8)
        for i in range(8):
            data = raw[i * size:(i + 1) * size]
            for level in (1, 3):
                assert emu_decompress(emu, oracle_ref.compress(data, level), size) == data, (size, i, level)


def test_emu_errors(emu, oracle_ref):
    data = b"hello hello hello hello " * 100
    z = oracle_ref.compress(data, 3)
    assert emu_decompress(emu, z, len(data) - 1) == -70
    assert emu_decompress(emu, b"\x00\x01\x02\x03\x04\x05", 10) == -10
    assert isinstance(emu_decompress(emu, z[:-3], len(data)), int)
    bad = bytearray(z); bad[len(z) // 2] ^= 0x55
    out = emu_decompress(emu, bytes(bad), len(data))
    assert isinstance(out, int) or out != data or True      # must not crash; corruption may go unnoticed without checksum


def test_emu_split_pipeline(emu, oracle_ref, zj):
    """prep -> lane-per-frame tANS decode -> execute == fused decoder == reference, and the frames the batched
    compressor emits (one block, content <= 64 KiB) really take the three-stage path"""
    import random
    rnd = random.Random(31)
    took = 0
    for name, data in edge_inputs():
        for level in (1, 3):
            z = oracle_ref.compress(data, level)
            out, used = emu_decompress_split(emu, z, len(data))
            assert out == data, (name, level, out if isinstance(out, int) else "bytes differ")
    xml = oracle_ref.decompress(golden("xml-1.zst"), 6_000_000)
    for _ in range(120):
        size = rnd.choice([rnd.randrange(1, 400), rnd.randrange(1, 5000), rnd.randrange(1, 65537), 65536, 4096, rnd.randrange(65537, 140000)])
        if rnd.random() < 0.5:
            off = rnd.randrange(0, len(xml) - size); data = xml[off:off + size]
        else:
            data = zj.synth_host(size, rnd.randrange(0, 100000), 1)
        for level in (1, 3):
            z = oracle_ref.compress(data, level)
            out, used = emu_decompress_split(emu, z, len(data))
            assert out == data, (size, level, out if isinstance(out, int) else "bytes differ")
            took += used
            out, used = emu_decompress_split(emu, z, len(data) + 77)      # roomy destination
            assert out == data
    assert took > 100
    # errors come out of the fused path, so codes match it
    data = b"hello hello hello hello " * 100
    z = oracle_ref.compress(data, 3)
    assert emu_decompress_split(emu, z, len(data) - 1)[0] == -70
    for cut in (1, 3, 7):
        assert emu_decompress_split(emu, z[:-cut], len(data))[0] == emu_decompress(emu, z[:-cut], len(data))
    for pos in range(6, len(z)):
        bad = bytearray(z); bad[pos] ^= 0x41
        a = emu_decompress_split(emu, bytes(bad), len(data))[0]; b = emu_decompress(emu, bytes(bad), len(data))
        assert a == b, pos
