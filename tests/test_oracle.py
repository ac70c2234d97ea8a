"""CPU (-m "not gpu"): the oracle restatement is pinned against the reference's golden vectors and
against oracle/_ref (the reference's own libzstd 1.5.7 compiled from its sources)."""
import hashlib

import pytest

from conftest import golden, XML_SHA256_PREFIX
from util import edge_inputs

GOLDEN_XML = ["xml-1.zst", "xml-3.zst", "xml-9.zst", "xml-advanced.zst"]


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
    with pytest.raises(oracle_port.ZstdOracleError) as e:
        oracle_port.decompress(b"\x00\x01\x02\x03\x04\x05", 10)
    assert e.value.code == 10
