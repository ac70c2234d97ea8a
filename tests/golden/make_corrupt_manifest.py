"""Writes tests/golden/corrupt/manifest.json: what the REFERENCE answers for each corrupted frame in that directory.

The frames were found by tools/fuzz_emu_decode.py (seeds 3 and 4: one flipped bit in a frame the reference compressed) or made
by hand (truncation; raw_rle_blocks_beyond_window.zst: a 1 KiB window descriptor in front of a 5 120-byte raw block and a 5 000-byte
RLE block — out of spec, but the one-shot frame loop bounds those block types by the destination only, zstd_decompress.c:1020-1026;
window_1GiB_declared.zst: a 2^30 window descriptor, which only ZSTD_decompressStream limits, :2231;
sequence_stream_runs_dry.zst: the sequence bit stream ends two sequences early and the reference's garbage sequences stop on
dstSize_tooSmall).  For each, the manifest records the answer of
  * "portable": the reference's decoder built with its own HUF_DISABLE_FAST_DECODE switch (oracle/_ref/libzstd_ref_portable.so,
    `make -C oracle refportable`) — the loops every platform without the 64-bit fast Huffman path runs.  THIS is the contract
    the product decoder is tested against: bytes (sha256) or refusal.
  * "default": the stock x86-64 build (oracle/_ref/libzstd_ref.so).  Its fast Huffman loops skip the end-of-stream check
    (N/decompress/huf_decompress.c:873-888 vs :697), so it accepts some frames the portable loops refuse; recorded for the
    record, not asserted.
Run where /root/reference (hence both builds) exists:  python tests/golden/make_corrupt_manifest.py"""
import hashlib, json, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import ref

def answer(fn, frame, cap):
    try:
        out = fn(frame, cap)
    except ref.ZstdRefError as e:
        return {"error": str(e)}
    return {"size": len(out), "sha256": hashlib.sha256(out).hexdigest()}

d = os.path.join(HERE, "corrupt")
man = {}
for name in sorted(os.listdir(d)):
    if not name.endswith(".zst"):
        continue
    z = open(os.path.join(d, name), "rb").read()
    cap = ref.lib().ZSTD_getFrameContentSize(z, len(z))
    if cap >= (1 << 62):
        cap = 1 << 16
    man[name] = {"capacity": cap, "portable": answer(ref.decompress_portable, z, cap), "default": answer(ref.decompress, z, cap)}
json.dump(man, open(os.path.join(d, "manifest.json"), "w"), indent=1, sort_keys=True)
print(json.dumps(man, indent=1, sort_keys=True))
