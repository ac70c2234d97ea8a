"""The numbers DESIGN.md and README.md quote for the round's final lines are the ones in the committed bench lines (profiles/r06_bench_config*.json), and those lines are
of the committed sources: same build stamp as the library's, as the counter passes (profiles/r06_pmc_traffic.json) and as the text."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line(name):
    with open(os.path.join(ROOT, "profiles", name)) as f:
        return json.loads(f.read().strip().splitlines()[-1])


def test_design_quotes_the_committed_lines(zj):
    design = open(os.path.join(ROOT, "DESIGN.md")).read()
    readme = open(os.path.join(ROOT, "README.md")).read()
    stamp = zj.build_stamp()
    metric = _line("r06_bench_configmetric.json")
    assert metric["library"]["build_stamp"] == stamp, "the final lines are of another build than the committed sources"
    assert stamp in design and stamp in readme
    # the headline row: value (compress, decompress; step)
    row = "**%.1f GiB/s** (compress %.1f, decompress %.1f; step %.1f ms)" % (metric["value"], metric["compress_GiBps_per_gpu"], metric["decompress_GiBps_per_gpu"], metric["ms_per_step"])
    assert row in design, row
    assert "compress %.1f, decompress %.1f, both ways **%.1f GiB/s**" % (metric["compress_GiBps_per_gpu"], metric["decompress_GiBps_per_gpu"], metric["value"]) in readme
    r = metric["roofline"]
    assert r["kernel"] == "zj_enc_match_run_kernel" and r["traffic"] and "**%.1f GB = %.1f ×**" % (r["traffic"] / 1e9, r["traffic"] / r["algorithmic_bytes_per_launch"]) in design
    assert "**%.1f GB/s = %.3f %% of 8 TB/s**" % (r["achieved"], 100 * r["frac"]) in design
    assert metric["parity"]["all_frames_byte_identical_to_reference"] == {"frames_compared": 65536, "of": 65536, "identical": True, "sizes_equal": True}
    # every config's line: same build, whole-batch identity where the config compresses, a traffic figure for its dominant kernel
    for name, quoted in (("r06_bench_config2.json", "**%.1f GiB/s** (call"), ("r06_bench_config3.json", "**%.1f GiB/s** (its frames decompress at")):
        d = _line(name)
        assert d["library"]["build_stamp"] == stamp and (quoted % d["value"]) in design, (name, d["value"])
    for name in ("r06_bench_config1.json", "r06_bench_config1_4096.json", "r06_bench_config3.json", "r06_bench_config4.json", "r06_bench_config5shape.json"):
        d = _line(name)
        assert d["library"]["build_stamp"] == stamp, name
        assert d["parity"]["all_frames_byte_identical_to_reference"]["identical"] is True, name
    for name in ("r06_bench_config1.json", "r06_bench_config2.json", "r06_bench_config3.json", "r06_bench_config4.json", "r06_bench_config5shape.json", "r06_bench_config5_two_chunks.json"):
        assert _line(name)["roofline"]["traffic"], name
