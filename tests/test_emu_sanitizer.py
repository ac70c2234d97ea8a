"""The kernel bodies under AddressSanitizer + UndefinedBehaviorSanitizer (tests/emu `make asan`): the lane-serial build with every
entry point working on exact-size heap copies of the source and the destination (-DEMU_EXACT), driven by a few seconds of each
randomised differential stress in tools/.  A read past the last source byte, a write past the capacity, or an out-of-range table
index in the match finders / entropy stages / decoders stops the run here instead of corrupting a neighbour frame on the GPU."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _asan_runtime():
    p = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


@pytest.mark.parametrize("tool", [["fuzz_emu_encode.py"], ["fuzz_emu_decode.py"], ["fuzz_emu_level4.py"], ["fuzz_emu_multiblock.py"],
                                  ["fuzz_emu_cdict_copy.py"], ["fuzz_emu_dict.py", "decode"], ["fuzz_emu_wave.py"], ["fuzz_emu_tight.py"],
                                  ["fuzz_emu_need.py", "NEEDMODE=1"], ["fuzz_emu_need.py", "NEEDMODE=4"], ["fuzz_emu_l3wave.py"]], ids=lambda t: "-".join(t))
def test_emu_bodies_under_sanitizers(tool):
    extra = dict(a.split("=", 1) for a in tool[1:] if "=" in a)           # NAME=value entries are environment for the tool, the rest its arguments
    tool = [a for a in tool if "=" not in a]
    rt = _asan_runtime()
    if rt is None:
        pytest.skip("no libasan in this image")
    emu = os.path.join(ROOT, "tests", "emu")
    subprocess.check_call(["make", "-s", "-C", emu, "all", "asan"])
    env = dict(os.environ, ZJNI_EMU_LIB=os.path.join(emu, "libzjni_emu_asan.so"), LD_PRELOAD=rt,
               ZJNI_EMU_WAVE_LIB=os.path.join(emu, "libzjni_emu_wave_asan.so"),
               ASAN_OPTIONS="detect_leaks=0:halt_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1", **extra)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool[0])] + tool[1:] + ["31337", "6"], env=env, cwd=ROOT,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-4000:]
    assert "bad 0" in r.stdout or "diffs {}" in r.stdout or "mismatches=0" in r.stdout, r.stdout[-2000:]
