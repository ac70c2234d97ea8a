"""oracle/ — CPU checkers.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
The product path (zstd-jni_amd/) never does.

  oracle.ref   — the reference's own libzstd 1.5.7 (oracle/_ref/libzstd_ref.so, built by oracle/Makefile
                 from /root/reference/src/main/native; travels prebuilt to the GPU box)
  oracle.port  — our plain-C restatement (oracle/zstd_oracle_*.c -> oracle/libzstd_oracle.so)
"""
from . import ref, port  # noqa: F401
