"""Opt-in (PT_RUN_SANITIZE=1; about two minutes of compiling): the CPU suite against ASan + UBSan builds of every library that
runs on the host -- libptcore's host side (scene conversion, broad-phase records, BVH builder, shard API), the C++ mirror of the Go
host layer, and the CPU checker.  The recipe is tools/sanitize.py; the pool offers no GPU sanitizer,
so the kernels themselves are covered by the oracle comparisons, not by this."""
import os

import pytest


@pytest.mark.skipif(not os.environ.get("PT_RUN_SANITIZE") or os.environ.get("PT_SANITIZE_CHILD"),
                    reason="opt-in: PT_RUN_SANITIZE=1 (and never from inside the sanitized run itself)")
def test_cpu_suite_is_clean_under_asan_and_ubsan():
    import subprocess
    import sys

    from conftest import ROOT

    assert subprocess.run([sys.executable, os.path.join(ROOT, "tools", "sanitize.py"), "test"]).returncode == 0
