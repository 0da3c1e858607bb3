"""The C ABI's concurrency contract (SURVEY.md 8(b) "Threading"; pkg/providers/instancetype/suite_test.go:2445-2479 calls
List() from 10 000 goroutines) under ThreadSanitizer and AddressSanitizer + UBSan: tests/native/abi_threads.cpp -- solves on
one context from four threads, consolidation rounds on a second, catalog readers, an in-place catalog writer and a
three-context device group, all at once, every result equal to the serial run -- linked with the library's host sources and
the TEST-ONLY emulation HAL.  Host code only (GPU sanitizers are not available on this pool); built by
__graft_entry__.build() through tests/native/Makefile."""
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("san", ["tsan", "asan"])
def test_threaded_abi_under_sanitizer(san):
    exe = os.path.join(HERE, "native", "_build", "abi_threads_" + san)
    if not os.path.exists(exe):
        pytest.skip("sanitizer build missing: run __graft_entry__.build()")
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1 exitcode=66", ASAN_OPTIONS="detect_leaks=1 exitcode=67", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([exe, "2"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env, timeout=600)
    out = r.stdout.decode("utf-8", "replace")
    assert r.returncode == 0 and "abi_threads ok" in out, out[-4000:]
    assert "WARNING: ThreadSanitizer" not in out and "ERROR: AddressSanitizer" not in out and "runtime error" not in out, out[-4000:]
