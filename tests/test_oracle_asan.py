"""The oracle and the data generator under AddressSanitizer + UBSan (CPU build)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_pipelines_are_sanitizer_clean():
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "asan_check"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "asan_check ok" in r.stdout
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr
