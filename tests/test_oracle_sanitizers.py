"""AddressSanitizer + UBSan run of the CPU oracle and the synthetic encoder (CPU build only: GPU sanitizers are
not available on this pool).  A restatement of C# int arithmetic is full of shift-count / signed-overflow traps
(SURVEY.md App. B Q12/Q13); the fuzz driver feeds garbage and mutated packets and must finish with no report."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_and_encoder_under_asan_ubsan():
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "fuzz_asan"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    r = subprocess.run([os.path.join(ROOT, "oracle", "fuzz_asan"), "3000"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "bad_status=0" in r.stdout and "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr
