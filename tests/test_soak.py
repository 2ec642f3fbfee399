"""tools/soak.py on the GPU: the module under four frames in flight, three databases interleaved, every output of every frame compared
on the GPU with the serial result; and the check itself (a one-ulp poisoned expectation is counted exactly)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

gpu = pytest.mark.gpu


def soak(*flags):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "soak.py"), *flags], capture_output=True, text=True, timeout=600, cwd=ROOT)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, (r.stdout[-1000:], r.stderr[-2000:])
    return r.returncode, json.loads(lines[0])


@gpu
@pytest.mark.parametrize("precision", ["fp32", "fp16x4"])
def test_soak_clean(precision):
    rc, d = soak("--frames", "1500", "--precision", precision)
    assert rc == 0 and d["frames_with_any_output_differing_from_the_serial_result"] == 0 and d["frames_in_flight"] == 4
    assert d["allocated_bytes_before_after"][1] <= d["allocated_bytes_before_after"][0]
    assert d["reserved_bytes_before_after"][1] <= d["reserved_bytes_before_after"][0]


@gpu
def test_soak_check_counts_a_one_ulp_difference():
    rc, d = soak("--frames", "600", "--poison")
    n = d["poisoned_self_test"]["frames_replaying_the_poisoned_expectation"]
    assert rc == 1 and n > 0 and d["frames_with_any_output_differing_from_the_serial_result"] == n
