"""The number tables of DESIGN.md section 7 (round 6), profiles/HISTORY.md (rounds 4, 5) and profiles/README.md are GENERATED from the files under profiles/ (tools/gen_results.py);
this test fails when they are out of date (prose that quotes a profile must follow from the committed file)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


import pytest


@pytest.mark.parametrize("tag", ["r6", "r5", "r4"])
def test_generated_result_blocks_match_the_profile_files(tag):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_results.py"), tag, "--check"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
