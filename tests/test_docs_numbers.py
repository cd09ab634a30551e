"""The number tables of DESIGN.md section 7c and profiles/README.md are GENERATED from the files under profiles/ (tools/gen_results.py);
this test fails when they are out of date (prose that quotes a profile must follow from the committed file)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_generated_result_blocks_match_the_profile_files():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_results.py"), "r4", "--check"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
