"""GPU: the parity suite is sensitive at the level the kernels deliver.  A build whose FFT twiddle table is wrong in ONE
entry by 1e-9 (NATAC_FAULT_TWIDDLE, natac_api.hip: ensure_fft) must fail the golden / oracle comparisons -- with the north-star 1e-5 as
the asserted tolerance it passed all of them."""
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SUBSET = ["tests/test_gpu_golden.py", "tests/test_gpu_configs.py::test_config4_10kb_tiles_match_oracle",
          "tests/test_gpu_bg_ext.py", "tests/test_host_api_gpu.py::test_nuc_helper_matches_reference"]


def _run(env_extra):
    env = dict(os.environ, **env_extra)
    env.pop("NATAC_TRACK_STATS", None)
    r = subprocess.run([sys.executable, "-m", "pytest", "-m", "gpu", "-q", "-p", "no:cacheprovider"] + SUBSET, cwd=ROOT, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1500)
    tail = r.stdout.strip().splitlines()[-1]
    failed = int((re.search(r"(\d+) failed", tail) or [0, 0])[1])
    passed = int((re.search(r"(\d+) passed", tail) or [0, 0])[1])
    return failed, passed, r.stdout


def test_a_twiddle_entry_off_by_1e_9_fails_the_parity_tests():
    failed, passed, out = _run({"NATAC_FAULT_TWIDDLE": "37:1e-9"})
    assert failed >= 3, out[-3000:]
    names = set(re.findall(r"FAILED (tests/[^ ]+)", out))
    assert any("test_nuc_tracks_match_reference" in n for n in names), names
    # and the same subset is green without the fault (the failures above are the fault's, not the subset's)
    failed0, passed0, out0 = _run({})
    assert failed0 == 0 and passed0 >= failed + passed, out0[-3000:]
