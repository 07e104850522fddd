"""A short run of tools/soak.py inside the suite: 240 random cases (u32 / u64 keys, tuples; nine distributions;
sizes 2^8...2^22; random `end_bit`, sub-array offsets, `direct_mode` 0/1/2 with lowered thresholds) against
torch.sort.  The driver itself takes a seed and a case count for longer runs."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("seed", [20261004, 7])
def test_soak_short(seed):
    env = dict(os.environ, SOAK_LOGN_MAX="22")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "soak.py"), str(seed), "240"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    assert "all 240 cases ok" in r.stdout
