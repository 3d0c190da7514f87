"""The spatial GI filter's block -> tile walk (csrc/device/xcd.h xcdWalk2: columns of XCDs x chunks per XCD) is a launch geometry: whatever it is, every tile is
filtered exactly once and the output bytes are the same. One child process per walk (the library reads the hooks once), ragged sizes on both sides of the
width the default walk switches at (1536 .. 3071 texels: two columns)."""
import os
import subprocess
import sys

import pytest

CHILD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "spatial_walk_child.py")
WALKS = [None, (1, 2), (2, 1), (2, 3), (4, 5), (8, 2)]


def run(tw, th, walk):
    env = dict(os.environ)
    env.pop("PLR_SPATIAL_SPLIT_X", None)
    env.pop("PLR_SPATIAL_CHUNKS", None)
    if walk:
        env["PLR_SPATIAL_SPLIT_X"], env["PLR_SPATIAL_CHUNKS"] = str(walk[0]), str(walk[1])
    p = subprocess.run([sys.executable, CHILD, str(tw), str(th)], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("pass ")]
    assert len(lines) == 2, p.stdout
    return lines


@pytest.mark.gpu
@pytest.mark.parametrize("tw,th", [(1992, 437), (1000, 333)])
def test_gpu_every_walk_of_the_spatial_filter_writes_the_same_bytes(tw, th):
    ref = run(tw, th, None)  # the default walk: two columns of XCDs at 1992 texels, one at 1000
    for line in ref:  # (a filter that wrote nothing would also agree with itself)
        assert float(line.split()[4]) > 0.99 and float(line.split()[6]) > 0.9, line
    for walk in WALKS[1:]:
        assert run(tw, th, walk) == ref, "walk splitX %d chunks %d differs from the default walk at %d x %d" % (walk[0], walk[1], tw, th)
