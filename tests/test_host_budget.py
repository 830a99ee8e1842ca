"""The host-thread budget of a rank (VERDICT r04 #6a): one process per GPU on one node, the ranks share the host --
tm_host_threads divides the container's CPU grant by $LOCAL_WORLD_SIZE.  CPU only (the library loads without a GPU)."""
import os
import subprocess
import sys

from tests import helpers

SNIPPET = "from enzymm_amd import _native; print(_native.host_threads())"


def _threads(env_extra):
    env = {k: v for k, v in os.environ.items() if k not in ("ENZYMM_AMD_HOST_THREADS", "LOCAL_WORLD_SIZE")}
    env.update(env_extra)
    out = subprocess.run([sys.executable, "-c", SNIPPET], cwd=str(helpers.ROOT), env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-2000:]
    return int(out.stdout.strip().splitlines()[-1])


def test_host_threads_follow_the_ranks_of_the_node():
    alone = _threads({})
    assert 1 <= alone <= 16
    assert _threads({"LOCAL_WORLD_SIZE": "1"}) == alone
    assert _threads({"LOCAL_WORLD_SIZE": "2"}) == max(1, alone // 2)
    assert _threads({"LOCAL_WORLD_SIZE": "8"}) == max(1, alone // 8)
    assert _threads({"LOCAL_WORLD_SIZE": "8", "ENZYMM_AMD_HOST_THREADS": "5"}) == 5  # an explicit setting wins
