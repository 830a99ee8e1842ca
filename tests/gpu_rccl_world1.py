"""RCCL leg of the hit-table gather on ONE GPU: a world-size-1 "nccl" group, rows sent from the device table. Not a test
(tests/test_gpu_round2.py::test_rccl_gather_world_size_1 is)."""
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
from tests import helpers  # noqa: E402
from enzymm_amd import _native, distributed, pyjess, synthetic  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
idx, cut = helpers.default_selection()
jess = pyjess.Jess(helpers.product_library_templates())
packed = pyjess._packed(jess._templates, jess._device)
batch = pyjess.make_batch(synthetic.structures(64), jess._device)
flags = _native.TM_FLAG_BEST_MATCH | _native.TM_FLAG_IGNORE_CHAIN
res = _native.query(packed.native, batch, idx, 2.0, cut, cut, 10000, flags, retry=False)
n = res.hit_count()
got = distributed.gather_result(res, n, 1000, device="cuda:0", force_collective=True)
want = res.hits()
want["structure"] += np.uint32(1000)
assert n > 20 and got.tobytes() == want.tobytes(), (n, len(got))
# the pass table of bench.py's N > 1 loop over the same backend: rows appended on the device, one gather
table = distributed.DeviceHitTable(8192, device="cuda:0")
table.append(res, 1000)
res.wait()
again = table.gather(force_collective=True)
assert again.tobytes() == want.tobytes(), len(again)
print("RCCL world-size-1 gather ok:", n, "rows")
dist.destroy_process_group()
