"""First contact with the N > 1 RCCL route on whatever the box has (launched with torchrun by tests/test_gpu_parity.py, gloo rendezvous).

Every rank builds an `ActionExchange` and asks for route "auto": the engine-side communicator (`ph_comm_init` = ncclCommInitRank with
nranks = WORLD_SIZE, the unique id through the process group's store) is attempted FIRST.
  * ranks on distinct GPUs: the communicator forms, `ph_all_gather_i32` must reproduce torch.distributed's gather, and the timed
    comparison picks a route;
  * ranks sharing one GPU (the 1-GPU test boxes): RCCL refuses the duplicate device -- every rank must come back from
    ncclCommInitRank with an error (not hang), agree that the native route is out (`rccl_verified` False everywhere), and land on the
    SAME fallback route, which must then carry a real exchange.
Either way: no rank may hang, all ranks print the same route, and the joint action is right.  The reference call site this replaces:
pantheonrl/common/multiagentenv.py:149-170."""
import json
import os
import sys

import numpy as np
import torch as th
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pantheonrl_amd import _native as nat  # noqa: E402
from pantheonrl_amd import dist as pdist  # noqa: E402

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
n_dev = th.cuda.device_count()
dev_index = int(os.environ.get("LOCAL_RANK", "0")) % n_dev
th.cuda.set_device(dev_index)
dist.init_process_group("gloo")
device = th.device("cuda", dev_index)
os.environ["PANTHEON_RCCL_WITH_GLOO"] = "1"
E, A_LOCAL, T = 128, 1, 8
ctx = nat.Context(dev_index)
epoch = th.zeros(1, dtype=th.int64, device=device)
ex = pdist.ActionExchange(A_LOCAL, E, device)
route = ex.setup(ctx, epoch, T, route="auto")
shared = ex.ranks_on_device
log = dict(ex.route_log)
# every rank reached here: nobody hung in ncclCommInitRank.  The verdicts are collective: identical on all ranks
verdicts = [None] * world
dist.all_gather_object(verdicts, (route, bool(log.get("rccl_verified")), shared))
assert all(v == verdicts[0] for v in verdicts), verdicts
if shared > 1:
    assert not log.get("rccl_verified"), "two ranks on one device formed an RCCL communicator?"
    assert route in ("p2p", "torch"), route
else:
    assert log.get("rccl_verified"), "ranks on distinct GPUs: the engine-side RCCL all-gather must verify against torch.distributed"
    assert route in ("rccl", "p2p"), route
# a real exchange over the chosen route
stream = th.cuda.current_stream(device)
ctx.set_stream(stream.cuda_stream)
for t in range(T):
    ex.local.copy_(th.arange(A_LOCAL * E, dtype=th.int32, device=device).view(A_LOCAL, E) + 1000 * rank + 7 * t)
    if route == "p2p":
        joint = ex.p2p_step(t, in_band=True).clone()
    else:
        joint = ex.gather_inplace().clone()
    th.cuda.synchronize(device)
    want = np.concatenate([np.arange(A_LOCAL * E, dtype=np.int32).reshape(A_LOCAL, E) + 1000 * r + 7 * t for r in range(world)])
    assert np.array_equal(joint.cpu().numpy(), want), (rank, t, route)
    dist.barrier()
assert ex.p2p_timeouts() == 0
if rank == 0:
    print(json.dumps({"route": route, "ranks_on_device": shared, "rccl_verified": bool(log.get("rccl_verified")), "log": {
        k: (v if isinstance(v, (int, float, str, bool)) else str(v)) for k, v in log.items()}}))
dist.barrier()
dist.destroy_process_group()
