"""Worker for tests/test_dist_gloo.py: the model-load collective of bench.py (rank 0 parses the
.april file and exports the packed blob, every other rank builds its model from the broadcast)
on the gloo backend with host tensors.  Launched by torch.distributed.run, one process per rank."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import april_asr_amd as A  # noqa: E402


def main():
    model_path, out_dir = sys.argv[1], sys.argv[2]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    if rank == 0:
        model = A.Model.load_host_only(model_path)
        blob = torch.from_numpy(model.export_blob())
        size = torch.tensor([blob.numel()], dtype=torch.int64)
    else:
        size = torch.zeros(1, dtype=torch.int64)
    dist.broadcast(size, 0)
    if rank != 0:
        blob = torch.empty(int(size.item()), dtype=torch.uint8)
    dist.broadcast(blob, 0)
    if rank != 0:
        model = A.Model.from_blob(blob.numpy(), init_gpu=False)
    # session sharding of the bench: rank r owns global sessions [r*B, (r+1)*B)
    B = 4
    mine = list(range(rank * B, (rank + 1) * B))
    again = model.export_blob()
    info = dict(rank=rank, name=model.get_name(), params=int(model.dims.param_count), vocab=int(model.dims.vocab),
                tokens=[model.token(i) for i in range(model.dims.vocab)], blob_bytes=int(again.size),
                blob_sum=int(np.frombuffer(again.tobytes(), np.uint8).astype(np.uint64).sum()), sessions=mine)
    with open(os.path.join(out_dir, "rank%d.json" % rank), "w") as f:
        json.dump(info, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
