"""Worker for the multi-rank replication-harness test (launched by tests/test_distributed_cpu.py): `try_count`
replications over the ranks of a gloo group, each fitted by an oracle-backed VIRT on CPU."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.oracle_backend import OracleBackend          # noqa: E402
from vipsy_amd import harness, vi                       # noqa: E402


def main():
    folder, out_path, try_count = sys.argv[1], sys.argv[2], int(sys.argv[3])
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1:
        torch.distributed.init_process_group(backend="gloo", rank=rank, world_size=world)
    out = harness.multiprocess_article_test_load_data_util(
        "2pl", 40, 6, 1, vi_class=vi.VIRT, try_count=try_count, folder=folder, device=torch.device("cpu"),
        vi_class_kwargs={"backend": OracleBackend(), "observed_lists": False},
        vi_fit_kwargs={"optim": vi.Adam({"lr": 5e-2}), "max_iter": 5})
    with open(out_path + ".%d" % rank, "w") as f:
        json.dump(out, f)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
