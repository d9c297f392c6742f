"""The N > 1 path on CPU: world_size-2 `gloo`.  spp sharding + ONE all-reduce of the float
accumulation buffers (SURVEY §8 e).  The launcher's render() is driven with a stand-in scene whose
render_into is the CPU oracle (test infrastructure), so what is exercised is exactly the host logic
the GPU ranks run: shard arithmetic, disjoint keyed sample ranges, the collective, resolve."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import ray_tracing_amd as rt
from oracle_lib import OracleScene

W, H, SPP = 24, 20, 10


class OracleBackedScene:
    """Duck-types DeviceScene for renderer.render(): adds the oracle's samples into a CPU tensor."""
    torch_device = torch.device("cpu")

    def __init__(self, name):
        self.oc = OracleScene(name, W / H)

    def render_into(self, accum, spp_begin, spp_count, seed=rt.DEFAULT_SEED, max_depth=50, precision="f64", stats=None):
        acc64, _ = self.oc.render(accum.shape[1], accum.shape[0], spp_begin, spp_count, seed=seed, max_depth=max_depth, threads=2)
        accum += torch.from_numpy(acc64).to(torch.float32)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sc = OracleBackedScene("cornell_box")
        # the launchers' exchange object (renderer.Exchange): without a GPU the C-ABI route cannot come up on any rank, ALL ranks
        # must agree on the torch.distributed route (the selection logic is what runs here), and say why
        ex = rt.Exchange(rank, world, dist, "cpu")
        assert ex.route == "torch" and ex.comm is None and "torch.distributed used" in ex.note and "no GPU" in ex.note
        assert rt.Exchange(rank, world, dist, "cpu", prefer="torch").note is None  # asked for: nothing to explain

        def f32_sum(buf):  # what rt_allreduce_accum can do: float32, in place
            assert buf.dtype == torch.float32 and buf.is_contiguous()
            return ex.all_reduce(buf)

        acc = rt.render(sc, W, H, SPP, rank=rank, world_size=world, all_reduce=f32_sum, spp_per_launch=3)
        st = torch.tensor([rank + 1, 10, 0, 0], dtype=torch.int64)
        ex.all_reduce_stats(st)
        assert st.tolist() == [sum(range(1, world + 1)), 10 * world, 0, 0]
        ex.close()
        np.save(os.path.join(out_dir, f"acc_{rank}.npy"), acc.numpy())
    finally:
        dist.destroy_process_group()


def test_spp_shard_partition():
    for spp in (1, 7, 64, 1024, 4096):
        for n in (1, 2, 3, 4, 8):
            parts = [rt.spp_shard(spp, r, n) for r in range(n)]
            assert parts[0][0] == 0 and sum(c for _, c in parts) == spp
            for (b0, c0), (b1, _) in zip(parts, parts[1:]):
                assert b0 + c0 == b1                      # contiguous, disjoint
            assert max(c for _, c in parts) - min(c for _, c in parts) <= 1


def test_two_rank_gloo_render_equals_single_rank(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a0, a1 = np.load(tmp_path / "acc_0.npy"), np.load(tmp_path / "acc_1.npy")
    assert np.array_equal(a0, a1)                         # all-reduce: every rank holds the sum
    single = rt.render(OracleBackedScene("cornell_box"), W, H, SPP).numpy()
    assert np.allclose(a0, single, rtol=1e-5, atol=1e-6)  # union of the shards == the unsharded render
    assert single.sum() > 0


def test_bench_step_ranges_are_disjoint():
    # bench.py: step i of rank r covers [(i*world + r)*S, +S)
    S, K = 256, 4
    for world in (1, 2, 4, 8):
        starts = sorted((i * world + r) * S for i in range(K) for r in range(world))
        assert starts == list(range(0, S * K * world, S))
