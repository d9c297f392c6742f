"""Progressive rendering + checkpoints of the accumulation buffer: host logic on CPU.
The stand-in scene adds the CPU oracle's samples (test infrastructure), as in test_distributed_cpu.py."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import ray_tracing_amd as rt
from test_distributed_cpu import H, SPP, W, OracleBackedScene, _free_port


def test_checkpoint_roundtrip_and_damage(tmp_path):
    acc = torch.rand((5, 7, 3), dtype=torch.float32)
    path = str(tmp_path / "a.rtck")
    rt.save_checkpoint(path, acc, {"scene": "x", "seed": 3, "max_depth": 50, "spp_done": 12})
    arr, meta = rt.load_checkpoint(path)
    assert np.array_equal(arr, acc.numpy()) and meta["spp_done"] == 12 and (meta["width"], meta["height"]) == (7, 5)
    raw = bytearray(open(path, "rb").read())
    raw[-5] ^= 0x40
    open(path, "wb").write(bytes(raw))
    with pytest.raises(ValueError):
        rt.load_checkpoint(path)
    open(path, "wb").write(b"not a checkpoint at all")
    with pytest.raises(ValueError):
        rt.load_checkpoint(path)


def test_passes_and_resume_equal_one_go(tmp_path):
    sc = OracleBackedScene("cornell_box")
    whole = rt.render(sc, W, H, SPP).numpy()
    seen = []
    acc, done = rt.render_progressive(sc, W, H, SPP, 4, spp_per_launch=3, on_pass=lambda d, t: seen.append(d))
    assert done == SPP and seen == [4, 8, 10]
    assert np.allclose(acc.numpy(), whole, rtol=1e-5, atol=1e-6)
    # stop after 6 samples, continue to 10 from the file
    path = str(tmp_path / "c.rtck")
    part, d6 = rt.render_progressive(sc, W, H, 6, 3, checkpoint=path, scene_name="cornell_box")
    arr, meta = rt.load_checkpoint(path)
    assert d6 == 6 and meta["spp_done"] == 6 and np.array_equal(arr, part.numpy())
    full, d10 = rt.render_progressive(sc, W, H, SPP, 3, checkpoint=path, resume=True, scene_name="cornell_box")
    assert d10 == SPP and rt.load_checkpoint(path)[1]["spp_done"] == SPP
    assert np.allclose(full.numpy(), whole, rtol=1e-5, atol=1e-6)
    # a checkpoint of another render is refused
    with pytest.raises(ValueError):
        rt.render_progressive(sc, W, H, SPP, 3, checkpoint=path, resume=True, scene_name="cornell_box", seed=7)
    with pytest.raises(ValueError):
        rt.render_progressive(sc, W, H, SPP, 3, checkpoint=path, resume=True, scene_name="cornell_smoke")


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sc = OracleBackedScene("cornell_box")
        path = os.path.join(out_dir, "two.rtck")
        ex = rt.Exchange(rank, world, dist, "cpu")  # (falls back to torch.distributed on CPU: tests/test_distributed_cpu.py)

        def f32_sum(buf):  # the exchange of the C-ABI sums float32 buffers and nothing else: so must every caller
            assert buf.dtype == torch.float32 and buf.is_contiguous()
            return ex.all_reduce(buf)

        # first 5 samples here ...
        rt.render_progressive(sc, W, H, 5, 5, rank=rank, world_size=world, all_reduce=f32_sum, checkpoint=path, scene_name="cornell_box")
        dist.barrier()
        # ... the rest continued from the file, in passes of 3 split over the two ranks
        acc, done = rt.render_progressive(sc, W, H, SPP, 3, rank=rank, world_size=world, all_reduce=f32_sum, checkpoint=path,
                                          resume=True, scene_name="cornell_box")
        assert done == SPP
        # ranks that disagree on where to resume (a checkpoint only rank 0 can see) are told so instead of hanging in unequal passes
        lonely = os.path.join(out_dir, "only_rank0.rtck")
        if rank == 0:
            rt.save_checkpoint(lonely, acc, dict(scene="cornell_box", seed=rt.DEFAULT_SEED, max_depth=50, spp_done=4, precision="f32"))
        dist.barrier()
        try:
            rt.render_progressive(sc, W, H, SPP, 3, rank=rank, world_size=world, all_reduce=f32_sum,
                                  checkpoint=lonely if rank == 0 else lonely + ".absent", resume=True, scene_name="cornell_box")
            raise AssertionError("ranks with different resume points must be refused")
        except ValueError as e:
            assert "disagree" in str(e)
        np.save(os.path.join(out_dir, f"acc_{rank}.npy"), acc.numpy())
    finally:
        dist.destroy_process_group()


def test_two_rank_progressive_resume(tmp_path):
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    a0, a1 = np.load(tmp_path / "acc_0.npy"), np.load(tmp_path / "acc_1.npy")
    assert np.array_equal(a0, a1)
    whole = rt.render(OracleBackedScene("cornell_box"), W, H, SPP).numpy()
    assert np.allclose(a0, whole, rtol=1e-5, atol=1e-6)
    arr, meta = rt.load_checkpoint(str(tmp_path / "two.rtck"))
    assert meta["spp_done"] == SPP and np.allclose(arr, whole, rtol=1e-5, atol=1e-6)
