"""Multi-process sharding + merge on CPU (gloo, world_size 2 and 3): the same DistributedRenderer
that drives RCCL on GPUs, with a stand-in backend whose "render" adds a known function of
(pixel, global sample index) - so the merged buffer can be checked exactly and shown to be
independent of the number of ranks."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from raytracing_amd.distributed import DistributedRenderer, shard_samples

W, H, TOTAL = 6, 5, 11


def test_shard_samples_partition():
    for total in (0, 1, 7, 256, 1000):
        for world in (1, 2, 3, 8):
            parts = [shard_samples(total, r, world) for r in range(world)]
            assert sum(c for _, c in parts) == total
            assert parts[0][0] == 0
            for (b0, c0), (b1, _) in zip(parts, parts[1:]):
                assert b1 == b0 + c0
            assert max(c for _, c in parts) - min(c for _, c in parts) <= 1
    with pytest.raises(ValueError):
        shard_samples(4, 2, 2)


def _value(sample_indices):
    x = np.arange(W)[:, None, None, None]
    y = np.arange(H)[None, :, None, None]
    c = np.arange(3)[None, None, :, None]
    s = np.asarray(sample_indices)[None, None, None, :]
    return ((x * 7 + y * 3 + c + 1) * (s + 1)).sum(axis=3).astype(np.float32)


class FakeBackend:
    """Adds f(pixel, sample) for each requested sample into the tensor it is pointed at."""

    def __init__(self, accum):
        self.accum = accum
        self.calls = []

    def render(self, width, height, sample_begin, sample_count, max_bounces, seed, accum_ptr=0, stream=0):
        assert accum_ptr == self.accum.data_ptr() and (width, height) == (W, H)
        self.calls.append((sample_begin, sample_count))
        self.accum += torch.from_numpy(_value(range(sample_begin, sample_begin + sample_count)))

    def synchronize(self, stream=0):
        pass


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    accum = torch.zeros((W, H, 3), dtype=torch.float32)
    be = FakeBackend(accum)
    dr = DistributedRenderer(be, accum)
    assert (dr.rank, dr.world) == (rank, world)
    merged = dr.render(W, H, TOTAL, 8, 1234, sample_offset=5)
    assert be.calls == [(5 + shard_samples(TOTAL, rank, world)[0], shard_samples(TOTAL, rank, world)[1])]
    np.save(os.path.join(out_dir, f"r{rank}.npy"), merged.numpy())
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world", [2, 3])
def test_gloo_merge_equals_single_rank(tmp_path, world):
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    want = _value(range(5, 5 + TOTAL))
    for r in range(world):
        got = np.load(tmp_path / f"r{r}.npy")
        assert np.array_equal(got, want)   # every rank holds the full sum; integers, so exact


class FakeCommBackend(FakeBackend):
    """A backend that brings its own communicator, as HipBackend does (rt_comm_init / rt_merge):
    here the "library" all-reduce is gloo's, so the hand-shake around it can be checked on CPU."""
    fail_on_rank = None
    fail_where = "comm_init"     # or "unique_id" (rank 0 cannot make the id) or "ready" (a rank cannot open the library)
    entered_init = False

    @staticmethod
    def comm_unique_id():
        if FakeCommBackend.fail_where == "unique_id" and FakeCommBackend.fail_on_rank == 0:
            raise RuntimeError("cannot open librccl.so")
        return bytes(range(128))

    def comm_ready(self):
        return not (FakeCommBackend.fail_where == "ready" and dist.get_rank() == FakeCommBackend.fail_on_rank)

    def comm_init(self, n_ranks, rank, unique_id):
        FakeCommBackend.entered_init = True
        if FakeCommBackend.fail_where == "comm_init" and rank == FakeCommBackend.fail_on_rank:
            raise RuntimeError("no RCCL here")
        assert unique_id == bytes(range(128))
        self.comm = (n_ranks, rank, unique_id)

    def merge(self, width, height, accum_ptr=0, stream=0):
        assert accum_ptr == self.accum.data_ptr() and (width, height) == (W, H)
        self.merged_by_library = True
        dist.all_reduce(self.accum)


def _comm_worker(rank, world, port, out_dir, fail_on_rank, fail_where="comm_init"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    FakeCommBackend.fail_on_rank = fail_on_rank
    FakeCommBackend.fail_where = fail_where
    accum = torch.zeros((W, H, 3), dtype=torch.float32)
    be = FakeCommBackend(accum)
    dr = DistributedRenderer(be, accum)
    ok = dr.use_library_merge()
    assert ok == (fail_on_rank is None)                      # all ranks or none
    if ok:
        assert be.comm == (world, rank, bytes(range(128)))   # rank 0's id reached every rank
    if fail_where != "comm_init":
        # a failure that is known BEFORE the communicator is made keeps every rank out of comm_init (inside RCCL it
        # is a collective: a rank entering it alone would wait for the others for ever)
        assert not FakeCommBackend.entered_init and dr.library_merge_error
    merged = dr.render(W, H, TOTAL, 8, 1234)
    assert getattr(be, "merged_by_library", False) == ok
    np.save(os.path.join(out_dir, f"r{rank}.npy"), merged.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("fail_on_rank", [None, 1])
def test_library_merge_handshake(tmp_path, fail_on_rank):
    """use_library_merge: the communicator id travels from rank 0 to all, the ranks agree on whether
    the library merges (a rank that cannot set RCCL up sends everyone to the torch.distributed path),
    and the merged sums are the same either way."""
    mp.spawn(_comm_worker, args=(2, _free_port(), str(tmp_path), fail_on_rank), nprocs=2, join=True)
    want = _value(range(0, TOTAL))
    for r in range(2):
        assert np.array_equal(np.load(tmp_path / f"r{r}.npy"), want)


@pytest.mark.parametrize("fail_on_rank,fail_where", [(0, "unique_id"), (1, "ready"), (0, "ready")])
def test_library_merge_handshake_failures_before_the_communicator(tmp_path, fail_on_rank, fail_where):
    """Rank 0 cannot make the communicator id (librccl.so does not open there), or some rank cannot open the
    library: the ranks still make the same collectives - rank 0 broadcasts None, readiness is gathered before
    anyone enters comm_init - so nobody hangs, all fall back to torch.distributed, and the sums are right."""
    mp.spawn(_comm_worker, args=(2, _free_port(), str(tmp_path), fail_on_rank, fail_where), nprocs=2, join=True)
    want = _value(range(0, TOTAL))
    for r in range(2):
        assert np.array_equal(np.load(tmp_path / f"r{r}.npy"), want)
