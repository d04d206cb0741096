"""Multi-GPU layout rehearsed on CPU: gloo, world_size 2."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from buzzdetect_amd import sharding


def test_round_robin_partition_is_exact():
    for world in (1, 2, 3, 8):
        seen = []
        for r in range(world):
            mine = sharding.shard_indices(1000, r, world)
            assert all(sharding.owner_of(i, world) == r for i in mine)
            seen += mine
        assert sorted(seen) == list(range(1000))
    assert len(sharding.shard_indices(1000, 0, 8)) == 125          # config 4: 125 files per rank
    with pytest.raises(ValueError):
        sharding.shard_indices(10, 2, 2)


def test_interleave_inverts_sharding():
    items = [torch.full((1,), float(i)) for i in range(11)]
    per_rank = [[items[i] for i in sharding.shard_indices(11, r, 3)] for r in range(3)]
    back = sharding.interleave_round_robin(per_rank)
    assert [int(t.item()) for t in back] == list(range(11))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        files = sharding.shard_indices(5, rank, world)                 # 5 "files", ragged over 2 ranks
        rows = [torch.full((3 + f, 13), float(f)) for f in files]      # file f yields 3+f windows
        local = torch.cat(rows, 0)
        got = sharding.gather_rows(local, dst=0)
        if rank == 0:
            assert got is not None and len(got) == world
            per_rank = []
            for r, t in enumerate(got):
                fs = sharding.shard_indices(5, r, world)
                sizes = [3 + f for f in fs]
                assert t.shape == (sum(sizes), 13)
                per_rank.append(list(torch.split(t, sizes)))
            ordered = sharding.interleave_round_robin(per_rank)
            assert [int(t[0, 0].item()) for t in ordered] == [0, 1, 2, 3, 4]
            assert [t.shape[0] for t in ordered] == [3, 4, 5, 6, 7]
            open(os.path.join(out_dir, "ok"), "w").write("ok")
        else:
            assert got is None
    finally:
        dist.destroy_process_group()


def test_gather_rows_world_size_2_gloo(tmp_path):
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok").exists()


def test_gather_rows_single_process_is_identity():
    t = torch.arange(26.0).reshape(2, 13)
    assert sharding.gather_rows(t)[0] is t


# ---------------------------------------------------------------------------------------------------
# strong scaling over a fixed set of recordings: round plan + one fixed-size gather per round
def test_plan_rounds_covers_every_batch_once():
    per_file = [1024, 1024, 1024, 678]
    for n_files, world in ((20, 8), (20, 3), (8, 8), (3, 8), (1, 2), (0, 4), (1000, 8)):
        rounds = sharding.plan_rounds(n_files, per_file, world)
        seen = sorted((u.file, u.batch) for rnd in rounds for us in rnd.units for u in us)
        assert seen == [(f, b) for f in range(n_files) for b in range(4)]
        for rnd in rounds:
            assert len(rnd.units) == world
            for us in rnd.units:
                offs = sharding.unit_offsets(us)
                assert all(o % sharding.ROW_ALIGN == 0 for o in offs) and offs[-1] <= rnd.rows
        full = n_files // world
        for g in range(full):                       # whole recordings: recording g*world + r on rank r
            for r in range(world):
                assert [(u.file, u.batch) for u in rounds[g].units[r]] == [(g * world + r, b) for b in range(4)]
            assert rounds[g].rows == 3752           # 3750 rows, blocks padded to 4-row boundaries
    assert len(sharding.plan_rounds(1000, per_file, 8)) == 125          # config 4: 125 rounds, no remainder
    # the remainder is dealt batch by batch, so no rank idles through a whole recording
    last = sharding.plan_rounds(20, per_file, 8)[-1]
    assert [len(us) for us in last.units] == [2] * 8


def _worker_rounds(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        per_file, n_files = [5, 5, 3], 5                    # 2 full rounds + a remainder dealt per batch
        rounds = sharding.plan_rounds(n_files, per_file, world)
        gathered = []
        for rnd in rounds:
            block = torch.full((rnd.rows, 13), -1.0)
            mine = rnd.units[rank]
            for u, at in zip(mine, sharding.unit_offsets(mine)):    # "predict": row value = file*100 + batch*10 + row
                block[at:at + u.windows] = (u.file * 100 + u.batch * 10 + torch.arange(u.windows, dtype=torch.float32))[:, None]
            g, work = sharding.gather_round(block, rnd.rows, dst=0)
            assert work is None
            if rank == 0:
                assert g.shape == (world, rnd.rows, 13)
                gathered.append(g)
            else:
                assert g is None
        if rank == 0:
            files = sharding.assemble_files(rounds, gathered, n_files)
            for f, t in enumerate(files):
                want = torch.cat([f * 100 + b * 10 + torch.arange(w, dtype=torch.float32) for b, w in enumerate(per_file)])
                assert t.shape == (13, 13) and torch.equal(t[:, 0], want) and torch.equal(t[:, 12], want)
            open(os.path.join(out_dir, "ok"), "w").write("ok")
        with pytest.raises(ValueError):
            sharding.gather_round(torch.zeros(3, 13), 4)
    finally:
        dist.destroy_process_group()


def test_gather_once_per_round_world_size_2_gloo(tmp_path):
    mp.spawn(_worker_rounds, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok").exists()


def test_bench_self_launch_builds_a_torchrun_command(monkeypatch):
    """`python bench.py --gpus N` from a plain shell must start N child ranks itself (and never exec)."""
    import importlib.util
    import subprocess
    import sys
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(os.path.dirname(__file__), "..", "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    calls = {}

    class Done:
        returncode = 7

    def fake_run(cmd, env=None, **kw):
        calls["cmd"], calls["env"] = cmd, env
        return Done()

    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setenv("BD_BENCH_REHEARSAL", "1")
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--steps", "3"])
    assert bench.main() == 7                               # the children's exit code is handed back
    cmd = calls["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=2" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "2", "--steps", "3"]
    assert calls["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    # without the rehearsal switch a box with fewer GPUs than ranks is refused before anything is started - and the
    # parent finds that out from the KFD topology, not through HIP (it must not touch a GPU before its children exist)
    monkeypatch.delenv("BD_BENCH_REHEARSAL")
    monkeypatch.setattr(bench, "visible_gpus", lambda: 1)
    calls.clear()
    assert bench.main() == 2 and not calls
    # a container that leases ONE GPU of an eight-GPU host can see all eight in the topology (round 5: eight ranks were started
    # on a one-GPU box): the runtime's own count, asked of a short-lived child, has the last word
    monkeypatch.setattr(bench, "visible_gpus", lambda: 8)
    monkeypatch.setattr(bench, "runtime_gpu_count", lambda: 1)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "1"])
    assert bench.main() == 2 and not calls
    monkeypatch.setattr(bench, "runtime_gpu_count", lambda: 8)
    assert bench.main() == 7 and "--nproc-per-node=8" in calls["cmd"]
    calls.clear()
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--steps", "3"])
    src = open(os.path.join(os.path.dirname(__file__), "..", "bench.py")).read()
    launch = src[src.index("def self_launch"):src.index("# ---", src.index("def self_launch"))]
    assert "import torch" not in launch and "device_count" not in launch


def _worker_round_gatherer(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rows_per_file = [5, 9, 2, 7, 4]                       # five recordings over two ranks: rounds (0,1) (2,3) (4,-)
        got = {}
        g = sharding.RoundGatherer(rows_per_file, 13, lambda f, rows: got.__setitem__(f, rows.copy()))
        assert (g.n_rounds, g.round_rows(0), g.round_rows(1), g.round_rows(2)) == (3, 12, 8, 4)
        mine = sharding.shard_indices(5, rank, world)
        for f in reversed(mine):                              # recordings finish out of order; gathers still go round by round
            g.submit(f, np.full((rows_per_file[f], 13), 10.0 * f) + np.arange(rows_per_file[f])[:, None])
        g.finish()
        if rank == 0:
            assert sorted(got) == [0, 1, 2, 3, 4]
            for f, rows in got.items():
                assert rows.shape == (rows_per_file[f], 13)
                assert np.array_equal(rows[:, 5], 10.0 * f + np.arange(rows_per_file[f]))
            open(os.path.join(out_dir, "ok"), "w").write("ok")
        else:
            assert not got
        with pytest.raises(ValueError):
            g.submit(rank ^ 1, np.zeros((rows_per_file[rank ^ 1], 13)))       # not this rank's recording
    finally:
        dist.destroy_process_group()


def test_round_gatherer_world_size_2_gloo(tmp_path):
    import numpy  # noqa: F401
    mp.spawn(_worker_round_gatherer, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok").exists()


def _worker_round_gatherer_failure(rank, world, port, out_dir):
    """a rank that cannot deliver a recording must still issue that round's collective."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rows_per_file = [5, 9, 2, 7, 4]
        got = {}
        g = sharding.RoundGatherer(rows_per_file, 13, lambda f, rows: got.__setitem__(f, rows.copy()))
        for f in sharding.shard_indices(5, rank, world):
            if f == 3:                                        # rank 1 could not read its second recording
                g.submit_failed(f, "unreadable")
            else:
                g.submit(f, np.full((rows_per_file[f], 13), float(f)))
        failed = g.finish()                                   # returns on BOTH ranks: nobody is left in a collective
        if rank == 0:
            assert failed == [3] and sorted(got) == [0, 1, 2, 4]
            assert all(np.all(rows == float(f)) for f, rows in got.items())
            open(os.path.join(out_dir, "ok"), "w").write("ok")
        else:
            assert failed == [] and g.failed_local == {3: "unreadable"}
    finally:
        dist.destroy_process_group()


def test_round_gatherer_undelivered_recording_does_not_hang(tmp_path):
    mp.spawn(_worker_round_gatherer_failure, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok").exists()


def _write_wav(path, seconds, rate=16000):
    import wave
    with wave.open(path, "wb") as w:
        w.setnchannels(1)
        w.setsampwidth(2)
        w.setframerate(rate)
        w.writeframes((np.arange(int(seconds * rate)) % 251).astype("<i2").tobytes())


def _worker_plan(rank, world, port, root):
    """on a resume rank 0 skips finished recordings; a rank with another view of the output folder must not
    derive a different plan (other block sizes, other owners): rank 0 plans, everybody receives that plan."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from buzzdetect_amd import analyze as A
        audio = os.path.join(root, "audio")
        todo = [(p, A.build_ident(p, audio)) for p in A.search_audio(audio)]
        assert len(todo) == 3
        dir_out = os.path.join(root, f"out{rank}")            # rank 0: recording "b" complete; rank 1: nothing there
        own = A.gather_plan(todo, dir_out, 15360, 96, 9.6)
        assert [i for _, i, _, _ in own] == (["a", "c"] if rank == 0 else ["a", "b", "c"])
        plan = A.broadcast_plan(own if rank == 0 else None, rank)
        assert [i for _, i, _, _ in plan] == ["a", "c"]
        assert [c for _, _, _, c in plan] == [[10, 10, 2], [10, 3]]      # 20.5 s and 12 s in 9.6 s chunks (H4: tail windows)
        assert [ch for _, _, ch, _ in plan][1] == [(0.0, 9.6), (9.6, 12.0)]
        if rank == 0:
            open(os.path.join(root, "ok"), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_gather_plan_is_rank_zeros_plan_on_resume(tmp_path):
    audio = tmp_path / "audio"
    audio.mkdir()
    for name, sec in (("a", 20.5), ("b", 15.0), ("c", 12.0)):
        _write_wav(str(audio / f"{name}.wav"), sec)
    (tmp_path / "out0").mkdir()
    (tmp_path / "out1").mkdir()
    (tmp_path / "out0" / "b_buzzdetect.csv").write_text("start,activation_ins_buzz\n0.0,1.0\n")
    mp.spawn(_worker_plan, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok").exists()


def _worker_round_gatherer_sink_raises(rank, world, port, out_dir):
    """rank 0's ``on_file`` (the CSV write) raises in round 1.  The round was gathered, so it is consumed; the
    later rounds are still gathered (rank 1 is never left waiting in a collective), the recording is listed as not
    delivered, and finish() re-raises the sink's error on rank 0 after the last round."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rows_per_file = [5, 9, 2, 7, 4, 3, 6]                 # rounds (0,1) (2,3) (4,5) (6,-)
        got = {}

        def on_file(f, rows):
            if f == 2:
                raise OSError(28, "No space left on device")
            got[f] = rows.copy()

        g = sharding.RoundGatherer(rows_per_file, 13, on_file)
        for f in sharding.shard_indices(len(rows_per_file), rank, world):
            g.submit(f, np.full((rows_per_file[f], 13), float(f)))
        if rank == 0:
            with pytest.raises(RuntimeError, match="writing recording 2 failed") as e:
                g.finish()
            assert isinstance(e.value.__cause__, OSError)
            assert g._next == g.n_rounds == 4                 # every round was gathered
            assert sorted(got) == [0, 1, 3, 4, 5, 6] and g.failed == [2] and list(g.sink_errors) == [2]
            open(os.path.join(out_dir, "ok"), "w").write("ok")
        else:
            assert g.finish() == []                           # returns: nothing is blocked
    finally:
        dist.destroy_process_group()


def test_round_gatherer_sink_error_does_not_strand_other_ranks(tmp_path):
    mp.spawn(_worker_round_gatherer_sink_raises, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok").exists()


def _worker_config4_world8(rank, world, port, out_dir):
    """BASELINE config 4's shape on CPU: 1000 recordings over 8 ranks = 125 rounds, the plan every rank derives for
    itself, one gather per round, recordings finishing out of order, and one delivery that fails on rank 5."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n_files = 1000
        rng = np.random.default_rng(4)
        rows_per_file = [int(v) for v in rng.integers(1, 40, n_files)]       # the same list on every rank
        per_file = (1024, 1024, 1024, 678)                    # config 2's batches of a 1 h recording (bench.py's plan)
        rounds = sharding.plan_rounds(n_files, per_file, world)
        assert len(rounds) == 125 and all(len(r.units) == world and r.rows == 3752 for r in rounds)   # 3750 windows, blocks aligned to 4 rows
        for gi in (0, 57, 124):                               # whole rounds: recording g * 8 + r on rank r
            assert [[(u.file, u.batch, u.windows) for u in units] for units in rounds[gi].units] == \
                [[(gi * 8 + r, b, w) for b, w in enumerate(per_file)] for r in range(8)]
        got = {}
        g = sharding.RoundGatherer(rows_per_file, 13, lambda f, rows: got.__setitem__(f, rows.copy()))
        assert g.n_rounds == 125
        mine = sharding.shard_indices(n_files, rank, world)
        assert mine == list(range(rank, n_files, world))
        order = list(mine)
        np.random.default_rng(rank).shuffle(order[:40])       # the first forty of a rank finish in any order
        order = [int(f) for f in order]
        lost = 8 * 60 + 5                                     # rank 5's recording of round 60
        for f in order:
            if f == lost:
                g.submit_failed(f, "unreadable")
            else:
                g.submit(f, np.full((rows_per_file[f], 13), float(f)) + np.arange(13)[None, :])
        failed = g.finish()
        if rank == 0:
            assert failed == [lost] and sorted(got) == [f for f in range(n_files) if f != lost]
            for f in (0, 1, 7, 8, 484, 486, 999):
                assert got[f].shape == (rows_per_file[f], 13) and np.array_equal(got[f][0], f + np.arange(13.0))
            open(os.path.join(out_dir, "ok"), "w").write("ok")
        else:
            assert failed == [] and not got
            assert g.failed_local == ({lost: "unreadable"} if rank == 5 else {})
    finally:
        dist.destroy_process_group()


def test_config4_shape_world_size_8_gloo(tmp_path):
    """RoundGatherer + plan_rounds at world size 8 with 1000 recordings (CPU only, no GPU minutes)."""
    mp.spawn(_worker_config4_world8, args=(8, _free_port(), str(tmp_path)), nprocs=8, join=True)
    assert (tmp_path / "ok").exists()
