"""Multi-GPU layout of the analyze path: one process per GPU, recordings dealt round-robin,
one gather of per-window logits to rank 0.

The reference has no distributed mode (threads over one queue, src/analyze.py:218-253); its unit of
independent work is the chunk (src/pipeline/assignments.py:35-41).  Windows never depend on another
chunk's audio (hazard H1: a chunk's last 240 samples are zero padding), so sharding whole files — or
whole chunks of one long file — changes no result.  Nothing but ``[W,13]`` f32 logits (52 B/window)
crosses ranks; weights are replicated.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_indices(n_items: int, rank: int, world_size: int) -> List[int]:
    """Item ``i`` belongs to rank ``i mod world_size`` (file-level round-robin)."""
    if not 0 <= rank < world_size:
        raise ValueError(f"rank {rank} outside world of {world_size}")
    return list(range(rank, n_items, world_size))


def owner_of(item: int, world_size: int) -> int:
    return item % world_size


def gather_rows(local: torch.Tensor, dst: int = 0, group=None) -> Optional[List[torch.Tensor]]:
    """Gather ``[rows_r, C]`` tensors of differing ``rows_r`` to ``dst``.

    One small all_gather of the row counts, then one padded gather of the payload (RCCL ``gather``
    needs equal shapes).  Returns the per-rank tensors on ``dst`` (trimmed), ``None`` elsewhere.
    """
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return [local]
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    counts = torch.zeros(world, dtype=torch.int64, device=local.device)
    mine = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
    dist.all_gather_into_tensor(counts, mine, group=group)
    rows = int(counts.max().item())
    cols = local.shape[1]
    padded = local
    if local.shape[0] != rows:
        padded = torch.zeros((rows, cols), dtype=local.dtype, device=local.device)
        padded[: local.shape[0]] = local
    padded = padded.contiguous()
    if rank == dst:
        bucket = [torch.empty((rows, cols), dtype=local.dtype, device=local.device) for _ in range(world)]
        dist.gather(padded, bucket, dst=dst, group=group)
        return [b[: int(c)] for b, c in zip(bucket, counts.tolist())]
    dist.gather(padded, None, dst=dst, group=group)
    return None


def interleave_round_robin(per_rank: Sequence[Sequence[torch.Tensor]]) -> List[torch.Tensor]:
    """Undo ``shard_indices``: per_rank[r][j] is item ``r + j * world`` -> items in original order."""
    world = len(per_rank)
    total = sum(len(p) for p in per_rank)
    out: List[torch.Tensor] = []
    for i in range(total):
        out.append(per_rank[i % world][i // world])
    return out


# ---------------------------------------------------------------------------------------------------
# Strong scaling over a fixed set of recordings (config 4): recordings go to ranks round-robin, one
# recording per rank per ROUND; what is left when the count is not a multiple of the world size is dealt
# at chunk (batch) level so that no rank idles through a whole recording.  Every round ends in ONE
# gather of equal-sized row blocks whose size is known on the host: no count exchange, no host sync.

@dataclass(frozen=True)
class Unit:
    """One launch set: batch ``batch`` (``windows`` rows) of recording ``file``."""
    file: int
    batch: int
    windows: int


@dataclass(frozen=True)
class Round:
    units: Tuple[Tuple[Unit, ...], ...]     # units[rank] = what that rank runs this round, in order
    rows: int                               # rows every rank contributes to the gather (padded)


ROW_ALIGN = 4      # a unit's rows start on a multiple of 4 rows: 4 x 13 f32 = 208 B keeps every block 16-byte aligned


def unit_offsets(units: Sequence[Unit]) -> List[int]:
    """Row offset of each unit inside its rank's block of a round (+ the block's used length as last item)."""
    at, out = 0, []
    for u in units:
        out.append(at)
        at += (u.windows + ROW_ALIGN - 1) // ROW_ALIGN * ROW_ALIGN
    return out + [at]


def plan_rounds(n_files: int, batch_windows: Sequence[int], world_size: int) -> List[Round]:
    """``batch_windows`` = window counts of one recording's batches (config 2: 1024, 1024, 1024, 678)."""
    if world_size < 1 or n_files < 0 or not batch_windows:
        raise ValueError("plan_rounds: bad argument")
    per_file = tuple(int(w) for w in batch_windows)
    rounds: List[Round] = []
    full = n_files // world_size
    for g in range(full):
        units = tuple(tuple(Unit(g * world_size + r, b, w) for b, w in enumerate(per_file))
                      for r in range(world_size))
        rounds.append(Round(units, unit_offsets(units[0])[-1]))
    left = [Unit(f, b, w) for f in range(full * world_size, n_files) for b, w in enumerate(per_file)]
    if left:
        units = tuple(tuple(left[r::world_size]) for r in range(world_size))
        rounds.append(Round(units, max(unit_offsets(us)[-1] for us in units)))
    return rounds


def gather_round(local: torch.Tensor, rows: int, dst: int = 0, out: Optional[torch.Tensor] = None, group=None,
                 async_op: bool = False):
    """Gather every rank's ``[rows, C]`` block (``local`` may hold fewer rows: the rest of the block is
    whatever ``local``'s buffer holds — callers hand in a ``rows``-sized buffer) to ``dst``.
    Returns ``(out [world, rows, C] on dst / None elsewhere, work handle or None)``."""
    if local.shape[0] != rows:
        raise ValueError(f"gather_round: block has {local.shape[0]} rows, the round was planned with {rows}")
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local.unsqueeze(0), None
    world = dist.get_world_size(group)
    local = local.contiguous()
    if dist.get_rank(group) == dst:
        if out is None:
            out = torch.empty((world,) + tuple(local.shape), dtype=local.dtype, device=local.device)
        work = dist.gather(local, [out[r] for r in range(world)], dst=dst, group=group, async_op=async_op)
        return out, work
    work = dist.gather(local, None, dst=dst, group=group, async_op=async_op)
    return None, work


def assemble_files(rounds: Sequence[Round], gathered: Sequence[torch.Tensor], n_files: int) -> List[torch.Tensor]:
    """On the destination rank: per-recording ``[W, C]`` rows in recording order from the per-round
    ``[world, rows, C]`` gathers."""
    parts: List[List[Tuple[int, torch.Tensor]]] = [[] for _ in range(n_files)]
    for rnd, g in zip(rounds, gathered):
        for r, units in enumerate(rnd.units):
            for u, at in zip(units, unit_offsets(units)):
                parts[u.file].append((u.batch, g[r, at:at + u.windows]))
    return [torch.cat([t for _, t in sorted(p, key=lambda bt: bt[0])]) if p else torch.empty(0) for p in parts]


class RoundGatherer:
    """Config 4's collective inside a running analysis: recordings complete in any order on every rank, but the
    gathers - one per ROUND of ``world`` recordings (recording i belongs to rank i mod world, round i // world) - are
    collectives and must be issued in the same order everywhere.  ``submit`` hands in the rows of a finished recording;
    rounds are gathered as soon as they and all earlier rounds are ready; on ``dst`` every gathered recording is passed
    to ``on_file(file_index, rows)``.  Block sizes come from ``rows_per_file``, which must be the SAME list on every rank
    (rank 0 plans, the plan is broadcast: ``analyze._analyze_gathered``): no count exchange per round.

    A rank that cannot deliver a recording it owns (unreadable file, a failure in its pipeline) still has to take part in
    that round's collective, or every other rank blocks in it forever: ``submit_failed`` contributes the block with its
    status row set (the ``ROW_ALIGN`` rows appended to every block; row 0, column 0: 0 = rows valid, 1 = not delivered);
    ``dst`` does not pass such a recording on and lists it in ``failed``.  The same goes for ``on_file`` itself: if it
    raises on ``dst``, the recording is listed in ``failed`` (reason in ``sink_errors``), every later round is still
    gathered, and ``finish`` re-raises after the last one - no rank is ever left waiting in a collective."""

    def __init__(self, rows_per_file: Sequence[int], n_cols: int, on_file, device="cpu", dst: int = 0, group=None):
        self.rows_per_file = [int(r) for r in rows_per_file]
        self.n_files, self.n_cols = len(self.rows_per_file), int(n_cols)
        self.on_file, self.device, self.dst, self.group = on_file, torch.device(device), dst, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.n_rounds = (self.n_files + self.world - 1) // self.world
        self._ready = {}                      # round -> rows of this rank's recording in it (None: not delivered)
        self._next = 0
        self.submitted = set()                # recordings of this rank handed in so far (either way)
        self.failed: List[int] = []           # on dst: recordings some rank could not deliver
        self.failed_local = {}                # recording -> reason, on the rank that owns it
        self.sink_errors = {}                 # on dst: recording -> exception its ``on_file`` raised

    def round_rows(self, g: int) -> int:
        files = range(g * self.world, min((g + 1) * self.world, self.n_files))
        m = max(self.rows_per_file[f] for f in files)
        return (m + ROW_ALIGN - 1) // ROW_ALIGN * ROW_ALIGN

    def submit(self, file_index: int, rows) -> None:
        if owner_of(file_index, self.world) != self.rank:
            raise ValueError(f"recording {file_index} does not belong to rank {self.rank}")
        rows = torch.as_tensor(rows, dtype=torch.float32)
        if tuple(rows.shape) != (self.rows_per_file[file_index], self.n_cols):
            raise ValueError(f"recording {file_index}: got {tuple(rows.shape)} rows, planned "
                             f"({self.rows_per_file[file_index]}, {self.n_cols})")
        self._ready[file_index // self.world] = rows
        self.submitted.add(file_index)
        self._pump()

    def submit_failed(self, file_index: int, reason: str = "") -> None:
        """This rank owns ``file_index`` and will not deliver its rows: take part in the round with the status row set."""
        if owner_of(file_index, self.world) != self.rank:
            raise ValueError(f"recording {file_index} does not belong to rank {self.rank}")
        self._ready[file_index // self.world] = None
        self.submitted.add(file_index)
        self.failed_local[file_index] = reason
        self._pump()

    def _mine(self, g: int) -> Optional[int]:
        f = g * self.world + self.rank
        return f if f < self.n_files else None

    def _pump(self) -> None:
        while self._next < self.n_rounds:
            g = self._next
            mine = self._mine(g)
            if mine is not None and g not in self._ready:
                return                        # this rank's recording of the round is still being analysed
            n = self.round_rows(g)
            block = torch.zeros((n + ROW_ALIGN, self.n_cols), dtype=torch.float32, device=self.device)
            if mine is not None:
                rows = self._ready.pop(g)
                if rows is None:
                    block[n, 0] = 1.0
                else:
                    block[: rows.shape[0]] = rows.to(self.device)
            out, _ = gather_round(block, n + ROW_ALIGN, dst=self.dst, group=self.group)
            # The round's collective is done: the round is consumed whatever happens to its rows now.  If ``on_file``
            # (rank 0's CSV write) raises - disk full, permissions - the recording is listed as not delivered with the
            # reason, the pump goes on issuing the later rounds (the other ranks are already on their way into them and
            # would otherwise block until the process-group timeout), and ``finish`` re-raises the first such error
            # AFTER the last round.
            self._next += 1
            if self.rank == self.dst:
                host = out.cpu()
                for r in range(self.world):
                    f = g * self.world + r
                    if f < self.n_files:
                        if float(host[r, n, 0]) != 0.0:
                            self.failed.append(f)
                        else:
                            try:
                                self.on_file(f, host[r, : self.rows_per_file[f]].numpy())
                            except Exception as exc:      # noqa: BLE001 - kept, re-raised by finish()
                                self.failed.append(f)
                                self.sink_errors[f] = exc

    def finish(self) -> List[int]:
        """Every local recording has been submitted (``submit`` or ``submit_failed``): run the remaining rounds (a rank
        without a recording in the last round contributes an empty block).  On ``dst`` returns the recordings that were
        not delivered."""
        self._pump()
        if self._next != self.n_rounds:
            raise RuntimeError(f"rank {self.rank}: rounds {self._next}..{self.n_rounds - 1} never became ready")
        if self.sink_errors:
            first = min(self.sink_errors)
            raise RuntimeError(f"rank {self.rank}: writing recording {first} failed ({len(self.sink_errors)} recording(s) in "
                               f"all, every round was still gathered): {self.sink_errors[first]!r}") from self.sink_errors[first]
        return list(self.failed)
