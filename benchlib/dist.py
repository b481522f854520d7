"""The default workload (BASELINE config 3; config 5 at N = 8) at any GPU
count: one process per GPU, torch.distributed over RCCL (backend "nccl"),
rows partitioned by contiguous ranges, x replicated, the fragments of y
exchanged every step.

`run_rank` is what every rank executes; it is a sequence of small steps on a
`RankJob`: build the shard, agree on the kernel, check rows against the host
generator, time K steps of the PLAIN arrangement (each rank's kernel, then
ONE in-place all-gather / grouped send-recv of y), print the line.  That line
goes out at once -- complete, flagged `provisional`, naming `legs_pending` --
and only then the optional legs run: the exchange alone and its
alternatives, the overlapped arrangements (`value_best` beside `value` when
one wins), the fixed-problem reading (`config.strong`), the nnz-balanced
partition of the nlpkkt160-shaped matrix, the library's own multi-GPU path.
Each leg runs inside `benchlib.legs.LegRunner.run`: local work first, the
ranks agree on failure before any collective of the leg, a leg that outlives
its limit ends the run WITH the line (watchdog), and the final line repeats
the main measurement plus whatever the legs added."""
import json
import os
import sys
import time

from .common import *  # noqa: F401,F403
from .common import (FAMILIES, MATRIX_SEED, METRIC, ROOT, ROWS_PER_GPU, X_SEED,
                     cgroup_cpu_stat, check_rows, config4_file, host_cpus,
                     kernel_source_ident, measured_traffic, roofline_dict,
                     secondary_roofline, stat_delta, strong_one_gpu,
                     strong_speedup_of, workload_name)
from .cpu import cpu_baseline
from .legs import LEG_BUDGET_S, LegRunner
from .single import extra_measurements, single_matrix_bench, window_variants

# an alternative arrangement replaces nothing: it is reported as `value_best`
# when it beats the plain one by this margin (5 + 5 timed steps decide; a
# closer call is noise -- ADVICE r05)
ARRANGEMENT_MARGIN = 0.97


class RankJob:
    """state of one rank's run of the default workload"""

    def __init__(self, args, omp_team):
        import numpy as np
        self.args, self.omp_team = args, omp_team
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if self.world > 1 or args.force_exchange:
            # ranks: torch.distributed over RCCL.  torch FIRST: the binding
            # then shares the ROCm runtime torch's wheel has mapped (one
            # runtime per process; `config.rocm` says which, and flags a
            # major.minor mismatch with the build)
            import torch
            import torch.distributed as dist
        else:
            # one GPU: no torch in the process -- device memory, stream and
            # events come from the library's C-ABI (benchlib.devshim) and the
            # library runs on the ROCm runtime it was built for
            from . import devshim as torch
            dist = None
        import spmv_scpa_amd as S
        from spmv_scpa_amd import dist as D
        self.np, self.torch, self.dist, self.S, self.D = np, torch, dist, S, D
        self.stat0 = cgroup_cpu_stat()
        self.t_start = time.time()
        if not torch.cuda.is_available() or S.device_count() == 0:
            raise SystemExit("bench.py needs an MI355X: no GPU visible "
                             "(there is no CPU fallback)")
        if args.backend == "gloo":  # rehearsal: ranks may share a card
            self.local_rank %= max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(self.local_rank)
        S.set_device(self.local_rank)
        self.dev = torch.device("cuda", self.local_rank)
        self.use_dist = self.world > 1 or args.force_exchange
        self.legs = None  # LegRunner, once the process group exists

    # ------------------------------------------------------------ plumbing
    def init_process_group(self):
        if self.use_dist:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29531")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
            # last resort only: the legs' own watchdog (benchlib.legs) ends a
            # run whose collective never completes long before this does
            import datetime
            limit = datetime.timedelta(seconds=300)
            if self.args.backend == "gloo":
                self.dist.init_process_group("gloo", timeout=limit)
            else:
                self.dist.init_process_group("nccl", device_id=self.dev,
                                             timeout=limit)
        self.legs = LegRunner(
            self.rank, self.world, self.t_start, self.dist,
            self.dev if self.args.backend == "nccl" else None, self.use_dist)
        if self.use_dist:
            # from here on a run that gets stuck BEFORE its line exists (a
            # first collective that never completes) ends with a record that
            # names the phase, not with the process group's timeout
            from .common import METRIC as metric
            self.legs.failure_record = lambda: {
                "metric": metric, "unit": "GFLOP/s", "steps": self.args.steps,
                "warmup": self.args.warmup, "higher_is_better": True}
            if os.environ.get("SPMV_BENCH_MAIN_LIMIT"):  # tests
                self.legs.main_limit_s = float(
                    os.environ["SPMV_BENCH_MAIN_LIMIT"])
            self.legs.start_watchdog()

    def sync(self):
        self.torch.cuda.synchronize()

    def barrier(self):
        self.sync()
        if self.use_dist:
            self.dist.barrier()
        self.sync()

    def stream(self):
        return self.torch.cuda.current_stream().cuda_stream

    def max_over_ranks(self, value):
        if not self.use_dist:
            return float(value)
        t = self.torch.tensor([float(value)], dtype=self.torch.float64,
                              device=self.dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def time_steps(self, sh, n):
        """barrier-bracketed wall time of n steps, max over ranks, ms/step"""
        self.barrier()
        t0 = time.perf_counter()
        for _ in range(n):
            sh.step()
        self.barrier()
        return self.max_over_ranks(time.perf_counter() - t0) * 1e3 / n

    # --------------------------------------------------------- the workload
    def define_workload(self):
        a, world = self.args, self.world
        L = a.shards_per_gpu
        if a.strong:
            if 8 % world:
                raise SystemExit("--strong needs 1, 2, 4 or 8 GPUs")
            L = 8 // world
        self.L, self.Mshard, self.K = L, a.rows_per_gpu, a.nnz_row
        self.kind = FAMILIES[a.family]
        Mglob = a.rows_per_gpu * L * world
        self.Mglob = self.Nglob = Mglob
        self.W = a.window if a.window > 0 else 2 * self.Nglob  # >= 2N: anywhere
        # the row ranges: equal counts, or near-equal ENTRY counts cut from
        # the generator's row lengths (csr.h partition_synth_rows_nnz: the
        # multi-GPU form of the reference's partition_csr_rows)
        if a.partition == "nnz":
            if L != 1:
                raise SystemExit("--partition nnz holds one shard per GPU")
            self.starts = [int(v) for v in self.S.partition_synth_rows_nnz(
                self.kind, Mglob, self.Nglob, self.K, self.W, MATRIX_SEED,
                world)]
        else:
            self.starts = self.D.even_row_partition(Mglob, world)
        self.row0 = self.starts[self.rank]
        self.Mloc = self.starts[self.rank + 1] - self.row0
        self.ragged = a.partition == "nnz" and any(
            self.starts[k] != k * (Mglob // world) for k in range(world + 1))
        if self.ragged:
            self.Mshard = self.Mloc

    def alloc_vectors(self):
        t = self.torch
        self.x = t.empty(self.Nglob, dtype=t.float64, device=self.dev)
        self.y = t.zeros(self.Mglob, dtype=t.float64, device=self.dev)
        self.S.dev_fill_synth(self.x.data_ptr(), self.Nglob, X_SEED, 0,
                              self.stream())
        self.sync()

    def build_shards(self, count, rows, first_row=None, ncols=None, w=None,
                     kind=None, col_major=None):
        """`count` logical shards of `rows` rows starting at `first_row`
        -> (handles, true entries, stored slots)"""
        a, S = self.args, self.S
        first_row = self.row0 if first_row is None else first_row
        ncols = self.Nglob if ncols is None else ncols
        w = self.W if w is None else w
        kind = self.kind if kind is None else kind
        out, nnz, stored = [], 0, 0
        for j in range(count):
            dA = S.CsrDevice.generate(kind, rows, ncols, self.K, w,
                                      first_row + j * rows, MATRIX_SEED)
            nnz += dA.NZ
            if a.format == "hll":
                if col_major is None:
                    col_major = True if a.kernel in (-1, 4) else \
                        S.HLL_KERNEL_COL_MAJOR[a.kernel]
                m = dA.to_hll(col_major)
                stored += m.slots
                dA.release()
            else:
                m = dA
                stored += dA.NZ
            out.append(m)
        return out, nnz, stored

    def labels(self):
        S = self.S
        return ((S.HLL_KERNEL_LABELS, "hll_") if self.args.format == "hll"
                else (S.CSR_KERNEL_LABELS, "csr_"))

    def pick_kernel(self):
        """--blocked-pin, --kernel, or the measured selector on the first
        shard; with ranks, rank 0's pick -- kernel id and, for the blocked
        path, schedule and tile height -- is broadcast and built everywhere
        (the pick decides how the exchange is arranged: every rank must issue
        the same collectives)"""
        a, S, D = self.args, self.S, self.D
        labels, _ = self.labels()
        mat = self.mats[0]
        self.tuned, self.t_tune = None, None
        if a.blocked_pin and self.use_dist:
            raise SystemExit("--blocked-pin pins ONE rank's layout for the "
                             "profiling passes: single GPU only")
        self.pinned = bool(a.blocked_pin)
        if self.pinned:
            kernel = (S.HLL_KERNEL_PANELS if a.format == "hll"
                      else S.CSR_KERNEL_PANELS)
            for m in self.mats:
                m.build_panels_pinned(a.blocked_pin)
        elif a.kernel >= 0:
            kernel = a.kernel
        else:
            t0 = time.time()
            kernel, self.tuned = mat.autotune(
                self.x.data_ptr(), self.y.data_ptr() + 8 * self.row0)
            self.t_tune = time.time() - t0
            if self.use_dist:
                mine = D.Pick(kernel, mat.panels_schedule(),
                              mat.panels_tile_rows() or 0)
                pick = D.agree_on_pick(self.dist, mine, self.dev)
                kernel = pick.kernel
                if labels[kernel] == "tile_panels" and not pick.same_build(mine):
                    mat.build_panels(0, pick.schedule, pick.tile_rows)
        self.kernel = kernel
        self.blocked = labels[kernel] == "tile_panels"
        if self.blocked and mat.panels_info() is None:
            mat.build_panels(0)
        self.sweep = self.blocked and mat.panels_schedule() == "sweep"

    def exchange_settings(self):
        a, D = self.args, self.D
        # the PLAIN arrangement is the line's: the whole shard's kernel, then
        # ONE exchange.  Row chunks (the overlapped "staged" pipeline) only on
        # request (--chunks k); without it they are an optional leg whose
        # result is reported beside `value` (alternative_arrangement)
        chunks = a.chunks if a.chunks > 0 else 1
        if self.blocked or self.L > 1 or (a.format == "csr"
                                          and self.kernel == 4):
            chunks = 1  # the blocked path runs whole shards only; with logical
            #             shards the shard is the unit of overlap; the CSR
            #             stream kernel's row-block table covers the whole
            #             shard (a row sub-range would fall back to sub-wave)
        halo = 0
        if a.exchange == "halo":
            halo = a.halo_rows
            if halo <= 0:
                if a.family == "banded":
                    halo = self.K
                elif a.window > 0 and a.family != "stencil":
                    halo = (self.W + 1) // 2
                else:
                    raise SystemExit("--exchange halo needs --halo-rows (or a "
                                     "column window)")
            halo = -(-halo // D.HACK) * D.HACK
        self.chunks, self.halo = chunks, halo

    def make_sharded(self, ms, rows_total=None, xx=None, yy=None, chunks=None):
        a = self.args
        rows_total = self.Mshard * self.L if rows_total is None else rows_total
        mode = "halo" if self.halo else (
            a.ragged_exchange if self.ragged else None)
        return self.D.ShardedSpmv(
            ms if len(ms) > 1 else ms[0], self.kernel, self.rank, self.world,
            None if self.ragged else rows_total,
            self.x if xx is None else xx, self.y if yy is None else yy,
            waves_per_block=a.waves,
            chunks=self.chunks if chunks is None else chunks,
            force_exchange=a.force_exchange, mode=mode, halo_rows=self.halo,
            starts=self.starts if self.ragged else None, backend=self.torch)

    # ---------------------------------------- the overlapped arrangements
    def alternative_kind(self):
        """which overlapped arrangement this run could try beside the plain
        one (None: none applies) -- decided from facts every rank shares"""
        a, D = self.args, self.D
        if not self.use_dist or self.L != 1 or self.halo or self.ragged:
            return None
        if self.sweep:
            return "sweep_split" if self.Mshard % (2 * D.HACK) == 0 else None
        nsplit = 2 if a.force_exchange and self.world == 1 else 4
        if self.blocked:
            return "logical_shards" if self.Mshard % (nsplit * D.HACK) == 0 \
                else None
        if a.chunks > 0 or (a.format == "csr" and self.kernel == 4):
            return None  # chunks were asked for / the stream kernel runs whole
        return "row_chunks" if self.Mshard % (4 * D.HACK) == 0 else None

    def build_alternative(self):
        """LOCAL half of the arrangement leg -> (kind, shards or None,
        ShardedSpmv, nnz, slots, text):
        sweep_split     two logical shards, each swept by a grid that leaves
                        --reserve-cus compute units to RCCL, the all-gather of
                        the first half beside the sweep of the second (the
                        sweep launch is persistent and wants its whole grid
                        resident, so by default the exchange FOLLOWS it)
        logical_shards  blocked chain / steps: the rank's rows as 4 logical
                        shards, the all-gather of shard c under the kernel of
                        c+1 (at 8 GPUs the exchange, 560 MB in per GPU, is
                        longer than the kernel of a matrix with locality)
        row_chunks      direct kernels: 4 row chunks, chunk c all-gathered
                        (chunk-major staging) under the kernel of c+1"""
        a, kind = self.args, self.alternative_kind()
        if kind == "sweep_split":
            alt, nnz, slots = self.build_shards(2, self.Mshard // 2)
            try:
                for m in alt:
                    m.build_panels(0, "sweep", reserve_cus=a.reserve_cus)
                sh = self.make_sharded(alt)
            except Exception:
                for m in alt:
                    m.release()
                raise
            return (kind, alt, sh, nnz, slots,
                    "2 logical shards on %d fewer CUs, the all-gather of the "
                    "first beside the sweep of the second" % a.reserve_cus)
        if kind == "logical_shards":
            n = 2 if a.force_exchange and self.world == 1 else 4
            alt, nnz, slots = self.build_shards(n, self.Mshard // n)
            try:
                for m in alt:  # the tuned schedule and tile height
                    m.build_panels_like(self.mats[0])
                sh = self.make_sharded(alt)
            except Exception:
                for m in alt:
                    m.release()
                raise
            return (kind, alt, sh, nnz, slots,
                    "%d logical shards, the all-gather of shard c under the "
                    "kernel of c+1" % n)
        sh = self.make_sharded(self.mats, chunks=4)
        return (kind, None, sh, self.nnz_local, self.slots,
                "4 row chunks, the all-gather of chunk c (staged) under the "
                "kernel of c+1")

    def time_alternative(self, kind, alt, sh_alt, nnz, slots, text):
        """COLLECTIVE half: 5 + 5 steps, max over ranks (every rank sees the
        same two numbers, hence takes the same branch); when the alternative
        wins by the margin, exactly K steps of it are timed like the main
        measurement -> `value_best`.  The plain arrangement stays `value`."""
        for s_ in (self.sharded, sh_alt):
            s_.step()
        t_plain = self.time_steps(self.sharded, 5)
        t_alt = self.time_steps(sh_alt, 5)
        rec = {"plain_ms_per_step": round(t_plain, 5),
               "alternative": text, "alternative_kind": kind,
               "alternative_ms_per_step": round(t_alt, 5),
               "margin": ARRANGEMENT_MARGIN, "winner": "plain"}
        if t_alt < ARRANGEMENT_MARGIN * t_plain:
            wall, kern, _ = self.timed_steps(sh_alt)
            ms = self.max_over_ranks(wall) * 1e3 / self.args.steps
            rec.update(winner=kind, best_ms_per_step=round(ms, 5),
                       best_kernel_ms_avg=round(float(self.np.mean(kern)), 5),
                       best_exchange=sh_alt.mode)
        return rec

    def release_alternative(self, built=None):
        if built and built[1]:
            for m in built[1]:
                m.release()

    # ------------------------------------------------------ check and timing
    def check_result(self):
        """rows of y recomputed from the workload definition by the product's
        HOST generator: own rows, and -- N > 1 -- rows every OTHER rank
        computed (the exchange)"""
        np, torch = self.np, self.torch
        self.sharded.step()
        self.sync()
        rng = np.random.default_rng(1234 + self.rank)
        rows = np.concatenate([[0, self.Mloc - 1],
                               rng.integers(0, self.Mloc, 256)]) + self.row0
        if self.world > 1:
            extra = []
            for r in range(self.world):
                if r == self.rank:
                    continue
                lo, hi = self.starts[r], self.starts[r + 1]
                if self.halo:  # only what lies within the halo of my rows
                    _, recv = self.sharded.halo_slices(r)
                    if recv:
                        extra.append(np.array([recv[0], recv[1] - 1]))
                elif hi > lo:
                    extra.append(np.array([lo, (lo + hi) // 2, hi - 1]))
            rows = np.concatenate([rows] + extra)
        got = self.y[torch.as_tensor(rows, device=self.dev)].cpu().numpy()
        # a rank whose rows are wrong must not leave the others waiting in
        # the next barrier: the ranks agree, then ALL of them stop
        err = None
        try:
            self.checked = check_rows(self.S, self.kind, self.Nglob, self.K,
                                      self.W, got, rows)
        except SystemExit as e:
            err = e
        bad = self.legs.ranks_where(err is not None)
        if bad:
            raise SystemExit("parity check failed on rank(s) %s%s" % (
                bad, ": %s" % (err,) if err is not None else ""))

    def timed_steps(self, sh=None):
        """K steps between barrier + synchronize on both sides; per-step
        events on the launch stream and host timestamps after each enqueue.
        -> (wall seconds, kernel ms per step, host seconds between enqueues)"""
        torch, n = self.torch, self.args.steps
        sh = self.sharded if sh is None else sh
        ev = [(torch.cuda.Event(enable_timing=True),
               torch.cuda.Event(enable_timing=True)) for _ in range(n)]
        stamps = [0.0] * (n + 1)
        self.barrier()
        t0 = time.perf_counter()
        stamps[0] = t0
        for k in range(n):
            sh.step(events=ev[k])
            stamps[k + 1] = time.perf_counter()
        self.barrier()
        wall = time.perf_counter() - t0
        return (wall, [a.elapsed_time(b) for a, b in ev],
                [stamps[k + 1] - stamps[k] for k in range(n)])

    def measure(self):
        np, a = self.np, self.args
        for _ in range(a.warmup):
            self.sharded.step()
        stat2 = cgroup_cpu_stat()
        elapsed, kern_ms, enq = self.timed_steps()
        self.stat3 = cgroup_cpu_stat()
        attempts = [{"ms_per_step": round(elapsed * 1e3 / a.steps, 5),
                     "kernel_ms_avg": round(float(np.mean(kern_ms)), 5),
                     "max_enqueue_ms": round(max(enq) * 1e3, 4),
                     "throttled": stat_delta(stat2, self.stat3)}]
        # At N = 1 a step is one launch, so wall / step must equal the
        # event-timed kernel; a gap means the HOST stalled inside the timed
        # region (round 2: CFS throttling, 4.4 ms/step).  Then -- once, in the
        # same process -- K steps are timed again AS A DIAGNOSTIC
        # (host.retry_ms_per_step, top-level "host_stall_retry": true):
        # `value` always is the FIRST attempt, exactly K timed steps, never a
        # best-of-two (lines must stay comparable across rounds).
        gap = elapsed * 1e3 / a.steps - float(np.mean(kern_ms))
        self.retried = False
        if (self.world == 1 and not a.force_exchange
                and gap > 0.05 * float(np.mean(kern_ms))):
            e2, k2, q2 = self.timed_steps()
            stat4 = cgroup_cpu_stat()
            self.retried = True
            attempts.append({"ms_per_step": round(e2 * 1e3 / a.steps, 5),
                             "kernel_ms_avg": round(float(np.mean(k2)), 5),
                             "max_enqueue_ms": round(max(q2) * 1e3, 4),
                             "throttled": stat_delta(self.stat3, stat4),
                             "diagnostic_only": True})
        self.elapsed, self.kern_ms, self.enq = elapsed, kern_ms, enq
        self.attempts = attempts

    def exchange_alone(self, sh=None, iters=10):
        """the collectives of one step without the kernels, ms (max over ranks)"""
        sh = self.sharded if sh is None else sh
        for _ in range(2):
            sh.exchange_only()
        self.barrier()
        t1 = time.perf_counter()
        for _ in range(iters):
            sh.exchange_only()
        self.barrier()
        return self.max_over_ranks(time.perf_counter() - t1) * 1e3 / iters

    def build_exchange_alternatives(self):
        """LOCAL: the same fragments by the other ways the library can move
        them (dist.RaggedExchange on this run's row ranges): grouped send /
        recv, one broadcast per rank, all-gather padded to the longest
        fragment + compaction -- so that one scaling run prices every
        exchange"""
        D = self.D
        if self.halo:
            return {}
        return {mode: D.ShardedSpmv(
            self.mats[0], self.kernel, self.rank, self.world, None, self.x,
            self.y, chunks=1, mode=mode,
            force_exchange=self.args.force_exchange,
            starts=[int(v) for v in self.starts] if self.ragged
            else _never_even(self.starts),
            compute=lambda a, b, out=None: None)
            for mode in ("p2p", "bcast", "padded")}

    def time_exchange_alternatives(self, shs):
        """COLLECTIVE: each alternative's exchange alone, ms"""
        return {mode: round(self.exchange_alone(sh, 5), 5)
                for mode, sh in shs.items()} or None

    def reduce_over_ranks(self):
        torch, dist = self.torch, self.dist
        if self.use_dist:
            t = torch.tensor([self.elapsed, float(self.nnz_local)],
                             dtype=torch.float64, device=self.dev)
            tm = t[:1].clone()
            dist.all_reduce(tm, op=dist.ReduceOp.MAX)
            ts = t[1:].clone()
            dist.all_reduce(ts, op=dist.ReduceOp.SUM)  # ragged / kkt: nnz differs
            self.elapsed, self.nnz_global = float(tm.item()), int(ts.item())
            per = torch.zeros(self.world, dtype=torch.float64, device=self.dev)
            mine = torch.tensor([float(self.nnz_local)], dtype=torch.float64,
                                device=self.dev)
            if self.args.backend == "gloo":
                host = torch.zeros(self.world, dtype=torch.float64)
                dist.all_gather_into_tensor(host, mine.cpu())
                per = host
            else:
                dist.all_gather_into_tensor(per, mine)
            self.nnz_per_rank = [int(v) for v in per.tolist()]
        else:
            self.nnz_global = self.nnz_local
            self.nnz_per_rank = [int(self.nnz_local)]
        self.ms_per_step = self.elapsed * 1e3 / self.args.steps
        self.value = 2.0 * self.nnz_global / (self.ms_per_step * 1e6)

    def release_everything(self):
        """free this rank's HBM (before the native child takes the devices)"""
        for m in getattr(self, "mats", []):
            m.release()
        self.mats = []
        self.sharded = None
        self.x = self.y = None
        self.torch.cuda.empty_cache()
        self.barrier()


def _never_even(starts):
    """`starts` as a list ShardedSpmv treats as ragged even when the ranges
    happen to be equal (the alternatives are timed on the even partition too)"""
    return [int(v) for v in starts]


def run_rank(args, argv, omp_team):
    """one rank of `bench.py --gpus N` (N >= 1)"""
    if args.config != 3:
        return run_matrix_rank(args, argv, omp_team)
    job = RankJob(args, omp_team)
    S, np, torch, dist = job.S, job.np, job.torch, job.dist
    job.init_process_group()
    legs = job.legs
    job.define_workload()

    # ---- build the shard(s) in HBM (device-side generator + converter) ----
    t_setup = time.time()
    legs.phase("generate + convert the shard in HBM")
    job.alloc_vectors()
    job.mats, job.nnz_local, job.slots = job.build_shards(job.L, job.Mshard)
    legs.phase("kernel selector + agreement on rank 0's pick (first "
               "collective: a broadcast)")
    job.pick_kernel()
    if job.blocked:
        for m in job.mats[1:]:  # the tuned shard's schedule and tile height
            if m.panels_info() is None:
                m.build_panels_like(job.mats[0])
    labels, prefix = job.labels()
    kname = prefix + labels[job.kernel]
    job.sync()
    job.exchange_settings()
    job.sharded = job.make_sharded(job.mats)
    mat = job.mats[0]
    pinfo = mat.panels_info() if job.blocked else None
    # kernel launches per step and GPU; the sweep schedule's phase counters
    # are zeroed by a hipMemsetAsync ahead of its one launch (panels.hip)
    launches = (pinfo["steps"] if pinfo else 1) * job.L
    memsets = job.L if job.sweep else 0
    # per step and GPU; one launch per logical shard.  Priced for
    # the kernel that runs: the blocked copy of an HLL handle stores no
    # padding (spmv_hll_kernel_bytes); same number when the format pads nothing
    alg_bytes = sum(m.kernel_bytes(job.kernel) for m in job.mats)
    job.sync()
    t_setup = time.time() - t_setup

    # ---- the main measurement: the PLAIN arrangement, exactly K steps ----
    legs.phase("first step + result check (first exchange of y)")
    job.check_result()
    stat1 = cgroup_cpu_stat()
    legs.phase("warm-up + K timed steps")
    job.measure()
    kern_ms = job.kern_ms
    # what joined, on which cards, and every rank's own kernel time
    legs.phase("describe the job + reductions over the ranks")
    rccl = per_rank = None
    if job.use_dist:
        rccl, per_rank = describe_job(S, torch, dist, job.dev, job.local_rank,
                                      job.world, args.backend, kern_ms)
    job.reduce_over_ranks()
    legs.phase(None)

    a, world, L, Mshard = args, job.world, job.L, job.Mshard
    # everything the line needs from the device is read NOW: the line must be
    # printable from the watchdog thread while the main thread sits in a leg
    sched_now = mat.panels_schedule() if job.blocked else None
    blocked_layout = mat.panels_describe() if job.blocked else None
    workload = workload_name(a.family, a.format, job.Mglob // world * 1,
                             job.Nglob, job.Mglob, job.K, a.window, job.W, L,
                             Mshard)
    traffic, why = (measured_traffic(workload, kname, sched_now) if world == 1
                    else (None, "single-GPU profiles only"))
    roof = roofline_dict(alg_bytes, kern_ms, kname, job.nnz_local, traffic, why)
    if roof["traffic_layout"] and roof["traffic_layout"] == blocked_layout:
        roof["traffic_layout"] = "same as config.blocked_layout"
    if per_rank:  # rank 0's events above; every rank's mean here
        roof["kernel_ms_per_rank"] = [round(v, 5) for v in per_rank]
        roof["kernel_ms_min_rank"] = round(min(per_rank), 5)
        roof["kernel_ms_max_rank"] = round(max(per_rank), 5)
    if world == 1 and job.sweep:  # the schedule for rows that reach beyond an L2
        roof["secondary"] = secondary_roofline(workload, kname,
                                               float(np.mean(kern_ms)),
                                               sched_now)
    ms_per_step = job.ms_per_step
    out = {
        "metric": METRIC,
        "value": round(job.value, 2),
        "unit": "GFLOP/s",
        "n_gpus": world,
        "steps": a.steps,
        "warmup": a.warmup,
        "ms_per_step": round(ms_per_step, 5),
        "higher_is_better": True,
        "scaling": "strong" if a.strong else "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "backend": ("gloo REHEARSAL (ranks share GPUs, host-staged "
                        "exchange: timings are not measurements)"
                        if a.backend == "gloo" else "nccl (RCCL)")
            if job.use_dist else None,
            "workload": workload,
            "kernel": kname,
            "kernel_choice": "pinned layout (--blocked-pin)" if job.pinned
            else "autotuned (spmv_%s_autotune)" % a.format
            if job.tuned is not None else "fixed by --kernel",
            # host seconds the selector took; its phase log when that is > 1 s
            "tune_s": round(job.t_tune, 2) if job.t_tune is not None else None,
            "tune_log": (mat.tune_log() or "").splitlines()
            if job.t_tune is not None and job.t_tune > 1.0 and job.L == 1
            else None,
            "blocked_schedule": sched_now,
            "blocked_layout": blocked_layout,
            # what --blocked-pin takes to run this layout again
            "blocked_pin": mat.panels_pin() if job.blocked else None,
            # y bitwise reproducible from launch to launch?  (the direct
            # kernels always; the blocked path when its copy was built
            # deterministic -- the default on sweep layouts)
            "deterministic": (not job.blocked) or
            "deterministic" in (blocked_layout or ""),
            "kernel_source": kernel_source_ident(kname),
            "kernel_launches_per_step": launches,
            # sweep schedule: its phase counters are zeroed on the stream
            # ahead of every launch (hipMemsetAsync, panels.hip)
            "memsets_per_step": memsets,
            "rows_per_gpu": job.Mloc if job.ragged else Mshard * L,
            "logical_shards_per_gpu": L,
            "nnz_per_row": job.K, "nnz_global": job.nnz_global,
            "stored_slots_per_gpu": job.slots,
            "partition": ("nnz-balanced contiguous row ranges (32-aligned; "
                          "reference csr.c:218-276), %s"
                          % ("ragged fragments" if job.ragged else
                             "which for these rows ARE the equal row counts")
                          if a.partition == "nnz" else
                          "contiguous row ranges of equal row counts")
            + ", x replicated, y exchanged over RCCL" if world > 1
            else "single GPU",
            "row_starts": job.starts if world > 1 and job.ragged else None,
            "nnz_per_rank": job.nnz_per_rank if world > 1 else None,
            "chunks": job.chunks, "exchange": job.sharded.mode,
            # the arrangement `value` was measured with; the overlapped ones
            # are an optional leg (config.arrangements, value_best)
            "exchange_arrangement": (
                "plain: every rank's kernel, then ONE exchange of y"
                if job.use_dist and job.chunks == 1 and L == 1 else
                "as asked for: %d row chunks / %d logical shards per GPU, the "
                "exchange of one under the kernel of the next"
                % (job.chunks, L) if job.use_dist else None),
            "arrangements": None,
            "exchange_ms_alone": None,
            "exchange_alternatives_ms": None,
            "rccl": rccl,
            "halo_rows": job.halo or None,
            "rows_per_s": round(job.Mglob / (ms_per_step * 1e-3), 1),
            "strong": None,
            "family_variants": None,
            "partition_kkt": None,
            "rocm": S.rocm_runtime_report(),
        },
        "roofline": roof,
        "host": {
            "host_gap_ms": round(ms_per_step - float(np.mean(kern_ms)), 5),
            "max_enqueue_ms": round(max(job.enq) * 1e3, 4),
            "timing_attempts": job.attempts,
            "retry_ms_per_step": job.attempts[1]["ms_per_step"]
            if job.retried else None,
            "omp_team": job.omp_team,
            "cpu_quota": host_cpus()[1],
            # CFS periods / throttled periods of this cgroup: over the result
            # check, and over the whole run up to the end of the timed steps
            "cfs_check": stat_delta(job.stat0, stat1),
            "cfs_total": stat_delta(job.stat0, job.stat3),
        },
        "setup_s": round(t_setup, 2),
        "phases_s": dict(legs.phases),  # the main measurement, by phase
        "rows_checked": job.checked,
    }
    if job.retried:
        out["host_stall_retry"] = True
    if out["config"]["rocm"].get("mismatch"):
        out["rocm_mismatch"] = True
    single = world == 1 and L == 1 and not a.force_exchange

    # ---- which optional legs this run will attempt, in order ----
    plan = []
    if job.use_dist:
        plan += ["exchange_alone", "exchange_alternatives"]
        if job.alternative_kind() and not a.no_arrangement_choice:
            plan.append("arrangement")
    # (--force-exchange at one rank runs them too: the same legs over 1-rank
    # RCCL collectives -- the nccl backend's code paths on a 1-GPU box)
    want_strong = (job.use_dist and not a.strong and not a.no_strong_leg
                   and 8 % world == 0 and a.family == "random"
                   and a.window <= 0 and not job.ragged)
    if want_strong:
        plan.append("strong")
    want_family = (job.use_dist and not a.no_family_leg and not job.ragged
                   and not a.strong and L == 1 and not job.halo
                   and a.family == "random" and a.window <= 0)
    if want_family:
        plan.append("family_variants")
    if job.use_dist and not a.no_partition_leg:
        plan.append("partition_kkt")
    # the library's OWN multi-GPU path (mgpu.hip), in a child process once
    # every rank has freed its HBM -- unless a GPU-free parent of ours does
    # that after the ranks have exited (bench.py orchestrate)
    want_native = (world > 1 and not a.no_native_leg
                   and not a.strong and a.shards_per_gpu == 1
                   and not os.environ.get("SPMV_BENCH_PARENT_RUNS_NATIVE"))
    if want_native:
        plan.append("native_mgpu")
    if (single and not a.no_extras and a.family == "random"
            and a.window <= 0 and a.format == "hll"):
        plan.append("variants")
    if world == 1 and not a.no_cpu_baseline:
        plan.append("cpu_baseline")
    if single and not a.no_extras:
        plan.append("extras")
    legs.announce(plan)

    def current_line(provisional=False):
        """the line as it stands (callable from the watchdog thread)"""
        with legs.lock:
            line = json.loads(json.dumps(out))  # a snapshot
            if world > 1:
                line.setdefault("native", None)
            line["legs_failed"] = list(legs.failed)
            line["legs_skipped"] = list(legs.skipped)
            line["legs_s"] = dict(legs.seconds)
            if world > 1 and per_rank:
                # y-throughput (global rows per second) kernel
                # only, kernel + exchange one after the other, as measured
                # (the plain arrangement) and the best arrangement timed
                kmax = max(per_rank)
                exch = line["config"]["exchange_ms_alone"]
                best = line.get("ms_per_step_best")
                line["config"]["y_rows_per_s"] = {
                    "kernel_only": round(job.Mglob / (kmax * 1e-3), 1),
                    "kernel_then_exchange": round(
                        job.Mglob / ((kmax + exch) * 1e-3), 1) if exch else None,
                    "measured": round(job.Mglob / (ms_per_step * 1e-3), 1),
                    "best_arrangement": round(job.Mglob / (best * 1e-3), 1)
                    if best else None}
            # the >= 6x target is a FIXED-problem reading (80M x 80M on N
            # GPUs vs 1): top level, so a scaling run reads without digging
            line["strong_speedup"] = strong_speedup_of(
                line, line["config"]["strong"], world)
            if line.get("ms_per_step_best") and line["strong_speedup"] and \
                    not (line["config"]["strong"] or {}).get("speedup_vs_1gpu"):
                line["strong_speedup_best"] = round(
                    line["strong_speedup"] * line["ms_per_step"]
                    / line["ms_per_step_best"], 3)
            if provisional:
                line["provisional"] = True
                line["legs_pending"] = list(legs.pending)
            return line

    def emit(provisional=False):
        print(json.dumps(current_line(provisional)), flush=True)

    # ---- the line goes out NOW; whatever happens in a leg, it stands ----
    legs.emit_final = emit  # from here on the watchdog guards LEGS
    if job.rank == 0:
        emit(provisional=True)

    def note(key, value, where=None):
        with legs.lock:
            (out["config"] if where is None else where)[key] = value

    # ---- legs with collectives -------------------------------------------
    if job.use_dist:
        # the exchange by itself (kernel only / serial / overlapped)
        v = legs.run("exchange_alone", job.exchange_alone)
        note("exchange_ms_alone", round(v, 5) if v else None)
        v = legs.run("exchange_alternatives", job.time_exchange_alternatives,
                     prepare=[job.build_exchange_alternatives])
        note("exchange_alternatives_ms", v)
    if "arrangement" in plan:
        built = []

        def build():
            built.append(job.build_alternative())
            return built[0]
        rec = legs.run("arrangement", lambda b: job.time_alternative(*b),
                       prepare=[build],
                       cleanup=lambda *_: job.release_alternative(
                           built[0] if built else None))
        note("arrangements", rec)
        if rec:
            with legs.lock:
                best = rec.get("best_ms_per_step") or ms_per_step
                out["ms_per_step_best"] = round(best, 5)
                out["value_best"] = round(
                    2.0 * job.nnz_global / (best * 1e6), 2)
                out["best_arrangement"] = (
                    rec["alternative"] if rec["winner"] != "plain" else
                    "plain (the alternative -- %s -- did not win by %.0f %%)"
                    % (rec["alternative"], 100 * (1 - ARRANGEMENT_MARGIN)))
    # ---- N > 1: the fixed-problem reading of config 5 (80M x 80M, 8 logical
    # shards of 10M rows, 8/N per GPU), so that a scaling run can be read
    # against the ">= 6x y-throughput at 8 GPUs" target: rows/s of the SAME
    # problem at every N; the 1-GPU denominator is a committed measurement.
    if want_strong:
        st = StrongLeg(job)
        note("strong", legs.run("strong", st.run, prepare=[st.build],
                                cleanup=lambda *_: st.release()))
    # ---- N > 1: the banded and the W = 2^20 members of the family at this N
    if want_family:
        fv = FamilyVariantLeg(job)
        note("family_variants", legs.run(
            "family_variants", fv.run, prepare=[fv.build],
            cleanup=lambda *_: fv.release()))
    # ---- N > 1: the nlpkkt160-shaped matrix over the same ranks, even rows
    # vs nnz-balanced rows (reference csr.c:218-276)
    if "partition_kkt" in plan:
        kk = KktLeg(job, a.kkt_n)
        note("partition_kkt", legs.run(
            "partition_kkt", kk.run, prepare=[kk.make_file, kk.load],
            cleanup=lambda *_: kk.release(), limit_s=90.0))

    # ---- the ranks part ----------------------------------------------------
    if want_native and not legs.broken:
        # every rank frees its HBM (barrier inside), the process group goes
        # away, ranks 1.. exit -- a rank left waiting in an RCCL barrier would
        # spin a kernel on its GPU under the native child's measurement --
        # and rank 0, alone, starts the child
        legs.run("release_for_native", job.release_everything, limit_s=30.0)
    if job.use_dist and not legs.broken:
        legs._current = ("destroy_process_group", time.time(), 30.0)
        try:
            dist.destroy_process_group()
        except Exception as e:  # noqa: BLE001 - the line matters, not this
            legs.failed.append("destroy_process_group: %r" % (e,))
        legs._current = None
    if job.rank != 0:
        legs.finished = True
        if legs.broken:
            os._exit(0)  # a dead peer: leave without the runtime's teardown
        return
    if want_native:
        from .native import native_leg
        remaining = legs.deadline_s - legs.spent() - 15.0
        nat = None
        if remaining > 25.0:
            nat = legs.run(
                "native_mgpu",
                lambda: native_leg(args, job.world,
                                   timeout_s=max(20.0, min(180.0, remaining))),
                start_by=LEG_BUDGET_S + 30, collective=False,
                limit_s=max(30.0, min(190.0, remaining + 5.0)))
        else:
            legs.skipped.append("native_mgpu (%.0f s of the run's budget left)"
                                % remaining)
        note("native", nat, out)
    elif world > 1:
        note("native", None, out)

    # ---- single GPU: what rides on the line (no collectives) ---------------
    if "variants" in plan:
        v = legs.run("variants", lambda: window_variants(
            S, torch, job.x, job.y, job.Mloc, job.Nglob, job.K, a.family),
            collective=False, start_by=1e9, limit_s=1e9)
        note("variants", v, roof)
    if "cpu_baseline" in plan:
        v = legs.run("cpu_baseline", lambda: cpu_baseline(
            S, job.kind, job.Mloc, job.Nglob, job.K, job.W,
            a.cpu_csv_dir or os.path.join(ROOT, "gpurun_out", "cpu_baseline"),
            "%s%dM" % (a.family, job.Mloc // 1_000_000)),
            collective=False, start_by=1e9, limit_s=1e9)
        note("cpu_baseline", v, out)
    if "extras" in plan:
        v = legs.run("extras", lambda: extra_measurements(
            S, torch, mat, job.x, job.y, job.Mloc, job.Nglob, job.K),
            collective=False, start_by=1e9, limit_s=1e9)
        note("extras", v, out)
    legs.finish()
    if legs.broken:
        os._exit(0)


def describe_job(S, torch, dist, dev, local_rank, world, backend, kern_ms):
    """-> (config.rccl dict, [every rank's mean kernel ms]).  Collective: all
    ranks call it.  nranks_joined = an all-reduce of ones (what the
    communicator really spans), devices = PCI bus id per rank (two ranks on
    one card would show here), version = the RCCL torch drives."""
    import numpy as np
    ones = torch.ones(1, dtype=torch.float64, device=dev)
    dist.all_reduce(ones)
    mine = torch.tensor([float(np.mean(kern_ms))], dtype=torch.float64,
                        device=dev)
    allk = torch.zeros(world, dtype=torch.float64, device=dev)
    if backend == "gloo":  # rehearsal: no GPU all-gather in gloo
        host = torch.zeros(world, dtype=torch.float64)
        dist.all_gather_into_tensor(host, mine.cpu())
        allk = host
    else:
        dist.all_gather_into_tensor(allk, mine)
    try:
        bus = S.device_pci_bus_id(local_rank)
    except OSError:
        bus = "?"
    ids = [None] * world
    dist.all_gather_object(ids, bus)
    ver = None
    if backend == "nccl":
        try:
            ver = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:  # noqa: BLE001 - informational
            ver = None
    return ({"backend": "nccl (RCCL)" if backend == "nccl" else backend,
             "version": ver, "library_links": S.rccl_version(),
             "nranks_joined": int(round(float(ones.item()))),
             "devices": ids},
            [float(v) for v in allk.tolist()])


class StrongLeg:
    """The fixed 80M x 80M problem at this N: 8/N logical shards of 10M rows
    per GPU with global columns, built with rank 0's pick.  At N = 8 this IS
    the weak-scaling workload (one shard per GPU), so nothing is rebuilt.
    build() is local, run() holds the collectives (benchlib.legs)."""

    def __init__(self, job):
        self.job, self.ms_, self.xs, self.ys, self.sh = job, [], None, None, None
        rows = job.args.rows_per_gpu
        self.rows, self.total = rows, 8 * rows
        # the committed denominator is the FULL-size problem's
        self.one_ms, self.one_src = (
            strong_one_gpu() if rows == ROWS_PER_GPU else
            (None, "not the 10M-rows-per-shard problem"))
        self.same = job.world == 8 and job.Mglob == self.total
        self.per = 8 // job.world

    def build(self):
        if self.same:
            return None
        job, S, torch = self.job, self.job.S, self.job.torch
        self.xs = torch.empty(self.total, dtype=torch.float64, device=job.dev)
        self.ys = torch.zeros(self.total, dtype=torch.float64, device=job.dev)
        S.dev_fill_synth(self.xs.data_ptr(), self.total, X_SEED, 0,
                         job.stream())
        self.ms_, _, _ = job.build_shards(
            self.per, self.rows, job.rank * self.per * self.rows, self.total,
            2 * self.total)
        if job.blocked:
            for m in self.ms_:
                m.build_panels_like(job.mats[0])
        self.sh = job.make_sharded(self.ms_, self.per * self.rows, self.xs,
                                   self.ys)
        return None

    def run(self, _=None):
        if self.same:
            return {"problem": "80M x 80M, 8 shards of 10M rows: identical to "
                               "this line's workload at N = 8",
                    "one_gpu_ms_per_step": self.one_ms,
                    "one_gpu_source": self.one_src,
                    "note": "speedup vs 1 GPU = one_gpu_ms_per_step / "
                            "ms_per_step of this line"}
        self.sh.step()
        ms = self.job.time_steps(self.sh, 5)
        return {"problem": "80M x 80M fixed, %d logical shards of 10M rows "
                           "per GPU" % self.per,
                "ms_per_step": round(ms, 4),
                "rows_per_s": round(self.total / (ms * 1e-3), 1),
                "one_gpu_ms_per_step": self.one_ms,
                "one_gpu_source": self.one_src,
                "speedup_vs_1gpu": round(self.one_ms / ms, 3)
                if self.one_ms else None}

    def release(self):
        for m in self.ms_:
            m.release()
        self.ms_, self.sh, self.xs, self.ys = [], None, None, None


class FamilyVariantLeg:
    """N > 1: the OTHER members of the synthetic family through the same plain
    path at this N (north star: "GFLOP/s and achieved HBM GB/s on synthetic
    banded/random sparse matrices reported at 1/2/4/8 GPUs"; at N = 1 they
    ride on the line as roofline.variants): the banded matrix and the random
    one with columns within 2^20 of the diagonal, same rows per GPU and
    entries per row as the line's workload, autotuned (rank 0's pick for
    all), rows of y checked against the host generator, 5 timed steps of
    kernel + ONE all-gather.  With locality the kernel time does not grow
    with N, so these are the members whose weak scaling the exchange bounds,
    not the 80M columns.  build() is local, run() holds the collectives."""

    VARIANTS = (("banded", "banded", 0), ("W=2^20", "random", 1 << 20))

    def __init__(self, job):
        self.job, self.built = job, []

    def build(self):
        job, S = self.job, self.job.S
        for tag, fam, window in self.VARIANTS:
            kind = FAMILIES[fam]
            W = window if window > 0 else 2 * job.Nglob
            ms_, nnz, slots = job.build_shards(1, job.Mshard, w=W, kind=kind,
                                               col_major=True)
            m = ms_[0]
            self.built.append([tag, fam, kind, W, m, nnz, None])
            best, _ = m.autotune(job.x.data_ptr(),
                                 job.y.data_ptr() + 8 * job.row0)
            self.built[-1][6] = best
        return None

    def run(self, _=None):
        job, S, D, torch, np = (self.job, self.job.S, self.job.D,
                                self.job.torch, self.job.np)
        out = {}
        for tag, fam, kind, W, m, nnz, best in self.built:
            mine = D.Pick(best, m.panels_schedule(), m.panels_tile_rows() or 0)
            pick = D.agree_on_pick(job.dist, mine, job.dev)
            kernel = pick.kernel
            labels = S.HLL_KERNEL_LABELS if job.args.format == "hll" \
                else S.CSR_KERNEL_LABELS
            blocked = labels[kernel] == "tile_panels"
            rebuild = blocked and (m.panels_info() is None
                                   or not pick.same_build(mine))
            together(job, (lambda: m.build_panels(
                0, pick.schedule, pick.tile_rows)) if rebuild
                else (lambda: None))
            sh = together(job, lambda: D.ShardedSpmv(
                m, kernel, job.rank, job.world, job.Mshard, job.x, job.y,
                chunks=1, force_exchange=job.args.force_exchange,
                backend=torch))
            sh.step()
            job.sync()

            def check():  # own rows + a row of every other rank's range
                rng = np.random.default_rng(4321 + job.rank)
                rows = np.concatenate(
                    [rng.integers(0, job.Mloc, 32) + job.row0,
                     [job.starts[r] for r in range(job.world)]])
                got = job.y[torch.as_tensor(rows, device=job.dev)].cpu().numpy()
                try:
                    return check_rows(S, kind, job.Nglob, job.K, W, got, rows)
                except SystemExit as e:
                    raise RuntimeError(str(e))
            checked = together(job, check)
            ev = [(torch.cuda.Event(enable_timing=True),
                   torch.cuda.Event(enable_timing=True)) for _ in range(5)]
            job.barrier()
            t0 = time.perf_counter()
            for k in range(5):
                sh.step(events=ev[k])
            job.barrier()
            ms = job.max_over_ranks(time.perf_counter() - t0) * 1e3 / 5
            kms = job.max_over_ranks(
                float(np.mean([a.elapsed_time(b) for a, b in ev])))
            t = torch.tensor([float(nnz)], dtype=torch.float64, device=job.dev)
            job.dist.all_reduce(t)
            nnz_glob = int(t.item())
            b = m.kernel_bytes(kernel)
            out[tag] = {
                "workload": workload_name(fam, job.args.format, job.Mshard,
                                          job.Nglob, job.Mglob, job.K,
                                          0 if W >= 2 * job.Nglob else W, W),
                "kernel": ("hll_" if job.args.format == "hll" else "csr_")
                + labels[kernel],
                "layout": m.panels_describe() if blocked else None,
                "ms_per_step": round(ms, 5),
                "value_gflops": round(2.0 * nnz_glob / (ms * 1e6), 2),
                "kernel_ms_max_rank": round(kms, 5),
                "kernel_frac_of_8TBps": round(b / (kms * 1e6) / 8000.0, 4),
                "rows_per_s": round(job.Mglob / (ms * 1e-3), 1),
                "rows_checked": checked}
        return out

    def release(self):
        for item in self.built:
            item[4].release()
        self.built = []


# ------------------------------------------------- a host matrix over N ranks
class PartitionedRun:
    """rows of host matrix A over the ranks (even / nnz), CSR shards, one
    kernel for all (rank 0's measured pick unless given).  prepare() is LOCAL
    (slice, upload, x, y, the selector); run(steps) holds the collectives ->
    dict with per-rank rows / entries / kernel ms, ms_per_step, exchange."""

    def __init__(self, job, A, partition, xchg, kernel=None, layout=None):
        """layout: (schedule, tile rows) of the blocked copy to build when
        `kernel` is the blocked path and was picked elsewhere (the nnz run
        repeats the even run's pick: same kernel AND same layout)"""
        S, D = job.S, job.D
        self.job, self.A, self.xchg, self.kernel = job, A, xchg, kernel
        self.layout = layout
        self.M, self.N = A.contents.M, A.contents.N
        self.IRP, _, _ = S.csr_arrays(A)
        self.starts = (D.nnz_row_partition(self.IRP, job.world)
                       if partition == "nnz"
                       else D.even_row_partition(self.M, job.world))
        self.per_nnz, self.balance = D.partition_balance(self.IRP, self.starts)
        # handled as ragged even if the cut happens to be even
        self.ragged = partition == "nnz"
        self.dA = self.x = self.y = None
        self.tuned = None

    def prepare(self):
        job, S, torch = self.job, self.job.S, self.job.torch
        st, rank = self.starts, job.rank
        sl = S.csr_row_slice(self.A, st[rank], st[rank + 1])
        try:
            self.dA = S.CsrDevice.upload(sl)
        finally:
            S.csr_free(sl)
        # the reference's x
        self.x = torch.from_numpy(S.vec_random(self.N)).to(job.dev)
        # even partition: y padded to equal fragments for the in-place gather
        self.y = torch.zeros(self.M if self.ragged else st[1] * job.world,
                             dtype=torch.float64, device=job.dev)
        if self.kernel is None:
            self.tuned, _ = self.dA.autotune(
                self.x.data_ptr(), self.y.data_ptr() + 8 * st[rank])
        return self

    def run(self, steps):
        job, S, D, torch, np = (self.job, self.job.S, self.job.D,
                                self.job.torch, self.job.np)
        world, rank, dA, st = job.world, job.rank, self.dA, self.starts
        kernel = self.kernel

        def build():  # LOCAL work inside the collective half: agreed on
            if kernel == S.CSR_KERNEL_PANELS and dA.panels_info() is None:
                if self.layout and self.layout[0]:
                    dA.build_panels(0, self.layout[0], self.layout[1])
                else:
                    dA.build_panels(0)
            force = job.args.force_exchange  # 1-rank RCCL rehearsal
            if self.ragged:
                return D.ShardedSpmv(dA, kernel, rank, world, None, self.x,
                                     self.y, chunks=1, mode=self.xchg,
                                     starts=st, force_exchange=force)
            return D.ShardedSpmv(dA, kernel, rank, world, st[1], self.x,
                                 self.y, chunks=1, force_exchange=force)

        if kernel is None:
            kernel = self.tuned
            if job.use_dist:
                mine = D.Pick(kernel, dA.panels_schedule(),
                              dA.panels_tile_rows() or 0)
                pick = D.agree_on_pick(job.dist, mine, job.dev)
                kernel = pick.kernel
                # EVERY rank enters the agreement, whether or not it is the
                # one that has to rebuild (rank 0 never is)
                rebuild = (kernel == S.CSR_KERNEL_PANELS
                           and not pick.same_build(mine))
                together(job, (lambda: dA.build_panels(
                    0, pick.schedule, pick.tile_rows)) if rebuild
                    else (lambda: None))
        self.kernel = kernel
        sh = together(job, build)
        if kernel == S.CSR_KERNEL_PANELS:  # what a later run repeats
            self.layout = (dA.panels_schedule(), dA.panels_tile_rows() or 0)
        yy = sh.y
        sh.step()
        job.sync()

        def check():  # own rows and rows of every other rank vs the HOST matrix
            rng = np.random.default_rng(77 + rank)
            rows = rng.integers(0, self.M, 64)
            got = yy[torch.as_tensor(rows, device=job.dev)].cpu().numpy()
            _, JA, AS = S.csr_arrays(self.A)
            xh = self.x.cpu().numpy()
            for g, r in zip(got, rows):
                c = JA[self.IRP[r]:self.IRP[r + 1]]
                v = AS[self.IRP[r]:self.IRP[r + 1]]
                t = v * xh[c]
                if abs(g - t.sum()) > 1e-6 * max(abs(t.sum()),
                                                 1e-3 * np.abs(t).sum()):
                    raise RuntimeError("parity check failed on row %d" % r)
            return len(rows)
        nrows = together(job, check)
        ev = [(torch.cuda.Event(enable_timing=True),
               torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        job.barrier()
        t0 = time.perf_counter()
        for k in range(steps):
            sh.step(events=ev[k])
        job.barrier()
        ms = job.max_over_ranks(time.perf_counter() - t0) * 1e3 / steps
        kms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
        allk = torch.zeros(world, dtype=torch.float64, device=job.dev)
        if job.use_dist:
            mine = torch.tensor([kms], dtype=torch.float64, device=job.dev)
            if job.args.backend == "gloo":  # rehearsal: host-staged
                host = torch.zeros(world, dtype=torch.float64)
                job.dist.all_gather_into_tensor(host, mine.cpu())
                allk = host
            else:
                job.dist.all_gather_into_tensor(allk, mine)
        else:
            allk[0] = kms
        exch = job.exchange_alone(sh, 5) if job.use_dist else None
        return {"rows_per_rank": [st[k + 1] - st[k] for k in range(world)],
                "nnz_per_rank": self.per_nnz,
                "nnz_max_over_min": round(self.balance, 3),
                "kernel": "csr_" + S.CSR_KERNEL_LABELS[kernel],
                "kernel_ms_per_rank": [round(float(v), 5)
                                       for v in allk.tolist()],
                "ms_per_step": round(ms, 5),
                "exchange": sh.mode,
                "exchange_ms_alone": round(exch, 5) if exch else None,
                "alg_bytes": dA.algorithmic_bytes,
                "rows_checked": nrows}

    def release(self):
        if self.dA is not None:
            self.dA.release()
        self.dA = self.x = self.y = None


def together(job, fn):
    """LOCAL work in the middle of a collective sequence: run fn() here, then
    let the ranks agree -- if it failed anywhere, EVERY rank raises (and so
    leaves the sequence at the same point) instead of one rank leaving the
    others inside the next collective"""
    err = res = None
    try:
        res = fn()
    except Exception as e:  # noqa: BLE001 - re-raised below, on every rank
        err = e
    bad = job.legs.ranks_where(err is not None)
    if bad:
        raise RuntimeError("failed on rank(s) %s%s" % (
            bad, ": %r" % (err,) if err is not None else ""))
    return res


class KktLeg:
    """config.partition_kkt: the nlpkkt160-shaped matrix (42 entries per
    state row in the upper half, 15 per constraint row below) over this
    run's ranks, equal ROWS vs near-equal ENTRIES per GPU -- per-rank
    entries, per-rank kernel ms, ms per step of each.  make_file / load are
    local (benchlib.legs prepare steps), run holds the collectives."""

    def __init__(self, job, kkt_n, mtx=""):
        self.job, self.kkt_n, self.mtx = job, kkt_n, mtx
        self.A, self.info, self.runs = None, {}, []

    def make_file(self):
        """rank 0 makes sure the file and its .bin sidecar exist (writes /
        parses once); the others load the sidecar afterwards"""
        if self.job.rank == 0:
            S = self.job.S
            path, self.info = config4_file(self.mtx, self.kkt_n)
            S.csr_free(S.io_load_csr_cached(path))  # writes the sidecar
        return None

    def load(self, _=None):
        S = self.job.S
        path, info2 = config4_file(self.mtx, self.kkt_n)
        t0 = time.time()
        self.A = S.io_load_csr_cached(path)
        self.info = dict(info2, **self.info)
        self.info["load_s"] = round(time.time() - t0, 2)
        return None

    def run(self, *_):
        job, A = self.job, self.A
        even = PartitionedRun(job, A, "even", "p2p")
        self.runs.append(even)
        together(job, even.prepare)
        r_even = even.run(5)
        even.release()
        nnz = PartitionedRun(job, A, "nnz", job.args.ragged_exchange,
                             even.kernel, even.layout)
        self.runs.append(nnz)
        together(job, nnz.prepare)
        r_nnz = nnz.run(5)
        for d in (r_even, r_nnz):
            d.pop("alg_bytes", None)
        return {"matrix": "%dx%d, %d entries (%s)" % (
                    A.contents.M, A.contents.N, A.contents.NZ,
                    self.info["source"]),
                "even_rows": r_even, "nnz_balanced": r_nnz,
                "speedup_nnz_over_even": round(
                    r_even["ms_per_step"] / r_nnz["ms_per_step"], 3)}

    def release(self):
        for r in self.runs:
            r.release()
        self.runs = []
        if self.A is not None:
            self.job.S.csr_free(self.A)
            self.A = None


def run_matrix_rank(args, argv, omp_team):
    """`--config 4 --gpus N` (N > 1; N = 1 is benchlib.single) and
    `--config 2`: a host matrix, one rank per GPU, --partition even | nnz"""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1 and not args.force_exchange:
        job = RankJob(args, omp_team)
        return single_matrix_bench(args, job.S, job.torch, job.dev)
    if args.config != 4:
        raise SystemExit("--config %d is a single-GPU line" % args.config)
    job = RankJob(args, omp_team)
    job.init_process_group()
    t0 = time.time()
    job.legs.phase("config 4: write / load the matrix on every rank")
    kk = KktLeg(job, args.kkt_n, args.mtx)
    together(job, kk.make_file)
    together(job, kk.load)
    A, info = kk.A, kk.info
    run = PartitionedRun(job, A, args.partition, args.ragged_exchange,
                         args.kernel if args.kernel >= 0 else None)
    job.legs.phase("config 4: upload the row ranges + selector")
    together(job, run.prepare)
    job.legs.phase("config 4: pick agreement, first exchange, K timed steps")
    res = run.run(args.steps)
    run.release()
    job.legs.phase("config 4: describe the job")
    rccl, _ = describe_job(job.S, job.torch, job.dist, job.dev, job.local_rank,
                           job.world, args.backend, [1.0])
    M, N, NZ = A.contents.M, A.contents.N, A.contents.NZ
    name = A.contents.name.decode()
    kk.release()
    if job.rank == 0:
        kname = res["kernel"]
        alg = res.pop("alg_bytes")
        kmax = max(res["kernel_ms_per_rank"])
        out = {"metric": METRIC,
               "value": round(2.0 * NZ / (res["ms_per_step"] * 1e6), 2),
               "unit": "GFLOP/s", "n_gpus": job.world, "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": res["ms_per_step"],
               "higher_is_better": True, "scaling": "strong",
               "vs_baseline": None, "dtype": "f64",
               "data": "synthetic" if "generated" in info["source"] else "file",
               "config": dict(
                   {"backend": "nccl (RCCL)" if args.backend == "nccl"
                    else "gloo REHEARSAL",
                    "workload": "%s.mtx %dx%d, %d nnz, CSR over %d GPUs "
                                "(BASELINE config 4: nlpkkt160; %s)"
                                % (name, M, N, NZ, job.world, info["source"]),
                    "kernel": kname, "partition": args.partition,
                    "rccl": rccl, "rocm": job.S.rocm_runtime_report()},
                   **res, **info),
               "roofline": {"bound": "hbm", "unit": "GB/s", "peak": 8000.0,
                            "achieved": round(alg / (kmax * 1e6), 1),
                            "frac": round(alg / (kmax * 1e6) / 8000.0, 4),
                            "traffic": None, "kernel": kname,
                            "note": "rank 0's shard bytes over the slowest "
                                    "rank's kernel time"},
               "setup_s": round(time.time() - t0, 2),
               "rows_checked": res["rows_checked"] * job.world}
        print(json.dumps(out))
        sys.stdout.flush()
    job.legs.finished = True  # the line is out: the watchdog stands down
    if job.use_dist:
        job.dist.destroy_process_group()
