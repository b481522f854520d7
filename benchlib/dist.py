"""The default workload (BASELINE config 3; config 5 at N = 8) at any GPU
count: one process per GPU, torch.distributed over RCCL (backend "nccl"),
rows partitioned by contiguous ranges, x replicated, the fragments of y
exchanged every step.

`run_rank` is what every rank executes; it is a sequence of small steps on a
`RankJob` (build the shard, agree on the kernel, arrange the exchange, check
rows against the host generator, time K steps, optional legs, report).  The
optional legs of an N > 1 line -- the fixed-problem reading (`config.strong`),
the exchange alone and its alternatives, the nnz-balanced partition of the
nlpkkt160-shaped matrix, the library's own multi-GPU path -- each run inside
`optional_leg`: a leg that fails is named in `legs_failed` and costs nothing
else of the line."""
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
from .single import extra_measurements, single_matrix_bench, window_variants

# wall budget of the optional legs of an N > 1 line (seconds): the driver
# gives a bench run 600 s; the main measurement takes ~15 s, so the legs
# together must stay well under half of that.  A leg is skipped (and named in
# legs_skipped) when the time already spent exceeds its start-by mark.
LEG_BUDGET_S = 240.0


class Legs:
    """bookkeeping of the optional legs: failures are recorded, never raised;
    all ranks must take the same skip decision (collectives inside the legs),
    so the clock that decides is rank 0's, broadcast by the caller"""

    def __init__(self, t0):
        self.t0 = t0
        self.failed, self.skipped, self.seconds = [], [], {}

    def spent(self):
        return time.time() - self.t0


class RankJob:
    """state of one rank's run of the default workload"""

    def __init__(self, args, omp_team):
        import numpy as np
        import torch
        import torch.distributed as dist
        import spmv_scpa_amd as S
        from spmv_scpa_amd import dist as D
        self.np, self.torch, self.dist, self.S, self.D = np, torch, dist, S, D
        self.args, self.omp_team = args, omp_team
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
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
        self.legs = Legs(self.t_start)

    # ------------------------------------------------------------ plumbing
    def init_process_group(self):
        if not self.use_dist:
            return
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        # a collective that never completes (a rank died in a leg) ends this
        # rank after 5 minutes instead of holding the GPUs until the caller's
        # own limit
        import datetime
        limit = datetime.timedelta(seconds=300)
        if self.args.backend == "gloo":
            self.dist.init_process_group("gloo", timeout=limit)
        else:
            self.dist.init_process_group("nccl", device_id=self.dev,
                                         timeout=limit)

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
        t = self.torch.tensor([float(value)], dtype=self.torch.float64,
                              device=self.dev)
        if self.use_dist:
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

    def optional_leg(self, name, fn, start_by=LEG_BUDGET_S, collective=True):
        """run fn() unless the legs' wall budget is spent; a failure is
        recorded in legs_failed and None is returned.  The legs hold
        collectives, so every rank must take the same skip decision: the
        slowest rank's clock (an all-reduce) decides -- except for a leg only
        one rank runs (collective=False)."""
        late = self.max_over_ranks(self.legs.spent()) \
            if self.use_dist and collective else self.legs.spent()
        if late > start_by:
            self.legs.skipped.append("%s (%.0f s spent, starts by %.0f s)"
                                     % (name, late, start_by))
            return None
        t0 = time.time()
        try:
            return fn()
        except Exception as e:  # noqa: BLE001 - an optional figure
            self.legs.failed.append("%s: %r" % (name, e))
            return None
        finally:
            self.legs.seconds[name] = round(time.time() - t0, 1)

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

    def build_shards(self, count, rows, first_row=None, ncols=None, w=None):
        """`count` logical shards of `rows` rows starting at `first_row`
        -> (handles, true entries, stored slots)"""
        a, S = self.args, self.S
        first_row = self.row0 if first_row is None else first_row
        ncols = self.Nglob if ncols is None else ncols
        w = self.W if w is None else w
        out, nnz, stored = [], 0, 0
        for j in range(count):
            dA = S.CsrDevice.generate(self.kind, rows, ncols, self.K, w,
                                      first_row + j * rows, MATRIX_SEED)
            nnz += dA.NZ
            if a.format == "hll":
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

    def chain_logical_shards(self):
        """the blocked path runs whole matrices only: hold the rank's rows as
        `nsplit` logical shards (4, like the row chunks of the direct kernels)
        so that the all-gather of one shard runs under the kernel of the next
        -- at 8 GPUs the exchange (560 MB in per GPU) is longer than the
        kernel of a matrix with locality"""
        D = self.D
        nsplit = 2 if self.args.force_exchange and self.world == 1 else 4
        if not (self.blocked and not self.sweep and self.use_dist
                and self.L == 1 and not self.ragged
                and self.Mshard % (nsplit * D.HACK) == 0):
            return
        model = self.mats[0]
        for m in self.mats[1:]:
            m.release()
        self.L, self.Mshard = nsplit, self.Mshard // nsplit
        self.mats, self.nnz_local, self.slots = self.build_shards(
            self.L, self.Mshard)
        for m in self.mats:  # the tuned schedule and tile height
            m.build_panels_like(model)
        model.release()
        self.arrangement = ("chain: %d logical shards, all-gather of shard c "
                            "under the kernel of c+1" % self.L)

    def exchange_settings(self):
        a, D = self.args, self.D
        chunks = a.chunks if a.chunks > 0 else (4 if self.world > 1 else 1)
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

    def make_sharded(self, ms, rows_total=None, xx=None, yy=None):
        a = self.args
        rows_total = self.Mshard * self.L if rows_total is None else rows_total
        mode = "halo" if self.halo else (
            a.ragged_exchange if self.ragged else None)
        return self.D.ShardedSpmv(
            ms if len(ms) > 1 else ms[0], self.kernel, self.rank, self.world,
            None if self.ragged else rows_total,
            self.x if xx is None else xx, self.y if yy is None else yy,
            waves_per_block=a.waves, chunks=self.chunks,
            force_exchange=a.force_exchange, mode=mode, halo_rows=self.halo,
            starts=self.starts if self.ragged else None)

    def choose_sweep_arrangement(self):
        """The sweep launch is persistent and wants its whole grid resident
        (phase counters), so by default the exchange FOLLOWS the kernel.
        Alternative: two logical shards, each swept by a grid that leaves
        --reserve-cus compute units free, the all-gather of the first half
        running beside the sweep of the second.  Whether RCCL's kernels and
        the persistent grid share the chip well is a property of the node:
        both arrangements are timed here (5 steps each, max over ranks) and
        the faster one is kept -- the same "choose by measurement" rule as the
        kernel selector, and every rank sees the same reduced times."""
        a, D = self.args, self.D
        if not (self.sweep and self.use_dist and self.L == 1 and not self.halo
                and not self.ragged and self.Mshard % (2 * D.HACK) == 0):
            return
        try:
            alt, nnz_alt, slots_alt = self.build_shards(2, self.Mshard // 2)
            for m in alt:
                m.build_panels(0, "sweep", reserve_cus=a.reserve_cus)
            sh_alt = self.make_sharded(alt)
            for s_ in (self.sharded, sh_alt):
                s_.step()
            t_serial = self.time_steps(self.sharded, 5)
            t_split = self.time_steps(sh_alt, 5)
            self.arrangement = (
                "sweep: exchange after the kernel %.3f ms/step vs 2 logical "
                "shards on %d fewer CUs with overlapped all-gather %.3f "
                "ms/step" % (t_serial, a.reserve_cus, t_split))
            if t_split < t_serial:
                for m in self.mats:
                    m.release()
                self.mats, self.sharded = alt, sh_alt
                self.L, self.Mshard = 2, self.Mshard // 2
                self.nnz_local, self.slots = nnz_alt, slots_alt
                self.arrangement += " -> overlapped"
            else:
                for m in alt:
                    m.release()
                self.arrangement += " -> exchange after the kernel"
        except OSError as e:
            self.arrangement = "sweep: overlapped arrangement not built (%s)" % e

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
        self.checked = check_rows(self.S, self.kind, self.Nglob, self.K, self.W,
                                  got, rows)

    def timed_steps(self):
        """K steps between barrier + synchronize on both sides; per-step
        events on the launch stream and host timestamps after each enqueue.
        -> (wall seconds, kernel ms per step, host seconds between enqueues)"""
        torch, n = self.torch, self.args.steps
        ev = [(torch.cuda.Event(enable_timing=True),
               torch.cuda.Event(enable_timing=True)) for _ in range(n)]
        stamps = [0.0] * (n + 1)
        self.barrier()
        t0 = time.perf_counter()
        stamps[0] = t0
        for k in range(n):
            self.sharded.step(events=ev[k])
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

    def exchange_alternatives(self):
        """the same fragments by the other ways the library can move them
        (dist.RaggedExchange on this run's row ranges): grouped send / recv,
        one broadcast per rank, all-gather padded to the longest fragment +
        compaction -- so that one scaling run prices every exchange"""
        D = self.D
        if self.halo:
            return None
        out = {}
        for mode in ("p2p", "bcast", "padded"):
            sh = D.ShardedSpmv(self.mats[0], self.kernel, self.rank, self.world,
                               None, self.x, self.y, chunks=1, mode=mode,
                               force_exchange=self.args.force_exchange,
                               starts=[int(v) for v in self.starts]
                               if self.ragged else _never_even(self.starts),
                               compute=lambda a, b, out=None: None)
            out[mode] = round(self.exchange_alone(sh, 5), 5)
        return out

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
    job.define_workload()

    # ---- build the shard(s) in HBM (device-side generator + converter) ----
    t_setup = time.time()
    job.alloc_vectors()
    job.mats, job.nnz_local, job.slots = job.build_shards(job.L, job.Mshard)
    job.arrangement = None
    job.pick_kernel()
    job.chain_logical_shards()
    if job.blocked:
        for m in job.mats[1:]:  # the tuned shard's schedule and tile height
            if m.panels_info() is None:
                m.build_panels_like(job.mats[0])
    labels, prefix = job.labels()
    kname = prefix + labels[job.kernel]
    job.sync()
    job.exchange_settings()
    job.sharded = job.make_sharded(job.mats)
    job.choose_sweep_arrangement()
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

    job.check_result()
    stat1 = cgroup_cpu_stat()
    job.measure()
    kern_ms = job.kern_ms

    # the exchange by itself (kernel only / serial / overlapped)
    exch_ms = exch_alt = None
    if job.use_dist:
        exch_ms = job.optional_leg("exchange_alone", job.exchange_alone)
        exch_alt = job.optional_leg("exchange_alternatives",
                                    job.exchange_alternatives)

    # what joined, on which cards, and every rank's own kernel time
    rccl = per_rank = None
    if job.use_dist:
        rccl, per_rank = describe_job(S, torch, dist, job.dev, job.local_rank,
                                      job.world, args.backend, kern_ms)
    job.reduce_over_ranks()

    # ---- N > 1: the fixed-problem reading of config 5 (80M x 80M, 8 logical
    # shards of 10M rows, 8/N per GPU), so that a scaling run can be read
    # against the ">= 6x y-throughput at 8 GPUs" target: rows/s of the SAME
    # problem at every N; the 1-GPU denominator is a committed measurement.
    strong = None
    if (job.world > 1 and not args.strong and not args.no_strong_leg
            and 8 % job.world == 0 and args.family == "random"
            and args.window <= 0 and not job.ragged):
        strong = job.optional_leg("strong", lambda: strong_leg(job))
    # ---- N > 1: the nlpkkt160-shaped matrix over the same ranks, even rows
    # vs nnz-balanced rows (reference csr.c:218-276)
    kkt = None
    if job.world > 1 and not args.no_partition_leg:
        kkt = job.optional_leg("partition_kkt",
                               lambda: kkt_partition_leg(job, args.kkt_n))
    # ---- N > 1: the library's OWN multi-GPU path (mgpu.hip), in a child
    # process once every rank has freed its HBM -- unless a GPU-free parent
    # of ours does that after the ranks have exited (bench.py orchestrate)
    native = None
    want_native = (job.world > 1 and not args.no_native_leg
                   and not args.strong and args.shards_per_gpu == 1
                   and not os.environ.get("SPMV_BENCH_PARENT_RUNS_NATIVE"))
    # everything the line needs from the device is read before the release
    blocked_desc = {
        "blocked_schedule": mat.panels_schedule() if job.blocked else None,
        "blocked_layout": mat.panels_describe() if job.blocked else None,
        "blocked_pin": mat.panels_pin() if job.blocked else None,
        "tune_log": (mat.tune_log() or "").splitlines()
        if job.t_tune is not None and job.t_tune > 1.0 and job.L == 1
        and job.arrangement is None else None}
    exchange_mode = job.sharded.mode
    if want_native:
        # every rank frees its HBM (barrier inside), the process group goes
        # away, ranks 1.. exit -- a rank left waiting in an RCCL barrier would
        # spin a kernel on its GPU under the native child's measurement --
        # and rank 0, alone, starts the child
        job.release_everything()
    if job.use_dist:
        dist.destroy_process_group()
    if job.rank != 0:
        return
    if want_native:
        from .native import native_leg
        native = job.optional_leg(
            "native_mgpu", lambda: native_leg(args, job.world),
            start_by=LEG_BUDGET_S + 60, collective=False)

    a, world, L, Mshard = args, job.world, job.L, job.Mshard
    workload = workload_name(a.family, a.format, job.Mglob // world * 1,
                             job.Nglob, job.Mglob, job.K, a.window, job.W, L,
                             Mshard)
    sched_now = blocked_desc["blocked_schedule"]
    traffic, why = (measured_traffic(workload, kname, sched_now) if world == 1
                    else (None, "single-GPU profiles only"))
    roof = roofline_dict(alg_bytes, kern_ms, kname, job.nnz_local, traffic, why)
    if roof["traffic_layout"] and \
            roof["traffic_layout"] == blocked_desc["blocked_layout"]:
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
            "tune_log": blocked_desc["tune_log"],
            "blocked_schedule": sched_now,
            "blocked_layout": blocked_desc["blocked_layout"],
            # what --blocked-pin takes to run this layout again
            "blocked_pin": blocked_desc["blocked_pin"],
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
            "chunks": job.chunks, "exchange": exchange_mode,
            "exchange_arrangement": job.arrangement,
            "exchange_ms_alone": round(exch_ms, 5) if exch_ms else None,
            "exchange_alternatives_ms": exch_alt,
            "rccl": rccl,
            "halo_rows": job.halo or None,
            "rows_per_s": round(job.Mglob / (ms_per_step * 1e-3), 1),
            "strong": strong,
            "partition_kkt": kkt,
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
        "rows_checked": job.checked,
    }
    if world > 1 and per_rank:
        # y-throughput (global rows per second) kernel only, kernel
        # + exchange one after the other, and as measured (the arrangement
        # that ran overlaps what it can)
        kmax = max(per_rank)
        out["config"]["y_rows_per_s"] = {
            "kernel_only": round(job.Mglob / (kmax * 1e-3), 1),
            "kernel_then_exchange": round(
                job.Mglob / ((kmax + exch_ms) * 1e-3), 1) if exch_ms else None,
            "measured": round(job.Mglob / (ms_per_step * 1e-3), 1)}
    if world > 1:
        out["native"] = native
        out["legs_failed"] = job.legs.failed
        out["legs_skipped"] = job.legs.skipped
        out["legs_s"] = job.legs.seconds
    if job.retried:
        out["host_stall_retry"] = True
    # the >= 6x target is a FIXED-problem reading (80M x 80M on N GPUs vs 1):
    # top level, so a scaling run can be read without digging
    out["strong_speedup"] = strong_speedup_of(out, strong, world)
    single = world == 1 and L == 1 and not a.force_exchange
    if (single and not a.no_extras and a.family == "random"
            and a.window <= 0 and a.format == "hll"):
        roof["variants"] = window_variants(S, torch, job.x, job.y, job.Mloc,
                                           job.Nglob, job.K, a.family)
    if world == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(
            S, job.kind, job.Mloc, job.Nglob, job.K, job.W,
            a.cpu_csv_dir or os.path.join(ROOT, "gpurun_out", "cpu_baseline"),
            "%s%dM" % (a.family, job.Mloc // 1_000_000))
    if single and not a.no_extras:
        out["extras"] = extra_measurements(S, torch, mat, job.x, job.y,
                                           job.Mloc, job.Nglob, job.K)
    print(json.dumps(out))
    sys.stdout.flush()


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


def strong_leg(job):
    """The fixed 80M x 80M problem at this N: 8/N logical shards of 10M rows
    per GPU with global columns, built with rank 0's pick.  At N = 8 this IS
    the weak-scaling workload (one shard per GPU), so nothing is rebuilt."""
    args, S, torch = job.args, job.S, job.torch
    rows, total = args.rows_per_gpu, 8 * args.rows_per_gpu
    # the committed denominator is the FULL-size problem's
    one_ms, one_src = (strong_one_gpu() if rows == ROWS_PER_GPU else
                       (None, "not the 10M-rows-per-shard problem"))
    if job.world == 8 and job.Mglob == total:
        return {"problem": "80M x 80M, 8 shards of 10M rows: identical to "
                           "this line's workload at N = 8",
                "one_gpu_ms_per_step": one_ms,
                "one_gpu_source": one_src,
                "note": "speedup vs 1 GPU = one_gpu_ms_per_step / "
                        "ms_per_step of this line"}
    per = 8 // job.world
    xs = torch.empty(total, dtype=torch.float64, device=job.dev)
    ys = torch.zeros(total, dtype=torch.float64, device=job.dev)
    S.dev_fill_synth(xs.data_ptr(), total, X_SEED, 0, job.stream())
    ms_, _, _ = job.build_shards(per, rows, job.rank * per * rows, total,
                                 2 * total)
    try:
        if job.blocked:
            for m in ms_:
                m.build_panels_like(job.mats[0])
        sh = job.make_sharded(ms_, per * rows, xs, ys)
        sh.step()
        ms = job.time_steps(sh, 5)
    finally:
        for m in ms_:
            m.release()
        del xs, ys
    return {"problem": "80M x 80M fixed, %d logical shards of 10M rows "
                       "per GPU" % per,
            "ms_per_step": round(ms, 4),
            "rows_per_s": round(total / (ms * 1e-3), 1),
            "one_gpu_ms_per_step": one_ms, "one_gpu_source": one_src,
            "speedup_vs_1gpu": round(one_ms / ms, 3) if one_ms else None}


# ------------------------------------------------- a host matrix over N ranks
def load_matrix_on_every_rank(job, mtx, kkt_n):
    """BASELINE config 4's input on every rank: rank 0 makes sure the file
    and its .bin sidecar exist (writes / parses once), the others then load
    the sidecar.  -> (sparse_csr pointer, info)"""
    S = job.S
    info = {}
    if job.rank == 0:
        path, info = config4_file(mtx, kkt_n)
        S.csr_free(S.io_load_csr_cached(path))  # writes the sidecar
    job.barrier()
    path, info2 = config4_file(mtx, kkt_n)
    t0 = time.time()
    A = S.io_load_csr_cached(path)
    info = dict(info2, **info)
    info["load_s"] = round(time.time() - t0, 2)
    return A, info


def partitioned_matrix_run(job, A, partition, xchg, steps, kernel=None):
    """rows of host matrix A over the ranks (even / nnz), CSR shards, one
    kernel for all (rank 0's measured pick unless given), `steps` timed steps
    -> dict with per-rank rows / entries / kernel ms, ms_per_step, exchange"""
    S, D, torch, np = job.S, job.D, job.torch, job.np
    world, rank = job.world, job.rank
    M, N = A.contents.M, A.contents.N
    IRP, _, _ = S.csr_arrays(A)
    starts = (D.nnz_row_partition(IRP, world) if partition == "nnz"
              else D.even_row_partition(M, world))
    per_nnz, balance = D.partition_balance(IRP, starts)
    ragged = partition == "nnz"  # handled as ragged even if the cut is even
    sl = S.csr_row_slice(A, starts[rank], starts[rank + 1])
    dA = S.CsrDevice.upload(sl)
    S.csr_free(sl)
    x = torch.from_numpy(S.vec_random(N)).to(job.dev)  # the reference's x
    # even partition: y padded to equal fragments for the in-place all-gather
    y = torch.zeros(M if ragged else starts[1] * world, dtype=torch.float64,
                    device=job.dev)
    try:
        if kernel is None:
            kernel, _ = dA.autotune(x.data_ptr(),
                                    y.data_ptr() + 8 * starts[rank])
            if job.use_dist:
                mine = D.Pick(kernel, dA.panels_schedule(),
                              dA.panels_tile_rows() or 0)
                pick = D.agree_on_pick(job.dist, mine, job.dev)
                kernel = pick.kernel
                if (kernel == S.CSR_KERNEL_PANELS
                        and not pick.same_build(mine)):
                    dA.build_panels(0, pick.schedule, pick.tile_rows)
        if kernel == S.CSR_KERNEL_PANELS and dA.panels_info() is None:
            dA.build_panels(0)
        if ragged:
            sh = D.ShardedSpmv(dA, kernel, rank, world, None, x, y, chunks=1,
                               mode=xchg, starts=starts)
        else:
            sh = D.ShardedSpmv(dA, kernel, rank, world, starts[1], x, y,
                               chunks=1)
        yy = sh.y
        sh.step()
        job.sync()
        # own rows and rows of every other rank against the HOST matrix
        rng = np.random.default_rng(77 + rank)
        rows = rng.integers(0, M, 64)
        got = yy[torch.as_tensor(rows, device=job.dev)].cpu().numpy()
        _, JA, AS = S.csr_arrays(A)
        xh = x.cpu().numpy()
        for g, r in zip(got, rows):
            c, v = JA[IRP[r]:IRP[r + 1]], AS[IRP[r]:IRP[r + 1]]
            t = v * xh[c]
            if abs(g - t.sum()) > 1e-6 * max(abs(t.sum()),
                                             1e-3 * np.abs(t).sum()):
                raise RuntimeError("parity check failed on row %d" % r)
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
            job.dist.all_gather_into_tensor(
                allk, torch.tensor([kms], dtype=torch.float64, device=job.dev))
        else:
            allk[0] = kms
        exch = job.exchange_alone(sh, 5) if job.use_dist else None
        return {"rows_per_rank": [starts[k + 1] - starts[k]
                                  for k in range(world)],
                "nnz_per_rank": per_nnz,
                "nnz_max_over_min": round(balance, 3),
                "kernel": "csr_" + S.CSR_KERNEL_LABELS[kernel],
                "kernel_ms_per_rank": [round(float(v), 5)
                                       for v in allk.tolist()],
                "ms_per_step": round(ms, 5),
                "exchange": sh.mode,
                "exchange_ms_alone": round(exch, 5) if exch else None,
                "alg_bytes": dA.algorithmic_bytes,
                "rows_checked": len(rows)}, kernel
    finally:
        dA.release()
        del x, y


def kkt_partition_leg(job, kkt_n):
    """config.partition_kkt: the nlpkkt160-shaped matrix (42 entries per
    state row in the upper half, 15 per constraint row below) over this
    run's ranks, equal ROWS vs near-equal ENTRIES per GPU -- per-rank
    entries, per-rank kernel ms, ms per step of each"""
    A, info = load_matrix_on_every_rank(job, "", kkt_n)
    try:
        even, kernel = partitioned_matrix_run(job, A, "even", "p2p", 5)
        nnz, _ = partitioned_matrix_run(job, A, "nnz",
                                        job.args.ragged_exchange, 5, kernel)
        for d in (even, nnz):
            d.pop("alg_bytes", None)
        return {"matrix": "%dx%d, %d entries (%s)" % (
                    A.contents.M, A.contents.N, A.contents.NZ, info["source"]),
                "even_rows": even, "nnz_balanced": nnz,
                "speedup_nnz_over_even": round(
                    even["ms_per_step"] / nnz["ms_per_step"], 3)}
    finally:
        job.S.csr_free(A)


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
    A, info = load_matrix_on_every_rank(job, args.mtx, args.kkt_n)
    res, kernel = partitioned_matrix_run(
        job, A, args.partition, args.ragged_exchange, args.steps,
        args.kernel if args.kernel >= 0 else None)
    rccl, _ = describe_job(job.S, job.torch, job.dist, job.dev, job.local_rank,
                           job.world, args.backend, [1.0])
    M, N, NZ = A.contents.M, A.contents.N, A.contents.NZ
    name = A.contents.name.decode()
    job.S.csr_free(A)
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
    if job.use_dist:
        job.dist.destroy_process_group()
