"""Fault isolation of a bench line's OPTIONAL legs (GPU-free: testable with
two gloo ranks on CPU, tests/test_bench_legs.py).

A bench run is one main measurement plus optional legs (the exchange alone,
alternative arrangements, the fixed-problem reading, the nnz-balanced
partition, the library's own multi-GPU path ...).  The rules that keep the
main measurement's line from being lost to a leg:

 1. The line is printed -- complete, flagged `provisional`, naming
    `legs_pending` -- right after the main measurement, before any leg.
 2. A leg is `prepare` steps (LOCAL work that may fail on one rank only:
    allocations, builds, file I/O, the selector) followed by one `run` (the
    collectives).  After every prepare step the ranks all-reduce a failed
    flag: if any rank failed, EVERY rank skips the leg's collectives and the
    leg is named in `legs_failed` with the failing ranks.
 3. What cannot be agreed on -- a rank that dies or hangs INSIDE a leg's
    collectives -- is bounded: a watchdog thread per rank holds a deadline for
    the running leg and one for the whole run; when it passes, rank 0 prints
    the final line from what it has (`legs_failed` names the leg and the
    deadline) and every rank leaves with os._exit(0) -- well before the
    process group's own timeout.  Nothing is ever exec'ed.
 4. Once a collective has failed the group is `broken`: no further
    collective leg is started.
 5. BEFORE the line exists the same watchdog bounds the main measurement: the
    run names the phase it is in (`phase()`), and if the line is not out
    MAIN_LIMIT_S after the start -- a first collective that never completes
    on a node nobody has seen -- rank 0 prints a failure record (`value`
    null, `failed_in`, seconds per phase) and every rank exits with code 3.
"""
import os
import sys
import threading
import time

# seconds from process start after which no optional leg may START
LEG_BUDGET_S = 200.0
# seconds from process start at which the watchdog ends the run (rank 0 three
# seconds earlier, so its line is out before a peer's exit breaks a collective)
RUN_DEADLINE_S = 270.0
# default bound of ONE leg (its prepare steps + run)
LEG_LIMIT_S = 75.0
# seconds from process start by which the MAIN measurement of a run with
# collectives must have printed its line: past it, rank 0 prints a failure
# record that names the phase the run is stuck in and every rank leaves with
# exit code 3 -- instead of the process group's timeout and a traceback
MAIN_LIMIT_S = 240.0


class InjectedFailure(RuntimeError):
    """SPMV_BENCH_INJECT (tests): a failure placed into a named leg"""


def injected(name, rank, phase):
    """SPMV_BENCH_INJECT="leg:rank:mode[,leg:rank:mode...]" (harness knob of
    the rehearsal tests): mode `prepare` raises in the leg's first prepare step
    on that rank, `run` raises at the start of its run, `hang` sleeps there
    forever, `die` ends the process there (exit code 13)."""
    spec = os.environ.get("SPMV_BENCH_INJECT", "")
    for item in spec.split(","):
        parts = item.strip().split(":")
        if len(parts) != 3 or parts[0] != name or int(parts[1]) != rank:
            continue
        mode = parts[2]
        if mode == phase == "prepare" or (mode == "run" and phase == "run"):
            raise InjectedFailure("injected into %s on rank %d (%s)"
                                  % (name, rank, mode))
        if phase == "run" and mode == "hang":
            while True:
                time.sleep(1.0)
        if phase == "run" and mode == "die":
            os._exit(13)


class LegRunner:
    """bookkeeping, agreement and deadlines of the optional legs of one rank.

    dist / device: torch.distributed and the device collectives' tensors live
    on (None: CPU tensors, the gloo backend); emit_final(): called once, by
    rank 0 only, to print the final line -- from the main thread at the end of
    a healthy run, or from the watchdog at a deadline."""

    def __init__(self, rank, world, t0=None, dist=None, device=None,
                 use_dist=False, emit_final=None, budget_s=LEG_BUDGET_S,
                 deadline_s=RUN_DEADLINE_S):
        self.rank, self.world = rank, world
        self.t0 = time.time() if t0 is None else t0
        self.dist, self.device, self.use_dist = dist, device, use_dist
        self.emit_final = emit_final
        self.budget_s, self.deadline_s = budget_s, deadline_s
        self.failed, self.skipped, self.seconds = [], [], {}
        self.pending = []
        self.broken = False       # a collective failed: no more collectives
        self.lock = threading.RLock()
        self.finished = False     # the final line is out
        self._current = None      # (name, started, limit_s)
        self._watchdog = None
        self.main_limit_s = MAIN_LIMIT_S
        self.phases = {}          # main measurement: seconds per phase
        self._phase = None        # (name, started)
        self.failure_record = None  # () -> dict, the line of a stuck main run

    # ------------------------------------------------------------ plumbing
    def spent(self):
        return time.time() - self.t0

    def _tensor(self, values):
        import torch
        return torch.tensor(values, dtype=torch.float64, device=self.device)

    def max_over_ranks(self, value):
        if not self.use_dist or self.broken:
            return float(value)
        t = self._tensor([float(value)])
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def ranks_where(self, flag):
        """-> sorted ranks on which `flag` is true (one all-reduce); on a
        broken or absent group: this rank alone"""
        if not self.use_dist or self.broken:
            return [self.rank] if flag else []
        t = self._tensor([0.0] * self.world)
        if flag:
            t[self.rank] = 1.0
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return [r for r, v in enumerate(t.tolist()) if v > 0.5]

    # -------------------------------------------------- the main measurement
    def phase(self, name):
        """the main measurement enters phase `name` (None: it is done)"""
        now = time.time()
        if self._phase:
            self.phases[self._phase[0]] = round(
                self.phases.get(self._phase[0], 0.0) + now - self._phase[1], 2)
        self._phase = (name, now) if name else None
        if name and name.startswith("first step"):
            injected("main", self.rank, "run")  # tests: a rank that hangs here

    def _main_stuck(self):
        """the line is not out and the main limit has passed: say where"""
        with self.lock:
            if self.finished:
                return
            self.finished = True
            ph = self._phase
            rec = {"value": None, "failed": True, "n_gpus": self.world,
                   "failed_in": ph[0] if ph else None,
                   "failed_after_s": round(self.spent(), 1),
                   "stuck_in_phase_s": round(time.time() - ph[1], 1)
                   if ph else None,
                   "phases_s": dict(self.phases),
                   "why": "the main measurement did not complete within "
                          "%.0f s; rank %d was in this phase (a collective "
                          "that never completes looks like this)"
                          % (self.main_limit_s, self.rank)}
            try:
                if self.failure_record:
                    rec = dict(self.failure_record(), **rec)
                if self.rank == 0:
                    import json
                    print(json.dumps(rec), flush=True)
            finally:
                sys.stderr.write("bench.py rank %d: stuck in %r\n"
                                 % (self.rank, rec["failed_in"]))
                sys.stderr.flush()
                os._exit(3)

    # ------------------------------------------------------------ watchdog
    def start_watchdog(self):
        """from now on a leg that outlives its limit, or a run that outlives
        deadline_s, ends with rank 0's final line and exit code 0"""
        if self._watchdog is not None:
            return
        self._watchdog = threading.Thread(target=self._watch, daemon=True,
                                          name="bench-leg-watchdog")
        self._watchdog.start()

    def _watch(self):
        lead = 3.0 if self.rank == 0 else 0.0
        while not self.finished:
            time.sleep(0.25)
            cur = self._current
            now = time.time()
            why = None
            if self.emit_final is None:  # the line does not exist yet
                if now - self.t0 > self.main_limit_s - lead:
                    self._main_stuck()
                    return
                continue
            if cur and now - cur[1] > cur[2] - lead:
                why = ("%s: still running after its %.0f s limit on rank %d "
                       "(deadline; the remaining legs were dropped)"
                       % (cur[0], cur[2], self.rank))
            elif now - self.t0 > self.deadline_s - lead:
                why = ("%s: the run's %.0f s deadline passed on rank %d"
                       % (cur[0] if cur else "run", self.deadline_s,
                          self.rank))
            if why:
                self._deadline(why)
                return

    def _deadline(self, why):
        with self.lock:
            if self.finished:
                return
            self.finished = True
            self.failed.append(why)
            if self._current:
                self.seconds[self._current[0]] = round(
                    time.time() - self._current[1], 1)
            try:
                if self.rank == 0 and self.emit_final:
                    self.emit_final()
            finally:
                sys.stdout.flush()
                sys.stderr.write("bench.py rank %d: %s\n" % (self.rank, why))
                sys.stderr.flush()
                os._exit(0)

    def finish(self):
        """healthy end: rank 0 prints the final line (once)"""
        with self.lock:
            if self.finished:
                return False
            self.finished = True
            if self.rank == 0 and self.emit_final:
                self.emit_final()
            sys.stdout.flush()
            return True

    # ---------------------------------------------------------------- legs
    def announce(self, names):
        """legs that will be attempted (for the provisional line)"""
        self.pending = list(names)

    def run(self, name, run, prepare=(), start_by=None, collective=True,
            limit_s=LEG_LIMIT_S, cleanup=None):
        """One optional leg.  prepare: callables of LOCAL work, called in
        order, each followed by the ranks' agreement on failure; their results
        are passed to run(*results).  A leg never raises: a failure is named
        in `failed` (with the ranks it happened on) and None is returned.
        cleanup(*results so far) always runs (release what prepare built)."""
        if name in self.pending:
            self.pending.remove(name)
        start_by = self.budget_s if start_by is None else start_by
        collective = collective and self.use_dist
        if collective and os.environ.get("SPMV_BENCH_LEG_LIMIT"):
            limit_s = float(os.environ["SPMV_BENCH_LEG_LIMIT"])  # tests
        if collective and self.broken:
            self.skipped.append("%s (an earlier collective failed)" % name)
            return None
        try:
            late = self.max_over_ranks(self.spent()) if collective \
                else self.spent()
        except Exception as e:  # noqa: BLE001 - the group is gone
            self.broken = True
            self.failed.append("%s: budget agreement failed: %r" % (name, e))
            return None
        if late > start_by:
            self.skipped.append("%s (%.0f s spent, starts by %.0f s)"
                                % (name, late, start_by))
            return None
        t0 = time.time()
        self._current = (name, t0, limit_s)
        built = []
        try:
            for k, step in enumerate(prepare):
                err = None
                try:
                    if k == 0:
                        injected(name, self.rank, "prepare")
                    built.append(step(*built))
                except (Exception, SystemExit) as e:  # noqa: BLE001 - an
                    err = e                           # optional figure
                bad = self._agree(name, err) if collective else \
                    ([self.rank] if err is not None else [])
                if bad is None:
                    return None
                if bad:
                    self.failed.append(
                        "%s: preparation failed on rank(s) %s%s; every rank "
                        "skipped the leg" % (
                            name, bad, ": %r" % (err,) if err is not None
                            else ""))
                    return None
            err = result = None
            try:
                injected(name, self.rank, "run")
                result = run(*built)
            except (Exception, SystemExit) as e:  # noqa: BLE001 (a leg's
                err = e                 # own parity check may SystemExit)
            bad = self._agree(name, err) if collective else \
                ([self.rank] if err is not None else [])
            if bad is None:
                return None
            if bad:
                self.failed.append("%s: failed on rank(s) %s%s" % (
                    name, bad, ": %r" % (err,) if err is not None else ""))
                return None
            return result
        finally:
            self._current = None
            self.seconds[name] = round(time.time() - t0, 1)
            if cleanup is not None:
                try:
                    cleanup(*built)
                except Exception as e:  # noqa: BLE001
                    self.failed.append("%s: cleanup: %r" % (name, e))

    def _agree(self, name, err):
        """-> ranks that failed; None when the agreement itself failed (the
        group is then marked broken and the failure recorded)"""
        try:
            return self.ranks_where(err is not None)
        except Exception as e:  # noqa: BLE001 - a peer is gone
            self.broken = True
            self.failed.append("%s: %s; then the ranks could not agree: %r"
                               % (name, "failed here: %r" % (err,)
                                  if err is not None else "fine here", e))
            return None
