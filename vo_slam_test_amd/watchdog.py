"""Deadline for waits on other ranks (bench.py, tests/test_dist_gloo.py): a daemon thread that ENDS THIS PROCESS with status 70
when a collective that was announced with `arm()` has not been followed by `mark()` / `disarm()` for `timeout` seconds.  A hung
barrier or all-reduce must end the run, not the GPU box; os._exit ends the process, nothing is exec'ed (a process that has
initialised HIP must never be replaced by another program).  Host-side utility: no torch, no GPU."""
from __future__ import annotations

import os
import sys
import threading
import time


class CollectiveWatchdog:
    EXIT_STATUS = 70

    def __init__(self, timeout: float, rank: int = 0, poll: float = 2.0, exit_fn=None, name: str = "bench.py"):
        self.timeout, self.rank, self.poll, self.name = float(timeout), int(rank), float(poll), name
        self._exit = exit_fn if exit_fn is not None else os._exit
        self.t, self.what, self.armed = time.time(), "start", False
        self._thread = None

    def mark(self, what: str):
        """progress: a collective has completed (or is about to be waited for)"""
        self.t, self.what = time.time(), what

    def arm(self, what: str):
        """from here on the process waits for other ranks: the deadline runs"""
        self.mark(what)
        self.armed = True

    def disarm(self, what: str | None = None):
        if what is not None:
            self.mark(what)
        self.armed = False

    def expired(self) -> bool:
        return self.armed and time.time() - self.t > self.timeout

    def _run(self):
        while True:
            time.sleep(self.poll)
            if self.expired():
                sys.stderr.write(f"{self.name} rank {self.rank}: no collective completed for {self.timeout:.0f} s "
                                 f"(last: {self.what}) -- giving up\n")
                sys.stderr.flush()
                self._exit(self.EXIT_STATUS)
                return

    def start(self):
        if self._thread is None:
            self._thread = threading.Thread(target=self._run, daemon=True)
            self._thread.start()
        return self
