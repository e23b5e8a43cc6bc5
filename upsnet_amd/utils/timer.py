"""Wall-clock timers with the reference's semantics (lib/utils/timer.py:16-42): tic()/toc() returning
the running average, plus the per-call samples so that p50 can be reported."""
import time


class Timer(object):
    def __init__(self):
        self.total_time = 0.
        self.calls = 0
        self.start_time = 0.
        self.diff = 0.
        self.average_time = 0.
        self.samples = []

    def tic(self):
        self.start_time = time.time()

    def toc(self, average=True):
        self.diff = time.time() - self.start_time
        self.total_time += self.diff
        self.calls += 1
        self.samples.append(self.diff)
        self.average_time = self.total_time / self.calls
        return self.average_time if average else self.diff

    def p50(self):
        s = sorted(self.samples)
        return s[len(s) // 2] if s else 0.
