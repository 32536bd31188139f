"""Rank-0 console + file logger with the method set ev-NSFnet/train.py uses
(logger.py:8-54: info / warning / error / header / stage / close, get_logger)."""
import os
import time


class RankZeroLogger:
    def __init__(self, name="PINN", rank=0, log_dir="logs"):
        self.name, self.rank, self.t0, self.fh = name, rank, time.time(), None
        if rank == 0 and log_dir:
            os.makedirs(log_dir, exist_ok=True)
            self.fh = open(os.path.join(log_dir, "%s_%s.log" % (name, time.strftime("%Y%m%d_%H%M%S"))), "w",
                           encoding="utf-8")

    def _write(self, tag, msg):
        if self.rank:
            return
        line = "%-5s| %s" % (tag, msg)
        print(line, flush=True)
        if self.fh:
            self.fh.write(line + "\n")
            self.fh.flush()

    def info(self, msg): self._write("INFO", msg)
    def warning(self, msg): self._write("WARN", msg)
    def error(self, msg): self._write("ERROR", msg)

    def header(self, title):
        for s in ("=" * 60, title, "=" * 60):
            self.info(s)

    def stage(self, name, alpha, epochs, lr):
        self.info("%s: alpha=%s, epochs=%s, lr=%.2e" % (name, alpha, format(epochs, ","), lr))

    def close(self):
        if self.fh:
            self.fh.close()
            self.fh = None


_SINGLETON = None


def get_logger(name="PINN", rank=0):
    global _SINGLETON
    if _SINGLETON is None:
        _SINGLETON = RankZeroLogger(name, rank)
    return _SINGLETON
