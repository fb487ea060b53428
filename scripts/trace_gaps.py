#!/usr/bin/env python
"""Idle time between consecutive gradient launches of the iteration graphs, from a rocprofv3 kernel-trace database:
usage: trace_gaps.py <results.db>.  Prints the distribution of (start of gradient launch k+1) - (end of gradient launch k) and of
the launch durations, over the steady-state part of the run."""
import sqlite3
import sys

import numpy as np

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
g = [(s, e) for n, s, e in rows if "ppo_grad" in n]
g = g[len(g) // 4:]          # skip warm-up
gaps = np.array([g[i + 1][0] - g[i][1] for i in range(len(g) - 1)], dtype=np.float64) / 1e3
dur = np.array([e - s for s, e in g], dtype=np.float64) / 1e3
inner = gaps[gaps < 100]     # gaps inside a train phase (the rest are rollouts between iterations)
print(f"{len(g)} gradient launches; duration mean {dur.mean():.2f} us, median {np.median(dur):.2f}")
print(f"start(k+1) - end(k) inside a train phase: n {len(inner)}, mean {inner.mean():.2f} us, median {np.median(inner):.2f}, "
      f"p10 {np.percentile(inner, 10):.2f}, p90 {np.percentile(inner, 90):.2f}  (negative = the launches overlap)")
pitch = np.array([g[i + 1][0] - g[i][0] for i in range(len(g) - 1)], dtype=np.float64) / 1e3
print(f"launch pitch inside a train phase: mean {pitch[pitch < 100].mean():.2f} us")
