#!/usr/bin/env python3
"""Timeline of k_best8's work units from a UGP_STATS=1 UGP_TRACE=<file> run (one 48-byte record per unit:
wave, tile, chunk range, acquire / start / end on the device-wide 100 MHz clock, restarts, splits).
    python tools/analysis/unit_trace.py gpurun_out/trace.bin
Prints the launch span, how many waves hold a unit over time, where the time of the last waves goes, and the
duration of units by kind."""
import sys
import numpy as np

r = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 6)
wave = (r[:, 0] >> 32).astype(np.int64)
tile = ((r[:, 0] & 0xFFFFFFFF) >> 4).astype(np.int64)
heavy = (r[:, 0] & 1).astype(bool)
dyn = ((r[:, 0] >> 1) & 1).astype(bool)
c0 = (r[:, 1] >> 32).astype(np.int64)
c_end = (r[:, 1] & 0xFFFFFFFF).astype(np.int64)
c1 = (r[:, 5] & 0xFFFFFF).astype(np.int64)
t_pull, t_start, t_end = (r[:, k].astype(np.int64) for k in (2, 3, 4))
restarts = (r[:, 5] >> 32).astype(np.int64)
splits = ((r[:, 5] >> 24) & 0xFF).astype(np.int64)
t0 = t_pull.min()
us = lambda t: (t - t0) / 100.0
span = us(t_end.max())
n_waves = len(np.unique(wave))
print("units %d (own region %d, split-off %d), waves %d, span %.1f us" % (len(r), heavy.sum(), dyn.sum(), n_waves, span))
busy = (t_end - t_start).sum() / 100.0
wait = (t_start - t_pull).sum() / 100.0
print("in-unit time %.1f wave-ms, acquisition %.1f wave-ms, capacity %.1f wave-ms -> utilisation %.0f %%" % (busy / 1e3, wait / 1e3, span * n_waves / 1e3, 100 * busy / (span * n_waves)))
# active waves over time
nb = 40
edges = np.linspace(0, span, nb + 1)
act = np.zeros(nb)
for b in range(nb):
    lo, hi = edges[b], edges[b + 1]
    ov = np.clip(np.minimum(us(t_end), hi) - np.maximum(us(t_start), lo), 0, None)
    act[b] = ov.sum() / (hi - lo)
print("waves inside a unit, by %.0f-us slice:" % (span / nb))
print(" ".join("%d" % a for a in act))
for name, m in (("own region, static", heavy & ~dyn), ("own region, split-off", heavy & dyn), ("other, static", ~heavy & ~dyn), ("other, split-off", ~heavy & dyn)):
    if m.sum() == 0:
        continue
    d = (t_end - t_start)[m] / 100.0
    a = (t_start - t_pull)[m] / 100.0
    ch = (c_end - c0)[m]
    print("%-24s n=%6d  duration us: mean %7.1f p50 %7.1f p90 %7.1f max %8.1f | acquire mean %6.1f | chunks walked mean %6.1f (given %6.1f) | restarts mean %6.1f | splits %d"
          % (name, m.sum(), d.mean(), np.median(d), np.percentile(d, 90), d.max(), a.mean(), ch.mean(), (c1 - c0)[m].mean(), restarts[m].mean(), splits[m].sum()))
# the last units to finish
order = np.argsort(-t_end)[:12]
print("last units to end:")
for i in order:
    print("  end %8.1f us  start %8.1f  dur %7.1f  tile %3d chunks [%d,%d) of [%d,%d) %s%s restarts %d" % (us(t_end[i]), us(t_start[i]), (t_end[i] - t_start[i]) / 100.0, tile[i], c0[i], c_end[i], c0[i], c1[i], "own " if heavy[i] else "", "split-off" if dyn[i] else "static", restarts[i]))
# when does each wave end its last unit
last = np.zeros(wave.max() + 1)
np.maximum.at(last, wave, us(t_end))
last = last[last > 0]
print("waves' last unit ends: p10 %.0f p50 %.0f p90 %.0f max %.0f us" % tuple(np.percentile(last, [10, 50, 90, 100])))
