#!/usr/bin/env python3
"""CPU model of the resident loop's timeline from the per-iteration solver-trip counts of every branch (scripts/trip_trace.py):
what would a deeper speculation past the stop decision buy?  Tile w of 64 branches (storage order: by from-bus) finishes iteration i
at start(w, i) + A + C * (largest trip count in the tile); a bus is ready when every tile with a branch or generator on it has finished
(+ H for the hand-over); a tile starts iteration i + 1 when the buses of all its branches are ready AND the stop decision of
iteration i - 1 - D is out (D = 0: today's loop; the decision of a pass is out when every tile has made that pass, + CT).
    python scripts/slack_model.py gpurun_out/trips/trips_case13659pegase.npz"""
import sys
import numpy as np

f = np.load(sys.argv[1])
trips, outer = f["trips"].astype(np.int64), f["outer"]
br = f["brBusIdx"].astype(np.int64)
br = br - br.min()
fr, to = br[0::2], br[1::2]
gb = f["gen_bus"].astype(np.int64)
gb = gb - min(gb.min(), 1) if gb.min() >= 1 else gb
nbus = int(f["nbus"])
L = fr.size
order = np.argsort(fr, kind="stable")
tile_of = np.empty(L, np.int64)
tile_of[order] = np.arange(L) // 64
nbt = (L + 63) // 64
ngt = (gb.size + 63) // 64
gen_tile = nbt + np.arange(gb.size) // 64
NT = nbt + ngt
# bus <- tiles incidence as pairs
pb = np.concatenate([fr, to, gb])
pt = np.concatenate([tile_of, tile_of, gen_tile])
pairs = np.unique(np.stack([pb, pt], 1), axis=0)
pb, pt = pairs[:, 0], pairs[:, 1]
A, Cc, H, PRO, CT, GEN = 11.0, 6.1, 1.5, 4.0, 3.0, 3.0   # us (profiles/r04_resident_wave_timeline.txt, r04_phase_cycles_resident_vs_two_launch.txt)


def simulate(D, rows):
    n = len(rows)
    start = np.zeros(NT)
    hist_pro = []           # time at which every tile has made the pass of index k
    t_end = 0.0
    for k, i in enumerate(rows):
        tmax = np.zeros(nbt)
        np.maximum.at(tmax, tile_of, trips[i])
        dur = np.concatenate([A + Cc * tmax, np.full(ngt, GEN)])
        hist_pro.append((start + PRO).max() + CT)
        fin = start + dur
        ready = np.zeros(nbus)
        np.maximum.at(ready, pb, fin[pt])
        ready += H
        nxt = np.zeros(NT)
        np.maximum.at(nxt, pt, ready[pb])
        kk = k - D      # decision needed before index k + 1 starts: the pass of index k - D
        if kk >= 0:
            nxt = np.maximum(nxt, hist_pro[kk])
        start = nxt
        t_end = fin.max()
    return t_end


tot = {}
for D in (0, 1, 2, 3, 1000):
    t = 0.0
    for o in np.unique(outer):
        rows = np.nonzero(outer == o)[0]
        t += simulate(D, rows)
    tot[D] = t
n = trips.shape[0]
print("%s: %d iterations, %d branch tiles + %d generator tiles" % (sys.argv[1], n, nbt, ngt))
print("sum over iterations of A + C * (largest trip count of the grid): %.2f ms" % (1e-3 * sum(A + Cc * trips[i].max() for i in range(n))))
for D, t in tot.items():
    print("speculation depth %4d: %.2f ms -> %.0f iterations/s" % (D, 1e-3 * t, n / (1e-6 * t)))
# how persistent are the stragglers?  the same branch the slowest in consecutive iterations
am = trips.argmax(1)
print("slowest branch unchanged from one iteration to the next: %.0f %% of the iterations" % (100.0 * (am[1:] == am[:-1]).mean()))
