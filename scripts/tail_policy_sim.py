"""Developer: discrete-event model of the time-sliced tile queue on configs[1]'s per-tile evaluation counts
(gpurun_out/tail_policy_data.npz from scripts/tail_policy_data.py): makespan in evaluations under several policies."""
import sys, os, heapq
from collections import deque
import numpy as np
d = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "tail_policy_data.npz"))
E = d["n_eval_20"].astype(int); I = d["n_iter_20"].astype(int)
T = len(E); W = 512
print("tiles", T, "mean evals", E.mean(), "max", E.max(), "ideal makespan", E.sum() / W)
print("corr(n_eval_4, n_eval_20)", np.corrcoef(d["n_eval_4"], E)[0, 1], "corr(n_eval_6, n_eval_20)", np.corrcoef(d["n_eval_6"], E)[0, 1])

def sim(policy, slice_len=4, order=None, alone_speedup=1.0):
    # event-driven: a workgroup that becomes free takes the head of the ring (FIFO of (tile, evaluations done, time it was queued));
    # when the ring is empty it waits for the next tile to be queued.  alone_speedup is not modelled per CU: every evaluation is
    # one time unit.  Returns (makespan, busy fraction).
    ring = deque((t, 0, 0.0) for t in (order if order is not None else range(T)))
    idle = [(0.0, w) for w in range(W)]          # workgroups waiting for a tile, by the time they became free
    heapq.heapify(idle)
    running = []                                 # (slice end, w, tile, done after the slice)
    end = 0.0
    busy = 0.0
    while ring or running:
        # hand queued tiles to waiting workgroups
        while ring and idle:
            tf, w = heapq.heappop(idle)
            t, done, tq = ring.popleft()
            start = max(tf, tq)
            run = min(slice_len, E[t] - done)
            heapq.heappush(running, (start + run, w, t, done + run))
            busy += run
        if not running: break
        now, w, t, done = heapq.heappop(running)
        end = max(end, now)
        if done >= E[t]:
            heapq.heappush(idle, (now, w))
        elif (not ring) or policy(t, done):      # nobody waits, or predicted long: keep the tile
            run = min(slice_len, E[t] - done)
            heapq.heappush(running, (now + run, w, t, done + run))
            busy += run
        else:
            ring.append((t, done, now))
            heapq.heappush(idle, (now, w))
    return end, busy / (W * end)

print("run to completion, given order", sim(lambda t, dn: True))
print("round robin, slices of 4      ", sim(lambda t, dn: False))
print("LPT oracle, run to completion ", sim(lambda t, dn: True, order=np.argsort(-E)))
for thr in (10, 12, 14, 16, 20):
    print(f"keep a tile once it has used >= {thr} evaluations", sim(lambda t, dn, thr=thr: dn >= thr))

def sim_prio(key, slice_len=4):
    # the ring as a priority queue: a free workgroup takes the queued tile with the largest key(tile, evaluations done)
    import itertools
    cnt = itertools.count()
    pq = [(-key(t, 0), next(cnt), t, 0, 0.0) for t in range(T)]
    heapq.heapify(pq)
    idle = [(0.0, w) for w in range(W)]
    heapq.heapify(idle)
    running = []
    end = busy = 0.0
    while pq or running:
        while pq and idle:
            tf, w = heapq.heappop(idle)
            _, _, t, done, tq = heapq.heappop(pq)
            start = max(tf, tq)
            run = min(slice_len, E[t] - done)
            heapq.heappush(running, (start + run, w, t, done + run)); busy += run
        if not running: break
        now, w, t, done = heapq.heappop(running)
        end = max(end, now)
        if done >= E[t]:
            heapq.heappush(idle, (now, w))
        elif not pq:
            run = min(slice_len, E[t] - done)
            heapq.heappush(running, (now + run, w, t, done + run)); busy += run
        else:
            heapq.heappush(pq, (-key(t, done), next(cnt), t, done, now))
            heapq.heappush(idle, (now, w))
    return end, busy / (W * end)

for sl in (1, 2, 4, 8):
    print(f"round robin, slices of {sl}", sim(lambda t, dn: False, slice_len=sl))
rate = E / np.maximum(I, 1)
print("priority: predicted remaining = (20 - iterations done) x evaluations per iteration so far (uniform-rate model)",
      sim_prio(lambda t, dn: (20 - dn / rate[t]) * rate[t] if dn > 0 else 1e9))
print("priority: evaluations per iteration so far (slow tiles first)", sim_prio(lambda t, dn: rate[t] if dn > 0 else 1e9))
print("priority: true remaining (oracle)", sim_prio(lambda t, dn: E[t] - dn))
for tau in (12, 16, 20, 24, 28, 32, 40):
    print(f"FIFO ring, a tile is kept (not suspended) while (20 - iterations done) x evaluations per iteration so far >= {tau}",
          sim(lambda t, dn, tau=tau: (20 - dn / rate[t]) * rate[t] >= tau))

# the same rule with what a workgroup really knows at a slice end: iterations done as a function of evaluations done, piecewise
# linear through the tile's measured (iterations, evaluations) at budgets 2, 4, 6 and at its end
e2, e4, e6 = d["n_eval_2"].astype(float), d["n_eval_4"].astype(float), d["n_eval_6"].astype(float)
def iters_done(t, dn):
    xs = [0.0, e2[t], e4[t], e6[t], float(E[t])]
    ys = [0.0, min(2, I[t]), min(4, I[t]), min(6, I[t]), float(I[t])]
    for k in range(1, 5):
        if xs[k] <= xs[k - 1]: xs[k] = xs[k - 1] + 1e-9
    return float(np.interp(dn, xs, ys))
for tau in (12, 16, 20, 24, 28, 32):
    def pol(t, dn, tau=tau):
        it = max(iters_done(t, dn), 0.5)
        return (20 - it) * (dn / it) >= tau
    print(f"the same rule on measured early progress, threshold {tau}", sim(pol))
print("priority ring: fewest iterations done first", sim_prio(lambda t, dn: -iters_done(t, dn)))
print("priority ring: (20 - iterations done) x evals per iteration so far, measured progress",
      sim_prio(lambda t, dn: (20 - iters_done(t, dn)) * (dn / max(iters_done(t, dn), 0.5)) if dn > 0 else 1e9))
print("priority ring: (20 - iterations done) x evals per iteration after the first two iterations",
      sim_prio(lambda t, dn: ((20 - iters_done(t, dn)) * ((dn - e2[t]) / max(iters_done(t, dn) - 2, 0.5)) if dn > e2[t] else 1e9)))
for tau in (8, 12, 16, 20, 24):
    def pol2(t, dn, tau=tau):
        it = iters_done(t, dn)
        if dn <= e2[t] or it <= 2: return False
        return (20 - it) * ((dn - e2[t]) / (it - 2)) >= tau
    print(f"FIFO + keep, rate after the first two iterations, threshold {tau}", sim(pol2))

def sim_buckets(K, Bw, keyf, slice_len=4, first_slice=None):
    # K FIFO rings by predicted remaining evaluations (ring b: [b Bw, (b+1) Bw), the last one open-ended) + the ring of unstarted
    # tiles, served first; a free workgroup takes from the highest non-empty ring
    rings = [deque() for _ in range(K + 1)]
    for t in range(T): rings[K].append((t, 0, 0.0))
    idle = [(0.0, w) for w in range(W)]
    heapq.heapify(idle)
    running = []
    end = busy = 0.0
    def pop():
        for b in range(K, -1, -1):
            if rings[b]: return rings[b].popleft()
        return None
    nq = T
    while nq or running:
        while nq and idle:
            tf, w = heapq.heappop(idle)
            t, done, tq = pop(); nq -= 1
            start = max(tf, tq)
            run = min(first_slice if (first_slice and done == 0) else slice_len, E[t] - done)
            heapq.heappush(running, (start + run, w, t, done + run)); busy += run
        if not running: break
        now, w, t, done = heapq.heappop(running)
        end = max(end, now)
        if done >= E[t]:
            heapq.heappush(idle, (now, w))
        elif nq == 0:
            run = min(slice_len, E[t] - done)
            heapq.heappush(running, (now + run, w, t, done + run)); busy += run
        else:
            b = int(min(K - 1, max(0.0, keyf(t, done)) // Bw))
            rings[b].append((t, done, now)); nq += 1
            heapq.heappush(idle, (now, w))
    return end, busy / (W * end)

def key_meas(t, dn):
    it = max(iters_done(t, dn), 0.5)
    return (20 - it) * (dn / it)
for K, Bw in ((2, 24), (2, 32), (4, 16), (8, 8), (16, 4), (8, 12), (4, 24)):
    print(f"{K} rings of width {Bw} evaluations, measured progress", sim_buckets(K, Bw, key_meas), "slices of 8:", sim_buckets(K, Bw, key_meas, slice_len=8))
