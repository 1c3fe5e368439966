"""How well do K3's waves pack the chip?  Event simulation of a headline step from the measured IPM iteration counts (CPU only).
    python tools/k3_packing_sim.py gpurun_out/<tag>/k3_iters_rocket_landing_4096.npz [slots = 2048] [ms per wave-iteration cold, warm = 3.6, 4.8]
Input: the [iter_max][B] matrix of IPM iterations per launch and problem (tools/ipm_iter_stats.py with K3_ITERS_DUMP).  A wave's duration is
its iteration count x the wave-iteration time (first launch: cold, no refinement; later launches: warm, refinement on).  Policies:
  lockstep(P)   -- today's loop: P sub-batches, one stream each; a stream's next launch starts when ITS slowest wave has finished;
                   blocks of concurrent launches take free slots in submission order;
  lpt(P)        -- the same with the blocks of a launch ordered by decreasing iteration count (an oracle for any predictor);
  capped(P, C)  -- a launch ends a wave after C iterations, the solve continues in the stream's next launch (other problems move on);
  async         -- every problem runs its own loop (a persistent kernel): list scheduling of the per-problem chains.
Prints the makespan of each against the work bound (total wave time / slots) and the chain bound (slowest problem alone)."""
import heapq
import sys
import numpy as np


def run_launches(queues, slots):
    """queues: per stream a list of launches, a launch = array of block durations (ms) in dispatch order.  A stream's launch k + 1 is
    submitted when all blocks of its launch k have finished (+ gap).  Free slots go to the pending launch submitted first."""
    free = [0.0] * slots
    heapq.heapify(free)
    t_ready = [0.0] * len(queues)          # time the stream's next launch may start
    nxt = [0] * len(queues)
    end = 0.0
    while True:
        cand = [(t_ready[s], s) for s in range(len(queues)) if nxt[s] < len(queues[s])]
        if not cand:
            break
        t_sub, s = min(cand)
        blocks = queues[s][nxt[s]]
        nxt[s] += 1
        fin = t_sub
        for d in blocks:
            t0 = max(heapq.heappop(free), t_sub)
            heapq.heappush(free, t0 + d)
            fin = max(fin, t0 + d)
        t_ready[s] = fin + 1.0              # K4 / K1 / K2 / copies between two K3 launches of a stream: ~1 ms when nothing else runs
        end = max(end, fin)
    return end


def main():
    d = np.load(sys.argv[1])
    it = d["iters"].astype(float)            # [J][B]
    slots = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
    tc, tw = (float(v) for v in sys.argv[3].split(",")) if len(sys.argv) > 3 else (3.6, 4.8)
    J, B = it.shape
    tit = np.full(J, tw); tit[0] = tc
    dur = it * tit[:, None]                  # wave durations, ms
    work = dur.sum() / slots
    chain = dur.sum(axis=0).max()
    print("J = %d launches of B = %d problems; IPM iterations mean %.1f, per-launch max %s" % (J, B, it.mean(), it.max(axis=1).astype(int).tolist()))
    print("work bound %.0f ms (total wave time / %d slots), chain bound %.0f ms (slowest problem alone), sum of per-launch maxima %.0f ms"
          % (work, slots, chain, dur.max(axis=1).sum()))
    if "k3_ms" in d:
        print("measured K3 launch times of this run (ONE stream of %d): total %.0f ms" % (B, d["k3_ms"].sum()))
    for P in (1, 2, 4, 8):
        parts = np.array_split(np.arange(B), P)
        q = [[dur[j, p] for j in range(J)] for p in parts]
        print("lockstep(%d streams): %.0f ms" % (P, run_launches(q, slots)))
        q = [[np.sort(dur[j, p])[::-1] for j in range(J)] for p in parts]
        print("lpt(%d streams):      %.0f ms" % (P, run_launches(q, slots)))
    for P in (2, 4):
        parts = np.array_split(np.arange(B), P)
        for C in (8, 12, 16, 24):
            q = []
            rounds = 0
            for p in parts:
                rem = it[:, p].copy()             # remaining iterations of the current solve of every problem
                jcur = np.zeros(len(p), dtype=int)
                left = rem[0].copy()
                launches = []
                while (jcur < J).any():
                    act = jcur < J
                    run = np.where(act, np.minimum(left, C), 0.0)
                    # problems whose solve needs 0 iterations still take a launch (set-up + exit): 0.2 of an iteration
                    launches.append((np.maximum(run, 0.2 * act) * np.where(jcur == 0, tc, tw))[act])
                    left = left - run
                    done = act & (left <= 0)
                    jcur = jcur + done
                    nxtj = np.minimum(jcur, J - 1)
                    left = np.where(done, rem[nxtj, np.arange(len(p))], left)
                q.append(launches)
                rounds = max(rounds, len(launches))
            print("capped(%d streams, C = %d): %.0f ms in %d rounds" % (P, C, run_launches(q, slots), rounds))
    # async: per-problem chains, list scheduling (a problem's next solve is ready when its previous one ends)
    free = [0.0] * slots
    heapq.heapify(free)
    ev = [(0.0, b, 0) for b in range(B)]
    heapq.heapify(ev)
    end = 0.0
    while ev:
        t, b, j = heapq.heappop(ev)
        t0 = max(heapq.heappop(free), t)
        t1 = t0 + dur[j, b]
        heapq.heappush(free, t1)
        end = max(end, t1)
        if j + 1 < J:
            heapq.heappush(ev, (t1, b, j + 1))
    print("async (per-problem loops): %.0f ms" % end)


if __name__ == "__main__":
    main()
