#!/usr/bin/env python3
"""What could another hand-out ORDER of the trace kernel's units buy?  (the round-3 review's item 3: last frame's unit costs
fed into the next frame's order; replaces the static schedule of screen.h:63-64.)

Input: what every 16 x 4-pixel unit of one launch cost the wave that traced it (PWN_OPT_WAVE_LOG + PWN_DBG_UNIT_COST=file:
u16 per unit, 40 ns each, pwn_trace_params.unit_cost).  The tool replays the kernel's scheduler on those costs -- W resident
waves, 64 queues, queue q holds the units u = q (mod 64), a wave's first unit is its place among the home waves of its queue,
the next ticket is drawn BEFORE the current unit is traced, a wave whose home queue is empty helps with the next open one --
for several orders of the units inside a queue:

  kernel     rows from the frame's middle row outwards (trace_kernel.hip), what the launch did
  lpt        longest unit first per queue, from these same costs (the best a cost-fed order can know: last frame = this frame)
  rows_lpt   unit ROWS by their total cost, longest first (a permutation small enough for LDS: H / 4 entries)

and prints each makespan beside the two bounds no order can beat: total work / W, and the longest single unit (a wave that
walks a mirror hall to the step limit is one dependent chain).

    python3 tools/unit_order_sim.py COSTFILE W H [Y0 Y1] [--waves N]
"""
import heapq
import sys

import numpy as np

Q = 64


def middle_out_row(k, rows_u, mid):
    a, b = mid, rows_u - 1 - mid
    m = min(a, b)
    j = (k + 1) >> 1
    if k <= 2 * m:
        return mid - j if (k & 1) else mid + j
    return mid - (k - b) if a > b else mid + (k - a)


def simulate(cost_of_ticket, nwaves, draw_ahead=True):
    """cost_of_ticket[q] = costs of queue q's units in hand-out order.  Returns (makespan, mean busy)."""
    pos = [0] * Q                         # next ticket of each queue
    qlen = [len(c) for c in cost_of_ticket]
    # static first tickets: wave i -> queue i % Q, ticket i // Q; the counters start behind them
    heap = []
    busy = 0.0
    waves = []
    for i in range(nwaves):
        q, t = i % Q, i // Q
        waves.append([q, t if t < qlen[q] else None])
    for q in range(Q):
        pos[q] = min(qlen[q], (nwaves + Q - 1 - q) // Q)

    def draw(q):
        """next ticket for a wave whose current queue is q (helping the next open one when q is empty): (queue, ticket) or None"""
        for d in range(Q):
            qq = (q + d) % Q
            if pos[qq] < qlen[qq]:
                t = pos[qq]
                pos[qq] += 1
                return qq, t
        return None

    # event loop: (time the wave is free, wave)
    state = []
    for i, (q, t) in enumerate(waves):
        if t is None:
            nxt = draw(q)
            if nxt is None:
                continue
            q, t = nxt
        nxt = draw(q) if draw_ahead else None
        c = cost_of_ticket[q][t]
        busy += c
        heapq.heappush(heap, (c, i))
        state.append(None)
        waves[i] = [q, nxt]
    end = 0.0
    while heap:
        now, i = heapq.heappop(heap)
        end = max(end, now)
        q, nxt = waves[i]
        if draw_ahead:
            if nxt is None:
                continue
            qq, t = nxt
            nn = draw(qq)
            waves[i] = [qq, nn]
        else:
            d = draw(q)
            if d is None:
                continue
            qq, t = d
            waves[i] = [qq, None]
        c = cost_of_ticket[qq][t]
        busy += c
        heapq.heappush(heap, (now + c, i))
    return end, busy / max(nwaves, 1)


def simulate_parked(cost_of_ticket, nwaves):
    """The kernel's order with the ticket drawn ahead PARKED where the three sibling waves of the workgroup can take it (DESIGN.md
    8, the round-3 review's item 7): a wave that runs dry takes a sibling's parked, not yet started unit instead of leaving; the
    sibling, done with its current unit, finds its ticket gone and draws again.  Upper bound: stealing costs nothing here."""
    pos = [0] * Q
    qlen = [len(c) for c in cost_of_ticket]
    for q in range(Q):
        pos[q] = min(qlen[q], (nwaves + Q - 1 - q) // Q)

    def draw(q):
        for d in range(Q):
            qq = (q + d) % Q
            if pos[qq] < qlen[qq]:
                t = pos[qq]
                pos[qq] += 1
                return qq, t
        return None
    cur_q = [i % Q for i in range(nwaves)]
    parked = [None] * nwaves
    heap = []
    busy = 0.0
    for i in range(nwaves):
        q, t = i % Q, i // Q
        first = (q, t) if t < qlen[q] else draw(q)
        if first is None:
            continue
        parked[i] = draw(first[0])
        c = cost_of_ticket[first[0]][first[1]]
        busy += c
        cur_q[i] = first[0]
        heapq.heappush(heap, (c, i))
    end = 0.0
    while heap:
        now, i = heapq.heappop(heap)
        end = max(end, now)
        nxt = parked[i]
        parked[i] = None
        if nxt is None:
            nxt = draw(cur_q[i])
        if nxt is None:
            # dry: a sibling's parked ticket (workgroup = waves 4k .. 4k + 3)
            g = i - (i & 3)
            for sidx in range(g, min(g + 4, nwaves)):
                if sidx != i and parked[sidx] is not None:
                    nxt = parked[sidx]
                    parked[sidx] = None
                    break
        if nxt is None:
            continue
        cur_q[i] = nxt[0]
        parked[i] = draw(nxt[0])
        c = cost_of_ticket[nxt[0]][nxt[1]]
        busy += c
        heapq.heappush(heap, (now + c, i))
    return end, busy / max(nwaves, 1)


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    nwaves = 5120
    if "--waves" in sys.argv:
        nwaves = int(sys.argv[sys.argv.index("--waves") + 1])
    cost = np.fromfile(args[0], np.uint16).astype(np.float64) * 0.04          # microseconds
    w, h = int(args[1]), int(args[2])
    y0, y1 = (int(args[3]), int(args[4])) if len(args) > 4 else (0, h)
    units_x = (w + 15) // 16
    rows_u = (y1 - y0 + 3) // 4
    units = units_x * rows_u
    assert len(cost) >= units, (len(cost), units)
    cost = cost[:units]
    mid = min(max(((h >> 1) - y0) >> 2, 0), rows_u - 1)
    # cost[unit] is indexed by the kernel's unit number (ticket * 64 + q), whose place in the frame is row middle_out(unit // units_x)
    sat = int((cost >= 65535 * 0.04 - 1e-9).sum())
    by_q = [[] for _ in range(Q)]
    for u in range(units):
        by_q[u % Q].append(cost[u])
    total, longest = float(cost.sum()), float(cost.max())
    print("%d units (%d x %d), %d waves, %d queues; unit cost mean %.2f us, median %.2f, p99 %.2f, max %.2f%s" % (
        units, units_x, rows_u, nwaves, Q, cost.mean(), np.median(cost), np.percentile(cost, 99), longest,
        " (%d saturated entries)" % sat if sat else ""))
    print("bounds: total work / waves = %.1f us, longest unit = %.1f us -> no order ends before %.1f us" % (total / nwaves, longest, max(total / nwaves, longest)))
    res = {}
    res["kernel (middle-out rows)"] = simulate(by_q, nwaves)
    res["kernel order, ticket drawn when the unit is done"] = simulate(by_q, nwaves, draw_ahead=False)
    res["kernel order, the drawn-ahead ticket parked for siblings"] = simulate_parked(by_q, nwaves)
    res["lpt per queue"] = simulate([sorted(c, reverse=True) for c in by_q], nwaves)
    res["lpt per queue, drawn when done"] = simulate([sorted(c, reverse=True) for c in by_q], nwaves, draw_ahead=False)
    # rows by total cost, longest first: unit (k, ux) of the kernel's numbering sits in row middle_out(k); renumber the rows
    row_cost = cost.reshape(rows_u, units_x).sum(axis=1)             # indexed by k (the hand-out rank of the row)
    order = np.argsort(-row_cost, kind="stable")
    re = cost.reshape(rows_u, units_x)[order].reshape(-1)
    by_q2 = [[] for _ in range(Q)]
    for u in range(units):
        by_q2[u % Q].append(re[u])
    res["rows by cost, longest first"] = simulate(by_q2, nwaves)
    base = res["kernel (middle-out rows)"][0]
    for name, (end, busy) in res.items():
        print("%-52s span %8.1f us  (%.3f of the kernel's order; mean wave busy %.1f us = %.2f of the span)" % (name, end, end / base, busy, busy / end))


if __name__ == "__main__":
    main()
