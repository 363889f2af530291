"""Timeline model of the blocked Cholesky of csrc/small.hip as a DATAFLOW between the waves (the round-4 experiment kept in
tools/experiments/potrf_flow.inc): wave 0 runs the serial chain [panel product of (k, k-1), column k-1 update of (k, k), 16x16
factorisation], NWK worker waves own the other tiles (owner = tile index mod NWK) and apply panel products, trailing updates,
the inverse's block substitution and the ride-along products in column order, handing tiles over through flags.  The model
executes exactly those per-wave programs with a cost per product (OP), per chain step (DIAG, W0OP) and per unsatisfied poll
(POLL); it reports the makespan, what the chain waited, and whether the programs can deadlock.  python tools/flow_model.py
Measured costs on MI355X (tools/worker_bench.py, in-kernel stamps): a product 760 cycles alone, ~1200 in situ; diag16 4000."""
import itertools, sys
def lt(i, j): return i * (i + 1) // 2 + j
def unpack(p):
    i = 0
    while (i + 1) * (i + 2) // 2 <= p: i += 1
    return i, p - i * (i + 1) // 2

def simulate(NT, NWK, with_inv, extra_cost, owner_a, owner_x, owner_g, OP=800, DIAG=4000, W0OP=500, POLL=150, verbose=False):
    NTT = NT * (NT + 1) // 2
    t_flag = {}                      # flag name -> time set
    def ready(names):                # time all flags are set (None if some missing)
        ts = [t_flag.get(n) for n in names]
        return None if any(t is None for t in ts) else (max(ts) if ts else 0)
    # programs: list of (waits, cost, sets)
    prog = {w: [] for w in range(NWK)}
    chain = []
    # wave 0
    chain.append(([], DIAG, [("inv", 0)]))
    for k in range(1, NT):
        chain.append(([("rdyS", k)], W0OP, [("L", k, k - 1)]))         # panel (k, k-1)
        chain.append(([("rdyD", k)], W0OP + DIAG, [("inv", k)]))       # update + diag16
    for w in range(NWK):
        P = prog[w]
        for c in range(NT):
            # (a) panel ops of own tiles (i, c), i >= c + 2
            for i in range(c + 2, NT):
                if owner_a(i, c) == w:
                    P.append(([("inv", c)], OP, [("L", i, c)]))
            # (b) X finals (c, i), i < c
            if with_inv:
                for i in range(c):
                    if owner_x(c, i) == w:
                        P.append(([("inv", c)], OP, [("X", c, i)]))
            # (c) A updates with column c, own tiles (i, j), j > c
            for j in range(c + 1, NT):
                for i in range(j, NT):
                    if owner_a(i, j) == w:
                        sets = []
                        if i == j and c == j - 2: sets.append(("rdyD", j))
                        if i == j + 1 and c == j - 1: sets.append(("rdyS", i))
                        if i == j and c == j - 1: continue               # wave 0's own update
                        waits = [("L", i, c)] + ([("L", j, c)] if j != i else [])
                        P.append((waits, OP, sets))
            # (d) X updates with column c: tiles (j, i), j > c >= i
            if with_inv:
                for j in range(c + 1, NT):
                    for i in range(c + 1):
                        if owner_x(j, i) == w:
                            waits = [("L", j, c)] + ([("X", c, i)] if i < c else [("inv", c)])
                            P.append((waits, OP, []))
            # (e) extra(c): own G tiles (i, j) with i <= c
            n = sum(1 for p in range(NTT) if owner_g(p) == w and unpack(p)[0] <= c)
            if n and extra_cost:
                waits = [("inv", c)] + ([("X", c, i) for i in range(c)] if with_inv else [])
                P.append((waits, extra_cost * n, []))
    # tiles with no worker updates are ready from the start
    for j in range(1, NT):
        if j - 2 < 0: t_flag[("rdyD", j)] = 0
    t_flag[("rdyS", 1)] = 0
    # L(i, i) is implied by inv(i); sub-diag L(k, k-1) by wave 0
    pcs = {w: 0 for w in range(NWK)}; tw = {w: 0.0 for w in range(NWK)}
    pc0 = 0; t0 = 0.0
    wait0 = 0.0
    progress = True
    while progress:
        progress = False
        if pc0 < len(chain):
            waits, cost, sets = chain[pc0]
            r = ready(waits)
            if r is not None:
                if r > t0: wait0 += r - t0 + POLL
                t0 = max(t0, r + (POLL if r > t0 else 0)) + cost
                for s in sets: t_flag[s] = t0
                pc0 += 1; progress = True
        for w in range(NWK):
            while pcs[w] < len(prog[w]):
                waits, cost, sets = prog[w][pcs[w]]
                r = ready(waits)
                if r is None: break
                tw[w] = max(tw[w], r + (POLL if r > tw[w] else 0)) + cost
                for s in sets: t_flag[s] = tw[w]
                pcs[w] += 1; progress = True
    dead = pc0 < len(chain) or any(pcs[w] < len(prog[w]) for w in range(NWK))
    if verbose:
        print("chain end", t0, "chain waited", wait0, "workers end", [int(tw[w]) for w in range(NWK)], "ops", [len(prog[w]) for w in range(NWK)])
    return dead, max([t0] + list(tw.values())), t0, wait0

if __name__ == "__main__":
    NT = 6
    for NWK in (6, 7):
        for name, oa in (("rowmajor", lambda i, j: lt(i, j) % NWK), ("colmajor", lambda i, j: (j * NT + i) % NWK), ("i+2j", lambda i, j: (i + 2 * j) % NWK), ("i-j", lambda i, j: (i - j) % NWK), ("i", lambda i, j: i % NWK)):
            for with_inv in (True, False):
                dead, mk, t0, w0 = simulate(NT, NWK, with_inv, 800, oa, oa, lambda p: p % NWK)
                print(f"NWK={NWK} owner={name:9s} inv={with_inv}: deadlock={dead} makespan={mk:.0f} chain={t0:.0f} chain waited {w0:.0f}")
