"""Offline (no GPU): LDS cycles of the Chebyshev step's gathers (ds_read_b64: two 32-lane groups per wave instruction, bank
pair = (row slot) mod 32, equal addresses broadcast, N distinct addresses on one bank pair = N cycles) for the patch tables
dumped by scripts/dump_patch_tables.py, and what re-assigning the entries to slots / renumbering the rows would save."""
import sys

import numpy as np

z = np.load(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r03/patch_tables_3.npz")
NP, LS, W = int(z["NP"]), int(z["LS"]), int(z["W"])
pnloc, pnh, lcol = z["pnloc"], z["pnh"], z["lcol"].astype(np.int64)


def cycles(lc, nloc):
    """lc: [W][rows] addresses (slots); rows beyond nloc in an active wave read their own slot"""
    nw = -(-nloc // 64)
    tot = 0
    for w in range(nw):
        for half in range(2):
            r0 = 64 * w + 32 * half
            for q in range(W):
                a = np.unique(lc[q, r0:r0 + 32])
                tot += np.bincount(a % 32, minlength=32).max()
    return tot, nw * 2 * W


def greedy_slots(lc, nloc):
    """per 32-lane group, rows in order: assign the row's entries to the slots so that the group's slot-wise bank loads stay low"""
    out = lc.copy()
    nw = -(-nloc // 64)
    import itertools
    perms = list(itertools.permutations(range(W))) if W <= 6 else None
    for g in range(nw * 2):
        r0 = 32 * g
        load = np.zeros((W, 32), np.int64)              # distinct addresses per (slot, bank pair) so far
        seen = [set() for _ in range(W)]
        for r in range(r0, r0 + 32):
            ent = lc[:, r]
            best, bestp = None, None
            for pm in perms:
                # cost: sum over slots of the resulting max load increase (lexicographic: max load, then total)
                c = 0
                for s in range(W):
                    a = ent[pm[s]]
                    if a in seen[s]:
                        continue
                    c += (load[s, a % 32] + 1) ** 2
                if best is None or c < best:
                    best, bestp = c, pm
            for s in range(W):
                a = ent[bestp[s]]
                out[s, r] = a
                if a not in seen[s]:
                    seen[s].add(a); load[s, a % 32] += 1
    return out


tot0 = tot1 = tot2 = ideal = 0
ps = range(0, NP, 8)
for p in ps:
    nloc = int(pnloc[p])
    lc = lcol[p].copy()
    rows = np.arange(LS)
    for q in range(W):
        m = lc[q] < 0
        lc[q, m] = rows[m]
    c0, idl = cycles(lc, nloc)
    lch = np.where(lc >= LS, LS, lc)                      # every halo column -> one zero slot (broadcast)
    c1, _ = cycles(lch, nloc)
    c2, _ = cycles(greedy_slots(lch, nloc), nloc)
    tot0 += c0; tot1 += c1; tot2 += c2; ideal += idl
n = len(list(ps))
print(f"per patch and step (LDS cycles of the gathers): now {tot0 / n:.0f}, halo -> one slot {tot1 / n:.0f}, + greedy slot assignment {tot2 / n:.0f}, conflict-free {ideal / n:.0f}")


def greedy_entrywise(lc, nloc):
    """cheap form: a row's entries in their given order, each to the free slot where its bank pair is least loaded (an
    address the slot has already seen costs nothing)"""
    out = lc.copy()
    nw = -(-nloc // 64)
    for g in range(nw * 2):
        r0 = 32 * g
        load = np.zeros((W, 32), np.int64)
        seen = [set() for _ in range(W)]
        for r in range(r0, r0 + 32):
            free = list(range(W))
            for e in range(W):
                a = int(lc[e, r])
                best = min(free, key=lambda s: (0 if a in seen[s] else load[s, a % 32] + 1, s))
                free.remove(best)
                out[best, r] = a
                if a not in seen[best]:
                    seen[best].add(a); load[best, a % 32] += 1
    return out


tot3 = 0
for p in ps:
    nloc = int(pnloc[p])
    lc = lcol[p].copy()
    rows = np.arange(LS)
    for q in range(W):
        m = lc[q] < 0
        lc[q, m] = rows[m]
    lch = np.where(lc >= LS, LS, lc)
    tot3 += cycles(greedy_entrywise(lch, nloc), nloc)[0]
print(f"entry-wise greedy {tot3 / n:.0f}")


def greedy_matching(lc, nloc, passes=1):
    """per row: repeatedly the cheapest (entry, free slot) pair; cost 0 where the slot has the address already, else its bank load + 1"""
    out = lc.copy()
    nw = -(-nloc // 64)
    for g in range(nw * 2):
        r0 = 32 * g
        load = np.zeros((W, 32), np.int64)
        cnt = [dict() for _ in range(W)]                 # address -> rows that have it in this slot
        assign = {}
        for ps_ in range(passes):
            for r in range(r0, r0 + 32):
                if r in assign:                          # refinement: take the row out first
                    for s, a in enumerate(assign[r]):
                        cnt[s][a] -= 1
                        if cnt[s][a] == 0:
                            del cnt[s][a]; load[s, a % 32] -= 1
                ents = [int(x) for x in lc[:, r]]
                fe, fs = list(range(W)), list(range(W))
                res = [0] * W
                while fe:
                    best = None
                    for e in fe:
                        a = ents[e]
                        for s in fs:
                            c = 0 if a in cnt[s] else load[s, a % 32] + 1
                            if best is None or (c, e, s) < best[0]:
                                best = ((c, e, s), e, s)
                    _, e, s = best
                    fe.remove(e); fs.remove(s)
                    a = ents[e]
                    res[s] = a
                    if a not in cnt[s]:
                        cnt[s][a] = 0; load[s, a % 32] += 1
                    cnt[s][a] += 1
                assign[r] = res
        for r in range(r0, r0 + 32):
            out[:, r] = assign[r]
    return out


for passes in (1, 2):
    t = 0
    for p in ps:
        nloc = int(pnloc[p])
        lc = lcol[p].copy()
        rows = np.arange(LS)
        for q in range(W):
            m = lc[q] < 0
            lc[q, m] = rows[m]
        lch = np.where(lc >= LS, LS, lc)
        t += cycles(greedy_matching(lch, nloc, passes), nloc)[0]
    print(f"greedy matching, {passes} pass(es): {t / n:.0f}")
