"""The hand-over protocol of k_tile_chain (pyipm_amd/csrc/kernels_chain.hpp) as a discrete-event model, on the CPU.

The kernel runs the tile steps of a diagonal block as one launch of persistent workgroups: workgroup 0 (the chain) and units
(r, y) that own row tiles.  Nobody waits at a launch boundary any more -- every read of another workgroup's data is guarded by
progress words (chain: tiles inverted; unit: stages completed).  This model replays the kernel's wait conditions and its
read / write sets under RANDOM interleavings of the workgroups and asserts that whatever a stage reads has been written by
then -- inv(T), the -S rows W of a finished column tile, a row tile's diagonal tile -- and that the two conditions
factor_block hands to k_chain_wait (pyipm_newton.hip) cover what the rows kernels and the forward substitution of a sub-panel
read.  It checks the LOGIC of the protocol (ownership, stage counting, pieces of a chain); visibility across XCDs is the GPU
suite's business (tests/test_gpu_tile_blocked.py, tools/chain_stress.py).  Replaces the launch boundaries of the stepped
schedule that replaces LAPACK inside pyipm.py:1720-1721."""
import random

import pytest


def chain_ny(r, ta, cpy, nT=1 << 30):          # kernels_chain.hpp:chain_ny (row tiles r >= nT: extra rows below the diagonal block)
    rc = r if r < nT else nT - 1
    cols = rc - (ta - 1 if ta > 0 else 0)
    ny = (cols + cpy - 1) // cpy
    return 1 if ny < 1 else (4 if ny > 4 else ny)


def chain_first_row(ta):                       # kernels_chain.hpp:chain_first_row
    return ta + 1 if ta > 0 else 1


class Launch(object):
    """One k_tile_chain launch over the steps [ta, tb) of a block of nT tiles; `state` carries what earlier launches left."""

    def __init__(self, nT, ta, tb, cpy, state, nR=None):
        self.nT, self.ta, self.tb, self.cpy, self.st = nT, ta, tb, cpy, state
        self.nR = nT if nR is None else nR      # nR > nT (ta = 0, tb = nT): extra row tiles that take EVERY stage in this launch
        self.sfirst = ta - 1 if ta > 0 else -1
        self.rfirst = chain_first_row(ta)
        self.crit = ta                          # progress word of the chain: tiles inverted so far (word - base)
        self.t = ta                             # next step of the chain
        self.units = {}
        for r in range(self.rfirst, self.nR):
            for y in range(chain_ny(r, ta, cpy, nT)):
                s0 = ta - 1 if ta > 0 else 0
                self.units[(r, y)] = {"done": 0, "next": (-1 if ta == 0 else s0), "last": (tb - 1 if r >= nT else min(r - 2, tb - 2))}

    # ---- what the kernel polls ----
    def _need(self, tp):
        return tp - self.sfirst

    def chain_ready(self):
        t = self.t
        if t >= self.tb:
            return False
        if t == 0 or t < self.rfirst:
            return True
        need = self._need(t - 1)
        return need <= 0 or all(self.units[(t, y)]["done"] >= need for y in range(chain_ny(t, self.ta, self.cpy)))

    def unit_ready(self, key):
        r, y = key
        u = self.units[key]
        tp = u["next"]
        if tp == -1:
            return True                                            # saving W of column tile 0: nothing to wait for
        if tp > u["last"]:
            return False
        if tp >= self.ta and self.crit < tp + 1:
            return False
        need = self._need(tp)
        if need <= 0:
            return True
        ny = chain_ny(r, self.ta, self.cpy, self.nT)
        rc = min(r, self.nT - 1)
        for v in list(range(tp + 1, rc + 1)) + ([r] if r > rc else []):
            if not (v == r or v % ny == y):
                continue
            if v == r and tp % ny == y:
                continue                                           # my own column tile
            if v < self.rfirst:
                continue
            yo = tp % chain_ny(v, self.ta, self.cpy, self.nT)
            if self.units[(v, yo)]["done"] < need:
                return False
        return True

    # ---- what the kernel reads and writes ----
    def run_chain(self):
        t, st = self.t, self.st
        if t > 0:
            assert st["inv"][t - 1], ("chain", t, "inv(T) of the tile before")
            assert st["W"][(t, t - 1)], ("chain", t, "S of its rows in column tile t - 1")
            assert st["C"][(t, t)] == t - 1, ("chain", t, "diagonal tile with the stages before t - 1", st["C"][(t, t)])
            st["L"][(t, t - 1)] = True
            st["C"][(t, t)] = t                                     # stage t - 1 applied
        st["inv"][t] = True
        self.t += 1
        self.crit = self.t

    def run_unit(self, key):
        r, y = key
        u, st = self.units[key], self.st
        tp = u["next"]
        ny = chain_ny(r, self.ta, self.cpy, self.nT)
        if tp == -1:
            if y == 0:
                st["W"][(r, 0)] = True
            u["next"] = 0
        else:
            assert st["inv"][tp], (key, tp, "inv(T)")
            assert st["W"][(r, tp)], (key, tp, "S of the row tile in column tile tp")
            if y == 0:
                st["L"][(r, tp)] = True
            for v in range(tp + 1, min(r, self.nT - 1) + 1):
                if v % ny != y:
                    continue
                if v < r:
                    assert st["W"][(v, tp)], (key, tp, "Wn operand of column tile", v)
                assert st["C"][(r, v)] == tp, (key, tp, v, "its column tile has exactly the stages before", st["C"][(r, v)])
                st["C"][(r, v)] = tp + 1
                if v == tp + 1:
                    st["W"][(r, v)] = True                          # column tile tp + 1 of row r is final
            u["next"] = tp + 1
        u["done"] += 1

    def finished(self):
        return self.t >= self.tb and all(u["next"] > u["last"] for u in self.units.values())


def fresh_state(nT, nR=None):
    nR = nT if nR is None else nR
    return {"inv": [False] * nT, "W": {(r, c): False for r in range(nR) for c in range(nT)},
            "L": {(r, c): False for r in range(nR) for c in range(nT)}, "C": {(r, v): 0 for r in range(nR) for v in range(nT)}}


def rows_wait_ok(L, toff_next):
    """k_chain_wait(crit_need = toff, row0 = toff, unit_need = toff): in front of a sub-panel's rows kernels."""
    if L.crit < toff_next:
        return False
    return all(u["done"] >= toff_next for (r, y), u in L.units.items() if r >= toff_next)


def whole_wait_ok(L, toff_next):
    """k_chain_wait(tq, tq, tq), tq = toff + 1: in front of whoever reads the sub-panel as a whole (forward substitution)."""
    tq = toff_next + 1
    if L.crit < tq:
        return False
    return all(u["done"] >= tq for (r, y), u in L.units.items() if r >= tq)


@pytest.mark.parametrize("nT,sub,cpy", [(4, 2, 5), (8, 4, 5), (16, 4, 5), (16, 4, 2), (32, 4, 5), (32, 4, 9), (12, 4, 1), (5, 4, 5)])
def test_one_launch_per_block_under_random_interleavings(nT, sub, cpy):
    rnd = random.Random(nT * 100 + cpy)
    toffs = list(range(sub, nT, sub))                               # first tile of every sub-panel but the first
    for trial in range(20):
        st = fresh_state(nT)
        L = Launch(nT, 0, nT, cpy, st)
        rows_released, whole_released = set(), set()
        while not L.finished():
            ready = [("chain", None)] if L.chain_ready() else []
            ready += [("unit", k) for k in L.units if L.unit_ready(k)]
            assert ready, "deadlock"
            kind, key = rnd.choice(ready) if rnd.random() < 0.8 else ready[-1]     # (sometimes starve the chain: far rows first)
            L.run_chain() if kind == "chain" else L.run_unit(key)
            for tn in toffs:
                k0 = tn - sub                                                    # the sub-panel's tiles: [k0, tn)
                if tn not in rows_released and rows_wait_ok(L, tn):
                    rows_released.add(tn)
                    # rows kernels of the sub-panel: inv(T) of its tiles; W(v, t) inside the sub-panel; W of every LATER row tile
                    # in the sub-panel's column tiles (in-block updates); W of its row tiles in every EARLIER column tile
                    assert all(st["inv"][t] for t in range(k0, tn))
                    assert all(st["W"][(v, t)] for t in range(k0, tn) for v in range(t + 1, nT)), (tn, "W of the sub-panel's columns")
                    assert all(st["W"][(v, t)] for v in range(k0, tn) for t in range(0, min(v, k0)))
                if tn not in whole_released and whole_wait_ok(L, tn):
                    whole_released.add(tn)
                    assert all(st["L"][(r, t)] for t in range(k0, tn) for r in range(t + 1, nT)), (tn, "L of the sub-panel's columns")
        assert all(st["inv"]) and set(toffs) <= rows_released and set(toffs) <= whole_released
        assert all(st["C"][(r, r)] == r for r in range(nT))                      # every diagonal tile got exactly its r stages
        assert all(st["L"][(r, t)] for r in range(nT) for t in range(r))


@pytest.mark.parametrize("nT,sub,cpy", [(16, 4, 5), (32, 4, 3), (8, 2, 5)])
def test_a_chain_in_pieces_hands_the_state_from_launch_to_launch(nT, sub, cpy):
    """chain_whole = 0: one launch per sub-panel piece [ta, tb), tb = first tile of the next sub-panel + 1 (factor_block)."""
    rnd = random.Random(7 * nT + cpy)
    for trial in range(10):
        st = fresh_state(nT)
        ta = 0
        bounds = [min(t + 1, nT) for t in range(sub, nT, sub)] + [nT]
        for tb in bounds:
            if tb <= ta:
                continue
            L = Launch(nT, ta, tb, cpy, st)
            while not L.finished():
                ready = [("chain", None)] if L.chain_ready() else []
                ready += [("unit", k) for k in L.units if L.unit_ready(k)]
                assert ready, ("deadlock", ta, tb)
                kind, key = rnd.choice(ready)
                L.run_chain() if kind == "chain" else L.run_unit(key)
            # the state of the launch-per-tile schedule after launch tb - 1: every row tile >= tb has the stages up to tb - 2
            assert all(st["C"][(r, v)] == min(tb - 1, v) for r in range(tb, nT) for v in range(tb - 1, r + 1)), (ta, tb)
            ta = tb
        assert all(st["inv"]) and all(st["L"][(r, t)] for r in range(nT) for t in range(r))


@pytest.mark.parametrize("nT,nX,cpy", [(4, 4, 5), (4, 2, 5), (8, 4, 2), (16, 8, 5), (2, 4, 5), (4, 4, 1)])
def test_extra_rows_below_the_block_take_every_stage_in_the_same_launch(nT, nX, cpy):
    """ChainGeo::nR > nT (the per-panel schedule across GPUs, dist_slices = 2): row tiles below the diagonal block whose units
    apply every stage 0 .. nT - 1 to the block's column tiles as the chain publishes the tiles -- in place of a k_panel_rest
    launch behind the chain.  (They start on a word of their own, set when their rows' head is in place: any time, here.)"""
    rnd = random.Random(31 * nT + nX + cpy)
    nR = nT + nX
    for trial in range(20):
        st = fresh_state(nT, nR)
        L = Launch(nT, 0, nT, cpy, st, nR)
        held = set(k for k in L.units if k[0] >= nT) if trial % 2 else set()     # odd trials: the extra rows are released late
        steps = 0
        while not L.finished():
            ready = [("chain", None)] if L.chain_ready() else []
            ready += [("unit", k) for k in L.units if k not in held and L.unit_ready(k)]
            if not ready or (held and steps > 3 * nT):
                assert held, "deadlock"
                held = set()                                                    # the word goes up
                continue
            kind, key = rnd.choice(ready)
            L.run_chain() if kind == "chain" else L.run_unit(key)
            steps += 1
        assert all(st["inv"])
        assert all(st["L"][(r, t)] for r in range(nR) for t in range(min(r, nT)))
        assert all(st["C"][(r, v)] == v for r in range(nT, nR) for v in range(nT))    # column tile v got exactly its v stages
        assert all(st["C"][(r, r)] == r for r in range(nT))
