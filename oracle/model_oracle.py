"""Literal CPU restatement of the reference's model builders.  TEST INFRASTRUCTURE ONLY.

Every function takes the graph as ``nodes`` (list, in ``G.nodes`` order) and ``edges`` (list of
``(u, v, w)`` in ``G.edges`` order, w = float64 weight) -- exactly what the reference iterates --
and follows the cited reference lines statement by statement, including accumulation order, the
``defaultdict(int)`` start value and the DQM ``set_`` overwrite semantics.
"""
from collections import defaultdict
from itertools import combinations


def total_weight(edges):
    """``G.size(weight="weight")`` (BQM_clustering.py:29) -- networkx sums degree/2; the value stored
    in the fixture (total_weight_hex) is used for bit-exactness, this is the plain fallback."""
    return sum(w for _, _, w in edges)


def q_bqm(nodes, edges, gamma_factor, edges_weights=None, k=8):
    """BQM_clustering.py:29-47 (clustering_bqm).  Returns (Q dict, gamma)."""
    if edges_weights is None:
        edges_weights = total_weight(edges)
    nodes_len = len(nodes)
    gamma = gamma_factor * edges_weights / nodes_len          # :31
    Q = defaultdict(int)                                      # :36
    for u, v, w in edges:                                     # :38
        Q[(u, u)] += k * w                                    # :39
        Q[(v, v)] += k * w                                    # :40
        Q[(u, v)] += k * -2 * w                               # :41
    for i in nodes:                                           # :43
        Q[(i, i)] += gamma * (1 - len(nodes))                 # :44
    for i, j in combinations(nodes, 2):                       # :46
        Q[(i, j)] += 2 * gamma                                # :47
    return Q, gamma


def q_bqm_2(nodes, edges, gamma_factor, k, weights_sum=None):
    """BQM_clustering.py:210-236 (clustering_bqm_2).  Returns (Q dict, gamma, chain_strength)."""
    nodes_len = len(nodes)
    deg = defaultdict(int)
    for u, v, _ in edges:
        deg[u] += 1
        deg[v] += 1
    degrees = [deg[nd] for nd in nodes]                       # :212
    degrees_mean = sum(degrees) / len(degrees)                # :214
    weights = [w for _, _, w in edges]                        # :216
    if weights_sum is None:
        weights_sum = total_weight(edges)                     # :217
    weights_mean = sum(weights) / len(weights)                # :218
    chain_strength = weights_mean * degrees_mean * 2          # :220
    gamma = (weights_sum / nodes_len) * gamma_factor          # :222
    Q = defaultdict(int)                                      # :228
    for u, v, w in edges:                                     # :230
        Q[(u, u)] += k * w
        Q[(v, v)] += k * w
        Q[(u, v)] += k * -2 * w
    for i in nodes:                                           # :235
        Q[(i, i)] += gamma                                    # :236
    return Q, gamma, chain_strength


def q_bqm_3_cut_only(nodes, edges, k=8):
    """BQM_clustering.py:363-369 (clustering_bqm_3, the QUBO part before the slack constraint)."""
    Q = defaultdict(int)
    for u, v, w in edges:
        Q[(u, u)] += k * w
        Q[(v, v)] += k * w
        Q[(u, v)] += k * -2 * w
    return Q


def dqm_model(nodes, edges, num_of_clusters, gamma):
    """DQM_clustering.py:29-43 (clustering_dqm).  Returns (linear, quadratic):
    linear[node] = list of K biases; quadratic[(i, j)] = dict {(c, c): bias} with the reference's
    *set* (overwrite) semantics."""
    clusters = list(range(num_of_clusters))
    linear = {}
    quadratic = {}
    for node in nodes:                                        # :30-31 add_variable
        linear[node] = [0.0] * num_of_clusters
    for node in nodes:                                        # :33-34
        linear[node] = [gamma * (1 - len(nodes) / num_of_clusters) for _ in clusters]
    for i, j in combinations(nodes, 2):                       # :36-37
        quadratic[(i, j)] = {(c, c): 2 * gamma for c in clusters}
    for u, v, w in edges:                                     # :40-43
        key = (u, v) if (u, v) in quadratic else (v, u)
        quadratic[key] = {(c, c): -2 * w for c in clusters}   # set_quadratic overwrites
        linear[u] = [w for _ in clusters]                     # set_linear overwrites
        linear[v] = [w for _ in clusters]
    return linear, quadratic


def qubo_energy(Q, sample):
    """E = sum_{(u,v)} Q[u,v] x_u x_v in dict iteration order (the 'dict-sum' of SURVEY 8c)."""
    e = 0.0
    for (u, v), b in Q.items():
        if sample[u] and sample[v]:
            e += b
    return e


def bqm_energy_closed_form(nodes, edges, gamma, sample, k=8):
    """E = k*cut_w + gamma*(s^2 - n s)."""
    cut_w = sum(w for u, v, w in edges if sample[u] != sample[v])
    s = sum(1 for nd in nodes if sample[nd])
    n = len(nodes)
    return k * cut_w + gamma * (s * s - n * s)


def dqm_energy(linear, quadratic, labels):
    e = 0.0
    for nd, lab in labels.items():
        e += linear[nd][lab]
    for (i, j), tab in quadratic.items():
        li, lj = labels[i], labels[j]
        if (li, lj) in tab:
            e += tab[(li, lj)]
    return e


def cut_edges(edges, sample):
    return sum(1 for u, v, _ in edges if sample[u] != sample[v])


def slot_independent_order(rowptr, col, slot=64):
    """Restatement of the sweep-ordering pass of the native library (mi_sa_plan_slot_order): degree-descending
    greedy, each variable into the least-filled block that holds none of its neighbours (ties: lowest block), a
    block that does only as a last resort; blocks emitted in order, variables inside a block by original index."""
    import numpy as np
    rowptr = np.asarray(rowptr)
    col = np.asarray(col)
    n = len(rowptr) - 1
    nslots = (n + slot - 1) // slot
    if nslots <= 1 or n > (1 << 18):
        return np.arange(n, dtype=np.int64)
    cap = np.full(nslots, slot, dtype=np.int64)
    cap[-1] = n - slot * (nslots - 1)
    fill = np.zeros(nslots, dtype=np.int64)
    where = np.full(n, -1, dtype=np.int64)
    deg = np.diff(rowptr)
    big = np.int64(1) << 40
    for v in np.argsort(-deg, kind="stable"):
        key = fill * nslots + np.arange(nslots)              # least filled first, ties by slot index
        key = np.where(fill >= cap, 4 * big, key)            # full slots are never chosen
        nb = where[col[rowptr[v]:rowptr[v + 1]]]
        key[nb[nb >= 0]] += big                              # slots holding a neighbour: only as a last resort
        s = int(np.argmin(key))
        where[v] = s
        fill[s] += 1
    return np.lexsort((np.arange(n), where)).astype(np.int64)


def padded_slot_layout(rowptr, col, slot=64, max_slots=None):
    """Restatement of mi_sa_plan_slot_layout: the greedy pass of ``slot_independent_order`` with ``nslots`` blocks of
    ``slot`` seats each (no short last block), repeated with nslots = ceil(n / slot), then + max(1, nslots // 8) per
    try, until no variable shares a block with a neighbour; beyond ``max_slots`` the packed result stands; when holes were
    needed a DSATUR colouring is tried too and kept if it needs fewer blocks (graphs up to 2048 variables).  Returns
    (pos, nslots, clashes): pos[i] = block * slot + rank of i inside its block (by original index)."""
    import numpy as np
    rowptr = np.asarray(rowptr)
    col = np.asarray(col)
    n = len(rowptr) - 1
    s0 = (n + slot - 1) // slot
    if max_slots is None:
        max_slots = 3 * s0 + 4
    max_slots = max(max_slots, s0)
    order = np.argsort(-np.diff(rowptr), kind="stable")
    big = np.int64(1) << 40

    def greedy(nslots):
        fill = np.zeros(nslots, dtype=np.int64)
        where = np.full(n, -1, dtype=np.int64)
        clashes = 0
        for v in order:
            key = fill * nslots + np.arange(nslots)
            key = np.where(fill >= slot, 4 * big, key)
            nb = where[col[rowptr[v]:rowptr[v + 1]]]
            key[nb[nb >= 0]] += big
            s = int(np.argmin(key))
            clashes += int(key[s] >= big)
            where[v] = s
            fill[s] += 1
        return where, clashes

    nslots = s0
    first = None
    while True:
        where, clashes = greedy(nslots)
        if first is None:
            first = (where, clashes)
        if clashes == 0:
            break
        nxt = nslots + max(1, nslots // 8)
        if nxt > max_slots:
            where, clashes = first
            nslots = s0
            break
        nslots = nxt
    if clashes == 0 and nslots > s0 and n <= 2048:
        # small graphs: a saturation-degree colouring (DSATUR; ties by degree, then index; lowest colour with a free seat),
        # kept when it needs fewer blocks than the greedy layout
        C = nslots
        deg = np.diff(rowptr)
        seen = np.zeros((n, C), dtype=bool)
        sat = np.zeros(n, dtype=np.int64)
        colour = np.full(n, -1, dtype=np.int64)
        fill2 = np.zeros(C, dtype=np.int64)
        used, ok = 0, True
        for _ in range(n):
            cand = np.flatnonzero(colour < 0)
            key = sat[cand] * (int(deg.max()) + 1) + deg[cand]
            v = int(cand[np.argmax(key)])                       # (argmax returns the first maximum: lowest index)
            free = np.flatnonzero(~seen[v] & (fill2 < slot))
            if len(free) == 0:
                ok = False
                break
            c = int(free[0])
            colour[v] = c
            fill2[c] += 1
            used = max(used, c + 1)
            if used >= C:
                ok = False
                break
            nb = col[rowptr[v]:rowptr[v + 1]]
            fresh = nb[~seen[nb, c]]
            seen[fresh, c] = True
            np.add.at(sat, fresh, 1)
        if ok and s0 <= used < nslots:
            where, nslots = colour, used
    pos = np.empty(n, dtype=np.int64)
    seat = np.zeros(nslots, dtype=np.int64)
    for i in range(n):
        pos[i] = where[i] * slot + seat[where[i]]
        seat[where[i]] += 1
    return pos, nslots, clashes
