"""Host-side mirror of the reference's default ("medium") search path with the candidates'
distances computed in BULK on the GPU.

The reference walks its random-hyperplane forest with a priority queue of nodes
(lshTree.search, lshtree.go:283-351) and calls consider() (collection.go:583-629) for every
unvisited id of each leaf it reaches: one getDocument + decode + distance per candidate.  Here
the forest and the traversal stay on the host, unchanged -- the tree is the reference's data
structure -- and only the distances move: szg_distances (ScanIndex.distances) returns the
reference's own float64 value for a whole list of rows in one call.

Why the result is IDENTICAL, not "similar".  Inside the traversal, the candidates' distances
feed back only through two scalars: `radius` (read when a leaf on the far side of a hyperplane
is popped, :305-310) and `k_counter` (read when any node is popped, :312-314).  Neither changes
the ORDER in which nodes leave the queue -- that order depends on the query and the hyperplanes
alone -- they only decide which popped leaves are skipped and when the walk stops.  So the
sequence of leaves the reference scores is a subsequence (and a prefix in time) of the sequence
S the same walk produces with pruning and stopping switched off.  This module produces S
lazily, takes the next window of leaves from it, scores all their not-yet-visited ids in ONE
GPU call, and then replays lshTree.search's leaf loop and consider() over that window with the
reference's own pruning / stopping tests, point by point, in the reference's order.  Leaves the
replay prunes do not mark their ids visited (the reference `continue`s before its loop), ids of
leaves behind the stopping point are never looked at; their distances were speculative work,
not part of the answer.  Distances are the reference's float64 values bit for bit, the heap is
container/heap's, so Results, their order and PercentSearched equal the reference's on the
same forest.

go_acos restates Go's math.Acos (pure-Go Cephes code on amd64, math/asin.go, math/atan.go):
distanceToHyperplane (lshtree.go:55-74) uses it for the angular metric.
"""
import math

import numpy as np

EUCLIDEAN, COSINE = 0, 1
STOP_SEARCH, POINT_ACCEPTED, POINT_CHECKED, POINT_IGNORED = range(4)  # collection.go:19-24
SEARCH_K = 200  # lshtree.go:286


def _xatan(x):
    P0, P1, P2, P3, P4 = (-8.750608600031904122785e-01, -1.615753718733365076637e+01, -7.500855792314704667340e+01,
                          -1.228866684490136173410e+02, -6.485021904942025371773e+01)
    Q0, Q1, Q2, Q3, Q4 = (+2.485846490142306297962e+01, +1.650270098316988542046e+02, +4.328810604912902668951e+02,
                          +4.853903996359136964868e+02, +1.945506571482613964425e+02)
    z = x * x
    z = z * ((((P0 * z + P1) * z + P2) * z + P3) * z + P4) / (((((z + Q0) * z + Q1) * z + Q2) * z + Q3) * z + Q4)
    return x * z + x


def _satan(x):
    morebits, tan3pio8 = 6.123233995736765886130e-17, 2.41421356237309504880
    if x <= 0.66:
        return _xatan(x)
    if x > tan3pio8:
        return math.pi / 2 - _xatan(1 / x) + morebits
    return math.pi / 4 + _xatan((x - 1) / (x + 1)) + 0.5 * morebits


def go_acos(x):
    """math.Acos as Go computes it on amd64 (math/asin.go): Pi/2 - Asin(x)."""
    def asin(x):
        if x == 0:
            return x
        sign = x < 0
        if sign:
            x = -x
        if x > 1:
            return math.nan
        temp = math.sqrt(1 - x * x)
        temp = math.pi / 2 - _satan(temp / x) if x > 0.7 else _satan(x / temp)
        return -temp if sign else temp
    if x != x:
        return math.nan
    return math.pi / 2 - asin(x)


def _seq_dot(a, b):
    """sum of a[i]*b[i] accumulated left to right in float64 (numpy's cumsum is sequential)."""
    return float(np.cumsum(a * b)[-1]) if a.size else 0.0


def distance_to_hyperplane(method, vector, length, normal, b):
    """lshtree.go:55-74."""
    dist = _seq_dot(vector, normal) - b
    if method == EUCLIDEAN:
        if dist > 0:
            return dist, True
        return -dist, False
    with np.errstate(all="ignore"):
        ratio = float(np.float64(dist) / np.float64(length))  # IEEE division (x/0 -> inf / nan, as in Go)
    dist = go_acos(ratio) / math.pi
    if dist > 0.5:
        return 1 - dist, True
    return dist, False


class _GoHeap:
    """container/heap over (priority, payload) with Less = priority > (max-heap): the
    nodePriorityQueue of lshtree.go:353-381 and the resultPriorityQueue of collection.go:536-564."""

    def __init__(self):
        self.a = []

    def __len__(self):
        return len(self.a)

    def _less(self, i, j):
        return self.a[i][0] > self.a[j][0]

    def push(self, priority, payload):
        a = self.a
        a.append((priority, payload))
        j = len(a) - 1
        while True:
            i = (j - 1) // 2 if j > 0 else 0
            if i == j or not self._less(j, i):
                break
            a[i], a[j] = a[j], a[i]
            j = i

    def pop(self):
        a = self.a
        n = len(a) - 1
        a[0], a[n] = a[n], a[0]
        i = 0
        while True:
            j1 = 2 * i + 1
            if j1 >= n:
                break
            j = j1
            if j1 + 1 < n and self._less(j1 + 1, j1):
                j = j1 + 1
            if not self._less(j, i):
                break
            a[i], a[j] = a[j], a[i]
            i = j
        return a.pop()


class LshForest:
    """The reference's forest as flat arrays: roots[T]; per node left/right (-1 = leaf), normal[dim],
    b, and the leaf's ids (rows of the mirror)."""

    def __init__(self, roots, left, right, normals, b, ids_off, ids_cnt, ids, metric):
        self.roots = np.asarray(roots, dtype=np.int32)
        self.left = np.asarray(left, dtype=np.int32)
        self.right = np.asarray(right, dtype=np.int32)
        self.normals = np.asarray(normals, dtype=np.float64)
        self.b = np.asarray(b, dtype=np.float64)
        self.ids_off = np.asarray(ids_off, dtype=np.int64)
        self.ids_cnt = np.asarray(ids_cnt, dtype=np.int32)
        self.ids = np.asarray(ids, dtype=np.uint64)
        self.metric = int(metric)

    def leaf_ids(self, node):
        o = int(self.ids_off[node])
        return self.ids[o:o + int(self.ids_cnt[node])]


def _leaf_sequence(forest, query):
    """The leaves in the order lshTree.search pops them when nothing is pruned and nothing stops
    the walk, with the priority each was popped with: generator of (node, priority)."""
    q = np.asarray(query, dtype=np.float64)
    length = math.sqrt(float(np.cumsum(q * q)[-1])) if q.size else 0.0  # vectorLength, lshtree.go:30-36
    pq = _GoHeap()
    for r in forest.roots:
        pq.push(0.0, int(r))
    while len(pq):
        priority, node = pq.pop()
        if forest.left[node] < 0:
            yield node, priority
            continue
        dist, right = distance_to_hyperplane(forest.metric, q, length, forest.normals[node], float(forest.b[node]))
        l, r = int(forest.left[node]), int(forest.right[node])
        if right:
            pq.push(dist, r)
            pq.push(-dist, l)
        else:
            pq.push(dist, l)
            pq.push(-dist, r)


def search(forest, index, query, k=0, radius=0.0, allow=None, window_points=2048):
    """Search{Precision:"medium", K:k, Radius:radius} over `index` (a ScanIndex holding the rows the
    forest's ids name).  allow: bool per row (the Filter's verdicts) or None.
    Returns (rows uint64[n], dist float64[n], points_searched)."""
    q = np.ascontiguousarray(query, dtype=np.float64).reshape(-1)
    n_rows = index.rows
    visited = np.zeros(n_rows, dtype=bool)
    results = _GoHeap()                       # resultsPQ
    rad = radius if radius > 0 else 1.7976931348623157e308  # math.MaxFloat64, collection.go:686-689
    k_counter, point_accepted, searched = 0, False, 0
    seq = _leaf_sequence(forest, q)
    stopped = False
    while not stopped:
        # next window of the unpruned, unstopped leaf sequence; score its unseen ids in one call
        window, want, seen = [], [], set()
        for node, priority in seq:
            window.append((node, priority))
            for id_ in forest.leaf_ids(node):
                id_ = int(id_)
                if not visited[id_] and id_ not in seen:
                    seen.add(id_)
                    want.append(id_)
            if len(want) >= window_points:
                break
        if not window:
            break
        dist_of = {}
        if want:
            dist_of = dict(zip(want, index.distances(q, np.asarray(want, dtype=np.uint64))))
        # replay lshTree.search's loop body and consider() over the window, in order
        for node, priority in window:
            if priority < 0 and -priority > rad:      # lshtree.go:305-310 (the node is a leaf)
                continue
            if k_counter >= SEARCH_K:                  # :312-314
                stopped = True
                break
            for id_ in forest.leaf_ids(node):
                id_ = int(id_)
                if visited[id_]:
                    continue
                visited[id_] = True
                searched += 1                          # collection.go:589, before the filter
                signal = POINT_CHECKED
                if allow is not None and not allow[id_]:
                    signal = POINT_IGNORED             # :592-594
                else:
                    distance = float(dist_of[id_])
                    if radius > 0 and distance <= radius:      # :598-603
                        results.push(distance, id_)
                        signal = POINT_ACCEPTED
                    elif radius > 0:
                        signal = POINT_CHECKED                 # :604-605
                    elif k > 0:                                # :606-619
                        if len(results) <= k and (len(results) < k or results.a[0][0] > distance):
                            results.push(distance, id_)
                            if len(results) > k:
                                results.pop()
                            rad = results.a[0][0]              # :616
                            signal = POINT_ACCEPTED
                if signal == POINT_ACCEPTED:           # lshtree.go:323-334
                    k_counter = 0
                    point_accepted = True
                elif signal == POINT_CHECKED and point_accepted:
                    k_counter += 1
    n = len(results)
    rows = np.zeros(n, dtype=np.uint64)
    dist = np.zeros(n, dtype=np.float64)
    for i in range(n - 1, -1, -1):                     # collection.go:694-697
        d, r = results.pop()
        rows[i], dist[i] = r, d
    return rows, dist, searched
