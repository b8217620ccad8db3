"""Skeleton graph -> stacked adjacency.  Mirrors ``net.utils.graph.Graph`` of the reference
(net/utils/graph.py:4-142): same constructor, same ``.A`` (K, V, V), computed here with vectorised masks."""
from collections import deque

import numpy as np


def get_hop_distance(num_nodes, edges, max_hop=1):
    """Pairwise hop count (BFS), inf beyond ``max_hop`` (graph.py:108-120)."""
    nbr = [set() for _ in range(num_nodes)]
    for a, b in edges:
        nbr[a].add(b)
        nbr[b].add(a)
    dis = np.full((num_nodes, num_nodes), np.inf)
    for s in range(num_nodes):
        dis[s, s] = 0
        q = deque([(s, 0)])
        seen = {s}
        while q:
            u, d = q.popleft()
            if d == max_hop:
                continue
            for v in nbr[u]:
                if v not in seen:
                    seen.add(v)
                    dis[s, v] = d + 1
                    q.append((v, d + 1))
    return dis


def normalize_digraph(A):
    """A . D^-1 with D the column sums (graph.py:123-131)."""
    col = A.sum(0)
    inv = np.divide(1.0, col, out=np.zeros_like(col, dtype=np.float64), where=col > 0)
    return A * inv[None, :]


class Graph:
    def __init__(self, num_nodes, neighbor_links, strategy='uniform', layout='openpose', max_hop=1, dilation=1):
        self.max_hop = max_hop
        self.dilation = dilation
        self.num_nodes = num_nodes
        self.center = 0
        self.edges = [(i, i) for i in range(num_nodes)] + list(neighbor_links)
        self.hop_dis = get_hop_distance(num_nodes, self.edges, max_hop=max_hop)
        self.set_adjacency(strategy)

    def set_adjacency(self, strategy):
        hops = list(range(0, self.max_hop + 1, self.dilation))
        dis = self.hop_dis
        reach = np.isin(dis, hops).astype(np.float64)
        norm = normalize_digraph(reach)
        if strategy == 'uniform':
            self.A = norm[None]
        elif strategy == 'distance':
            self.A = np.stack([np.where(dis == h, norm, 0.0) for h in hops])
        elif strategy == 'spatial':
            c = dis[:, self.center]
            cj, ci = c[:, None], c[None, :]          # A[j, i]: j = row (source), i = column
            parts = []
            for h in hops:
                at = np.where(dis == h, norm, 0.0)   # dis is symmetric
                root, close, further = at * (cj == ci), at * (cj > ci), at * (cj < ci)
                if h == 0:
                    parts.append(root)
                else:
                    parts += [root + close, further]
            self.A = np.stack(parts)
        else:
            raise ValueError('The given strategy does not exist')

    def __str__(self):
        return str(self.A)
