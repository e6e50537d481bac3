"""Skeleton graph constants (reference: pyskl/utils/graph.py:58-187).

``A`` per mode, ``node_type`` (body part 0..4 per joint) and ``edge_type`` (class 0..14 per
ordered joint pair: the rank of the signed product of the two part codes) — the integer
tables K-B indexes its typed weight slices with.  ``mode='random'`` draws from the *numpy
global RNG* exactly like the reference (graph.py:185-187), so ``np.random.seed(s)`` before
``build_model`` reproduces the reference's initial ``A``.
"""
import numpy as np

LAYOUTS = {
    # name: (num_node, inward edges (child, parent), center, body-part id per joint)
    'nturgb+d': (25,
                 [(0, 1), (1, 20), (2, 20), (3, 2), (4, 20), (5, 4), (6, 5), (7, 6), (8, 20), (9, 8), (10, 9),
                  (11, 10), (12, 0), (13, 12), (14, 13), (15, 14), (16, 0), (17, 16), (18, 17), (19, 18),
                  (21, 7), (22, 7), (23, 11), (24, 11)],
                 20,
                 [0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 0, 1, 1, 2, 2]),
    'coco': (17,
             [(15, 13), (13, 11), (16, 14), (14, 12), (11, 5), (12, 6), (9, 7), (7, 5), (10, 8), (8, 6),
              (5, 0), (6, 0), (1, 0), (3, 1), (2, 0), (4, 2)],
             0,
             [0, 0, 0, 0, 0, 1, 2, 1, 2, 1, 2, 3, 4, 3, 4, 3, 4]),
    'openpose': (18,
                 [(4, 3), (3, 2), (7, 6), (6, 5), (13, 12), (12, 11), (10, 9), (9, 8), (11, 5), (8, 2), (5, 1),
                  (2, 1), (0, 1), (15, 0), (14, 0), (17, 15), (16, 14)],
                 1,
                 None),   # the reference defines no part table for openpose (graph.py:98-105)
}


def k_adjacency(A, k, with_self=False, self_factor=1):
    A = np.asarray(A)
    eye = np.eye(len(A), dtype=A.dtype)
    if k == 0:
        return eye
    Ak = (np.minimum(np.linalg.matrix_power(A + eye, k), 1)
          - np.minimum(np.linalg.matrix_power(A + eye, k - 1), 1))
    if with_self:
        Ak += self_factor * eye
    return Ak


def edge2mat(link, num_node):
    A = np.zeros((num_node, num_node))
    for i, j in link:
        A[j, i] = 1
    return A


def normalize_digraph(A, dim=0):
    deg = A.sum(dim)
    inv = np.zeros_like(deg)
    inv[deg > 0] = 1.0 / deg[deg > 0]
    return A @ np.diag(inv)


def get_hop_distance(num_node, edge, max_hop=1):
    adj = np.eye(num_node)
    for i, j in edge:
        adj[i, j] = adj[j, i] = 1
    hop = np.full((num_node, num_node), np.inf)
    reach = [np.linalg.matrix_power(adj, d) > 0 for d in range(max_hop + 1)]
    for d in range(max_hop, -1, -1):
        hop[reach[d]] = d
    return hop


def part_pair_classes(node_type):
    """edge_type[u, w]: rank of code(u)*code(w) among the distinct products, where
    code(v) = (p_v+1)*(-1)^(p_v+1) (graph.py:119-126).  5 parts -> 15 classes."""
    p = np.asarray(node_type).reshape(-1, 1) + 1
    code = p * np.power(-1, p)
    prod = code @ code.T
    uniq = np.unique(prod)
    et = np.zeros(prod.shape)
    for r, u in enumerate(uniq):
        et[prod == u] = r
    return et, uniq


class Graph:

    def __init__(self, layout='coco', mode='spatial', max_hop=1, nx_node=1, num_filter=3, init_std=0.02,
                 init_off=0.04):
        self.max_hop = max_hop
        self.layout = layout
        self.mode = mode
        self.num_filter = num_filter
        self.init_std = init_std
        self.init_off = init_off
        self.nx_node = nx_node
        assert nx_node == 1 or mode == 'random', "nx_node can be > 1 only if mode is 'random'"
        assert layout in ['openpose', 'nturgb+d', 'coco']
        self.get_layout(layout)
        self.hop_dis = get_hop_distance(self.num_node, self.inward, max_hop)
        assert hasattr(self, mode), f'Do Not Exist This Mode: {mode}'
        self.A = getattr(self, mode)()

    def get_layout(self, layout):
        if layout not in LAYOUTS:
            raise ValueError(f'Do Not Exist This Layout: {layout}')
        self.num_node, inward, self.center, parts = LAYOUTS[layout]
        self.inward = list(inward)
        if parts is not None:
            self.node_type = list(parts)
            self.edge_type, self.edge_type_num = part_pair_classes(parts)
        self.self_link = [(i, i) for i in range(self.num_node)]
        self.outward = [(j, i) for (i, j) in self.inward]
        self.neighbor = self.inward + self.outward

    def stgcn_spatial(self):
        adj = (self.hop_dis <= self.max_hop).astype(float)
        nadj = normalize_digraph(adj)
        hop, c = self.hop_dis, self.center
        out = []
        for h in range(self.max_hop + 1):
            sel = hop.T == h                      # sel[i, j] <=> hop[j, i] == h
            closer = (hop[:, c][None, :] >= hop[:, c][:, None])   # [i, j]: hop[j,c] >= hop[i,c]
            close = np.where((sel & closer).T, nadj, 0.0)
            far = np.where((sel & ~closer).T, nadj, 0.0)
            out.append(close)
            if h > 0:
                out.append(far)
        return np.stack(out)

    def spatial(self):
        iden = edge2mat(self.self_link, self.num_node)
        inw = normalize_digraph(edge2mat(self.inward, self.num_node))
        outw = normalize_digraph(edge2mat(self.outward, self.num_node))
        return np.stack((iden, inw, outw))

    def binary_adj(self):
        return edge2mat(self.inward + self.outward, self.num_node)[None]

    def random(self):
        num_node = self.num_node * self.nx_node
        return np.random.randn(self.num_filter, num_node, num_node) * self.init_std + self.init_off
