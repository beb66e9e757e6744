"""SemGCN 2D-keypoint encoder (the third modality).

Reference: /root/reference/pycontrast/networks/SGCN/{create_SGCN,skeleton_meta,graph_utils,
sem_gcn,sem_graph_conv}.py.  Same parameter names/shapes (``state_dict`` compatible) and the same
math; the adjacency is built with plain torch instead of scipy.sparse.  ``coco17`` is an addition
of this build (17-joint COCO skeleton used by the benchmark config).
"""
import math
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

# parents per joint (-1 = root)                                   skeleton_meta.py:3-23
SKELETONS = {
    'mpii': [1, 2, 6, 6, 3, 4, -1, 6, 7, 8, 11, 12, 8, 8, 13, 14],
    'coco_reduce': [1, 2, 9, 10, 3, 4, -1, 8, 9, 6, 6, 10, 11],
    # COCO-17: nose, eyes, ears, shoulders, elbows, wrists, hips, knees, ankles
    'coco17': [-1, 0, 0, 1, 2, 0, 0, 5, 6, 7, 8, 5, 6, 11, 12, 13, 14],
}


def num_joints(name):
    return len(SKELETONS[name])


def adjacency(name):
    """Symmetric skeleton adjacency + identity, row-normalised (graph_utils.py:27-46)."""
    if name not in SKELETONS:
        raise NotImplementedError(name)
    parents = SKELETONS[name]
    n = len(parents)
    a = torch.zeros(n, n)
    for child, parent in enumerate(parents):
        if parent >= 0:
            a[child, parent] = 1.0
            a[parent, child] = 1.0
    a = a + torch.eye(n)
    return a / a.sum(1, keepdim=True)


def graph_index(adj):
    """int32 CSR/CSC description of the skeleton edges (incl. self loops) in the row-major order of
    ``adj[adj > 0]`` -- what the fused layer kernel (hcm_sgc_forward/backward) walks."""
    mask = adj > 0
    J = mask.shape[0]
    rows, cols = mask.nonzero(as_tuple=True)                    # row-major
    row_ptr = torch.zeros(J + 1, dtype=torch.int32)
    row_ptr[1:] = torch.cumsum(mask.sum(1), 0).to(torch.int32)
    order = torch.argsort(cols * J + rows)                       # by column, then row
    csc_ptr = torch.zeros(J + 1, dtype=torch.int32)
    csc_ptr[1:] = torch.cumsum(mask.sum(0), 0).to(torch.int32)
    return [row_ptr, cols.to(torch.int32), csc_ptr, order.to(torch.int32), rows.to(torch.int32)]


_FUSED = True      # module attribute (tests compare against the eager modules through _fusable)


def _fusable(x, cout):
    return _FUSED and x.is_cuda and x.dtype == torch.float32 and cout in (64, 128) and x.shape[1] <= 32


class SemGraphConv(nn.Module):
    """sem_graph_conv.py:9-57: learned edge weights soft-maxed over the skeleton adjacency;
    self-loops use W[0], neighbours W[1]."""

    def __init__(self, in_features, out_features, adj, bias=True):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.W = nn.Parameter(torch.zeros(2, in_features, out_features))
        nn.init.xavier_uniform_(self.W.data, gain=1.414)
        self.register_buffer('adj', adj.clone(), persistent=False)
        self.register_buffer('m', adj > 0, persistent=False)
        # flat positions of the edges in row-major order (what ``adj[self.m] = self.e`` enumerates);
        # indexing with them instead of the boolean mask avoids a device->host sync (nonzero) and
        # keeps the layer capturable in a hipGraph
        self.register_buffer('m_idx', (adj > 0).flatten().nonzero().flatten(), persistent=False)
        for name, t in zip(('g_row_ptr', 'g_col_idx', 'g_csc_ptr', 'g_csc_edge', 'g_edge_row'), graph_index(adj)):
            self.register_buffer(name, t, persistent=False)
        self.e = nn.Parameter(torch.ones(1, int((adj > 0).sum())))
        if bias:
            self.bias = nn.Parameter(torch.zeros(out_features))
            bound = 1.0 / math.sqrt(out_features)
            self.bias.data.uniform_(-bound, bound)
        else:
            self.register_parameter('bias', None)

    def edge_weights(self):
        n = self.adj.shape[0]
        logits = torch.full((n * n,), -9e15, dtype=self.e.dtype, device=self.e.device)
        logits = logits.index_copy(0, self.m_idx, self.e.reshape(-1)).view(n, n)
        return F.softmax(logits, dim=1)

    def graph(self):
        return [self.g_row_ptr, self.g_col_idx, self.g_csc_ptr, self.g_csc_edge, self.g_edge_row]

    def forward(self, x):
        if _fusable(x, self.out_features):          # MI355X: GEMM + one fused kernel
            from ... import hip_ops
            return hip_ops.sgc_layer(x, self.W, self.e, self.bias, self.graph())
        a = self.edge_weights()
        eye = torch.eye(a.shape[0], dtype=a.dtype, device=a.device)
        out = torch.matmul(a * eye, torch.matmul(x, self.W[0])) + torch.matmul(a * (1 - eye), torch.matmul(x, self.W[1]))
        return out if self.bias is None else out + self.bias.view(1, 1, -1)


class _GraphConv(nn.Module):
    def __init__(self, adj, cin, cout):
        super().__init__()
        self.gconv = SemGraphConv(cin, cout, adj)
        self.bn = nn.BatchNorm1d(cout)
        self.relu = nn.ReLU()

    def forward(self, x):
        gc = self.gconv
        if _fusable(x, gc.out_features):            # SemGraphConv + BatchNorm1d + ReLU in one kernel
            from ... import hip_ops
            if self.bn.training and self.bn.track_running_stats:
                self.bn.num_batches_tracked.add_(1)
            return hip_ops.sgc_layer(x, gc.W, gc.e, gc.bias, gc.graph(), bn=self.bn, relu=True)
        x = gc(x).transpose(1, 2)
        return self.relu(self.bn(x).transpose(1, 2))


class _ResGraphConv(nn.Module):
    def __init__(self, adj, cin, cout, hid):
        super().__init__()
        self.gconv1 = _GraphConv(adj, cin, hid)
        self.gconv2 = _GraphConv(adj, hid, cout)

    def forward(self, x):
        return x + self.gconv2(self.gconv1(x))


class SemGCN(nn.Module):
    """sem_gcn.py:60-95 with nodes_group=None, p_dropout=0 (the only configuration create_sgcn uses)."""

    def __init__(self, adj, hid_dim, coords_dim=(2, 3), num_layers=4):
        super().__init__()
        self.gconv_input = nn.Sequential(_GraphConv(adj, coords_dim[0], hid_dim))
        self.gconv_layers = nn.Sequential(*[_ResGraphConv(adj, hid_dim, hid_dim, hid_dim) for _ in range(num_layers)])
        self.gconv_output = SemGraphConv(hid_dim, coords_dim[1], adj)

    def forward(self, x):
        return self.gconv_output(self.gconv_layers(self.gconv_input(x)))


def create_sgcn(name, hidden_dim, num_layers):
    """create_SGCN.py:6-14."""
    return SemGCN(adjacency(name), hidden_dim, coords_dim=(2, hidden_dim), num_layers=num_layers)
