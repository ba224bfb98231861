"""Hierarchical box tree of the context-aggregation stage.

Host-side restatement of lib/combine_tree/cluster_tree.py `cluster_tree2D.node_cluster`
(:504-556) with `cluster_algorithm` (:558-593), `get_cluster` (:595-607), `get_distance`
(:609-615) and `get_new_bbox_feat` (:617-628), plus what `get_A_and_link` (:400-430) and
`get_lca` (lib/modeling_rel/model_builder_rel.py:1614-1619) derive from it.  The reference
runs this in numpy on the host too; the MI355X change is what comes OUT of it: instead of
dense (nodes x nodes) adjacency matrices uploaded 8x per clip (cluster_tree.py:206-215) it
emits a parent array, CSR child lists and per-level node ranges once per batch of clips
(the layout trace_tree_gru consumes), and an ancestor table for O(1) LCA lookups.

Arithmetic (float32 distances, exp, argsort/argmin tie behaviour) is kept expression for
expression so that the tree is the one the reference builds.
"""
import math

import numpy as np

MAX_LEVELS = 12  # TRACE_TREE_MAX_LEVELS


def _sqdist(x, y):
    return ((x - y) ** 2).sum(-1)


def _cluster_layer(feat, l, r, cluster_num):
    """One agglomeration layer over nodes l..r -> (centres, rest, owner): `centres[k]` founds new node k,
    `rest[i]` joins new node `owner[i]`.  Same float32 expressions as the reference, evaluated on
    broadcast views instead of index meshes (element values and reduction order are identical)."""
    sub = feat[l:r + 1]
    distance = _sqdist(sub[:, None, :], sub[None, :, :])
    score = -np.sum(np.exp(-distance), axis=1)          # density score (:601)
    ids = np.argsort(score) + l                          # (:602-603)
    ids = np.concatenate((ids[0::2], ids[1::2]))         # even ranks become centres (:604-606)
    centres, rest = ids[:cluster_num], ids[cluster_num:]
    if len(rest):
        d = _sqdist(feat[centres][:, None, :], feat[rest][None, :, :])
        owner = np.argmin(d, axis=0)                     # nearest centre (:584)
    else:
        owner = np.zeros(0, np.int64)
    return centres, rest, owner


def build_tree(leaf_boxes, max_tree_height=7, partition=2, with_layers=False):
    """leaf_boxes (n,4) float32 -> dict(parent, level_off, boxes, n_leaf) for one clip.  with_layers: also
    `layers`, the reference's own structure (cluster_tree.py:511, :586-593): one dict per layer,
    {new node: [its centre, then the joined nodes in the order they were assigned]}."""
    n = leaf_boxes.shape[0]
    if n <= 0:
        raise ValueError("a clip needs at least one box")
    # node count is known up front: n leaves + the layer sizes
    sizes = []
    c = int(math.ceil(n))
    for _ in range(max_tree_height):
        c = int(math.ceil(float(c) / partition))
        sizes.append(c)
        if c <= 1:
            break
    if c > 1:                                            # force a root (:543-548)
        sizes.append(1)
    if len(sizes) + 1 > MAX_LEVELS:
        raise ValueError("tree deeper than %d levels" % MAX_LEVELS)
    total = n + sum(sizes)
    feat = np.empty((total, 4), leaf_boxes.dtype)
    feat[:n] = leaf_boxes
    parent = np.full(total, -1, np.int32)
    level_off = [0, n]
    layers = []
    l, r = 0, n - 1
    for k in sizes:
        centres, rest, owner = _cluster_layer(feat, l, r, k)
        new_ids = np.arange(r + 1, r + 1 + k)
        if with_layers:
            d = {int(new_ids[i]): [int(centres[i])] for i in range(k)}
            for i in range(len(rest)):
                d[int(new_ids[owner[i]])].append(int(rest[i]))
            layers.append(d)
        parent[centres] = new_ids
        nb = feat[centres].copy()                        # hull of each group (:617-628): min / max are exact
        if len(rest):
            parent[rest] = new_ids[owner]
            rb = feat[rest]
            np.minimum.at(nb[:, 0], owner, rb[:, 0])
            np.minimum.at(nb[:, 1], owner, rb[:, 1])
            np.maximum.at(nb[:, 2], owner, rb[:, 2])
            np.maximum.at(nb[:, 3], owner, rb[:, 3])
        feat[r + 1:r + 1 + k] = nb
        l, r = r + 1, r + k
        level_off.append(r + 1)
    out = dict(parent=parent, level_off=level_off, boxes=feat, n_leaf=n)
    if with_layers:
        out["layers"] = layers
    return out


def adjacency_and_link(n_tree, layers):
    """cluster_tree.py:400-430 get_A_and_link for one clip: dense child->parent matrix A (A[son, fa] = 1), its
    transpose A_T and the ancestor table `link` (n_tree, levels + 1; -1 where undefined) - what the reference's
    callers receive in the 13-tuple of cluster_tree3d.forward.  The HIP path itself uses the parent / CSR arrays."""
    A_T = np.zeros((n_tree, n_tree), np.float32)
    A = np.zeros((n_tree, n_tree), np.float32)
    rev = layers[::-1]
    link = -np.ones((n_tree, len(rev) + 1), dtype=np.int64)
    link[n_tree - 1, len(rev)] = n_tree - 1
    j = len(rev) - 1
    for layer in rev:
        for fa, son in layer.items():
            A_T[fa, son] = 1.0
            A[son, fa] = 1.0
            link[son, j + 1:] = link[fa, j + 1:]
            link[son, j] = son
        j -= 1
    return A, A_T, link


def layer_sizes(n, max_tree_height=7, partition=2):
    """Node counts of the layers above n leaves (cluster_tree.py:529-548, forced root included)."""
    sizes = []
    c = int(math.ceil(n))
    for _ in range(max_tree_height):
        c = int(math.ceil(float(c) / partition))
        sizes.append(c)
        if c <= 1:
            break
    if c > 1:
        sizes.append(1)
    if len(sizes) + 1 > MAX_LEVELS:
        raise ValueError("tree deeper than %d levels" % MAX_LEVELS)
    return sizes


def build_trees_same_n(leaf_boxes, max_tree_height=7, partition=2):
    """`build_tree` for B clips with the same number of boxes in one pass: leaf_boxes (B,n,4) -> dict(parent
    (B,total) int32, level_off (shared list), boxes (B,total,4), n_leaf).  Every numpy call of the per-clip
    version runs once over a leading clip axis; reductions keep their axis order (the density sum stays a
    last-axis pairwise sum, argsort / argmin run per row), so each clip's tree is bit-identical."""
    B, n = leaf_boxes.shape[0], leaf_boxes.shape[1]
    if n <= 0:
        raise ValueError("a clip needs at least one box")
    sizes = layer_sizes(n, max_tree_height, partition)
    total = n + sum(sizes)
    feat = np.empty((B, total, 4), leaf_boxes.dtype)
    feat[:, :n] = leaf_boxes
    parent = np.full((B, total), -1, np.int32)
    level_off = [0, n]
    bi = np.arange(B)[:, None]
    l, r = 0, n - 1
    for k in sizes:
        sub = feat[:, l:r + 1]
        distance = _sqdist(sub[:, :, None, :], sub[:, None, :, :])
        score = -np.sum(np.exp(-distance), axis=2)
        ids = np.argsort(score, axis=1) + l
        ids = np.concatenate((ids[:, 0::2], ids[:, 1::2]), axis=1)
        centres, rest = ids[:, :k], ids[:, k:]
        new_ids = np.arange(r + 1, r + 1 + k)
        parent[bi, centres] = new_ids[None, :]
        nb = feat[bi, centres]                          # fancy indexing: a copy
        if rest.shape[1]:
            rb = feat[bi, rest]
            d = _sqdist(nb[:, :, None, :], rb[:, None, :, :])
            owner = np.argmin(d, axis=1)
            parent[bi, rest] = new_ids[owner]
            bb = np.broadcast_to(bi, owner.shape)
            np.minimum.at(nb[:, :, 0], (bb, owner), rb[:, :, 0])
            np.minimum.at(nb[:, :, 1], (bb, owner), rb[:, :, 1])
            np.maximum.at(nb[:, :, 2], (bb, owner), rb[:, :, 2])
            np.maximum.at(nb[:, :, 3], (bb, owner), rb[:, :, 3])
        feat[:, r + 1:r + 1 + k] = nb
        l, r = r + 1, r + k
        level_off.append(r + 1)
    return dict(parent=parent, level_off=level_off, boxes=feat, n_leaf=n)


def split_trees(bt):
    """Per-clip tree dicts (views) of a `build_trees_same_n` result."""
    return [dict(parent=bt["parent"][b], level_off=bt["level_off"], boxes=bt["boxes"][b], n_leaf=bt["n_leaf"])
            for b in range(bt["parent"].shape[0])]


def lca_same_n(bt, X, Y):
    """`lca` for every clip of a `build_trees_same_n` result and one shared leaf-pair list: (B, P) node ids."""
    n = bt["n_leaf"]
    par = bt["parent"]
    B = par.shape[0]
    bi = np.arange(B)[:, None]
    rows = [np.broadcast_to(np.arange(n), (B, n))]
    for _ in range(len(bt["level_off"]) - 2):
        rows.append(par[bi, rows[-1]])
    anc = np.stack(rows, 0)                              # (levels, B, n)
    differ = (anc[:, :, X] != anc[:, :, Y]).sum(0)       # (B, P)
    differ = np.minimum(np.maximum(differ, 1), anc.shape[0] - 1)
    return anc[differ, bi, X[None, :]]


def forest_arrays_same_n(bt):
    """`forest_arrays` of a `build_trees_same_n` result (all trees have the same node count)."""
    par = bt["parent"]
    B, nt = par.shape
    base = (np.arange(B) * nt)[:, None]
    parent = np.where(par >= 0, par + base, -1)
    lo = np.asarray(bt["level_off"])
    n_levels = len(lo) - 1
    lo_pad = np.concatenate([lo, np.full(MAX_LEVELS + 1 - len(lo), lo[-1])])
    level_off = base + lo_pad[None, :]
    order = np.argsort(par, axis=1, kind="stable")       # the root (-1) sorts first: exactly one per tree
    child_idx = (order[:, 1:] + base).reshape(-1)
    counts = np.zeros((B, nt), np.int64)
    np.add.at(counts, (np.broadcast_to(np.arange(B)[:, None], (B, nt - 1)), par[np.arange(B)[:, None], order[:, 1:]]), 1)
    child_off = np.concatenate([[0], np.cumsum(counts.reshape(-1))])
    i32 = lambda a: np.ascontiguousarray(a, np.int32)
    return dict(tree_off=i32(np.arange(B + 1) * nt), level_off=i32(level_off).reshape(-1),
                n_levels=i32(np.full(B, n_levels)), parent=i32(parent.reshape(-1)), child_off=i32(child_off),
                child_idx=i32(child_idx))


def ancestors(tree):
    """(levels, n_leaf) table: row k = ancestor of each leaf at height k (row 0 = the leaf)."""
    n = tree["n_leaf"]
    par = tree["parent"]
    rows = [np.arange(n)]
    for _ in range(len(tree["level_off"]) - 2):
        rows.append(par[rows[-1]])
    return np.stack(rows, 0)


def lca(tree, X, Y):
    """Lowest common ancestor node id of leaf pairs (model_builder_rel.py:1614-1619)."""
    anc = ancestors(tree)
    differ = (anc[:, X] != anc[:, Y]).sum(0)             # ancestors agree from this height on
    differ = np.minimum(np.maximum(differ, 1), anc.shape[0] - 1)  # X == Y -> parent, as the reference's argmax+1
    return anc[differ, X]


def temporal_hulls(tree, temporal_boxes):
    """Per-frame hull boxes of the non-leaf nodes (cluster_tree3d.py:368-391): (n,T,4) -> (m,T,4)."""
    n = tree["n_leaf"]
    par = tree["parent"]
    n_tree = par.shape[0]
    T = temporal_boxes.shape[1]
    feat = np.zeros((n_tree, T, 4), np.float32)
    feat[:n] = temporal_boxes
    lo = tree["level_off"]
    for k in range(1, len(lo) - 1):
        for node in range(lo[k], lo[k + 1]):
            sons = np.nonzero(par == node)[0]
            sf = feat[sons]
            feat[node, :, 0] = sf[:, :, 0].min(0)
            feat[node, :, 1] = sf[:, :, 1].min(0)
            feat[node, :, 2] = sf[:, :, 2].max(0)
            feat[node, :, 3] = sf[:, :, 3].max(0)
    return feat[n:]


def forest_arrays(trees):
    """Concatenate per-clip trees into the int32 arrays trace_tree_gru takes (global node ids)."""
    tree_off = [0]
    parent, level_off, n_levels, child_off, child_idx = [], [], [], [0], []
    for t in trees:
        base = tree_off[-1]
        par = t["parent"]
        nt = par.shape[0]
        parent.append(np.where(par >= 0, par + base, -1))
        lo = [base + v for v in t["level_off"]]
        n_levels.append(len(lo) - 1)
        level_off.append(lo + [lo[-1]] * (MAX_LEVELS + 1 - len(lo)))
        order = np.argsort(par, kind="stable")
        order = order[par[order] >= 0]
        counts = np.bincount(par[par >= 0], minlength=nt)
        child_idx.append(order + base)
        child_off.extend((child_off[-1] + np.cumsum(counts)).tolist())
        tree_off.append(base + nt)
    i32 = lambda a: np.ascontiguousarray(a, np.int32)
    return dict(tree_off=i32(tree_off), level_off=i32(level_off).reshape(-1), n_levels=i32(n_levels),
                parent=i32(np.concatenate(parent)), child_off=i32(child_off),
                child_idx=i32(np.concatenate(child_idx)) if child_idx else i32([]))
