"""Topology ingestion: raw network description -> flattened `Topology` tables.

Stands in for the reference's offline prep (examples/create_topology.py:96-147, examples/graph_utils.py:10-116):
  * `.txt` files: first non-comment line = #nodes, second = #links, then "a b length" per link
  * SNDlib `.xml` files: node coordinates + links; geographical coordinates give haversine lengths (R = 6373 km,
    rounded to 3 decimals), planar coordinates give Euclidean lengths
  * k shortest paths by length (networkx `shortest_simple_paths`, the routine the reference uses, so that equal-length
    ties come out in the same order), best modulation per path (utils.py:84-96)
networkx is only needed here, never on the step path; the tables of the reference's two topologies are committed
under data/, and `tests/test_topology_io.py` checks that this module regenerates them exactly.
"""
import math
import os
import xml.dom.minidom
from itertools import islice

import numpy as np

from .topology import Modulation, Topology, get_best_modulation_format

DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")

# the reference's default modulation set (examples/create_topology.py:47-93)
DEFAULT_MODULATIONS = (
    Modulation("BPSK", 100_000, 1, 12.6, -14),
    Modulation("QPSK", 2_000, 2, 12.6, -17),
    Modulation("8QAM", 1_000, 3, 18.6, -20),
    Modulation("16QAM", 500, 4, 22.4, -23),
    Modulation("32QAM", 250, 5, 26.4, -26),
    Modulation("64QAM", 125, 6, 30.4, -29),
)


def _haversine_km(p1, p2):
    """examples/graph_utils.py:10-28 (note: the tuples are (x, y) = (lon, lat) but are used as (lat, lon) there)."""
    R = 6373.0
    lat1, lon1, lat2, lon2 = math.radians(p1[0]), math.radians(p1[1]), math.radians(p2[0]), math.radians(p2[1])
    dlon, dlat = lon2 - lon1, lat2 - lat1
    a = math.sin(dlat / 2) ** 2 + math.cos(lat1) * math.cos(lat2) * math.sin(dlon / 2) ** 2
    return R * (2 * math.atan2(math.sqrt(a), math.sqrt(1 - a)))


def read_txt(path):
    """-> (node_names, [(a, b, length, id)]) in file order (examples/graph_utils.py:89-116)."""
    lines = [ln for ln in open(path) if not ln.startswith("#")]
    n_nodes = int(lines[0])
    nodes = [str(i) for i in range(1, n_nodes + 1)]
    links = []
    for ln in lines[2:]:
        if len(ln) > 1:
            a, b, length = ln.replace("\n", "").split(" ")[:3]
            links.append((a, b, int(length), len(links)))
    return nodes, links


def read_sndlib_xml(path):
    """-> (node_names, [(a, b, length, id)]) (examples/graph_utils.py:31-86)."""
    doc = xml.dom.minidom.parse(path).documentElement
    ctype = doc.getElementsByTagName("nodes")[0].getAttribute("coordinatesType")
    pos, nodes = {}, []
    for node in doc.getElementsByTagName("node"):
        name = node.getAttribute("id")
        x = float(node.getElementsByTagName("x")[0].childNodes[0].data)
        y = float(node.getElementsByTagName("y")[0].childNodes[0].data)
        pos[name] = (x, y)
        nodes.append(name)
    links = []
    for link in doc.getElementsByTagName("link"):
        a = link.getElementsByTagName("source")[0].childNodes[0].data
        b = link.getElementsByTagName("target")[0].childNodes[0].data
        if ctype == "geographical":
            length = np.around(_haversine_km(pos[a], pos[b]), 3)
        else:
            length = np.around(math.sqrt((pos[a][0] - pos[b][0]) ** 2 + (pos[a][1] - pos[b][1]) ** 2), 3)
        links.append((a, b, length, link.getAttribute("id")))
    return nodes, links


def build_topology(path, name=None, modulations=DEFAULT_MODULATIONS, k_paths=5):
    """Raw file -> Topology, following examples/create_topology.py:96-147 step by step."""
    import networkx as nx  # offline prep only

    nodes, links = read_sndlib_xml(path) if path.endswith(".xml") else read_txt(path)
    g = nx.Graph()
    for n in nodes:
        g.add_node(n)
    for idx, (a, b, length, lid) in enumerate(links):
        g.add_edge(a, b, id=lid, index=idx, weight=1, length=length)
    node_names = list(g.nodes())
    n, e = len(node_names), g.number_of_edges()
    mods = list(modulations)
    per_pair = {}
    pid = 0
    for i1, n1 in enumerate(node_names):
        for i2, n2 in enumerate(node_names):
            if i1 < i2:
                paths = list(islice(nx.shortest_simple_paths(g, n1, n2, weight="length"), k_paths))
                objs = []
                for p in paths:
                    length = np.sum([g[p[i]][p[i + 1]]["length"] for i in range(len(p) - 1)])
                    objs.append((pid, p, length, mods.index(get_best_modulation_format(length, mods))))
                    pid += 1
                per_pair[(i1, i2)] = objs
                per_pair[(i2, i1)] = objs
    max_hops = max(len(p) - 1 for objs in per_pair.values() for _, p, _, _ in objs)
    n_paths = np.zeros((n, n), np.int32)
    hops = np.zeros((n, n, k_paths), np.int32)
    plinks = np.full((n, n, k_paths, max_hops), -1, np.int32)
    pnodes = np.full((n, n, k_paths, max_hops + 1), -1, np.int32)
    plen = np.zeros((n, n, k_paths), np.float64)
    ppid = np.full((n, n, k_paths), -1, np.int32)
    pmod = np.full((n, n, k_paths), -1, np.int32)
    for (s, d), objs in per_pair.items():
        n_paths[s, d] = len(objs)
        for q, (pid_, p, length, m) in enumerate(objs):
            hops[s, d, q] = len(p) - 1
            plen[s, d, q] = length
            ppid[s, d, q] = pid_
            pmod[s, d, q] = m
            for h in range(len(p) - 1):
                plinks[s, d, q, h] = g[p[h]][p[h + 1]]["index"]
            for h, nd in enumerate(p):
                pnodes[s, d, q, h] = node_names.index(nd)
    link_nodes = np.zeros((e, 2), np.int32)
    link_length = np.zeros(e, np.float64)
    link_ids = [None] * e
    order = np.zeros(e, np.int32)
    for it, (a, b) in enumerate(g.edges()):
        idx = g[a][b]["index"]
        order[it] = idx
        link_nodes[idx] = (node_names.index(a), node_names.index(b))
        link_length[idx] = g[a][b]["length"]
        link_ids[idx] = str(g[a][b]["id"])
    if name is None:
        name = os.path.splitext(os.path.basename(path))[0].upper()
    return Topology(name, node_names, k_paths, link_nodes, link_length, link_ids, order, n_paths, hops, plinks, pnodes,
                    plen, ppid, pmod, mods)


def save_topology(t, path):
    np.savez_compressed(
        path, name=np.array(t.name), node_names=np.array(t.node_names), k_paths=np.int32(t.k_paths),
        link_nodes=t.link_nodes, link_length=t.link_length, link_ids=np.array(t.link_ids),
        edge_iter_order=t.edge_iter_order, n_paths=t.n_paths, path_hops=t.path_hops, path_links=t.path_links,
        path_nodes=t.path_nodes, path_length=t.path_length, path_id=t.path_id, path_best_mod=t.path_best_mod,
        mod_name=np.array([m.name for m in t.modulations]),
        mod_max_length=np.array([m.maximum_length for m in t.modulations], np.float64),
        mod_se=np.array([m.spectral_efficiency for m in t.modulations], np.int32),
        mod_min_osnr=np.array([m.minimum_osnr for m in t.modulations], np.float64),
        mod_inband_xt=np.array([m.inband_xt for m in t.modulations], np.float64))


# ---- compatibility loader for the reference's pickled topologies -------------------------------------------------------
def load_reference_pickle(path):
    """A topology file written by the reference's examples/create_topology.py:184-185 (a pickled networkx graph whose
    graph["ksp"] holds `optical_rl_gym.utils.Path` objects and graph["modulations"] `optical_rl_gym.utils.Modulation`s)
    -> flattened `Topology`.  The reference package need not be importable: the two classes are resolved to stand-ins that
    just take the pickled attributes.  Needs networkx (the graph class inside the pickle)."""
    import pickle

    class _Attrs:
        def __setstate__(self, state):
            self.__dict__.update(state if isinstance(state, dict) else state[0] or {})

    # A pickle can name any importable callable: only what such a file legitimately holds is resolved — the networkx graph
    # and view classes, numpy scalars / arrays / dtypes, plain containers — and anything else is refused.
    safe_builtins = {"set", "frozenset", "list", "dict", "tuple", "int", "float", "complex", "str", "bytes", "bytearray", "bool",
                     "slice", "range", "object"}
    safe_numpy = {("numpy", "dtype"), ("numpy", "ndarray"), ("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar"),
                  ("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"),
                  ("numpy.core.numeric", "_frombuffer"), ("numpy._core.numeric", "_frombuffer")}

    class _Unpickler(pickle.Unpickler):
        def find_class(self, module, name):
            if module.startswith("optical_rl_gym"):
                return type(name, (_Attrs,), {})
            top = module.split(".")[0]
            if (top == "networkx" and module.startswith("networkx.classes.")) or (module, name) in safe_numpy or \
                    (module == "builtins" and name in safe_builtins) or \
                    (module == "collections" and name in ("OrderedDict", "defaultdict", "deque")) or \
                    (module, name) == ("copyreg", "_reconstructor"):
                return super().find_class(module, name)
            raise pickle.UnpicklingError("topology file names %s.%s, which a topology pickle has no business loading" % (module, name))

    with open(path, "rb") as f:
        g = _Unpickler(f).load()
    return topology_from_reference_graph(g)


def topology_from_reference_graph(g):
    """networkx graph in the reference's layout (graph["ksp"], ["k_paths"], ["modulations"], ["node_indices"], ["name"];
    edge attrs index / id / length: create_topology.py:138-147, graph_utils.py:77-84, 106-113) -> `Topology`."""
    nodes = list(g.graph["node_indices"])
    pos = {n: i for i, n in enumerate(nodes)}
    e, k = g.number_of_edges(), int(g.graph["k_paths"])
    mods = [Modulation(str(m.name), float(m.maximum_length), int(m.spectral_efficiency),
                       None if getattr(m, "minimum_osnr", None) is None else float(m.minimum_osnr),
                       None if getattr(m, "inband_xt", None) is None else float(m.inband_xt)) for m in g.graph["modulations"]]
    mod_index = {m.name: i for i, m in enumerate(mods)}
    link_nodes, link_length = np.zeros((e, 2), np.int32), np.zeros(e, np.float64)
    link_ids, order = [None] * e, np.zeros(e, np.int32)
    for it, (a, b) in enumerate(g.edges()):
        idx = g[a][b]["index"]
        order[it] = idx
        link_nodes[idx] = (pos[a], pos[b])
        link_length[idx] = g[a][b]["length"]
        link_ids[idx] = str(g[a][b].get("id", idx))
    n = len(nodes)
    hmax = max(p.hops for paths in g.graph["ksp"].values() for p in paths)
    n_paths = np.zeros((n, n), np.int32)
    hops = np.zeros((n, n, k), np.int32)
    links = np.full((n, n, k, hmax), -1, np.int32)
    pnodes = np.full((n, n, k, hmax + 1), -1, np.int32)
    length = np.zeros((n, n, k), np.float64)
    pid = np.full((n, n, k), -1, np.int32)
    best = np.full((n, n, k), -1, np.int32)
    for (s, d), paths in g.graph["ksp"].items():
        si, di = pos[s], pos[d]
        n_paths[si, di] = len(paths)
        for ip, p in enumerate(paths):
            hops[si, di, ip] = p.hops
            length[si, di, ip] = p.length
            pid[si, di, ip] = p.path_id
            best[si, di, ip] = mod_index[p.best_modulation.name]
            for h in range(p.hops):
                links[si, di, ip, h] = g[p.node_list[h]][p.node_list[h + 1]]["index"]
            for h, nd in enumerate(p.node_list):
                pnodes[si, di, ip, h] = pos[nd]
    return Topology(str(g.graph["name"]), [str(x) for x in nodes], k, link_nodes, link_length, link_ids, order, n_paths, hops,
                    links, pnodes, length, pid, best, mods)


if __name__ == "__main__":
    import sys

    src = sys.argv[1]
    topo = build_topology(src, k_paths=int(sys.argv[2]) if len(sys.argv) > 2 else 5)
    out = os.path.join(DATA, os.path.splitext(os.path.basename(src))[0] + "_%d-paths_%d-modulations.npz" % (topo.k_paths, len(topo.modulations)))
    save_topology(topo, out)
    print("wrote", out, "N", topo.n_nodes, "E", topo.n_links, "Hmax", topo.max_hops)
