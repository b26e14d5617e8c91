"""Own topology ingestion (txt / SNDlib xml readers + KSP + flattening) regenerates, bit for bit, the tables that
oracle/gen_golden.py flattened from the reference's pickled graphs."""
import os

import numpy as np
import pytest

from optical_rl_gym_amd.topology import Topology
from optical_rl_gym_amd.topology_io import DATA, build_topology

nx = pytest.importorskip("networkx")


@pytest.mark.parametrize("raw,table", [("nsfnet_chen.txt", "nsfnet_chen"), ("germany50.xml", "germany50")])
def test_rebuild_matches_reference_tables(raw, table):
    built = build_topology(os.path.join(DATA, "topologies", raw))
    ref = Topology.load(table)
    assert built.node_names == ref.node_names and built.k_paths == ref.k_paths
    for f in ("link_nodes", "link_length", "edge_iter_order", "n_paths", "path_hops", "path_links", "path_nodes",
              "path_length", "path_id", "path_best_mod"):
        assert np.array_equal(getattr(built, f), getattr(ref, f)), f
    assert [m.name for m in built.modulations] == [m.name for m in ref.modulations]


def test_cost239_table_present():
    t = Topology.load("cost239")
    assert t.n_nodes == 11 and t.n_links == 26 and (t.n_paths[~np.eye(11, dtype=bool)] == 5).all()
